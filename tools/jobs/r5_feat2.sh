#!/bin/bash
out=gpurun_out/${1:-r5_feat2}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_features.py -x -q -m gpu > $out/tests_features.log 2>&1
tail -3 $out/tests_features.log
for opt in "" "--clustered" "--explicit"; do
for w in auto 1 0; do
  if [ $w = auto ]; then unset EK_FEAT_PAM_WINDOWS; else export EK_FEAT_PAM_WINDOWS=$w; fi
  echo "== windows $w $opt" | tee -a $out/feat_pam_probe.log
  EK_FEAT_PAM_VERBOSE=1 timeout 600 python3 tools/feat_pam_probe.py 200000 16 400 --no-host $opt 2>&1 | grep -v amdgpu.ids | tee -a $out/feat_pam_probe.log
  EK_FEAT_PAM_VERBOSE=1 timeout 600 python3 tools/feat_pam_probe.py 1000000 64 1000 --no-host $opt 2>&1 | grep -v amdgpu.ids | tee -a $out/feat_pam_probe.log
done; done
