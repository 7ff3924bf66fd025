#!/bin/bash
# round 5: feature-space PAM with windows
out=gpurun_out/${1:-r5_feat}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_features.py -x -q -m gpu > $out/tests_features.log 2>&1
tail -5 $out/tests_features.log
for w in 1 0; do
  EK_FEAT_PAM_VERBOSE=1 EK_FEAT_PAM_WINDOWS=$w timeout 600 python3 tools/feat_pam_probe.py 200000 16 400 --no-host 2>&1 | grep -v amdgpu.ids | tee -a $out/feat_pam_probe.log
  EK_FEAT_PAM_VERBOSE=1 EK_FEAT_PAM_WINDOWS=$w timeout 600 python3 tools/feat_pam_probe.py 1000000 64 1000 --no-host 2>&1 | grep -v amdgpu.ids | tee -a $out/feat_pam_probe.log
done
