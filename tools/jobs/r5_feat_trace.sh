#!/bin/bash
out=gpurun_out/${1:-r5_feat_trace}
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for w in 1 0; do
  export EK_FEAT_PAM_WINDOWS=$w
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace$w -- python3 tools/feat_pam_probe.py 1000000 64 1000 --no-host --clustered > $out/probe$w.log 2>&1
  f=$(find $out/trace$w -name "*kernel_trace.csv" | head -1)
  python3 tools/summarize_profile.py trace $f $out/kernel_summary_w$w.csv
  rm -rf $out/trace$w
  grep -v amdgpu $out/probe$w.log | tail -1
  head -18 $out/kernel_summary_w$w.csv | cut -c1-120
done
