#!/bin/bash
# Round 6, ninth call: MSM histogram with the trajectory staged per workgroup
out=gpurun_out/$1; shift
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_msm.py -q -m gpu -x > $out/tests.log 2>&1; tail -3 $out/tests.log
python3 tools/msm_probe.py 2>&1 | grep -v amdgpu.ids > $out/msm_probe.log; cat $out/msm_probe.log | cut -c1-220
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 bench.py --centers 200 --steps 1 --warmup 0 --no-cpu-baseline --pam-sweeps 0 > $out/bench_msm.json 2> $out/trace.err
fcsv=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 tools/summarize_profile.py trace $fcsv $out/kernel_summary_msm.csv
grep -E "msm" $out/kernel_summary_msm.csv | cut -c1-110
python3 -c "
import json; d=json.load(open('$out/bench_msm.json'))['msm']; print(d['counts_s_labels_resident_on_device'], d['counts_s_labels_from_host_arrays'], d['counts_equal_scipy'], d['counts_roofline'])"
rm -rf $out/trace
