#!/bin/bash
# Round 6, eighth call: MSM (scan with loads in flight, histogram gathered in LDS): tests, A/B, trace
out=gpurun_out/$1; shift
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_msm.py -q -m gpu -x --durations=5 > $out/tests.log 2>&1; tail -8 $out/tests.log
for f in 1 0; do
  EK_MSM_HIST_LDS=$f python3 tools/msm_probe.py 2>&1 | grep -v amdgpu.ids > $out/msm_probe_lds$f.log; cat $out/msm_probe_lds$f.log | cut -c1-220
done
for f in 1 0; do
EK_MSM_HIST_LDS=$f rocprofv3 --kernel-trace --output-format csv -d $out/trace$f -- python3 bench.py --centers 200 --steps 1 --warmup 0 --no-cpu-baseline --pam-sweeps 0 > $out/bench_msm_lds$f.json 2> $out/trace$f.err
fcsv=$(find $out/trace$f -name "*kernel_trace.csv" | head -1)
python3 tools/summarize_profile.py trace $fcsv $out/kernel_summary_msm_lds$f.csv
grep -E "msm" $out/kernel_summary_msm_lds$f.csv | cut -c1-110
python3 -c "
import json; d=json.load(open('$out/bench_msm_lds$f.json'))['msm']; print('lds $f', d['counts_s_labels_resident_on_device'], d['counts_s_labels_from_host_arrays'], d['counts_equal_scipy'])"
rm -rf $out/trace$f
done
