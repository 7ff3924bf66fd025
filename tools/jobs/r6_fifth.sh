#!/bin/bash
# Round 6, fifth call: sweep fix + auto rule: tests; probes (split of roles on a SIMD); stamps
out=gpurun_out/$1; shift
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
timeout 1500 python3 -m pytest tests/test_gpu_kcenters.py tests/test_gpu_sharded.py -q -m gpu -x --durations=5 > $out/tests.log 2>&1
tail -6 $out/tests.log
./tools/probes/split_probe 2>&1 | grep -v amdgpu.ids > $out/split_probe.log; cat $out/split_probe.log
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 tools/ms_probe.py 125000 300 3000 1 16 2 > $out/ms_trace.log 2>&1
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 tools/summarize_profile.py trace $f $out/kernel_summary_ms125k.csv
head -12 $out/kernel_summary_ms125k.csv | cut -c1-120
rm -rf $out/trace
rocprofv3 --kernel-trace --output-format csv -d $out/trace2 -- python3 tools/prof_spec.py 1000000 300 2000 -1 > $out/fit_trace.log 2>&1
f=$(find $out/trace2 -name "*kernel_trace.csv" | head -1)
python3 tools/summarize_profile.py trace $f $out/kernel_summary_1m.csv
head -12 $out/kernel_summary_1m.csv | cut -c1-120
rm -rf $out/trace2
