#!/bin/bash
out=gpurun_out/${1:-r5_pairs_ab}
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_golden.py -x -q -m gpu -k "pam or hybrid or kmedoids" > $out/tests_pam.log 2>&1; tail -2 $out/tests_pam.log
timeout 900 python3 tools/lab_pam.py enspara_amd/libenspara_hip.so enspara_amd/_variants/libld65.so --reps 3 2>&1 | grep -v amdgpu.ids | tee $out/lab_pam.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --pam-sweeps 1 --no-msm --steps 4 --warmup 0 > $out/bench_under_rocprof.json 2> $out/trace.err
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 tools/summarize_profile.py trace $f $out/kernel_summary.csv
rm -rf $out/trace
grep -E "pairs_kernel" $out/kernel_summary.csv | cut -c1-110
