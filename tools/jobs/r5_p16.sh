#!/bin/bash
out=gpurun_out/${1:-r5_p16}
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_golden.py tests/test_gpu_fuzz.py -x -q -m gpu -k "pam or hybrid or kmedoids or fuzz" > $out/tests_pam.log 2>&1
tail -12 $out/tests_pam.log
LAB_PAM_OPTS="21=1" timeout 900 python3 tools/lab_pam.py --reps 3 2>&1 | grep -v amdgpu.ids | tee $out/lab_pam.log
LAB_PAM_OPTS="21=1" timeout 900 python3 tools/lab_pam.py --reps 2 --n 200000 --centers 2000 --walk 1 2>&1 | grep -v amdgpu.ids | tee $out/lab_pam_walk.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --pam-sweeps 1 --no-msm --steps 4 --warmup 0 > $out/bench_under_rocprof.json 2> $out/trace.err
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 tools/summarize_profile.py trace $f $out/kernel_summary.csv
rm -rf $out/trace
grep -E "pairs" $out/kernel_summary.csv | cut -c1-110
timeout 900 python3 tools/fuzz_gpu3.py 20 17 2>&1 | grep -v amdgpu.ids | tail -2
