#!/bin/bash
# round 5: PAM windows, slots evaluated ahead -- tests, A/B, kernel trace
out=gpurun_out/${1:-r5_pam3}
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_golden.py tests/test_gpu_fuzz.py -x -q -m gpu -k "pam or hybrid or kmedoids or fuzz" > $out/tests_pam.log 2>&1
tail -5 $out/tests_pam.log
LAB_PAM_OPTS="19=1;19=0" timeout 900 python3 tools/lab_pam.py --reps 3 2>&1 | grep -v amdgpu.ids | tee $out/lab_pam.log
LAB_PAM_OPTS="19=1;19=0" timeout 900 python3 tools/lab_pam.py --reps 2 --n 200000 --centers 2000 --walk 1 2>&1 | grep -v amdgpu.ids | tee $out/lab_pam_walk.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --pam-sweeps 1 --no-msm > $out/bench_under_rocprof.json 2> $out/trace.err
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 tools/summarize_profile.py trace $f $out/kernel_summary.csv
rm -rf $out/trace
grep -E "pam|sp_|pw_|count_members|scan_counts|select_member" $out/kernel_summary.csv | cut -c1-110
LAB_PAM_OPTS="19=1" timeout 900 python3 tools/lab_pam.py enspara_amd/_variants/libspprof.so --reps 1 --sweeps 2 2>&1 | grep -v amdgpu.ids | tee $out/lab_pam_prof.log
timeout 1500 python3 tools/fuzz_gpu3.py 30 9 2>&1 | grep -v amdgpu.ids | tail -3 | tee $out/fuzz_gpu3.log
