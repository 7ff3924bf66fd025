#!/bin/bash
out=gpurun_out/${1:-r5_zc}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_golden.py tests/test_gpu_fuzz.py -x -q -m gpu -k "pam or hybrid or kmedoids or fuzz" > $out/tests_pam.log 2>&1
tail -3 $out/tests_pam.log
LAB_PAM_OPTS="20=1;20=0" timeout 900 python3 tools/lab_pam.py --reps 3 2>&1 | grep -v amdgpu.ids | tee $out/lab_pam.log
LAB_PAM_OPTS="20=1;20=0" timeout 900 python3 tools/lab_pam.py --reps 2 --n 200000 --centers 2000 --walk 1 2>&1 | grep -v amdgpu.ids | tee $out/lab_pam_walk.log
timeout 900 python3 tools/fuzz_gpu3.py 20 13 2>&1 | grep -v amdgpu.ids | tail -2
