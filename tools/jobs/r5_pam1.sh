#!/bin/bash
# round 5: the PAM sweep without a wait for the selected frames
out=gpurun_out/${1:-r5_pam1}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_golden.py tests/test_gpu_fuzz.py -x -q -m gpu -k "pam or hybrid or kmedoids or fuzz" > $out/tests_pam.log 2>&1
tail -3 $out/tests_pam.log
LAB_PAM_OPTS="16=1" timeout 900 python3 tools/lab_pam.py --reps 3 2>&1 | grep -v amdgpu.ids | tee $out/lab_pam.log
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "pam" > $out/tests_fullsize.log 2>&1
tail -3 $out/tests_fullsize.log
