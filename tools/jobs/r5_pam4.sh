#!/bin/bash
out=gpurun_out/${1:-r5_pam4}
mkdir -p $out
cd $GRAFT_REPO_ROOT
LAB_PAM_OPTS="19=1;19=0" timeout 900 python3 tools/lab_pam.py enspara_amd/_variants/libspprof.so --reps 1 --sweeps 2 2>&1 | grep -v amdgpu.ids | tee $out/lab_pam_prof.log
