#!/bin/bash
out=gpurun_out/${1:-r5_p16b}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1200 python3 tools/lab_pam.py enspara_amd/libenspara_hip.so enspara_amd/_variants/libdepth4.so enspara_amd/_variants/libdepth16.so --reps 3 2>&1 | grep -v amdgpu.ids | tee $out/lab_pam.log
