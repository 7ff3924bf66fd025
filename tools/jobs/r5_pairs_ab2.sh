#!/bin/bash
out=gpurun_out/${1:-r5_pairs_ab2}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1200 python3 tools/lab_pam.py enspara_amd/libenspara_hip.so enspara_amd/_variants/libch16.so enspara_amd/_variants/libch24.so enspara_amd/_variants/libch32.so --reps 3 2>&1 | grep -v amdgpu.ids | tee $out/lab_pam.log
