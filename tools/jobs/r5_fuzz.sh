#!/bin/bash
out=gpurun_out/${1:-r5_fuzz}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python3 tools/fuzz_gpu3.py 30 5 2>&1 | grep -v amdgpu.ids | tail -8 | tee $out/fuzz_gpu3.log
timeout 900 python3 tools/fuzz_gpu2.py 400 5 2>&1 | grep -v amdgpu.ids | tail -4 | tee $out/fuzz_gpu2.log
