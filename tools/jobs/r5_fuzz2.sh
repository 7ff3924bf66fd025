#!/bin/bash
out=gpurun_out/${1:-r5_fuzz2}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python3 tools/fuzz_gpu.py 3000 31 2>&1 | grep -v amdgpu.ids | tail -3 | tee $out/fuzz_gpu.log
timeout 900 python3 tools/fuzz_gpu3.py 60 23 2>&1 | grep -v amdgpu.ids | tail -3 | tee $out/fuzz_gpu3.log
timeout 600 python3 tools/fuzz_gpu2.py 800 7 2>&1 | grep -v amdgpu.ids | tail -2 | tee $out/fuzz_gpu2.log
