#!/bin/bash
# round 5: the queue's second float32 level against the build without it
out=gpurun_out/${1:-r5_level2}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_qcp_device.py tests/test_gpu_kcenters.py tests/test_gpu_golden.py -x -q -m gpu > $out/tests.log 2>&1
tail -3 $out/tests.log
V=enspara_amd/_variants
LAB_REPS=3 LAB_CONFIGS="1,0,16;1,1,-1" timeout 1200 python3 tools/lab_pass.py enspara_amd/libenspara_hip.so $V/libnolevel2.so --centers 5000 > $out/lab_1m.log 2>&1
grep -v amdgpu.ids $out/lab_1m.log
LAB_REPS=3 LAB_CONFIGS="1,0,16" timeout 600 python3 tools/lab_pass.py enspara_amd/libenspara_hip.so $V/libnolevel2.so --n 125000 --centers 3000 > $out/lab_125k.log 2>&1
grep -v amdgpu.ids $out/lab_125k.log
