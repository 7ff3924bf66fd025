#!/bin/bash
export GPU_MAX_HW_QUEUES=16
out=gpurun_out/two_phase; mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_sharded.py -x -q 2>&1 | tail -5 > $out/tests.log; cat $out/tests.log
ENSPARA_HIP_LIB=$PWD/enspara_amd/libenspara_hip_stamps.so timeout 300 python3 tools/ms_probe.py 125000 300 3000 1 16 2 2>&1 | grep -v amdgpu.ids > $out/stamps_125k.log; tail -16 $out/stamps_125k.log | cut -c1-200
timeout 300 python3 tools/ms_probe.py 125000 300 3000 1 16 3 2>&1 | grep -v amdgpu.ids | tail -2
timeout 300 python3 tools/ms_probe.py 250000 300 3000 2 16 2 2>&1 | grep -v amdgpu.ids | tail -2
for i in 1 2; do timeout 300 python3 tools/ms_probe.py 1000000 300 5000 8 -1 1 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-200; done
