#!/bin/bash
# round 3: parity of the 16-candidate rounds + first timings (1M and 125k frames)
out=gpurun_out/$1; mkdir -p $out
python3 -m pytest tests/test_gpu_kcenters.py tests/test_gpu_golden.py -x -q -m gpu > $out/tests.log 2>&1; tail -3 $out/tests.log
LAB_CONFIGS="1,0,8;1,0,16;1,1,-1" python3 tools/lab_pass.py --centers 3000 2>&1 | grep -v amdgpu.ids > $out/lab_1m.log; cat $out/lab_1m.log
LAB_CONFIGS="1,0,8;1,0,16;1,1,-1" python3 tools/lab_pass.py --n 125000 --centers 3000 2>&1 | grep -v amdgpu.ids > $out/lab_125k.log; cat $out/lab_125k.log
