#!/bin/bash
# the driver's own command on the final tree + the randomized runs kept under profiles/r06/fuzz
export GPU_MAX_HW_QUEUES=16
out=gpurun_out/final6; mkdir -p $out/fuzz
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_command.json 2> $out/bench_driver_command.err
python3 - <<PY
import json
d = json.load(open('$out/bench_driver_command.json'))
print('driver command', d['value'], d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline_one_center'].get('traffic'))
PY
timeout 600 python3 tools/fuzz_ms.py 400 21 2>&1 | grep -v amdgpu.ids > $out/fuzz/fuzz_ms_400_21.log; tail -1 $out/fuzz/fuzz_ms_400_21.log
timeout 600 python3 tools/fuzz_ms.py 400 22 2>&1 | grep -v amdgpu.ids > $out/fuzz/fuzz_ms_400_22.log; tail -1 $out/fuzz/fuzz_ms_400_22.log
EK_POISON=1 timeout 600 python3 tools/fuzz_ms.py 400 23 2>&1 | grep -v amdgpu.ids > $out/fuzz/fuzz_ms_400_23_poisoned.log; tail -1 $out/fuzz/fuzz_ms_400_23_poisoned.log
timeout 600 python3 tools/fuzz_gpu.py 2000 31 2>&1 | grep -v amdgpu.ids > $out/fuzz/fuzz_gpu_2000_31.log; tail -1 $out/fuzz/fuzz_gpu_2000_31.log
timeout 600 python3 tools/fuzz_gpu2.py 800 32 2>&1 | grep -v amdgpu.ids > $out/fuzz/fuzz_gpu2_800_32.log; tail -1 $out/fuzz/fuzz_gpu2_800_32.log
