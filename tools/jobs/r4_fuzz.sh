#!/bin/bash
# round 4: the randomized device-vs-oracle runs and the round stress test on the final build
out=gpurun_out/$1; mkdir -p $out
timeout 1500 python3 tools/fuzz_gpu.py 3000 4 2>&1 | grep -v amdgpu.ids | tail -4 > $out/fuzz_gpu.log; cat $out/fuzz_gpu.log
timeout 900 python3 tools/fuzz_gpu2.py 800 4 2>&1 | grep -v amdgpu.ids | tail -4 > $out/fuzz_gpu2.log; cat $out/fuzz_gpu2.log
timeout 900 python3 tools/fuzz_gpu3.py 25 4 2>&1 | grep -v amdgpu.ids | tail -4 > $out/fuzz_gpu3.log; cat $out/fuzz_gpu3.log
timeout 900 python3 tools/stress_rounds.py 10 2>&1 | grep -v amdgpu.ids | tail -8 > $out/stress_rounds.log; cat $out/stress_rounds.log
