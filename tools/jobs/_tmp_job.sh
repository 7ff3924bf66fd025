timeout 300 python3 tools/fuzz_gpu.py 3000 21 2>&1 | grep -v amdgpu | tail -1
timeout 300 python3 tools/fuzz_gpu2.py 800 21 2>&1 | grep -v amdgpu | tail -1
timeout 400 python3 tools/fuzz_gpu3.py 25 21 2>&1 | grep -v amdgpu | tail -1
timeout 300 python3 tools/stress_rounds.py 12 2>&1 | grep -v amdgpu | tail -7
