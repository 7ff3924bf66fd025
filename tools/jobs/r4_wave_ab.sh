#!/bin/bash
# round 4: forms of the 16-candidate pass -- parity tests, A/B of builds, wave stamps of measurement builds
#   r4_wave_ab.sh <out> "<libs>" "<stats libs>"
out=gpurun_out/$1; mkdir -p $out
timeout 1200 python3 -m pytest tests/test_gpu_kcenters.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -3 | tee $out/tests.log
for n in 1000000 125000; do
LAB_CONFIGS="1,0,16;1,1,-1" python3 tools/lab_pass.py $2 --n $n --centers 2000 2>&1 | grep -v amdgpu.ids > $out/lab_$n.log; cat $out/lab_$n.log
done
for lib in $3; do
LAB_CONFIGS="1,0,16" LAB_REPS=1 python3 tools/lab_pass.py $lib --n 1000000 --centers 1200 2>&1 | grep -v amdgpu.ids | tee -a $out/stamps.log | tail -3
done
