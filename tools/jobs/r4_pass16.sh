#!/bin/bash
# round 4: parity of the quad-row / broadcast-candidate 16-candidate pass, then
# A/B against the round-3 build on one box:  r4_pass16.sh <out> "<libs>"
out=gpurun_out/$1; mkdir -p $out
V="$2"
timeout 900 python3 -m pytest tests/test_gpu_kcenters.py tests/test_gpu_golden.py -m gpu -x -q -k "not pam" > $out/parity.log 2>&1
tail -3 $out/parity.log
for n in 1000000 125000; do
LAB_CONFIGS="1,0,16;1,1,-1" python3 tools/lab_pass.py $V --n $n --centers 2000 2>&1 | grep -v amdgpu.ids > $out/lab_$n.log; cat $out/lab_$n.log
done
