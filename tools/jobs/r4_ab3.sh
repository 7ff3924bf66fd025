#!/bin/bash
# round 4: A/B of library variants on one box (16-candidate pass, adaptive fit,
# 8-candidate pass) and the assign kernel:  r4_ab3.sh <out> "<libs>"
out=gpurun_out/$1; mkdir -p $out
V="$2"
for n in 1000000 125000; do
LAB_CONFIGS="1,0,16;1,1,-1;1,0,8" python3 tools/lab_pass.py $V --n $n --centers 2000 2>&1 | grep -v amdgpu.ids > $out/lab_$n.log; cat $out/lab_$n.log
done
for lib in $V; do
ENSPARA_HIP_LIB=$PWD/$lib python3 tools/quick_bench2.py 1000000 300 5000 assign 2>&1 | grep -E "variant" | sed "s|^|$lib |" | tee -a $out/assign.log
done
