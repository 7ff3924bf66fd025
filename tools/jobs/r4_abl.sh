#!/bin/bash
# round 4: ablation builds of the 16-candidate pass (-DEK_P16_ABLATE): r4_abl.sh <out> "<libs>"
out=gpurun_out/$1; mkdir -p $out
V="$2"
for n in 1000000 125000; do
LAB_CONFIGS="1,0,16" LAB_REPS=1 python3 tools/lab_pass.py $V --n $n --centers 400 2>&1 | grep -v "amdgpu.ids\|checksums" > $out/lab_$n.log; cat $out/lab_$n.log
done
