#!/bin/bash
# A/B of library variants: r3_ab.sh <out> "<libs>" "<configs>" <centers> <n...>
out=gpurun_out/$1; mkdir -p $out; shift
V="$1"; C="$2"; K=$3; shift 3
for n in "$@"; do
LAB_CONFIGS="$C" python3 tools/lab_pass.py $V --n $n --centers $K 2>&1 | grep -v amdgpu.ids | grep -v checksums > $out/lab_$n.log; cat $out/lab_$n.log
done
