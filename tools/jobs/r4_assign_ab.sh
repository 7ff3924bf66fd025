#!/bin/bash
# round 4: A/B of library builds on the assign kernels + their parity tests:  r4_assign_ab.sh <out> "<libs>"
out=gpurun_out/$1; mkdir -p $out
for lib in $2; do
ENSPARA_HIP_LIB=$PWD/$lib python3 tools/quick_bench2.py 1000000 300 5000 assign 2>&1 | grep -E "variant" | sed "s|^|$lib |" | tee -a $out/assign.log
ENSPARA_HIP_LIB=$PWD/$lib python3 tools/quick_bench2.py 1250000 500 2000 assign 2>&1 | grep -E "variant" | sed "s|^|$lib |" | tee -a $out/assign.log
done
timeout 900 python3 -m pytest tests -m gpu -x -q -k "assign or golden or nearest" 2>&1 | tail -3 | tee $out/tests.log
