#!/bin/bash
# Round 6: BASELINE.json configs[3]'s whole data set on one MI355X as eight contexts
# (tools/c4_one_gpu.py).  usage: r6_c4.sh <outdir-under-gpurun_out> [extra args]
out=gpurun_out/$1; shift
mkdir -p $out
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
timeout 1500 python3 tools/c4_one_gpu.py --out $out/c4_one_gpu.json "$@" > $out/c4_one_gpu.log 2>&1
echo "rc=$?" >> $out/c4_one_gpu.log
grep -v amdgpu.ids $out/c4_one_gpu.log | tail -25 | cut -c1-400
