#!/bin/bash
export GPU_MAX_HW_QUEUES=16
out=gpurun_out/two_phase; mkdir -p $out
for i in 1 2 3 4; do
  MS_TWO_PHASE=1 timeout 300 python3 tools/ms_probe.py 1000000 300 5000 8 -1 1 2>&1 | grep -v amdgpu.ids > $out/ms_8x125k_tp1_$i.log; grep -c DBG $out/ms_8x125k_tp1_$i.log; tail -3 $out/ms_8x125k_tp1_$i.log | cut -c1-200
done
