#!/bin/bash
# the exchange with the headers first: parity suite of the mailbox rounds, then the proxy
export GPU_MAX_HW_QUEUES=16
out=gpurun_out/two_phase; mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_sharded.py -x -q 2>&1 | tail -5 > $out/tests.log; cat $out/tests.log
for tp in 1 0; do
  MS_TWO_PHASE=$tp timeout 300 python3 tools/ms_probe.py 125000 300 3000 1 16 3 2>&1 | grep -v amdgpu.ids > $out/ms_125k_tp$tp.log; tail -3 $out/ms_125k_tp$tp.log | cut -c1-300
  MS_TWO_PHASE=$tp timeout 300 python3 tools/ms_probe.py 250000 300 3000 2 16 2 2>&1 | grep -v amdgpu.ids > $out/ms_2x125k_tp$tp.log; tail -3 $out/ms_2x125k_tp$tp.log | cut -c1-300
done
for i in 1 2 3 4 5 6; do
  MS_TWO_PHASE=1 timeout 300 python3 tools/ms_probe.py 1000000 300 5000 8 -1 1 2>&1 | grep -v amdgpu.ids > $out/ms_8x125k_tp1_$i.log; tail -3 $out/ms_8x125k_tp1_$i.log | cut -c1-200
done
MS_TWO_PHASE=0 timeout 300 python3 tools/ms_probe.py 1000000 300 5000 8 -1 1 2>&1 | grep -v amdgpu.ids > $out/ms_8x125k_tp0.log; tail -3 $out/ms_8x125k_tp0.log | cut -c1-200
