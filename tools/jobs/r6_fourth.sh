#!/bin/bash
# Round 6, fourth call: in-pass sweep, second form (only the states that change are written) + across shards
out=gpurun_out/$1; shift
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
timeout 1500 python3 -m pytest tests/test_gpu_kcenters.py tests/test_gpu_golden.py tests/test_gpu_sharded.py -q -m gpu -x --durations=5 > $out/tests.log 2>&1
tail -10 $out/tests.log
C="1,0,16,1,1,1;1,0,16,1,1,0;1,1,-1,1,1,1;1,1,-1,1,1,0"
LAB_REPS=3 LAB_CONFIGS=$C python3 tools/lab_pass.py --centers 3000 2>&1 | grep -v amdgpu.ids > $out/sweep_ab_1m.log; cut -c1-200 $out/sweep_ab_1m.log
LAB_REPS=3 LAB_CONFIGS=$C python3 tools/lab_pass.py --n 125000 --centers 3000 2>&1 | grep -v amdgpu.ids > $out/sweep_ab_125k.log; cut -c1-200 $out/sweep_ab_125k.log
for sw in 1 0; do
  MS_SWEEP=$sw python3 tools/ms_probe.py 125000 300 3000 1 16 3 2>&1 | grep -v amdgpu.ids > $out/ms_125k_sweep$sw.log; tail -3 $out/ms_125k_sweep$sw.log | cut -c1-200
done
MS_SWEEP=1 python3 tools/ms_probe.py 125000 300 3000 1 -1 3 2>&1 | grep -v amdgpu.ids > $out/ms_125k_ladder_sweep1.log; tail -2 $out/ms_125k_ladder_sweep1.log | cut -c1-200
