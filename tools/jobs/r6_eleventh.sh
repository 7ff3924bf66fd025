#!/bin/bash
# Round 6: in-pass sweep, a wave reduction only where the arg-max frame's own distance changed
out=gpurun_out/$1; shift
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
timeout 1500 python3 -m pytest tests/test_gpu_kcenters.py tests/test_gpu_golden.py tests/test_gpu_sharded.py -q -m gpu -x > $out/tests.log 2>&1
tail -4 $out/tests.log
C="1,0,16,1,1,2;1,0,16,1,1,0;1,1,-1,1,1,2;1,1,-1,1,1,0"
LAB_REPS=3 LAB_CONFIGS=$C python3 tools/lab_pass.py --centers 5000 2>&1 | grep -v amdgpu.ids > $out/sweep_ab_1m.log; cut -c1-200 $out/sweep_ab_1m.log
LAB_REPS=3 LAB_CONFIGS=$C python3 tools/lab_pass.py --n 125000 --centers 3000 2>&1 | grep -v amdgpu.ids > $out/sweep_ab_125k.log; cut -c1-200 $out/sweep_ab_125k.log
python3 tools/ms_probe.py 125000 300 3000 1 16 3 2>&1 | grep -v amdgpu.ids > $out/ms_125k_untraced.log; tail -3 $out/ms_125k_untraced.log | cut -c1-200
