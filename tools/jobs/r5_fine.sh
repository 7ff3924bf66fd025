#!/bin/bash
# round 5: maxima per 64 frames for the candidate pick -- parity, then 16 / 32 / ladder
out=gpurun_out/${1:-r5_fine}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_kcenters.py tests/test_gpu_qcp_device.py -x -q -m gpu > $out/tests_kcenters.log 2>&1
tail -5 $out/tests_kcenters.log
timeout 1500 python3 -m pytest tests/test_gpu_sharded.py -x -q -m gpu \
  -k "mailbox_rounds or two_device or eight_processes" > $out/tests_sharded.log 2>&1
tail -5 $out/tests_sharded.log
LAB_REPS=2 LAB_CONFIGS="1,0,16;1,0,32;1,1,-1;1,0,8" timeout 900 python3 tools/lab_pass.py --centers 5000 > $out/lab_1m.log 2>&1
cat $out/lab_1m.log
LAB_REPS=2 LAB_CONFIGS="1,0,16;1,0,32;1,1,-1" timeout 600 python3 tools/lab_pass.py --n 125000 --centers 3000 > $out/lab_125k.log 2>&1
cat $out/lab_125k.log
for T in 16 32 -1; do timeout 300 python3 tools/ms_probe.py 125000 300 3000 1 $T >> $out/ms_probe.log 2>&1; done
tail -12 $out/ms_probe.log
