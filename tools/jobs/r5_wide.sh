#!/bin/bash
# round 5: rounds of 32 candidates -- parity tests of the new form, then 16 / 32 /
# ladder timed on one box (lab_pass: same frames, checksums must agree)
out=gpurun_out/${1:-r5_wide}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_kcenters.py -x -q -m gpu \
  -k "candidates_per_pass or rounds_of_16 or tiny or duplicate or adaptive" > $out/tests_kcenters.log 2>&1
tail -5 $out/tests_kcenters.log
timeout 1500 python3 -m pytest tests/test_gpu_sharded.py -x -q -m gpu \
  -k "mailbox_rounds or two_device or eight_processes" > $out/tests_sharded.log 2>&1
tail -5 $out/tests_sharded.log
LAB_REPS=2 LAB_CONFIGS="1,0,16;1,0,32;1,1,-1" timeout 900 python3 tools/lab_pass.py --centers 5000 > $out/lab_1m.log 2>&1
cat $out/lab_1m.log
LAB_REPS=2 LAB_CONFIGS="1,0,16;1,0,32;1,1,-1" timeout 600 python3 tools/lab_pass.py --n 125000 --centers 3000 > $out/lab_125k.log 2>&1
cat $out/lab_125k.log
