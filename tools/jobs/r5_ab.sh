#!/bin/bash
# round 5: new GPU tests; the pick per 64 / per 256 frames A/B; per-label caps; where chains break
out=gpurun_out/${1:-r5_ab}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_qcp_device.py -x -q -m gpu > $out/tests_qcp.log 2>&1
tail -3 $out/tests_qcp.log
timeout 900 python3 -m pytest tests/test_gpu_sharded.py -x -q -m gpu -k "kmedoids_in_mpi or estimators" > $out/tests_km.log 2>&1
tail -3 $out/tests_km.log
V=enspara_amd/_variants
LAB_REPS=2 LAB_CONFIGS="1,1,-1,1,1;1,1,-1,1,0;1,0,16,1,1;1,0,16,1,0" timeout 1200 python3 tools/lab_pass.py enspara_amd/libenspara_hip.so $V/libper8.so $V/libper16.so $V/libper2k8.so --centers 5000 > $out/lab_1m.log 2>&1
grep -v amdgpu.ids $out/lab_1m.log
LAB_REPS=2 LAB_CONFIGS="1,1,-1,1,1;1,1,-1,1,0;1,0,16,1,1;1,0,16,1,0" timeout 600 python3 tools/lab_pass.py enspara_amd/libenspara_hip.so $V/libper8.so $V/libper16.so --n 125000 --centers 3000 > $out/lab_125k.log 2>&1
grep -v amdgpu.ids $out/lab_125k.log
LAB_REPS=1 LAB_CONFIGS="1,0,16,1,1" timeout 600 python3 tools/lab_pass.py $V/libstamps.so --centers 5000 > $out/stamps_1m.log 2>&1
grep -c "miss at" $out/stamps_1m.log
grep "miss at" $out/stamps_1m.log | awk '{r=$9; h=$14; ab=$NF; key=(r=="-1"?"notlisted":"listed") " hidden" h " above" ab; n[key]++} END {for (k in n) print n[k], k}' | sort -rn
grep "list:" $out/stamps_1m.log | awk '{e+=$2; l+=$4; n++} END {print "lists", n, "entries", e/n, "labels", l/n}'
