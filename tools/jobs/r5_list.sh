#!/bin/bash
# round 5: a list of 128 far frames for the greedy choice (variant builds) against 64
out=gpurun_out/${1:-r5_list}
mkdir -p $out
cd $GRAFT_REPO_ROOT
V=enspara_amd/_variants
ENSPARA_HIP_LIB=$PWD/$V/liblist128.so timeout 900 python3 -m pytest tests/test_gpu_kcenters.py -x -q -m gpu -k "candidates_per_pass or rounds_of_16 or tiny or triangle" > $out/tests_list128.log 2>&1
tail -3 $out/tests_list128.log
timeout 900 python3 -m pytest tests/test_gpu_kcenters.py tests/test_gpu_sharded.py -x -q -m gpu -k "candidates_per_pass or rounds_of_16 or tiny or triangle or mailbox_rounds or two_device" > $out/tests_default.log 2>&1
tail -3 $out/tests_default.log
LAB_REPS=2 LAB_CONFIGS="1,1,-1;1,0,16" timeout 1200 python3 tools/lab_pass.py enspara_amd/libenspara_hip.so $V/liblist128.so $V/liblist128per8.so --centers 5000 > $out/lab_1m.log 2>&1
grep -v amdgpu.ids $out/lab_1m.log
LAB_REPS=2 LAB_CONFIGS="1,1,-1;1,0,16" timeout 600 python3 tools/lab_pass.py enspara_amd/libenspara_hip.so $V/liblist128.so $V/liblist128per8.so --n 125000 --centers 3000 > $out/lab_125k.log 2>&1
grep -v amdgpu.ids $out/lab_125k.log
for lib in libenspara_hip.so _variants/liblist128.so _variants/liblist128per8.so; do
  for data in "walk:--data walk" "t500:--templates 500"; do
    n=${data%%:*}; a=${data#*:}
    ENSPARA_HIP_LIB=$PWD/enspara_amd/$lib timeout 600 python3 bench.py $a --no-cpu-baseline --pam-sweeps 0 --no-msm > $out/${n}_$(basename $lib .so).json 2> /dev/null
    python3 -c "
import json; d=json.loads(open('$out/${n}_$(basename $lib .so).json').read().strip().splitlines()[-1]); print('$n $lib', '%.4g' % d['value'], d['config']['passes_by_candidates'])"
  done
done
