#!/bin/bash
# round 5: the walk's early phase -- does a wider per-label cap of the candidate list help there?
out=gpurun_out/${1:-r5_walk}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_golden.py -x -q -m gpu -k "pam" > $out/tests_pam.log 2>&1
tail -3 $out/tests_pam.log
for lib in libenspara_hip.so _variants/libper8.so _variants/libper16.so; do
  ENSPARA_HIP_LIB=$PWD/enspara_amd/$lib timeout 600 python3 bench.py --data walk --triangle 1 --no-cpu-baseline --pam-sweeps 0 --no-msm > $out/walk_$(basename $lib .so).json 2> $out/walk_$(basename $lib .so).err
  python3 -c "
import json; d=json.loads(open('$out/walk_$(basename $lib .so).json').read().strip().splitlines()[-1]); print('$lib', '%.4g' % d['value'], d['config']['passes_by_candidates'])"
done
