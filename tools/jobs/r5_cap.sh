#!/bin/bash
# round 5: the pick's per-label cap by the yield (4 <-> 16)
out=gpurun_out/${1:-r5_cap}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_gpu_kcenters.py tests/test_gpu_golden.py -x -q -m gpu > $out/tests.log 2>&1
tail -3 $out/tests.log
B="--no-cpu-baseline --pam-sweeps 0 --no-msm"
for name in "default:" "walk:--data walk" "walk_tri:--data walk --triangle 1" "templates500:--templates 500" "c3:--frames 1250000 --atoms 500 --templates 20000 --centers 6000"; do
  n=${name%%:*}; a=${name#*:}
  timeout 900 python3 bench.py $a $B > $out/bench_$n.json 2> $out/bench_$n.err
  python3 -c "
import json; d=json.loads(open('$out/bench_$n.json').read().strip().splitlines()[-1]); print('$n', '%.4g' % d['value'], d['config']['passes_by_candidates'])"
done
for T in 16 -1; do timeout 300 python3 tools/ms_probe.py 125000 300 3000 1 $T 2>&1 | grep -v amdgpu.ids | tail -3; done
LAB_REPS=2 LAB_CONFIGS="1,1,-1;1,0,16" timeout 900 python3 tools/lab_pass.py --centers 5000 2>&1 | grep -v amdgpu.ids
