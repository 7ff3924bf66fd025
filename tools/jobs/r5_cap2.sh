#!/bin/bash
# round 5: the per-label cap's probing, less eager
out=gpurun_out/${1:-r5_cap2}
mkdir -p $out
cd $GRAFT_REPO_ROOT
LAB_REPS=2 LAB_CONFIGS="1,1,-1;1,0,16" timeout 900 python3 tools/lab_pass.py --centers 5000 2>&1 | grep -v amdgpu.ids
LAB_REPS=2 LAB_CONFIGS="1,1,-1;1,0,16" timeout 600 python3 tools/lab_pass.py --n 125000 --centers 3000 2>&1 | grep -v amdgpu.ids
B="--no-cpu-baseline --pam-sweeps 0 --no-msm"
for name in "walk:--data walk" "walk_tri:--data walk --triangle 1" "templates500:--templates 500"; do
  n=${name%%:*}; a=${name#*:}
  timeout 900 python3 bench.py $a $B > $out/bench_$n.json 2> $out/bench_$n.err
  python3 -c "
import json; d=json.loads(open('$out/bench_$n.json').read().strip().splitlines()[-1]); print('$n', '%.4g' % d['value'], d['config']['passes_by_candidates'])"
done
for T in 16 -1; do timeout 300 python3 tools/ms_probe.py 125000 300 3000 1 $T 2>&1 | grep -v amdgpu.ids | tail -3; done
