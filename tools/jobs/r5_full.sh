#!/bin/bash
# round 5: the whole GPU suite + the driver's bench line on one box
out=gpurun_out/${1:-r5_full}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1
tail -5 $out/gpu_tests.log
timeout 900 python3 bench.py --steps 20 --warmup 2 > $out/bench_default.json 2> $out/bench_default.err
python3 -c "
import json,sys
d=json.loads(open('$out/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['kernel'], d['roofline']['frac'], d['config']['passes_by_candidates'])
"
