#!/bin/bash
out=gpurun_out/${1:-r5_final}
mkdir -p $out
cd $GRAFT_REPO_ROOT
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
cut -c1-400 $out/bench_default.json
python3 -c "
import json; d=json.load(open('$out/bench_default.json')); print('traffic', d['roofline']['traffic'], 'khybrid', d['khybrid']['s_per_sweep_runs'], 'msm', d['msm']['top20_eigenpairs_s'])"
timeout 2400 python3 -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1
tail -4 $out/gpu_tests.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
