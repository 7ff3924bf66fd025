#!/bin/bash
out=gpurun_out/${1:-r5_bench}
mkdir -p $out
cd $GRAFT_REPO_ROOT
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
python3 -c "
import json; d=json.load(open('$out/bench_default.json')); print(d['value'], d['ms_per_step'], 'traffic', d['roofline']['traffic'], 'khybrid', d['khybrid']['s_per_sweep_runs'], 'msm', d['msm']['top20_eigenpairs_s'], d['msm']['eigenvalues_max_abs_diff_vs_arpack'])"
