#!/bin/bash
# Round 6, first call: the new tests, the bench line with its new legs, the configs[3] job
# (small, then whole).  usage: r6_first.sh <outdir-under-gpurun_out>
out=gpurun_out/$1; shift
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_c4gen.py tests/test_gpu_bigk.py -q -m gpu -x --durations=10 > $out/new_tests.log 2>&1
tail -18 $out/new_tests.log
timeout 600 python3 -m pytest tests/test_gpu_sharded.py -q -m gpu -x -k "mailbox_rounds_between_contexts" --durations=5 > $out/sharded_tests.log 2>&1
tail -8 $out/sharded_tests.log
timeout 600 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
python3 - <<PY
import json
d=json.load(open('$out/bench_default.json'))
print('value', d['value'], 'roofline', {k:v for k,v in d['roofline'].items() if not isinstance(v,(dict,str))})
print('one_center', d.get('roofline_one_center'))
print('five', d['khybrid'].get('five_sweeps'))
PY
bash tools/jobs/r6_c4.sh ${out#gpurun_out/}/c4_small --frames-per-shard 131072 --centers 3000 --check-centers 120 --templates 2000
if grep -q '"ok": true' $out/c4_small/c4_one_gpu.json 2>/dev/null; then
  bash tools/jobs/r6_c4.sh ${out#gpurun_out/}/c4_full
fi
