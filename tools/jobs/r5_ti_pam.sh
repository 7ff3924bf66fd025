#!/bin/bash
# round 5: triangle inequality in rounds; PAM window tables as bounds
out=gpurun_out/${1:-r5_ti_pam2}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python3 tools/probes/pam_bounds_debug.py > $out/pam_debug.log 2>&1
grep -v amdgpu.ids $out/pam_debug.log
timeout 1200 python3 -m pytest tests/test_gpu_kcenters.py -x -q -m gpu -k "triangle" > $out/tests_ti.log 2>&1
tail -4 $out/tests_ti.log
timeout 1500 python3 -m pytest tests/test_gpu_golden.py -q -m gpu -k "pam or hybrid or kmedoids" > $out/tests_pam.log 2>&1
tail -4 $out/tests_pam.log
timeout 600 python3 tools/ti_probe.py 2000 500 300 3000 > $out/ti_probe.log 2>&1
grep -v amdgpu.ids $out/ti_probe.log
for tri in 0 1; do
  timeout 600 python3 bench.py --data walk --triangle $tri --cpu-seconds 1 > $out/bench_walk_tri$tri.json 2> $out/bench_walk_tri$tri.err
  timeout 600 python3 bench.py --triangle $tri --cpu-seconds 1 > $out/bench_default_tri$tri.json 2> $out/bench_default_tri$tri.err
done
python3 - <<PY
import json
for name in ("walk_tri0","walk_tri1","default_tri0","default_tri1"):
    try:
        d=json.loads(open("$out/bench_%s.json"%name).read().strip().splitlines()[-1])
        print(name, "%.4g"%d["value"], d["config"]["passes_by_candidates"], d.get("triangle_inequality"))
    except Exception as e:
        print(name, "failed", e)
PY
timeout 900 python3 -m pytest tests/test_gpu_sharded.py -x -q -m gpu -k "bench_two_ranks or mailbox_rounds" > $out/tests_bench2.log 2>&1
tail -4 $out/tests_bench2.log
