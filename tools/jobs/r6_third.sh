#!/bin/bash
# Round 6, third call: the per-prefix maxima taken by the pass (option pass_sweep) -- parity
# tests, A/B at 10^6 and 125 000 frames, the bench line -- then configs[3] whole.
out=gpurun_out/$1; shift
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_gpu_kcenters.py tests/test_gpu_golden.py -q -m gpu -x --durations=5 > $out/tests.log 2>&1
tail -10 $out/tests.log
LAB_REPS=3 LAB_CONFIGS="1,0,16,1,1,1;1,0,16,1,1,0;1,1,-1,1,1,1;1,1,-1,1,1,0" python3 tools/lab_pass.py --centers 3000 2>&1 | grep -v amdgpu.ids > $out/sweep_ab_1m.log; cat $out/sweep_ab_1m.log | cut -c1-260
LAB_REPS=3 LAB_CONFIGS="1,0,16,1,1,1;1,0,16,1,1,0;1,1,-1,1,1,1;1,1,-1,1,1,0" python3 tools/lab_pass.py --n 125000 --centers 3000 2>&1 | grep -v amdgpu.ids > $out/sweep_ab_125k.log; cat $out/sweep_ab_125k.log | cut -c1-260
timeout 600 python3 bench.py --no-cpu-baseline --no-msm --pam-sweeps 0 > $out/bench_short.json 2> $out/bench_short.err
python3 -c "
import json; d=json.load(open('$out/bench_short.json')); print('value', d['value'], d['passes_over_frames'], d['config']['passes_by_candidates'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
bash tools/jobs/r6_c4.sh ${out#gpurun_out/}/c4_full
