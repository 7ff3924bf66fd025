#!/bin/bash
# Round 6, tenth call: a broken chain taken up at once (EK_OPT_MS_INLINE): tests, the proxy A/B
out=gpurun_out/$1; shift
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
timeout 1500 python3 -m pytest tests/test_gpu_sharded.py -q -m gpu -x --durations=5 > $out/tests.log 2>&1
tail -8 $out/tests.log
for il in 1 0; do
  MS_INLINE=$il python3 tools/ms_probe.py 125000 300 3000 1 16 3 2>&1 | grep -v amdgpu.ids > $out/ms_125k_inline$il.log; tail -4 $out/ms_125k_inline$il.log | cut -c1-250
done
MS_INLINE=1 python3 tools/ms_probe.py 125000 300 3000 1 -1 3 2>&1 | grep -v amdgpu.ids > $out/ms_125k_ladder_inline1.log; tail -3 $out/ms_125k_ladder_inline1.log | cut -c1-250
MS_INLINE=1 python3 tools/ms_probe.py 250000 300 3000 2 16 2 2>&1 | grep -v amdgpu.ids > $out/ms_2x125k_inline1.log; tail -3 $out/ms_2x125k_inline1.log | cut -c1-250
python3 bench.py --sharded --no-cpu-baseline --pam-sweeps 0 --no-msm > $out/bench_sharded_1rank_mailbox.json 2> /dev/null
python3 -c "
import json; d=json.load(open('$out/bench_sharded_1rank_mailbox.json')); print('sharded 1 rank', d['value'], d['per_rank'][0].get('exchanges'), d['per_rank'][0].get('exchanges_without_a_pass'))"
