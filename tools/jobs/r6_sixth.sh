#!/bin/bash
# Round 6, sixth call: the whole -m gpu suite with durations; sweep forms at 10^6 again;
# the complete-parity bench artefact (two consecutive PAM sweeps replayed by the oracle)
out=gpurun_out/$1; shift
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
C="1,0,16,1,1,2;1,0,16,1,1,0;1,1,-1,1,1,2;1,1,-1,1,1,0"
LAB_REPS=3 LAB_CONFIGS=$C python3 tools/lab_pass.py --centers 3000 2>&1 | grep -v amdgpu.ids > $out/sweep_ab_1m.log; cut -c1-200 $out/sweep_ab_1m.log
LAB_REPS=3 LAB_CONFIGS=$C python3 tools/lab_pass.py --n 125000 --centers 3000 2>&1 | grep -v amdgpu.ids > $out/sweep_ab_125k.log; cut -c1-200 $out/sweep_ab_125k.log
( time timeout 1700 python3 -m pytest tests -q -m gpu --durations=25 ) > $out/gpu_tests.log 2>&1
tail -40 $out/gpu_tests.log
timeout 1500 python3 bench.py --cpu-seconds 0 > $out/bench_full_parity.json 2> $out/bench_full_parity.err
python3 -c "
import json; d=json.load(open('$out/bench_full_parity.json')); c=d['cpu_baseline']; k=d['khybrid']['parity']
print('full parity:', c['centers_checked'], c['centers_match_gpu'], c['whole_fit_state_vs_gpu'], k['proposals_replayed_by_oracle'], k['medoids_match_gpu'], k['whole_sweep_state_vs_oracle'], k['second_sweep_vs_oracle'])"
