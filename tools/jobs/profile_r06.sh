#!/bin/bash
# Round 6: the numbers DESIGN.md / profiles/r06 quote, from one box and HEAD.
# usage: profile_r06.sh <outdir-under-gpurun_out> [part ...]
#   parts: tests bench full trace pmc c3 ms sweep msm probe driver fuzz c4
out=gpurun_out/$1; shift
parts=${@:-bench trace pmc c3 ms sweep msm probe}
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
has() { [[ " $parts " == *" $1 "* ]]; }
if has tests; then
  ( time timeout 1700 python3 -m pytest tests -q -m gpu --durations=25 ) > $out/gpu_tests.log 2>&1
  tail -3 $out/gpu_tests.log
fi
if has bench; then
  python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
  python3 bench.py --candidates 1 --no-cpu-baseline --pam-sweeps 0 --no-msm > $out/bench_candidates1.json 2> /dev/null
  python3 bench.py --candidates 8 --no-cpu-baseline --pam-sweeps 0 --no-msm > $out/bench_candidates8.json 2> /dev/null
  python3 bench.py --sharded --no-cpu-baseline --pam-sweeps 0 --no-msm > $out/bench_sharded_1rank_mailbox.json 2> /dev/null
  python3 - <<PY
import json
d = json.load(open('$out/bench_default.json'))
print('default', d['value'], d['ms_per_step'], {k: v for k, v in d['roofline'].items() if not isinstance(v, (dict, str))})
print('one center', {k: d['roofline_one_center'][k] for k in ('avg_launch_ms', 'achieved', 'frac', 'traffic')})
print('khybrid', d['khybrid']['s_per_sweep'], d['khybrid']['five_sweeps'])
print('msm', {k: v for k, v in d['msm'].items() if not isinstance(v, (dict, str))})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['centers_match_gpu'])
for f in ('candidates1', 'candidates8', 'sharded_1rank_mailbox'):
    e = json.load(open('$out/bench_%s.json' % f)); print(f, e['value'], e['roofline'].get('frac'))
PY
fi
if has full; then
  python3 bench.py --cpu-seconds 0 > $out/bench_full_parity.json 2> $out/bench_full_parity.err
fi
if has trace; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --pam-sweeps 1 --no-msm > $out/bench_under_rocprof.json 2> $out/trace.err
  f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
  python3 tools/summarize_profile.py trace $f $out/kernel_summary.csv
  cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv 2>/dev/null
  rm -rf $out/trace
  head -14 $out/kernel_summary.csv | cut -c1-110
fi
if has pmc; then
  for T in 16 8 1; do
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $c --output-format csv -d $out/pmc_${T}_$c -- python3 bench.py --centers 1000 --steps 4 --warmup 0 --candidates $T --no-cpu-baseline --pam-sweeps 0 --no-msm > $out/bench_under_pmc_${T}_$c.json 2> $out/pmc_${T}_$c.err
    done
  done
  python3 tools/summarize_profile.py pmc $(find $out/pmc_* -name "*counter_collection.csv") $out/pmc_summary.csv
  rm -rf $out/pmc_16_* $out/pmc_8_* $out/pmc_1_*
  grep -E "pass16|pass2|step_kernel" $out/pmc_summary.csv
fi
if has c3; then
  # BASELINE.json configs[3], one GPU's share: 1.25 M frames x 500 atoms
  C3="--frames 1250000 --atoms 500 --templates 20000 --centers 20000 --no-cpu-baseline --pam-sweeps 0 --no-msm"
  python3 bench.py $C3 > $out/bench_config3_shard.json 2> $out/bench_config3_shard.err
  cut -c1-300 $out/bench_config3_shard.json
fi
if has ms; then
  python3 tools/ms_probe.py 125000 300 3000 1 16 3 2>&1 | grep -v amdgpu.ids > $out/ms_125k_untraced.log
  python3 tools/ms_probe.py 125000 300 3000 1 -1 3 2>&1 | grep -v amdgpu.ids > $out/ms_125k_ladder_untraced.log
  MS_SWEEP=0 python3 tools/ms_probe.py 125000 300 3000 1 16 3 2>&1 | grep -v amdgpu.ids > $out/ms_125k_untraced_sweep_in_chain_kernel.log
  python3 tools/ms_probe.py 250000 300 3000 2 16 2 2>&1 | grep -v amdgpu.ids > $out/ms_2x125k_untraced.log
  # the exchange in one step (rounds 3-5: speculating offers, a broken chain offered again)
  MS_TWO_PHASE=0 python3 tools/ms_probe.py 125000 300 3000 1 16 3 2>&1 | grep -v amdgpu.ids > $out/ms_125k_untraced_one_step.log
  MS_TWO_PHASE=0 python3 tools/ms_probe.py 250000 300 3000 2 16 2 2>&1 | grep -v amdgpu.ids > $out/ms_2x125k_untraced_one_step.log
  MS_TWO_PHASE=0 python3 tools/ms_probe.py 1000000 300 5000 8 -1 1 2>&1 | grep -v amdgpu.ids > $out/ms_8x125k_one_step.log
  # where the chain and plan kernels' single workgroups spend their time (-DEK_MS_STAMPS)
  if [ -f enspara_amd/libenspara_hip_stamps.so ]; then
    ENSPARA_HIP_LIB=$PWD/enspara_amd/libenspara_hip_stamps.so python3 tools/ms_probe.py 125000 300 3000 1 16 2 2>&1 | grep -v amdgpu.ids > $out/ms_125k_stamps.log
  fi
  # the 8-way split of the headline case, whole fit: the decisions eight GPUs would take
  python3 tools/ms_probe.py 1000000 300 5000 8 -1 1 2>&1 | grep -v amdgpu.ids > $out/ms_8x125k.log
  python3 tools/ms_probe.py 1000000 300 3000 8 16 1 2>&1 | grep -v amdgpu.ids >> $out/ms_8x125k.log
  tail -8 $out/ms_8x125k.log | cut -c1-250
  rocprofv3 --kernel-trace --output-format csv -d $out/trace_ms -- python3 tools/ms_probe.py 125000 300 3000 1 16 2 > /dev/null 2>&1
  f=$(find $out/trace_ms -name "*kernel_trace.csv" | head -1)
  python3 tools/summarize_profile.py trace $f $out/kernel_summary_ms125k.csv
  rm -rf $out/trace_ms
  tail -2 $out/ms_125k_untraced.log; tail -2 $out/ms_2x125k_untraced.log; head -8 $out/kernel_summary_ms125k.csv | cut -c1-110
fi
if has sweep; then
  C="1,0,16,1,1,2;1,0,16,1,1,0;1,1,-1,1,1,2;1,1,-1,1,1,0"
  LAB_REPS=3 LAB_CONFIGS=$C python3 tools/lab_pass.py --centers 3000 2>&1 | grep -v amdgpu.ids > $out/sweep_ab_1m.log
  LAB_REPS=3 LAB_CONFIGS=$C python3 tools/lab_pass.py --n 125000 --centers 3000 2>&1 | grep -v amdgpu.ids > $out/sweep_ab_125k.log
  cut -c1-190 $out/sweep_ab_1m.log $out/sweep_ab_125k.log
fi
if has msm; then
  python3 tools/msm_probe.py 2>&1 | grep -v amdgpu.ids > $out/msm_probe.log; cat $out/msm_probe.log
  python3 tools/eig_probe.py 1000 2>&1 | grep -v amdgpu.ids > $out/eig_probe.log; cat $out/eig_probe.log
fi
if has probe; then
  ./tools/probes/split_probe 2>&1 | grep -v amdgpu.ids > $out/split_probe.log; cat $out/split_probe.log
fi
if has driver; then
  # the driver's own command (the kernel sources' hash must match profiles/traffic.json)
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_command.json 2> $out/bench_driver_command.err
  python3 -c "import json; d = json.load(open('$out/bench_driver_command.json')); print('driver command', d['value'], d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline_one_center'].get('traffic'))"
fi
if has fuzz; then
  mkdir -p $out/fuzz
  python3 tools/fuzz_ms.py 400 61 2>&1 | grep -v amdgpu.ids > $out/fuzz/fuzz_ms_400_61.log; tail -1 $out/fuzz/fuzz_ms_400_61.log
  FUZZ_ATOMS=300,500 FUZZ_SHARDS_MIN=3 python3 tools/fuzz_ms.py 150 50 2>&1 | grep -v amdgpu.ids > $out/fuzz/fuzz_ms_150_50_300_500_atoms.log; tail -1 $out/fuzz/fuzz_ms_150_50_300_500_atoms.log
  EK_POISON=1 python3 tools/fuzz_ms.py 400 63 2>&1 | grep -v amdgpu.ids > $out/fuzz/fuzz_ms_400_63_poisoned.log; tail -1 $out/fuzz/fuzz_ms_400_63_poisoned.log
  python3 tools/fuzz_gpu.py 2000 31 2>&1 | grep -v amdgpu.ids > $out/fuzz/fuzz_gpu_2000_31.log; tail -1 $out/fuzz/fuzz_gpu_2000_31.log
  python3 tools/fuzz_gpu2.py 800 32 2>&1 | grep -v amdgpu.ids > $out/fuzz/fuzz_gpu2_800_32.log; tail -1 $out/fuzz/fuzz_gpu2_800_32.log
fi
if has c4; then
  bash tools/jobs/r6_c4.sh ${out#gpurun_out/}/c4
fi
