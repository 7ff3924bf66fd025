#!/bin/bash
# Round 5: the numbers DESIGN.md / profiles/r05 quote, from one box and HEAD.
# usage: profile_r05.sh <outdir-under-gpurun_out> [part ...]   parts: bench trace pmc c3 data ms pam msm
out=gpurun_out/$1; shift
parts=${@:-tests bench full trace pmc c3 data ms pam ti msm}
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
has() { [[ " $parts " == *" $1 "* ]]; }
if has tests; then
  timeout 2400 python3 -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1
  tail -3 $out/gpu_tests.log
fi
if has bench; then
  python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
  python3 bench.py --candidates 32 --no-cpu-baseline --pam-sweeps 0 --no-msm > $out/bench_candidates32.json 2> /dev/null
  python3 bench.py --candidates 8 --no-cpu-baseline --pam-sweeps 0 --no-msm > $out/bench_candidates8.json 2> /dev/null
  python3 bench.py --candidates 1 --no-cpu-baseline --pam-sweeps 0 --no-msm > $out/bench_candidates1.json 2> /dev/null
  python3 bench.py --sharded --no-cpu-baseline --pam-sweeps 0 --no-msm > $out/bench_sharded_1rank_mailbox.json 2> /dev/null
  python3 bench.py --sharded --transport gather --no-cpu-baseline --pam-sweeps 0 --no-msm > $out/bench_sharded_1rank_gather.json 2> /dev/null
  cut -c1-300 $out/bench_default.json
fi
if has full; then
  # the oracle replays ALL 5000 centers (labels and distances compared) and the
  # complete PAM sweep: minutes of CPU work
  python3 bench.py --cpu-seconds 0 > $out/bench_full_parity.json 2> $out/bench_full_parity.err
  python3 -c "
import json; d=json.load(open('$out/bench_full_parity.json')); c=d['cpu_baseline']; k=d['khybrid']['parity']
print('full parity:', c['centers_checked'], c['centers_match_gpu'], c['whole_fit_state_vs_gpu'], k['proposals_replayed_by_oracle'], k['medoids_match_gpu'], k['whole_sweep_state_vs_oracle'])"
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
  for T in 16 8 32; do
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $c --output-format csv -d $out/pmc_${T}_$c -- python3 bench.py --centers 1000 --steps 4 --warmup 0 --candidates $T --no-cpu-baseline --pam-sweeps 0 --no-msm > $out/bench_under_pmc_${T}_$c.json 2> $out/pmc_${T}_$c.err
    done
  done
  python3 tools/summarize_profile.py pmc $(find $out/pmc_* -name "*counter_collection.csv") $out/pmc_summary.csv
  rm -rf $out/pmc_16_* $out/pmc_8_* $out/pmc_32_*
  grep -E "pass16|pass2|step_kernel" $out/pmc_summary.csv
fi
if has sq; then
  # SQ counters of the 16-candidate pass (instructions per wave, matrix pipe busy)
  B="python3 tools/prof_spec.py 1000000 300 320 16"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --output-format csv -d $out/sq_a -- $B > /dev/null 2> $out/sq_a.err
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $out/sq_b -- $B > /dev/null 2> $out/sq_b.err
  for d in a b; do
    f=$(find $out/sq_$d -name "*counter_collection.csv" | head -1)
    python3 tools/summarize_profile.py pmc $f $out/pass16_pmc_sq_$d.csv
  done
  rm -rf $out/sq_a $out/sq_b
  grep -E "pass16" $out/pass16_pmc_sq_a.csv $out/pass16_pmc_sq_b.csv
fi
if has c3; then
  # BASELINE.json configs[3], one GPU's share: 1.25 M frames x 500 atoms, 20 000 centers
  C3="--frames 1250000 --atoms 500 --templates 20000 --centers 20000 --no-cpu-baseline --pam-sweeps 0 --no-msm"
  python3 bench.py $C3 > $out/bench_config3_shard.json 2> $out/bench_config3_shard.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace3 -- python3 bench.py $C3 --centers 4000 > $out/bench_config3_under_rocprof.json 2> /dev/null
  f=$(find $out/trace3 -name "*kernel_trace.csv" | head -1)
  python3 tools/summarize_profile.py trace $f $out/kernel_summary_config3.csv
  rm -rf $out/trace3
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $out/pmc3_$c -- python3 bench.py $C3 --centers 600 --steps 4 --warmup 0 --candidates 16 > /dev/null 2> $out/pmc3_$c.err
  done
  python3 tools/summarize_profile.py pmc $(find $out/pmc3_* -name "*counter_collection.csv") $out/pmc_summary_config3.csv
  rm -rf $out/pmc3_*
  head -8 $out/kernel_summary_config3.csv | cut -c1-110; grep -E "pass16" $out/pmc_summary_config3.csv
  cut -c1-300 $out/bench_config3_shard.json
fi
if has data; then
  # the headline off the friendly data: 10 centers per template, and a time-ordered walk
  python3 bench.py --templates 500 --no-cpu-baseline --pam-sweeps 0 --no-msm > $out/bench_templates500.json 2> /dev/null
  python3 bench.py --data walk --no-cpu-baseline --pam-sweeps 0 --no-msm > $out/bench_walk.json 2> /dev/null
  python3 bench.py --data walk --triangle 1 --no-cpu-baseline --pam-sweeps 0 --no-msm > $out/bench_walk_triangle.json 2> /dev/null
  python3 bench.py --triangle 1 --no-cpu-baseline --pam-sweeps 0 --no-msm > $out/bench_default_triangle.json 2> /dev/null
  for f in templates500 walk walk_triangle default_triangle; do python3 -c "
import json; d=json.load(open('$out/bench_$f.json')); print('$f', d['value'], d['passes_over_frames'], d['centers_per_pass'], d['config']['passes_by_candidates'])"; done
fi
if has ms; then
  bash tools/jobs/trace_ms.sh ${out#gpurun_out/}/ms_125k 125000 300 3000 1 16 > $out/ms_125k.log 2>&1
  python3 tools/ms_probe.py 125000 300 3000 1 16 3 2>&1 | grep -v amdgpu.ids > $out/ms_125k_untraced.log
  python3 tools/ms_probe.py 125000 300 3000 1 -1 3 2>&1 | grep -v amdgpu.ids > $out/ms_125k_ladder_untraced.log
  python3 tools/ms_probe.py 125000 300 3000 1 32 3 2>&1 | grep -v amdgpu.ids > $out/ms_125k_rounds32_untraced.log
  python3 tools/ms_probe.py 250000 300 3000 2 16 2 2>&1 | grep -v amdgpu.ids > $out/ms_2x125k_untraced.log
  tail -4 $out/ms_125k_untraced.log; tail -3 $out/ms_2x125k_untraced.log
fi
if has pam; then
  LAB_PAM_OPTS="16=1;16=0" python3 tools/lab_pam.py --reps 2 2>&1 | grep -v amdgpu.ids > $out/pam_sweep_1m.log; cat $out/pam_sweep_1m.log
  # every slot of a window evaluated ahead of its turn (key 19) against round 4's form
  LAB_PAM_OPTS="19=1;19=0" python3 tools/lab_pam.py --reps 3 2>&1 | grep -v amdgpu.ids > $out/pam_ahead_ab_1m.log; cat $out/pam_ahead_ab_1m.log
  LAB_PAM_OPTS="19=1;19=0" python3 tools/lab_pam.py --reps 2 --n 200000 --centers 2000 --walk 1 2>&1 | grep -v amdgpu.ids > $out/pam_ahead_ab_walk200k.log; cat $out/pam_ahead_ab_walk200k.log
fi
if has ti; then
  python3 tools/ti_probe.py 2000 500 300 3000 2>&1 | grep -v amdgpu.ids > $out/ti_probe.log; cat $out/ti_probe.log
fi
if has up; then
  python3 tools/upload_probe.py 2>&1 | grep -v amdgpu.ids > $out/upload_probe.log; cat $out/upload_probe.log
fi
if has msm; then
  python3 tools/msm_probe.py 2>&1 | grep -v amdgpu.ids > $out/msm_probe.log; cat $out/msm_probe.log
  python3 tools/eig_probe.py 1000 2>&1 | grep -v amdgpu.ids > $out/eig_probe.log; cat $out/eig_probe.log
fi
