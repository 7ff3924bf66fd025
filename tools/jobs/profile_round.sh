#!/bin/bash
# The numbers DESIGN.md / profiles/ quote, from one box and HEAD:
#   bench_default.json          python3 bench.py                       (the driver's form)
#   kernel_summary.csv          rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --pam-sweeps 1
#   pmc_fetch/write summaries   rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)
# usage: profile_round.sh <outdir-under-gpurun_out>
out=gpurun_out/$1
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --pam-sweeps 1 > $out/bench_under_rocprof.json 2> $out/trace.err
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 tools/summarize_profile.py trace $f $out/kernel_summary.csv
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv 2>/dev/null
rm -rf $out/trace
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py --centers 1000 --steps 4 --warmup 0 --no-cpu-baseline --pam-sweeps 0 > $out/bench_under_pmc_$c.json 2> $out/pmc_$c.err
done
python3 tools/summarize_profile.py pmc $(find $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE -name "*counter_collection.csv") $out/pmc_summary.csv
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
head -12 $out/kernel_summary.csv | cut -c1-110
grep -E "pass2|step_kernel" $out/pmc_summary.csv
cut -c1-400 $out/bench_default.json
# nearest-center assignment (predict), 10^6 x 5000: a rocprof row for the MFMA kernel
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_assign -- python3 tools/quick_bench2.py 1000000 300 5000 assign > $out/assign.log 2> $out/assign.err
f=$(find $out/trace_assign -name "*kernel_trace.csv" | head -1)
python3 tools/summarize_profile.py trace $f $out/kernel_summary_assign.csv
rm -rf $out/trace_assign
grep -E "assign" $out/kernel_summary_assign.csv | cut -c1-110
grep -E "assign n=" $out/assign.log
# one pass per center (BASELINE.md's roofline case) and the torch.distributed driver with one rank
python3 bench.py --candidates 1 --no-cpu-baseline --pam-sweeps 0 > $out/bench_candidates1.json 2> $out/bench_candidates1.err
python3 bench.py --sharded --no-cpu-baseline --pam-sweeps 1 > $out/bench_sharded_1rank.json 2> $out/bench_sharded_1rank.err
# one PAM sweep under the kernel trace (tools/lab_pam.py child): per-kernel table + a window's timeline
bash tools/jobs/trace_pam.sh $1/pam > /dev/null 2>&1
python3 tools/lab_pam.py --reps 3 2>&1 | grep -v amdgpu.ids | tail -4 > $out/pam/lab_untraced.log
head -16 $out/pam/kernel_summary.csv | cut -c1-110; cat $out/pam/lab_untraced.log
# MSM transition counts at scale
python3 tools/msm_probe.py 2>&1 | grep -v amdgpu.ids > $out/msm_probe.log; cat $out/msm_probe.log
