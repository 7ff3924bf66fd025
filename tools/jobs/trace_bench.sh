#!/bin/bash
# rocprofv3 kernel trace of a short bench run; condensed per-kernel table
# usage: trace_bench.sh <outdir-under-gpurun_out> [bench args...]
out=gpurun_out/$1; shift
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --pam-sweeps 0 "$@" > $out/bench.json 2> $out/bench.err
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 tools/summarize_profile.py trace $f $out/kernel_summary.csv
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv 2>/dev/null
rm -rf $out/trace
head -30 $out/kernel_summary.csv
