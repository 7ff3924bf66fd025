#!/bin/bash
# kernel trace of the multi-shard rounds on a one-shard proxy: trace_ms.sh <out> <n> <atoms> <centers> <shards> <cands>
out=gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/ms_probe.py $2 $3 $4 $5 $6 1 > $out/probe.log 2> $out/trace.err
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 tools/summarize_profile.py trace $f $out/kernel_summary.csv
python3 tools/summarize_profile.py window $f $out/window.txt ek_ms_chain_kernel
rm -rf $out/trace
grep -v amdgpu.ids $out/probe.log
head -14 $out/kernel_summary.csv | cut -c1-120
head -30 $out/window.txt
