#!/bin/bash
out=gpurun_out/${1:-r5_p16c}
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export ENSPARA_HIP_LIB=$GRAFT_REPO_ROOT/enspara_amd/_variants/libnosolve.so
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --pam-sweeps 1 --no-msm --steps 4 --warmup 0 --cpu-seconds 1 > $out/bench_under_rocprof.json 2> $out/trace.err
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 tools/summarize_profile.py trace $f $out/kernel_summary.csv
rm -rf $out/trace
grep -E "pairs" $out/kernel_summary.csv | cut -c1-110
tail -3 $out/trace.err
