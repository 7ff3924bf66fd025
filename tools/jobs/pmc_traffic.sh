#!/bin/bash
# the two PMC passes behind profiles/traffic.json only (see profile_round.sh)
# usage: pmc_traffic.sh <outdir-under-gpurun_out>
out=gpurun_out/$1
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py --centers 1000 --steps 4 --warmup 0 --no-cpu-baseline --pam-sweeps 0 > $out/bench_under_pmc_$c.json 2> $out/pmc_$c.err
done
python3 tools/summarize_profile.py pmc $(find $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE -name "*counter_collection.csv") $out/pmc_summary.csv
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
grep -E "pass2|step_kernel" $out/pmc_summary.csv
