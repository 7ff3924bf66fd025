#!/bin/bash
# SQ / SQC counters of the multi-candidate pass kernel and of the one-center
# step kernel (rocprofv3 --pmc, one pass per counter set, no tracing)
out=gpurun_out/${1:-pmc}
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python3 bench.py --centers 400 --steps 2 --warmup 0 --candidates 8 --no-cpu-baseline --pam-sweeps 0"
B1="python3 bench.py --centers 100 --steps 2 --warmup 0 --candidates 1 --no-cpu-baseline --pam-sweeps 0"
run() {  # name, counters..., then the command after --
    local name=$1; shift
    rocprofv3 --pmc "$@" > $out/$name.json 2> $out/$name.err
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $out/pmc_a -- $B
run b SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_b -- $B
run c SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SALU --output-format csv -d $out/pmc_c -- $B
run s SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $out/pmc_s -- $B1
for d in a b c s; do
    f=$(find $out/pmc_$d -name "*counter_collection.csv" | head -1)
    python3 tools/summarize_profile.py pmc $f $out/sum_$d.csv
done
rm -rf $out/pmc_*
grep -E "pass2|kernel,counter" $out/sum_a.csv
grep -E "pass2" $out/sum_b.csv $out/sum_c.csv
grep step $out/sum_s.csv
tail -3 $out/a.err
