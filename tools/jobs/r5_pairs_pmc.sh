#!/bin/bash
# SQ counters of the PAM window's two distance kernels (what do they wait for?)
out=gpurun_out/${1:-r5_pairs_pmc}
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python3 bench.py --no-cpu-baseline --pam-sweeps 1 --no-msm --steps 4 --warmup 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $out/sq_a -- $B > /dev/null 2> $out/sq_a.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE --output-format csv -d $out/sq_b -- $B > /dev/null 2> $out/sq_b.err
for d in a b; do
  f=$(find $out/sq_$d -name "*counter_collection.csv" | head -1)
  python3 tools/summarize_profile.py pmc $f $out/pam_pmc_sq_$d.csv
done
rm -rf $out/sq_a $out/sq_b
grep -E "pairs_kernel|sp_window|sp_spec|pam_active" $out/pam_pmc_sq_a.csv $out/pam_pmc_sq_b.csv
