#!/bin/bash
# SQ counters of the PAM window kernels (ek_pam_pairs_kernel, ek_sp_window_kernel), 10^6 x 300, 5000 medoids
out=gpurun_out/${1:-pmc_pairs}
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python3 tools/sp_check.py --only-big"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_a -- $B > $out/a.log 2> $out/a.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD --output-format csv -d $out/pmc_b -- $B > $out/b.log 2> $out/b.err
for d in a b; do
  f=$(find $out/pmc_$d -name "*counter_collection.csv" | head -1)
  python3 tools/summarize_profile.py pmc $f $out/sum_$d.csv
done
rm -rf $out/pmc_a $out/pmc_b
grep -E "pairs|sp_window" $out/sum_a.csv $out/sum_b.csv
