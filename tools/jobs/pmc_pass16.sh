#!/bin/bash
# SQ counters of the 16-candidate MFMA pass (rocprofv3 --pmc, no tracing), 1M and 125k frames
out=gpurun_out/${1:-pmc16}
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for n in 1000000 125000; do
  B="python3 tools/prof_spec.py $n 300 320 16"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --output-format csv -d $out/pmc_a_$n -- $B > $out/a_$n.log 2> $out/a_$n.err
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_b_$n -- $B > $out/b_$n.log 2> $out/b_$n.err
  for d in a b; do
    f=$(find $out/pmc_${d}_$n -name "*counter_collection.csv" | head -1)
    python3 tools/summarize_profile.py pmc $f $out/sum_${d}_$n.csv
  done
  rm -rf $out/pmc_a_$n $out/pmc_b_$n
  grep -E "pass16" $out/sum_a_$n.csv $out/sum_b_$n.csv
done
tail -2 $out/a_125000.err
