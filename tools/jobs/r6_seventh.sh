#!/bin/bash
# Round 6, seventh call: sweep (common-case shortcut) A/B; SQ counters of the split probe;
# kernel trace of the MSM block
out=gpurun_out/$1; shift
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
C="1,0,16,1,1,2;1,0,16,1,1,0;1,0,16,1,1,2;1,0,16,1,1,0"
LAB_REPS=3 LAB_CONFIGS=$C python3 tools/lab_pass.py --centers 3000 2>&1 | grep -v amdgpu.ids > $out/sweep_ab_1m.log; cut -c1-200 $out/sweep_ab_1m.log
timeout 900 python3 -m pytest tests/test_gpu_kcenters.py -q -m gpu -x -k "per_prefix or candidates_per_pass or triangle" > $out/tests.log 2>&1; tail -3 $out/tests.log
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_split -- ./tools/probes/split_probe > $out/split_probe_under_pmc.log 2> $out/pmc_split.err
f=$(find $out/pmc_split -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' | tee $out/split_probe_pmc.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.OrderedDict()
for r in rows:
    by.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = ["SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]
print("dispatch " + " ".join("%26s" % n for n in names))
for d, c in by.items():
    print("%8s " % d + " ".join("%26.4g" % c.get(n, float("nan")) for n in names))
PY
rm -rf $out/pmc_split
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 bench.py --centers 200 --steps 1 --warmup 0 --no-cpu-baseline --pam-sweeps 0 > $out/bench_msm_trace.json 2> $out/trace.err
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 tools/summarize_profile.py trace $f $out/kernel_summary_msm.csv
grep -E "msm|kr_|rocclr" $out/kernel_summary_msm.csv | cut -c1-110
rm -rf $out/trace
