#!/bin/bash
out=gpurun_out/${1:-r5_gaps}
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --pam-sweeps 1 --no-msm --steps 4 --warmup 0 > $out/bench.json 2> $out/trace.err
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee $out/gaps.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows))
# PAM part: from the first ek_sp_window_kernel on
names = [e[2] for e in ev]
first = next(i for i, n in enumerate(names) if "ek_sp_spec_kernel" in n)
gap_after = collections.defaultdict(list)
for i in range(first, len(ev) - 1):
    gap_after[ev[i][2][:40]].append((ev[i + 1][0] - ev[i][1]) / 1000.0)
print("gap after kernel (us): mean over the sweeps' windows")
tot = 0
for k, v in sorted(gap_after.items(), key=lambda kv: -sum(kv[1])):
    if len(v) > 100:
        print("%-42s n=%4d mean %6.2f" % (k, len(v), sum(v) / len(v)))
        tot += sum(v) / 314.0
print("sum of mean gaps per window ~ %.1f us" % tot)
PY
rm -rf $out/trace
