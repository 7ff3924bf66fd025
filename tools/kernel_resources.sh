#!/bin/bash
# registers / spills / occupancy of the kernels of one source file:
#   tools/kernel_resources.sh ek_spec.hip [name filter]
f=$1; pat=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
  -fno-slp-vectorize -c "$(dirname "$0")/../enspara_amd/csrc/$f" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys, re, subprocess
cur = None; rows = []
for l in sys.stdin:
    m = re.search(r"remark: +(Function Name|TotalSGPRs|VGPRs|SGPRs Spill|VGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", l)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None: cur[k] = v
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n)
    print("%-48s vgpr %4s sgpr %4s spill v%s s%s occ %s lds %s" % (n[:48], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
' | grep -E "$pat"
