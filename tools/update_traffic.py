"""profiles/traffic.json from a PMC summary (tools/summarize_profile.py pmc):
HBM bytes per launch of the two distance kernels, tagged with the hash of the
kernel sources in this tree (bench.py quotes the figure only while they match).

  update_traffic.py <pmc_summary.csv> <path the summary is committed under>
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

# (cands32: a round of 32 is the two launches <true, 1> + <true, 2>; the entry holds
# their MEAN, the per-launch figure bench.py quotes for that form)
KERNELS = {"cands16": ["ek_pass16_kernel<true, 0>"],
           "cands32": ["ek_pass16_kernel<true, 1>", "ek_pass16_kernel<true, 2>"],
           "cands8": ["ek_pass2_kernel<8, true, true>"],
           "cands1": ["ek_step_kernel<2, 0, true>"]}


def main(summary, committed_as):
    rows = {(r["kernel"], r["counter"]): float(r["mean_excluding_noop_dispatches"])
            for r in csv.DictReader(open(summary))}
    path = os.path.join(ROOT, "profiles", "traffic.json")
    t = json.load(open(path))
    h = bench.kernel_source_hash()
    for key, kerns in KERNELS.items():
        if any((kern, "FETCH_SIZE") not in rows for kern in kerns):
            print("no rows for", kerns)
            continue
        fetch = sum(rows[(kern, "FETCH_SIZE")] for kern in kerns) / len(kerns)
        write = sum(rows[(kern, "WRITE_SIZE")] for kern in kerns) / len(kerns)
        e = t.setdefault(key, {"frames": 1000000, "atoms": 300, "kernel": " + ".join(kerns),
                               "correction": t["cands16"]["correction"],
                               "algorithmic_bytes_per_launch":
                                   1000000 * ((12 * 300 + 20 + 60) + (12 * 300 + 12 + 64)) / 2})
        e["FETCH_SIZE_raw_KB_per_launch"] = fetch
        e["WRITE_SIZE_raw_KB_per_launch"] = write
        e["hbm_read_bytes_per_launch"] = fetch * 1024 * 2     # gfx950: see "correction"
        e["hbm_write_bytes_per_launch"] = write * 1024
        e["hbm_bytes_per_launch"] = fetch * 1024 * 2 + write * 1024
        e["profile"] = committed_as
        e["kernel_source_sha256_16"] = h
        print(key, "%.4e B per launch (algorithmic %.4e)"
              % (e["hbm_bytes_per_launch"], e["algorithmic_bytes_per_launch"]))
    json.dump(t, open(path, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
