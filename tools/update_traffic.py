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

KERNELS = {"cands16": "ek_pass16_kernel<true>",
           "cands8": "ek_pass2_kernel<8, true, true>",
           "cands1": "ek_step_kernel<2, 0, true>"}


def main(summary, committed_as):
    rows = {(r["kernel"], r["counter"]): float(r["mean_excluding_noop_dispatches"])
            for r in csv.DictReader(open(summary))}
    path = os.path.join(ROOT, "profiles", "traffic.json")
    t = json.load(open(path))
    h = bench.kernel_source_hash()
    for key, kern in KERNELS.items():
        if (kern, "FETCH_SIZE") not in rows:
            print("no rows for", kern)
            continue
        fetch, write = rows[(kern, "FETCH_SIZE")], rows[(kern, "WRITE_SIZE")]
        e = t[key]
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
