"""Condense rocprofv3 output under gpurun_out/ into the small files kept in
profiles/: per-kernel time summary (dispatches that returned immediately --
rounds enqueued past the goal -- listed separately) and per-kernel means of a
PMC counter.
  summarize_profile.py trace <kernel_trace.csv> <out.csv>
  summarize_profile.py pmc <counter_collection.csv> [...] <out.csv>
  summarize_profile.py window <kernel_trace.csv> <out.txt> [marker kernel]
      timeline (start offset, duration, gap to the previous end; us) of the
      dispatches between two launches of the marker kernel in mid-run"""
import collections
import csv
import sys


def short(name):
    return name.split("(")[0].replace("void ", "")[:64]


def trace(path, out):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(list)
    for r in rows:
        agg[short(r["Kernel_Name"])].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    with open(out, "w") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "calls", "real_calls", "real_avg_us",
                    "real_total_us", "noop_calls", "noop_avg_us"])
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            big = max(v)
            # a dispatch that found nothing to do returns in a few us
            noop = [t for t in v if big > 50 and t < 0.05 * big]
            real = [t for t in v if t not in noop] if noop else v
            real = [t for t in v if not (big > 50 and t < 0.05 * big)]
            w.writerow([k, len(v), len(real),
                        "%.2f" % (sum(real) / max(len(real), 1)),
                        "%.1f" % sum(real), len(noop),
                        "%.2f" % (sum(noop) / len(noop)) if noop else ""])


def pmc(paths, out):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in paths:
        for r in csv.DictReader(open(path)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(
                float(r["Counter_Value"]))
    with open(out, "w") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "counter", "dispatches", "mean_all",
                    "mean_excluding_noop_dispatches"])
        for k in sorted(agg):
            for c, v in sorted(agg[k].items()):
                big = max(v)
                real = [t for t in v if t >= 0.05 * big] or v
                w.writerow([k, c, len(v), "%.3f" % (sum(v) / len(v)),
                            "%.3f" % (sum(real) / len(real))])


def window(path, out, marker="ek_count_members_multi_kernel"):
    rows = sorted(csv.DictReader(open(path)),
                  key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    if len(marks) < 3:
        open(out, "w").write("marker %s: %d launches\n" % (marker, len(marks)))
        return
    i0, i1 = marks[len(marks) // 2], marks[len(marks) // 2 + 1]
    t0 = int(rows[i0]["Start_Timestamp"])
    prev_end = t0
    busy = 0.0
    with open(out, "w") as fh:
        fh.write("%10s %9s %9s  kernel\n" % ("start_us", "dur_us", "gap_us"))
        for r in rows[i0:i1]:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            fh.write("%10.1f %9.2f %9.2f  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3,
                                                  (s - prev_end) / 1e3,
                                                  short(r["Kernel_Name"])))
            busy += (e - s) / 1e3
            prev_end = e
        span = (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3
        fh.write("span %.1f us, kernels busy %.1f us (%d dispatches)\n"
                 % (span, busy, i1 - i0))


if __name__ == "__main__":
    if sys.argv[1] == "trace":
        trace(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "window":
        window(*sys.argv[2:5])
    else:
        pmc(sys.argv[2:-1], sys.argv[-1])
