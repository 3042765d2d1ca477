"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel totals and per-frame cost.
usage: prof_summary.py DIR [FRAMES]   (FRAMES omitted or 0: counted in the trace, one tracking-finish launch per frame)"""
import csv
import glob
import sys

d = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 0
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
if frames <= 0:      # every frame ends its tracking with one k_icp_finish (or k_track_skip) launch
    frames = max(1, sum(int(r["Calls"]) for r in rows
                        if r["Name"].split("(")[0].split("::")[-1] in ("k_icp_finish", "k_track_skip")))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("%-28s %7s %10s %10s %6s %9s %9s" % ("kernel", "calls", "total_ms", "avg_us", "%", "min_us", "max_us"))
for r in rows:
    name = r["Name"].split("(")[0].replace("void ", "")[-28:]
    print("%-28s %7s %10.2f %10.1f %6.1f %9.1f %9.1f" % (
        name, r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
        100 * int(r["TotalDurationNs"]) / tot, int(r["MinNs"]) / 1e3, int(r["MaxNs"]) / 1e3))
print("total kernel time %.2f ms; %.1f us per frame over %d frames" % (tot / 1e6, tot / frames / 1e3, frames))
