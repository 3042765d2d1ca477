"""Launch boundaries around the integration's two launches, from a rocprofv3 kernel trace.

usage: launch_gaps.py KERNEL_TRACE_DIR
Prints, for the plan launch and the work launch that follows it: the idle time before the plan launch, its
duration, the idle time between the two, and the work launch's duration (median, mean, 10th and 90th percentile
over all integrations in the trace).  The idle time between the two plus the fixed cost of one launch (the
duration of a launch that returns at once, printed last) is what a fused single launch could save at most.
"""
import csv
import glob
import sys

import numpy as np


def short(name):
    return name.split("(")[0].replace("void ", "").replace("kfh::", "")


def main():
    rows = []
    for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    before, plan, between, work = [], [], [], []
    for i in range(1, len(rows) - 1):
        if rows[i][2] == "k_integrate_plan" and rows[i + 1][2].startswith("k_integrate") and "plan" not in rows[i + 1][2]:
            before.append(rows[i][0] - rows[i - 1][1])
            plan.append(rows[i][1] - rows[i][0])
            between.append(rows[i + 1][0] - rows[i][1])
            work.append(rows[i + 1][1] - rows[i + 1][0])
    def line(name, v):
        v = np.array(v, float) / 1e3
        print("%-34s n=%d median %.2f mean %.2f p10 %.2f p90 %.2f us" % (name, len(v), np.median(v), v.mean(),
                                                                        np.percentile(v, 10), np.percentile(v, 90)))
    line("idle before the plan launch", before)
    line("plan launch", plan)
    line("idle between plan and work launch", between)
    line("work launch", work)
    skip = [e - s for s, e, n in rows if n in ("k_track_skip", "k_icp_finish")]
    if skip:
        line("a launch with next to no work (k_icp_finish / k_track_skip)", skip)
    allgaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
    line("idle between any two launches", [g for g in allgaps if 0 <= g < 20000])


if __name__ == "__main__":
    main()
