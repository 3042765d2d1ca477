"""Condense rocprofv3 output directories (merged back under gpurun_out/) into the tracked files of profiles/.

usage: make_profiles.py TAG FRAMES KERNEL_TRACE_DIR FETCH_PMC_DIR WRITE_PMC_DIR [VOLUME]
       (FRAMES = 0: the frames are counted in the trace, one tracking-finish launch per frame)

Writes profiles/TAG_kernel_stats.csv (rocprofv3's own per-kernel statistics, verbatim),
profiles/TAG_kernel_summary.txt, profiles/TAG_pmc_per_kernel.json (mean counter value per launch and kernel)
and profiles/integrate_traffic.json (HBM bytes per k_integrate launch, which bench.py reports as
roofline.traffic).  On gfx950 FETCH_SIZE counts 64-byte units for requests that move 128 bytes, so the
byte figure is 2 x FETCH_SIZE KiB; WRITE_SIZE is exact.  Both corrections were checked against a launch
whose traffic is known (tools/dense_integrate.py: every voxel read and written once).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.split("(")[0].replace("void ", "").replace("kfh::", "")


def frames_in_trace(rows):
    """Frames a kernel trace holds: every frame ends its tracking with one k_icp_finish launch (or one k_track_skip
    when tracking is not due), whatever else the traced program ran."""
    return sum(int(r["Calls"]) for r in rows if short(r["Name"]) in ("k_icp_finish", "k_track_skip"))


def kernel_summary(d, frames):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    if frames <= 0:
        frames = max(1, frames_in_trace(rows))
    out = ["%-28s %7s %10s %10s %6s %9s %9s" % ("kernel", "calls", "total_ms", "avg_us", "%", "min_us", "max_us")]
    for r in rows:
        out.append("%-28s %7s %10.2f %10.1f %6.1f %9.1f %9.1f" % (
            short(r["Name"])[-28:], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
            100 * int(r["TotalDurationNs"]) / tot, int(r["MinNs"]) / 1e3, int(r["MaxNs"]) / 1e3))
    upload = sum(int(r["TotalDurationNs"]) for r in rows if short(r["Name"]) in ("k_stage_copy", "__amd_rocclr_copyBuffer"))
    out.append("total kernel time %.2f ms; %.1f us per frame over %d frames (%.1f us without the upload kernels of "
               "the host-frame leg)" % (tot / 1e6, tot / frames / 1e3, frames, (tot - upload) / frames / 1e3))
    return f, "\n".join(out) + "\n"


def pmc_means(d, skip=4):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, cs in agg.items():
        res[k] = {c: {"mean": float(np.mean(v[skip:] if len(v) > 2 * skip else v)), "launches": len(v)}
                  for c, v in cs.items()}
    return res


def dense(tag, base, volume=512):
    """profiles/TAG_dense_*: kernel summary of tools/dense_leg.py under the profiler and the counter bytes of its
    k_integrate launches next to the 8 B x V^3 they must move."""
    prof = os.path.join(ROOT, "profiles")
    stats_csv, text = kernel_summary(os.path.join(base, "dense_kt"), 0)
    with open(os.path.join(prof, tag + "_dense_kernel_summary.txt"), "w") as f:
        f.write(text)
    pmc = {}
    for d in (os.path.join(base, "dense_fetch"), os.path.join(base, "dense_write")):
        for k, cs in pmc_means(d).items():
            pmc.setdefault(k, {}).update(cs)
    name = sorted(k for k in pmc if k.startswith("k_integrate") and "plan" not in k)[-1]
    ki = pmc[name]
    fetch_kib, write_kib = ki["FETCH_SIZE"]["mean"], ki["WRITE_SIZE"]["mean"]
    with open(os.path.join(base, "dense_under_trace.json")) as f:
        leg = json.loads(f.read().strip().splitlines()[-1])
    out = {
        "volume": volume, "kernel": name, "leg_under_kernel_trace": leg,
        "FETCH_SIZE_KiB_per_launch": fetch_kib, "WRITE_SIZE_KiB_per_launch": write_kib,
        "correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE under-counts 128 B requests by half)",
        "hbm_bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0,
        "algorithmic_bytes_per_launch": 8.0 * volume ** 3,
    }
    out["traffic_over_algorithmic"] = out["hbm_bytes_per_launch"] / out["algorithmic_bytes_per_launch"]
    with open(os.path.join(prof, tag + "_dense_integrate.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(text)
    print(json.dumps({k: v for k, v in out.items() if k != "leg_under_kernel_trace"}, indent=1))


def gather(tag, base):
    """profiles/TAG_gather_efficiency.json: per kernel, what the lanes asked of the L1, what the L1 asked of the L2 and
    what came from beyond the L2 (the latter from TAG_pmc_per_kernel.json), plus the march's sample statistics."""
    prof = os.path.join(ROOT, "profiles")
    pmc = {}
    for d in (os.path.join(base, "sq"), os.path.join(base, "tcp")):
        for k, cs in pmc_means(d).items():
            pmc.setdefault(k, {}).update(cs)
    hbm = {}
    try:
        with open(os.path.join(prof, tag + "_pmc_per_kernel.json")) as f:
            hbm = json.load(f)
    except OSError:
        pass
    out = {}
    for k, cs in sorted(pmc.items()):
        if not any(x in k for x in ("raycast", "integrate", "icp_iteration", "preprocess")) or "TCP_TOTAL_CACHE_ACCESSES_sum" not in cs:
            continue
        acc, req = cs["TCP_TOTAL_CACHE_ACCESSES_sum"]["mean"], cs["TCP_TCC_READ_REQ_sum"]["mean"]
        rec = {"vector_load_instructions": cs.get("SQ_INSTS_VMEM_RD", {}).get("mean"),
               "valu_instructions": cs.get("SQ_INSTS_VALU", {}).get("mean"),
               "l1_line_accesses": acc, "l1_to_l2_read_requests": req,
               "l1_hit_rate": 1.0 - req / acc if acc else None, "l2_to_l1_bytes": req * 64.0}
        h = hbm.get(k, {})
        if "FETCH_SIZE" in h:
            rec["beyond_l2_fetch_bytes"] = 2.0 * h["FETCH_SIZE"]["mean"] * 1024.0
            rec["beyond_l2_write_bytes"] = h["WRITE_SIZE"]["mean"] * 1024.0
        out[k] = rec
    try:
        with open(os.path.join(base, "raycast_stats.txt")) as f:
            out["raycast_march_sample_statistics"] = [line.rstrip() for line in f if line.strip()]
    except OSError:
        pass
    out["note"] = ("means per launch on the benchmark stream (rocprofv3 --pmc, two passes of bench.py --steps 60); "
                   "l2_to_l1_bytes = 64 B x L1-to-L2 read requests; beyond_l2_* from the FETCH_SIZE / WRITE_SIZE passes "
                   "(FETCH_SIZE doubled per the gfx950 correction; it does not tell infinity-cache hits from HBM reads)")
    with open(os.path.join(prof, tag + "_gather_efficiency.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


def main():
    if sys.argv[1] == "--gather":
        return gather(sys.argv[2], sys.argv[3])
    if sys.argv[1] == "--dense":
        return dense(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 512)
    tag, frames, kt, fd, wd = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    volume = int(sys.argv[6]) if len(sys.argv) > 6 else 512
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    stats_csv, text = kernel_summary(kt, frames)
    shutil.copy(stats_csv, os.path.join(prof, tag + "_kernel_stats.csv"))
    with open(os.path.join(prof, tag + "_kernel_summary.txt"), "w") as f:
        f.write(text)
    pmc = {}
    for d in (fd, wd):
        for k, cs in pmc_means(d).items():
            pmc.setdefault(k, {}).update(cs)
    with open(os.path.join(prof, tag + "_pmc_per_kernel.json"), "w") as f:
        json.dump(pmc, f, indent=1, sort_keys=True)
    # the work launch of the integration (a template since the box groups: k_integrate<1>), not its plan launch
    name = sorted(k for k in pmc if k.startswith("k_integrate") and "plan" not in k)[-1]
    ki = pmc[name]
    fetch_kib, write_kib = ki["FETCH_SIZE"]["mean"], ki["WRITE_SIZE"]["mean"]
    traffic = {
        "volume": volume,
        "kernel": name,
        "source": tag + "_pmc_per_kernel.json",
        "FETCH_SIZE_KiB_per_launch": fetch_kib,
        "WRITE_SIZE_KiB_per_launch": write_kib,
        "correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE under-counts 128 B requests by half)",
        "hbm_bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0,
        "algorithmic_bytes_per_launch": 8.0 * volume ** 3,
    }
    with open(os.path.join(prof, "integrate_traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
    print(text)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
