#!/usr/bin/env python3
"""KFusion throughput benchmark on MI355X.

One "step" = one depth frame through the whole hot path (preprocessing -> tracking -> integration ->
raycasting), with the frame already resident in HBM.  Workload at N=1: BASELINE.json configs[1] —
640x480 depth, TUM fr1 depth intrinsics, 512^3 TSDF over 8 m, ICP pyramid 10/5/4, integration-rate 2 —
on the deterministic synthetic room stream (no dataset is available offline).  At N>1 every rank runs
its own sequence (the reference's design-space sweeps are one process per configuration,
framework/shared/src/SLAMBenchConfiguration.cpp:524-533), so scaling is weak and there is no data-path
collective; --mode zslab shards ONE sequence's volume over the ranks instead.

`--gpus N` with N > 1 started as a plain `python bench.py` (no WORLD_SIZE in the environment) launches the N
ranks itself through torch.distributed.run before anything touches the GPU; started under torch.distributed.run
it is one of the ranks.  The line reports the rank count the ranks counted among themselves, and the run fails
if that is not N.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (spec)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--volume", type=int, default=512, help="TSDF resolution per side")
    ap.add_argument("--csr", type=int, default=1, help="compute-size-ratio")
    ap.add_argument("--unique-frames", type=int, default=100,
                    help="rendered frames; the stream plays them forwards then backwards")
    ap.add_argument("--cpu-frames", type=int, default=16, help="frames of the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--cpu-frames-1t", type=int, default=9,
                    help="frames of the additional single-threaded CPU sample (0 = skip)")
    ap.add_argument("--sequences-per-gpu", type=int, default=1,
                    help="independent sequences (contexts, one stream each) interleaved on every GPU; the "
                         "default 1 is the benchmark, more measures how many sweep points one GPU can carry")
    ap.add_argument("--host-threads", action="store_true",
                    help="with --sequences-per-gpu S: one host thread per sequence instead of one thread that "
                         "interleaves them (the library calls release the interpreter lock)")
    ap.add_argument("--no-host-frames", action="store_true",
                    help="skip the second leg that hands every frame over in host memory (one H2D copy per frame; "
                         "reported as pcie_inclusive_frames_per_s, never as value)")
    ap.add_argument("--no-dense-leg", action="store_true",
                    help="skip the dense integration leg (roofline_dense: a volume whose every voxel is updated)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=["replicas", "zslab"], default="replicas",
                    help="N>1: one independent sequence per GPU (default), or ONE sequence whose volume is "
                         "sharded into z-slabs across the GPUs")
    ap.add_argument("--backend", default="auto",
                    help="torch.distributed backend: nccl (RCCL), gloo, or auto = nccl when every rank has a GPU of "
                         "its own, gloo when ranks share GPUs (a rehearsal on a smaller box)")
    ap.add_argument("--icp", choices=["replicated", "allreduce"], default="replicated",
                    help="--mode zslab: every rank tracks the whole image (no collective), or every rank tracks its "
                         "rows and the 32 ICP sums are all-reduced per iteration")
    ap.add_argument("--exchange", choices=["auto", "gather", "tiles"], default="auto",
                    help="--mode zslab: one all-gather of every rank's slab candidates (16 B per pixel and rank), or "
                         "an all-to-all that gives every rank the candidates of its 1/N of the image followed by an "
                         "all-gather of the selected shares (2/N of the bytes, two collectives); auto: the z-slab leg "
                         "behind the replicas times a few frames of each and keeps the faster one, --mode zslab takes "
                         "the all-gather")
    ap.add_argument("--force-exchange", action="store_true",
                    help="--mode zslab on ONE rank: still take the exchange path (slab march, all-gather, composite, "
                         "all-reduce) through a single-rank RCCL group, to time the collectives' fixed cost")
    ap.add_argument("--zslab-volume", type=int, default=2048,
                    help="N>1, replicas: TSDF resolution of the additional z-slab leg (BASELINE.json configs[3]: ONE "
                         "sequence whose 2048^3 volume is sharded over the ranks), reported under zslab_<volume>")
    ap.add_argument("--zslab-steps", type=int, default=100, help="timed frames of the additional z-slab leg")
    ap.add_argument("--no-zslab-leg", action="store_true", help="N>1, replicas: skip the additional z-slab leg")
    ap.add_argument("--no-group-leg", action="store_true",
                    help="N>1, replicas: skip the leg that runs the same z-slab configuration through the in-process "
                         "group (kfh_group) on rank 0")
    ap.add_argument("--no-icp-probe", action="store_true",
                    help="z-slab leg: do not run the extra frames with the other ICP mode (replicated <-> all-reduced sums)")
    ap.add_argument("--zslab-timeout", type=float, default=240.0,
                    help="seconds after which rank 0 prints the line without a z-slab result and exits with status 4 "
                         "(below --pg-timeout, which aborts the process)")
    ap.add_argument("--pg-timeout", type=float, default=300.0,
                    help="seconds after which a collective that does not complete ends the run (non-zero exit)")
    ap.add_argument("--dataset", default="auto",
                    help="compare the plugin with the reference's published kfusion-cpp log on a dataset file "
                         "(slambench_amd/refcheck.py): a file name the reference's logs know, 'auto' = every such "
                         "file found under KFH_DATASET_DIR or datasets/, 'none' = skip.  Reported as dataset_check")
    ap.add_argument("--track-stats-csv", default=os.path.join(ROOT, "profiles", "r3_kernel_stats.csv"),
                    help="rocprofv3 --kernel-trace --stats table of this command (tools/profile_round.sh): source of "
                         "the per-level launch durations behind roofline_track")
    ap.add_argument("--dense-traffic-json", default=os.path.join(ROOT, "profiles", "r3_dense_integrate.json"),
                    help="counter bytes of the dense integration leg (tools/make_profiles.py --dense)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "integrate_traffic.json"),
                    help="HBM bytes per integrate launch from the committed rocprofv3 PMC passes")
    return ap.parse_args()


def ping_pong(n_unique, count):
    """Frame order 0..n-1, n-2..1, 0.. — consecutive frames are always neighbours in time."""
    period = list(range(n_unique)) + list(range(n_unique - 2, 0, -1))
    return [period[i % len(period)] for i in range(count)]


def cpu_model():
    """First "model name" of /proc/cpuinfo, as the reference's log header prints it (ResultWriter.cpp:49-75)."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


CPU_STARTUP_FRAMES = 4     # a sequence's first frames neither track nor raycast: not part of the timed sample


def cpu_baseline(cfg, depth, frames, threads):
    """The oracle (a port of the kfusion C++ path) timed on this box's host cores: `frames` frames of the stream, of
    which the first CPU_STARTUP_FRAMES (no tracking, no raycast yet: cheaper than a steady frame) run untimed."""
    from oracle import kf_ref
    threads = kf_ref.set_threads(threads)
    ref = kf_ref.KFusionRef(cfg.to_c(kf_ref.Config))
    poses = []
    skip = min(CPU_STARTUP_FRAMES, frames - 1)
    phases = np.zeros(4)
    t0 = time.perf_counter()
    for i in range(frames):
        if i == skip:
            t0 = time.perf_counter()
        ref.process_frame(depth[i])
        if i >= skip:
            phases += ref.phase_seconds
        poses.append(ref.pose.copy())
    dt = time.perf_counter() - t0
    ref.close()
    return (frames - skip) / dt, np.stack(poses), phases / (frames - skip), threads


class stdout_to_stderr:
    """RCCL prints a version banner on stdout when its first communicator comes up; the benchmark's stdout must hold
    the one JSON line only, so the file descriptor is pointed at stderr while the process group starts."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks through torch.distributed.run and pass their
    output and exit code on.  Called before this process has imported torch or touched the GPU, so the ranks are
    fresh children and nothing is re-executed."""
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def dense_integrate_leg(KFusion, KFusionConfig, intrinsics, V, torch, band=False):
    """Integration of a volume whose EVERY voxel is updated: a cube that lies wholly inside the camera frustum, one
    metre in front of the camera, in front of a flat wall.  The first four frames of a sequence integrate
    unconditionally from the initial pose, so each of them reads and writes all V^3 voxels (8 B each: the weight
    changes every time); timed with HIP events on the pipeline's stream like the main leg.

    band=False: a 0.5 m cube with the wall at 3 m, far behind it: every voxel is free space (the cheapest update: the
    value stays 1, the weight moves).  band=True: the volume is a 0.17 m thick slab around a wall at 1.1 m, so every
    voxel lies inside the truncation band (|depth - z| < mu = 0.1 m) and takes the reference's full update: the ray
    scale with its square root and two divisions, and the running average with a third."""
    if band:
        size, direction, wall_mm = (0.5, 0.5, 0.17), (0.25, 0.25, -1.015), 1100
    else:
        size, direction, wall_mm = (0.5, 0.5, 0.5), (0.25, 0.25, -1.0), 3000
    cfg = KFusionConfig(input_size=(640, 480), intrinsics=intrinsics, volume_resolution=(V, V, V),
                        volume_size=size, volume_direction=direction)
    kf = KFusion(cfg)
    kf.set_timing(2)
    wall = torch.full((480, 640), wall_mm, dtype=torch.int16, device="cuda")
    ms, boxes = [], []
    for rep in range(4):
        if rep:
            kf.reset()
        b0, n0, vpb = kf.integrate_totals()
        for f in range(4):
            kf.process_frame(int(wall.data_ptr()))
        kf.sync()
        b1, n1, _ = kf.integrate_totals()
        assert n1 - n0 == 4, "the first four frames of a sequence must integrate"
        boxes.append((b1 - b0) / 4.0 * vpb)
        # the first integration of the first repetition also pays for code loading
        ms += [float(kf.phase_ms(f)[2]) for f in range(4) if rep or f]
    vox = float(np.mean(boxes))
    assert vox >= V ** 3, "the plan must keep every box of the dense volume (%g of %g voxels)" % (vox, V ** 3)
    from slambench_amd import native
    volume = kf.download(native.BUF_VOLUME)
    weights = volume[..., 1]
    assert int(weights.min()) == 4 and int(weights.max()) == 4, "every voxel must have been updated by every frame"
    if band:
        assert int(np.abs(volume[..., 0].astype(np.int32)).max()) < 32766, "every voxel must lie inside the truncation band"
    kf.close()
    t, tm = float(np.median(ms)) * 1e-3, float(np.mean(ms)) * 1e-3
    what = ("%d^3 voxels over a 0.5 x 0.5 x 0.17 m slab around a wall at 1.1 m: every voxel inside the truncation band "
            "(checked: every weight == number of frames, no value at +-1)" % V) if band else \
           ("%d^3 voxels over a 0.5 m cube inside the frustum, wall at 3 m: all voxels read and written "
            "(checked: every weight == number of frames)" % V)
    return {"kernel": "k_integrate_plan + k_integrate", "bound": "hbm", "achieved": 8.0 * V ** 3 / t / 1e9,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 8.0 * V ** 3 / t / 1e9 / HBM_PEAK_GBS,
            "frac_mean": 8.0 * V ** 3 / tm / 1e9 / HBM_PEAK_GBS, "traffic": None,
            "ms_per_launch": t * 1e3, "ms_per_launch_mean": tm * 1e3, "launches_timed": len(ms),
            "gvox_per_s": V ** 3 / t / 1e9,
            "note": "frac from the median, frac_mean from the mean of the HIP-event durations (plan + work launch) of "
                    "the timed integrations",
            "workload": what + ", 8 B x V^3 per launch"}


def track_roofline(stats_csv, compute_size, tracking_ms):
    """Track + reduce (one k_icp_iteration launch per ICP iteration) against the HBM roofline, per pyramid level:
    36 algorithmic bytes per pixel of the level (five 4-byte input depths the vertex and normal are rebuilt from, and
    the 16-byte packed model map entry the pixel gathers) / the launch's average duration.  The durations are those of
    a committed rocprofv3 kernel trace of this command (the pipeline has no per-launch events of its own);
    tracking_ms is this run's HIP-event time of the whole tracking phase on sampled frames."""
    import csv
    import re
    w, h = compute_size
    # block shapes the library picks per level for a 640 x 480 compute size (one instantiation per level)
    level_of_shape = {(7, 3): 0, (5, 1): 1, (3, 1): 2}
    out = {"kernel": "k_icp_iteration (trackKernel + reduceKernel + the previous iteration's solve)", "bound": "hbm",
           "peak": HBM_PEAK_GBS, "unit": "GB/s", "bytes_per_pixel": 36, "tracking_ms_per_frame_this_run": tracking_ms,
           "levels": None, "source": None}
    if (w, h) != (640, 480) or not os.path.exists(stats_csv):
        return out
    levels = {}
    with open(stats_csv) as f:
        for r in csv.DictReader(f):
            m = re.search(r"k_icp_iteration<\w+, (\d+), (\d+)>", r["Name"])
            if not m or (int(m.group(1)), int(m.group(2))) not in level_of_shape:
                continue
            lvl = level_of_shape[(int(m.group(1)), int(m.group(2)))]
            pixels = (w >> lvl) * (h >> lvl)
            avg_s = float(r["AverageNs"]) * 1e-9
            gbs = 36.0 * pixels / avg_s / 1e9
            levels["level%d" % lvl] = {"pixels": pixels, "launches": int(r["Calls"]), "avg_us": avg_s * 1e6,
                                       "achieved": gbs, "frac": gbs / HBM_PEAK_GBS}
    if levels:
        out["levels"] = levels
        out["achieved"] = levels.get("level0", {}).get("achieved")
        out["frac"] = levels.get("level0", {}).get("frac")
        out["source"] = ("profiles/%s: rocprofv3 --kernel-trace --stats of an earlier run of this command; the average "
                         "includes the launches of a converged level, which return at once; not measured in this run"
                         % os.path.basename(stats_csv))
    return out


def reduce_ranks(values, op, dist, torch, backend, world):
    """All-reduce a short list of floats over the ranks (sum or max); a world of one returns them as they are."""
    if world <= 1:
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX)
    return [float(v) for v in t.tolist()]


def zslab_leg(a, rank, world, dist, torch, backend, V, steps, solo=False):
    """BASELINE.json configs[3] in the ranks of this run: ONE synthetic 640x480 sequence whose V^3 volume is sharded
    into z-slabs over the ranks (integration per slab, per-slab ray segments exchanged and composited, ICP sums
    replicated or all-reduced).  Every rank returns the same record: frames/s of the job, voxel updates per second
    summed over the ranks, and the integrate roofline with boxes summed and launch time maximised over the ranks."""
    from slambench_amd import synth
    from slambench_amd.config import KFusionConfig, TUM_FR1_DEPTH_INTRINSICS
    from slambench_amd.metrics import ate
    from slambench_amd.sharded import SlabKFusion

    cfg = KFusionConfig(input_size=(640, 480), intrinsics=TUM_FR1_DEPTH_INTRINSICS, volume_resolution=(V, V, V))
    untimed = 8
    total = steps + untimed
    n_unique = max(8, min(40, total))
    depth, gt, ts = synth.make_sequence(n_unique, seed=1, intrinsics=TUM_FR1_DEPTH_INTRINSICS, noise=True)
    order = ping_pong(n_unique, total + 96)
    d_depth = torch.from_numpy(depth.view(np.int16)).cuda()
    ptrs = [int(d_depth[i].data_ptr()) for i in range(n_unique)]
    can_tile = (640 * 480) % world == 0
    exchange = a.exchange if a.exchange != "auto" else "gather"
    sharded = SlabKFusion(cfg, rank, world, dist, torch, icp=a.icp, force_exchange=solo, exchange=exchange)
    kf = sharded.kf
    kf.set_timing(2)                    # every integration of this short leg is timed
    for i in range(untimed):
        sharded.process_frame(ptrs[order[i]])
    kf.sync()
    probe = None
    if a.exchange == "auto" and world > 1 and can_tile:
        # which of the two exchanges is faster depends on the link rate and on the fixed cost of a collective at this
        # rank count: a few frames of each, timed on all ranks, decide (the maps and poses are the same either way)
        probe = {}
        at = untimed                    # the stream simply goes on, so that the frames keep tracking
        probe_error = None
        for mode in ("gather", "tiles", "gather", "tiles"):
            if mode == "tiles" and probe_error:
                continue
            sharded.exchange_mode = mode
            try:
                for i in range(4):                            # buffers and communicators of the mode come up untimed
                    sharded.process_frame(ptrs[order[at + i]])
                kf.sync()
                dist.barrier()
                tp = time.perf_counter()
                for i in range(4, 20):
                    sharded.process_frame(ptrs[order[at + i]])
                kf.sync()
            except Exception as exc:
                if mode == "gather":
                    raise
                # the second exchange is an option, not the leg: without it the all-gather runs the leg
                probe_error = "%s: %s" % (type(exc).__name__, exc)
                sys.stderr.write("bench.py: rank %d: tile exchange failed in the probe: %s\n" % (rank, probe_error))
                continue
            at += 20
            (t_mode,) = reduce_ranks([time.perf_counter() - tp], "max", dist, torch, backend, world)
            probe[mode] = min(probe.get(mode, 1e9), t_mode / 16 * 1e3)
        exchange = min(probe, key=probe.get)
        if probe_error:
            probe["tiles_error"] = probe_error
        sharded.exchange_mode = exchange
        kf.reset()                      # the probe frames are not part of the sequence
        kf.set_timing(2)
        for i in range(untimed):
            sharded.process_frame(ptrs[order[i]])
        kf.sync()
    boxes0, launches0, voxels_per_box = kf.integrate_totals()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(untimed, total):
        sharded.process_frame(ptrs[order[i]])
    kf.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    boxes1, launches1, _ = kf.integrate_totals()
    integ_ms, tracked = [], 0
    for f in range(untimed, total):
        st = kf.status(f)
        tracked += int(st["tracked"])
        if st["integrated"]:
            integ_ms.append(float(kf.phase_ms(f)[2]))
    n_int = launches1 - launches0
    t_int = float(np.mean(integ_ms)) * 1e-3 if integ_ms else 0.0
    n_fwd = min(total, n_unique)
    est = np.stack([kf.status(f)["pose"] for f in range(n_fwd)])
    ate_rmse = ate(est, gt[:n_fwd], ts[:n_fwd], ts[:n_fwd])["rmse"]
    elapsed, t_int, neg_tracked = reduce_ranks([elapsed, t_int, -tracked], "max", dist, torch, backend, world)
    (voxels,) = reduce_ranks([(boxes1 - boxes0) * voxels_per_box], "sum", dist, torch, backend, world)
    # the same stream a little further with the other ICP mode: the configuration names all-reduced sums, the default
    # replicates the tracking (no collective in the iteration chain)
    other = None
    if world > 1 and not a.no_icp_probe:
        other_mode = "allreduce" if sharded.icp == "replicated" else "replicated"
        sharded.icp = other_mode
        at, n_other = total, 40
        try:
            for i in range(8):
                sharded.process_frame(ptrs[order[at + i]])
            kf.sync()
            dist.barrier()
            tp = time.perf_counter()
            for i in range(8, 8 + n_other):
                sharded.process_frame(ptrs[order[at + i]])
            kf.sync()
            (t_other,) = reduce_ranks([time.perf_counter() - tp], "max", dist, torch, backend, world)
            ok = sum(int(kf.status(kf.frame - 1 - k)["tracked"]) for k in range(n_other))
            other = {"icp": other_mode, "frames_per_s": n_other / t_other, "ms_per_step": t_other / n_other * 1e3,
                     "steps": n_other, "tracked_fraction": ok / n_other}
        except Exception as exc:          # the leg's own result stands
            other = {"icp": other_mode, "error": "%s: %s" % (type(exc).__name__, exc)}
            sys.stderr.write("bench.py: rank %d: extra frames with icp=%s failed: %s\n" % (rank, other_mode, other["error"]))
    sharded.close()
    del d_depth
    vox_per_launch = voxels / n_int if n_int else None
    achieved = 8.0 * vox_per_launch / t_int / 1e9 if (n_int and t_int > 0) else None
    return {
        "workload": "ONE synthetic 640x480 sequence, %d^3 TSDF over 8 m in %d z-slabs, ICP 10/5/4, integration-rate 2 "
                    "(BASELINE.json configs[3])" % (V, world),
        "volume": V, "ranks": world, "backend": backend, "icp": a.icp,
        "exchange": "all-gather of the packed ray segments" if exchange == "gather" else
                    "all-to-all by image share + all-gather of the shares",
        "exchange_probe_ms_per_frame": probe,
        "steps": steps, "frames_per_s": steps / elapsed, "ms_per_step": elapsed / steps * 1e3,
        "scaling": "strong",
        "voxel_updates_per_s": voxels / elapsed,
        "integrate_voxels_per_launch_all_ranks": vox_per_launch,
        "integrate_ms_slowest_rank": t_int * 1e3 if t_int > 0 else None,
        "integrate_voxels_per_s_all_ranks": vox_per_launch / t_int if (n_int and t_int > 0) else None,
        "tracked_fraction": -neg_tracked / max(1, steps),
        "ate_rmse_m": ate_rmse,
        "other_icp_mode": other,
        "roofline": {"kernel": "k_integrate_plan + k_integrate", "bound": "hbm", "achieved": achieved,
                     "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                     "frac": achieved / (HBM_PEAK_GBS * world) if achieved else None, "traffic": None,
                     "note": "8 B x the voxels of the boxes the work launches of ALL ranks took per integration / the "
                             "slowest rank's mean HIP-event duration of its two launches / (ranks x 8 TB/s)"},
        "note": "frames/s of a sharded sequence is bounded by what every rank repeats (preprocessing, tracking); "
                "integrate_voxels_per_s_all_ranks (voxels all ranks integrate per launch / the slowest rank's launch "
                "time) is the figure that grows with the ranks; voxel_updates_per_s is voxels over wall time",
    }


def group_leg(a, world, n_devices, torch, V, steps):
    """The same configuration through the library's in-process group (kfh_group_*: what the plugin's --devices N
    uses, the harness being one process): `world` z-slab contexts on the devices 0..world-1 of THIS process (wrapping
    round when there are fewer devices), the members reading each other's ray segments in place over peer access.
    Rank 0 runs it while the other ranks wait."""
    from slambench_amd import synth
    from slambench_amd.config import KFusionConfig, TUM_FR1_DEPTH_INTRINSICS
    from slambench_amd.kfusion import GroupKFusion
    from slambench_amd.metrics import ate

    cfg = KFusionConfig(input_size=(640, 480), intrinsics=TUM_FR1_DEPTH_INTRINSICS, volume_resolution=(V, V, V))
    untimed = 8
    total = steps + untimed
    n_unique = max(8, min(40, total))
    depth, gt, ts = synth.make_sequence(n_unique, seed=1, intrinsics=TUM_FR1_DEPTH_INTRINSICS, noise=True)
    order = ping_pong(n_unique, total)
    devices = [m % n_devices for m in range(world)]
    home = torch.cuda.current_device()
    resident = {}                        # per device: the stream's frames in that device's memory
    for d in sorted(set(devices)):
        with torch.cuda.device(d):
            resident[d] = torch.from_numpy(depth.view(np.int16)).cuda()
    ptrs = [[int(resident[d][i].data_ptr()) for d in devices] for i in range(n_unique)]
    group = GroupKFusion(cfg, devices)
    try:
        for i in range(untimed):
            group.process_frame(ptrs[order[i]])
        group.sync()
        t0 = time.perf_counter()
        for i in range(untimed, total):
            group.process_frame(ptrs[order[i]])
        group.sync()
        elapsed = time.perf_counter() - t0
        first = group.members[0]
        tracked = sum(int(first.status(f)["tracked"]) for f in range(untimed, total))
        n_fwd = min(total, n_unique)
        est = np.stack([first.status(f)["pose"] for f in range(n_fwd)])
        ate_rmse = ate(est, gt[:n_fwd], ts[:n_fwd], ts[:n_fwd])["rmse"]
    finally:
        group.close()
        torch.cuda.set_device(home)
    return {
        "workload": "ONE synthetic 640x480 sequence, %d^3 TSDF over 8 m in %d z-slabs inside one process (kfh_group), "
                    "ICP replicated" % (V, world),
        "members": world, "devices": sorted(set(devices)),
        "exchange": os.environ.get("KFH_GROUP_EXCHANGE") or ("share" if world >= 4 else "gather"),
        "steps": steps, "frames_per_s": steps / elapsed, "ms_per_step": elapsed / steps * 1e3,
        "tracked_fraction": tracked / max(1, steps), "ate_rmse_m": ate_rmse,
    }


def main():
    a = parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("OMP_NUM_THREADS", "8")

    import torch
    import torch.distributed as dist
    from slambench_amd import synth
    from slambench_amd.config import KFusionConfig, TUM_FR1_DEPTH_INTRINSICS
    from slambench_amd.kfusion import KFusion
    from slambench_amd.metrics import ate

    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    n_devices = torch.cuda.device_count()
    device_index = local_rank % n_devices
    torch.cuda.set_device(device_index)
    backend = a.backend
    if backend == "auto":
        backend = "nccl" if world <= n_devices else "gloo"     # RCCL wants one device per rank
    ranks_seen = 1
    solo_exchange = a.mode == "zslab" and world == 1 and a.force_exchange
    if solo_exchange:
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
    if world > 1 or solo_exchange:
        with stdout_to_stderr():
            kw = {"device_id": torch.device("cuda", device_index)} if backend == "nccl" else {}
            if solo_exchange:
                kw.update(init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
            import datetime
            dist.init_process_group(backend, timeout=datetime.timedelta(seconds=a.pg_timeout), **kw)
            one = torch.ones(1, dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(one)                   # the ranks count themselves (and the communicator comes up)
            ranks_seen = int(one.item())
    zslab = a.mode == "zslab" and (world > 1 or solo_exchange)

    V = a.volume
    cfg = KFusionConfig(input_size=(640, 480), intrinsics=TUM_FR1_DEPTH_INTRINSICS, compute_size_ratio=a.csr,
                        volume_resolution=(V, V, V))
    # Tracking only starts with the fifth frame of a sequence (the first raycast happens on the fourth), so the
    # frames before the timed region are never fewer than 8: the timed region must hold the full pipeline.
    untimed = max(a.warmup, 8)
    total = a.steps + untimed
    n_unique = max(8, min(a.unique_frames, total))
    seed = 1 if zslab else 1 + rank          # sharded: every rank sees the same frames (broadcast input)
    depth, gt, ts = synth.make_sequence(n_unique, seed=seed, intrinsics=TUM_FR1_DEPTH_INTRINSICS, noise=True)
    order = ping_pong(n_unique, total + 340)
    d_depth = torch.from_numpy(depth.view(np.int16)).cuda()      # every frame resident in HBM
    frame_ptrs = [int(d_depth[i].data_ptr()) for i in range(n_unique)]

    if zslab:
        from slambench_amd.sharded import SlabKFusion
        if a.exchange == "auto":
            a.exchange = "gather"
        sharded = SlabKFusion(cfg, rank, world, dist, torch, icp=a.icp, force_exchange=solo_exchange,
                              exchange=a.exchange)
        kf = sharded.kf
        step = sharded.process_frame
    else:
        kf = KFusion(cfg)
        step = kf.process_frame
    # two HIP events around the integration phase, on the pipeline's stream: on every 16th frame (a record occupies
    # the stream for a few microseconds), on every frame when the run is too short to hold a few such frames
    timing_every = 16 if a.steps >= 96 else 1
    kf.set_timing(3 if timing_every == 16 else 2)
    # further independent sequences on the same GPU (replica mode only): own context, own stream, own frames
    extra = []
    if a.sequences_per_gpu > 1 and not zslab:
        for q in range(1, a.sequences_per_gpu):
            dq, _, _ = synth.make_sequence(n_unique, seed=100 * q + seed, intrinsics=TUM_FR1_DEPTH_INTRINSICS, noise=True)
            tq = torch.from_numpy(dq.view(np.int16)).cuda()
            extra.append((KFusion(cfg), tq, [int(tq[i].data_ptr()) for i in range(n_unique)]))
    if extra:
        first_step = step

        def step(ptr_index):           # one frame of every sequence, launches interleaved from this one thread
            first_step(frame_ptrs[ptr_index])
            for kq, _, ptrs in extra:
                kq.process_frame(ptrs[ptr_index])
    else:
        single_step = step

        def step(ptr_index):
            single_step(frame_ptrs[ptr_index])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        kf.sync()

    def sync_all():
        kf.sync()
        for kq, _, _ in extra:
            kq.sync()

    for i in range(untimed):
        step(order[i])
    sync_all()
    boxes0, launches0, voxels_per_box = kf.integrate_totals()
    barrier()
    if a.host_threads and extra:
        import threading
        seqs = [(kf, frame_ptrs)] + [(kq, ptrs) for kq, _, ptrs in extra]
        go = threading.Barrier(len(seqs) + 1)

        def run(k, ptrs):
            go.wait()
            for i in range(untimed, total):
                k.process_frame(ptrs[order[i]])
            k.sync()

        workers = [threading.Thread(target=run, args=sq) for sq in seqs]
        for t in workers:
            t.start()
        go.wait()
        t0 = time.perf_counter()
        for t in workers:
            t.join()
    else:
        t0 = time.perf_counter()
        for i in range(untimed, total):
            step(order[i])
        sync_all()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()
    boxes1, launches1, _ = kf.integrate_totals()

    # integrate launches of the timed region (HIP events recorded live on the pipeline's stream)
    first = max(untimed, total - 500)
    integ_ms = []
    tracked = 0
    for f in range(first, total):
        st = kf.status(f)
        tracked += int(st["tracked"])
        if st["integrated"] and f % timing_every == 0:
            integ_ms.append(float(kf.phase_ms(f)[2]))
    integ_ms = np.asarray(integ_ms)
    t_int = float(integ_ms.mean()) * 1e-3 if integ_ms.size else float("nan")
    zs = kf.cfg.slab if kf.cfg.slab != (0, 0) else (0, V)
    planes = min(zs[1] + 2, V) - max(zs[0] - 2, 0) if zslab else V
    # Voxels the work launches of the timed region really had to read and write: the plan launch keeps the boxes
    # (64 x 4 x voxels_per_box / 256 voxels) that can hold a voxel integrateKernel would update; the device counts them.
    n_int = launches1 - launches0
    voxels_taken = (boxes1 - boxes0) * voxels_per_box
    peak = HBM_PEAK_GBS
    if zslab and world > 1:
        # one volume over all ranks: boxes summed, launch time of the slowest rank, peak of all the GPUs
        (voxels_taken,) = reduce_ranks([voxels_taken], "sum", dist, torch, backend, world)
        (t_max,) = reduce_ranks([0.0 if np.isnan(t_int) else t_int], "max", dist, torch, backend, world)
        t_int = t_max if t_max > 0 else float("nan")
        planes = V
        peak = HBM_PEAK_GBS * world
    vox_per_launch = voxels_taken / n_int if n_int else None
    achieved = 8.0 * vox_per_launch / t_int / 1e9 if (n_int and t_int == t_int and t_int > 0) else None

    # accuracy on the forward part of the stream (ground truth is known for the synthetic camera)
    n_fwd = min(total, n_unique)
    est = np.stack([kf.status(f)["pose"] for f in range(n_fwd)]) if total <= 4096 else None
    ate_hip = ate(est, gt[:n_fwd], ts[:n_fwd], ts[:n_fwd])["rmse"] if est is not None else None

    # the tracking phase by HIP events on a few more frames (all four phases are timed then, which costs the stream
    # a few microseconds per event: after the timed region)
    tracking_ms = None
    after = total                       # next frame of the stream
    if not zslab and not extra:
        kf.set_timing(1)
        t_frames = []
        for i in range(32):
            step(order[after])
            after += 1
            t_frames.append(kf.frame - 1)
        kf.sync()
        tracking_ms = float(np.mean([kf.phase_ms(f)[1] for f in t_frames[8:]]))
        kf.set_timing(0)

    # frames handed over in host memory (after the records of the timed region have been read: it moves the
    # context's frame counter on)
    pcie_fps = None
    if not a.no_host_frames and not zslab:
        n_host = min(a.steps, 300)
        kf.sync()
        th = time.perf_counter()
        for i in range(n_host):
            kf.process_frame(depth[order[after + i]])     # numpy frame: kfh_upload_depth per frame
        kf.sync()
        pcie_fps = n_host / (time.perf_counter() - th)

    result = None
    if rank == 0:
        traffic, traffic_source = None, None
        if os.path.exists(a.traffic_json):
            try:
                with open(a.traffic_json) as f:
                    tj = json.load(f)
                if int(tj.get("volume", 0)) == V:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = ("profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of "
                                      "this command (2 x FETCH_SIZE + WRITE_SIZE); not measured in this run"
                                      % tj.get("source", os.path.basename(a.traffic_json)))
            except Exception:
                traffic = None
        result = {
            "metric": "KFusion frames/s @640x480 512^3 TSDF; integrate GVox/s; ATE delta vs ref",
            "value": (1 if zslab else world * (1 + len(extra))) * a.steps / elapsed,
            "unit": "frames/s",
            "n_gpus": ranks_seen,
            "devices_used": min(ranks_seen, n_devices),
            "steps": a.steps,
            "warmup": a.warmup,
            "untimed_frames": untimed,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if zslab else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "KFusion 640x480 depth (TUM fr1 intrinsics), %d^3 TSDF over 8 m, ICP 10/5/4, "
                            "integration-rate 2, synthetic room stream%s"
                            % (V, " (configs[3])" if (zslab and V == 2048) else
                               " (configs[1])" if (V == 512 and a.csr == 1 and not zslab) else ""),
                "compute_size": "%dx%d" % cfg.compute_size, "volume": V,
                "input": "every depth frame is resident in HBM before the timed region starts (value has no upload in "
                         "it); pcie_inclusive_frames_per_s is the same stream with one host-to-device upload per frame",
                "parallelism": ("one sequence, volume in %d z-slabs, raycast segments %s, ICP %s (%s)"
                                % (world, "all-gathered" if a.exchange == "gather" else
                                   "exchanged by image share (all-to-all + all-gather)", a.icp, backend)
                                if zslab else "%d sequence(s) per rank (replicas, no collective; %s)" % (1 + len(extra), backend)
                                if world > 1 else "%d sequence(s) on one GPU" % (1 + len(extra)) if extra else "single GPU"),
            },
            "integrate_gvox_per_s": (V * V * planes / t_int / 1e9) if t_int == t_int else None,
            "integrate_ms": t_int * 1e3 if t_int == t_int else None,
            "integrate_voxels_per_launch": vox_per_launch,
            "integrate_visited_fraction": (vox_per_launch / (V * V * planes)) if vox_per_launch else None,
            "tracked_fraction": tracked / max(1, total - first),
            "ate_rmse_m": ate_hip,
            "pcie_inclusive_frames_per_s": pcie_fps,
            "roofline": {
                "kernel": "k_integrate_plan + k_integrate",
                "bound": "hbm",
                "achieved": achieved,
                "peak": peak,
                "unit": "GB/s",
                "frac": achieved / peak if achieved else None,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "note": "achieved = 8 B x the voxels of the boxes the work launch took (counted by the device; the plan "
                        "launch drops boxes outside the camera frustum or behind the surface, which integrateKernel "
                        "leaves untouched) / mean HIP-event duration of the two launches, sampled in the timed region; "
                        "8 B x V^3 / that duration is integrate_gvox_per_s x 8, not a bandwidth",
            },
        }
        if a.dataset != "none" and world == 1:
            # real data, when somebody has put it on the machine: the only numeric pins the reference holds for this path
            from slambench_amd import refcheck
            names = sorted(refcheck.REFERENCE_HEADS) if a.dataset == "auto" else [a.dataset]
            checks = []
            for name in names:
                try:
                    checks.append(refcheck.check_dataset(name))
                except Exception as exc:
                    checks.append({"dataset": name, "status": "failed", "error": "%s: %s" % (type(exc).__name__, exc)})
            result["dataset_check"] = checks
        if not zslab:
            result["roofline_track"] = track_roofline(a.track_stats_csv, cfg.compute_size, tracking_ms)
        if not a.no_dense_leg and not zslab and world == 1:
            result["roofline_dense"] = dense_integrate_leg(KFusion, KFusionConfig, TUM_FR1_DEPTH_INTRINSICS, V, torch)
            result["roofline_dense_band"] = dense_integrate_leg(KFusion, KFusionConfig, TUM_FR1_DEPTH_INTRINSICS, V,
                                                                torch, band=True)
            dense = result["roofline_dense"]
            if os.path.exists(a.dense_traffic_json):
                try:
                    with open(a.dense_traffic_json) as f:
                        dj = json.load(f)
                    if int(dj.get("volume", 0)) == V:
                        dense["traffic"] = dj.get("hbm_bytes_per_launch")
                        dense["traffic_source"] = ("profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                                   "tools/dense_leg.py (the same leg as a program of its own); not "
                                                   "measured in this run" % os.path.basename(a.dense_traffic_json))
                except Exception:
                    pass
        if not a.no_cpu_baseline and a.cpu_frames > 0 and world == 1:
            threads = max(1, min(a.cpu_threads, os.cpu_count() or 1))
            n_cpu = min(a.cpu_frames, n_unique)
            fps_cpu, poses_cpu, phases, threads = cpu_baseline(cfg, depth, n_cpu, threads)
            d = np.linalg.norm(est[:n_cpu, :3, 3] - poses_cpu[:, :3, 3], axis=1) if est is not None else None
            a_cpu = ate(poses_cpu, gt[:n_cpu], ts[:n_cpu], ts[:n_cpu])["rmse"]
            a_hip = ate(est[:n_cpu], gt[:n_cpu], ts[:n_cpu], ts[:n_cpu])["rmse"] if est is not None else None
            result["cpu_baseline"] = {
                "value": fps_cpu, "unit": "frames/s", "cores": threads, "kind": "port",
                "cpu_model": cpu_model(), "host_cpus": os.cpu_count(),
                "sample": "frames %d to %d of the same stream (same config) through the CPU oracle, OpenMP; the "
                          "start-up frames before them run untimed" % (min(CPU_STARTUP_FRAMES, n_cpu - 1), n_cpu - 1),
                "phase_seconds": {"preprocessing": phases[0], "tracking": phases[1], "integration": phases[2],
                                  "raycasting": phases[3]},
            }
            if a.cpu_frames_1t > 0:
                # the single-threaded figure (what the reference calls kfusion-cpp next to kfusion-openmp)
                n1 = min(a.cpu_frames_1t, n_unique)
                fps_1t, _, phases_1t, _ = cpu_baseline(cfg, depth, n1, 1)
                result["cpu_baseline_single_thread"] = {
                    "value": fps_1t, "unit": "frames/s", "cores": 1, "kind": "port",
                    "sample": "frames %d to %d of the same stream through the CPU oracle, one thread; the start-up "
                              "frames before them run untimed" % (min(CPU_STARTUP_FRAMES, n1 - 1), n1 - 1),
                    "phase_seconds": {"preprocessing": phases_1t[0], "tracking": phases_1t[1],
                                      "integration": phases_1t[2], "raycasting": phases_1t[3]},
                }
            result["ate_delta_vs_cpu_m"] = abs(a_hip - a_cpu) if a_hip is not None else None
            result["max_position_delta_vs_cpu_m"] = float(d.max()) if d is not None else None
    kf.close()
    for kq, _, _ in extra:
        kq.close()
    del d_depth
    leg_failed = False
    if world > 1 and a.mode == "replicas" and not a.no_zslab_leg:
        if rank == 0:       # should the sharded leg end the job, the replica line is at least in the log
            sys.stderr.write("bench.py: replica leg done: " + json.dumps({k: result[k] for k in ("value", "n_gpus", "ms_per_step")}) + "\n")
        torch.cuda.empty_cache()
        watchdog = None
        if rank == 0:
            # a sharded leg that hangs (a collective that never completes) must not take the replica result with it:
            # before the process group's own timeout aborts the process, the line goes out with the failure in it
            import threading

            def give_up():
                result.setdefault("zslab_%d" % a.zslab_volume, {"error": "no result within %d s" % a.zslab_timeout})
                result.setdefault("zslab_%d_in_process" % a.zslab_volume, {"error": "no result within %d s" % a.zslab_timeout})
                print(json.dumps(result))
                sys.stdout.flush()
                os._exit(4)

            watchdog = threading.Timer(a.zslab_timeout, give_up)
            watchdog.daemon = True
            watchdog.start()
        try:
            leg = zslab_leg(a, rank, world, dist, torch, backend, a.zslab_volume, a.zslab_steps)
        except Exception as exc:            # the line still goes out, with the failure in it, and the run fails
            leg_failed = True
            leg = {"error": "%s: %s" % (type(exc).__name__, exc)}
            sys.stderr.write("bench.py: rank %d: z-slab leg failed: %s\n" % (rank, leg["error"]))
        if rank == 0:
            result["zslab_%d" % a.zslab_volume] = leg
        if not leg_failed and not a.no_group_leg:
            # the in-process group on the same devices, by rank 0 alone (the others have freed theirs and wait)
            torch.cuda.empty_cache()
            dist.barrier()
            if rank == 0:
                try:
                    result["zslab_%d_in_process" % a.zslab_volume] = group_leg(a, world, n_devices, torch, a.zslab_volume,
                                                                             a.zslab_steps)
                except Exception as exc:                # reported, but the run's other results stand
                    result["zslab_%d_in_process" % a.zslab_volume] = {"error": "%s: %s" % (type(exc).__name__, exc)}
                    sys.stderr.write("bench.py: in-process group leg failed: %s\n" % exc)
            dist.barrier()
        if watchdog is not None:
            watchdog.cancel()
    if world > 1 or solo_exchange:
        if not leg_failed:
            dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))
        sys.stdout.flush()
    if leg_failed:
        sys.exit(4)
    if ranks_seen != a.gpus:
        sys.stderr.write("bench.py: --gpus %d but %d rank(s) took part (WORLD_SIZE=%s)\n"
                         % (a.gpus, ranks_seen, os.environ.get("WORLD_SIZE")))
        sys.exit(3)


if __name__ == "__main__":
    main()
