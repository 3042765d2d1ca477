"""Design-space sweeps: one harness process per parameter point, one GPU per process, no communication.

The reference explores a plugin's parameter space by asking the harness for the space (`--dse`, reference
framework/shared/src/SLAMBenchConfiguration.cpp:524-533) and then running one `slambench` process per point.
This module is that outer loop for a multi-GPU node: points are handed to whichever GPU is free, each process
sees exactly one device through HIP_VISIBLE_DEVICES, and the logs are reduced to one row per point.
"""
import itertools
import os
import queue
import subprocess
import threading
from dataclasses import dataclass, field

import numpy as np

from . import harness


def parse_dse(text):
    """`--dse` output -> {long option: {"short-option":, "type":, "default":, "values": [...], "min":, "max":}}.  A
    library loaded under an identifier prints its arguments as "<id>-<name>" (the "libs:" line gives the identifier):
    the keys here are the plain names."""
    params, cur, prefix = {}, None, ""
    for line in text.splitlines():
        if line.startswith("libs:"):
            prefix = line[len("libs:"):].strip() + "-"
        elif line.startswith("argument:"):
            name = line[len("argument:"):]
            cur = params.setdefault(name[len(prefix):] if prefix != "-" and name.startswith(prefix) else name, {})
        elif cur is not None and ":" in line:
            k, v = line.split(":", 1)
            if k == "values":
                cur.setdefault("values", []).append(v)
            else:
                cur[k] = v
    return params


def grid_points(grid):
    """{"mu": ["0.05", "0.1"], "volume-resolution": ["128", "256"]} -> list of dicts, one per combination
    (first key varies slowest)."""
    keys = list(grid)
    return [dict(zip(keys, combo)) for combo in itertools.product(*[grid[k] for k in keys])]


@dataclass
class PointResult:
    point: dict
    gpu: int
    returncode: int
    stdout: str = ""
    stderr: str = ""
    summary: dict = field(default_factory=dict)


def summarise(log_text, gt=None, gt_ts=None):
    """Mean frame duration, frames, last position; ATE when ground truth is given."""
    props, cols, rows = harness.parse_log(log_text)
    out = {"frames": int(rows.shape[0])}
    if rows.shape[0]:
        out["mean_frame_s"] = float(rows[:, cols.index("Duration_Frame")].mean())
        out["last_position"] = [float(v) for v in rows[-1, -3:]]
        if gt is not None and len(gt):
            from .metrics import ate
            est = np.tile(np.eye(4), (rows.shape[0], 1, 1))
            est[:, :3, 3] = rows[:, -3:]
            out["ate_rmse_m"] = float(ate(est, gt, rows[:, 1], gt_ts)["rmse"])
    return out


def run_sweep(points, make_cmd, gpus, timeout=3600, env=None, summarise_fn=None):
    """Run make_cmd(point) -> argv for every point, at most one process per entry of `gpus` (device indices) at a
    time.  Results come back in the order of `points`."""
    free = queue.Queue()
    for g in gpus:
        free.put(g)
    results = [None] * len(points)

    def work(i):
        g = free.get()
        try:
            e = dict(os.environ if env is None else env)
            e["HIP_VISIBLE_DEVICES"] = str(g)
            try:
                r = subprocess.run(make_cmd(points[i]), capture_output=True, text=True, timeout=timeout, env=e)
                res = PointResult(points[i], g, r.returncode, r.stdout, r.stderr)
            except subprocess.TimeoutExpired as ex:
                res = PointResult(points[i], g, -9, ex.stdout or "", "timed out after %d s" % timeout)
            if res.returncode == 0 and summarise_fn is not None:
                res.summary = summarise_fn(res.stdout)
            results[i] = res
        finally:
            free.put(g)

    threads = []
    for i in range(len(points)):
        t = threading.Thread(target=work, args=(i,))
        t.start()
        threads.append(t)
    for t in threads:
        t.join()
    return results


def mini_harness_cmd(input_path, fixed_args=(), frame_limit=0):
    def make(point):
        # loaded under an identifier: the harness's own "-s" otherwise hides the plugin's options from volume-size on
        cmd = [harness.MINI_HARNESS_PATH, "-i", input_path, "-load", harness.PLUGIN_PATH + "=" + harness.LIBRARY_ID]
        if frame_limit:
            cmd += ["-fl", str(frame_limit)]
        args = list(fixed_args)
        for k, v in point.items():
            args += ["--" + k, str(v)]
        return cmd + harness.plugin_options_with_id(args)
    return make
