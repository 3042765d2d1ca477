"""Comparison of a plugin run with the reference's published kfusion-cpp logs.

The reference tree holds no kfusion source, but it holds the per-frame logs of its kfusion-cpp plugin on four
ICL-NUIM and four TUM sequences at the default parameters (icra2018_results/1080/violons_libkfusion-cpp_*.log: camera
position X, Y, Z with ten decimals, and the duration of every phase, for every frame).  The datasets themselves are
fetched from the network by the reference's build (framework/makefiles/dataset-utils.makefile:72,91,95) and are not
in this repository.  The first rows of four of those logs are kept under tests/golden/; this module finds a dataset
file when somebody has put one on the machine, runs it through the plugin boundary and compares what comes out with
those rows: the camera positions, and which frames integrated and raycasted.
"""
import os
import re
import subprocess

import numpy as np

from . import harness

_HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN_DIR = os.path.join(os.path.dirname(_HERE), "tests", "golden")

# dataset file (as the reference's dataset generators name it) -> head of the reference's log of that sequence
REFERENCE_HEADS = {
    "living_room_traj2_loop.slam": "ref_kfusion_cpp_lr_kt2_head.tsv",
    "living_room_traj1_loop.slam": "ref_kfusion_cpp_lr_kt1_head.tsv",
    "rgbd_dataset_freiburg1_xyz.slam": "ref_kfusion_cpp_fr1_xyz_head.tsv",
    "rgbd_dataset_freiburg2_xyz.slam": "ref_kfusion_cpp_fr2_xyz_head.tsv",
}
# the twelve parameters of the kfusion use-case as the reference's logs print them
PLUGIN_PROPERTIES = ("compute-size-ratio", "integration-rate", "tracking-rate", "rendering-rate", "icp-threshold", "mu",
                     "volume-size", "volume-direction", "volume-resolution", "pyramid-level1", "pyramid-level2",
                     "pyramid-level3")
SEARCH_SUBDIRS = ("", "ICL_NUIM", "TUM", "TUM/freiburg1", "TUM/freiburg2", "iclnuim", "tum")


def dataset_roots():
    """Where a dataset file is looked for: KFH_DATASET_DIR, then datasets/ under the working directory and under the
    repository (the layout of the reference: datasets/ICL_NUIM/..., datasets/TUM/freiburg1/...)."""
    roots = []
    if os.environ.get("KFH_DATASET_DIR"):
        roots.append(os.environ["KFH_DATASET_DIR"])
    for root in (os.path.join(os.getcwd(), "datasets"), os.path.join(os.path.dirname(_HERE), "datasets")):
        if os.path.realpath(root) not in [os.path.realpath(r) for r in roots]:
            roots.append(root)
    return roots


def find_dataset(name):
    """Path of the dataset file `name` under one of dataset_roots(), or None."""
    for root in dataset_roots():
        for sub in SEARCH_SUBDIRS:
            p = os.path.join(root, sub, name)
            if os.path.isfile(p):
                return p
    return None


def load_reference_head(path):
    """(properties, positions [n, 3], timestamps [n], integrated [n], raycasted [n]) of a reference log (or its first
    rows).  The reference's host timers bracket real work in its CPU build: a phase that did not run shows a few
    microseconds, one that ran shows milliseconds."""
    with open(path) as f:
        props, cols, rows = harness.parse_log(f.read())
    col = {c: i for i, c in enumerate(cols)}
    pos = rows[:, [col["X"], col["Y"], col["Z"]]]
    integrated = rows[:, col["Duration_Integration"]] > 1e-3
    raycasted = rows[:, col["Duration_Raycasting"]] > 1e-4
    return props, pos, rows[:, col["Timestamp"]], integrated, raycasted


def run_plugin(dataset_path, frames, plugin_args=(), timeout=3600):
    """The first `frames` frame-sets of a dataset file through kfh_mini_slambench -> c_sb_* -> the plugin, with the
    plugin's own defaults unless plugin_args say otherwise.  Returns (properties, positions, timestamps, integrated,
    raycasted); the last two come from the device's status records (KFH_LOG_SCHEDULE), as the plugin's phases only
    queue work and their host timers say nothing about what ran."""
    cmd = [harness.MINI_HARNESS_PATH, "-i", dataset_path, "-load", harness.PLUGIN_PATH, "-fl", str(frames)] + list(plugin_args)
    env = dict(os.environ, KFH_LOG_SCHEDULE="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    if r.returncode != 0:
        raise RuntimeError("kfh_mini_slambench failed (%d): %s" % (r.returncode, r.stderr[-2000:]))
    props, cols, rows = harness.parse_log(r.stdout)
    flags = {}
    for m in re.finditer(r"schedule frame (\d+) tracked (\d) integrated (\d) raycasted (\d)", r.stderr):
        flags[int(m.group(1))] = (m.group(3) == "1", m.group(4) == "1")
    n = rows.shape[0]
    if sorted(flags) != list(range(n)):
        raise RuntimeError("the plugin logged the schedule of %d frames, the harness processed %d" % (len(flags), n))
    col = {c: i for i, c in enumerate(cols)}
    return (props, rows[:, [col["X"], col["Y"], col["Z"]]], rows[:, col["Timestamp"]],
            np.array([flags[i][0] for i in range(n)]), np.array([flags[i][1] for i in range(n)]))


def compare(run, reference):
    """Per-frame comparison of run_plugin()'s result with load_reference_head()'s over the frames both hold."""
    props, pos, ts, integ, ray = run
    rprops, rpos, rts, rinteg, rray = reference
    n = min(len(pos), len(rpos))
    delta = np.linalg.norm(pos[:n] - rpos[:n], axis=1)
    return {
        "frames_compared": int(n),
        "properties_differ": sorted(k for k in PLUGIN_PROPERTIES if k in rprops and props.get(k) != rprops[k]),
        "timestamps_equal": bool(np.allclose(ts[:n], rts[:n], rtol=0, atol=1e-6)),
        "max_position_delta_m": float(delta.max()) if n else None,
        "mean_position_delta_m": float(delta.mean()) if n else None,
        "first_frame_beyond_1mm": int(np.argmax(delta > 1e-3)) if n and (delta > 1e-3).any() else None,
        "integration_schedule_equal": bool(np.array_equal(integ[:n], rinteg[:n])),
        "raycast_schedule_equal": bool(np.array_equal(ray[:n], rray[:n])),
    }


def check_dataset(name, frames=None):
    """The whole hook for one dataset file name: {"status": "absent", ...} when the file is not on the machine,
    otherwise the comparison of its first rows with the reference's log."""
    head = os.path.join(GOLDEN_DIR, REFERENCE_HEADS[name])
    path = find_dataset(name)
    if path is None:
        return {"dataset": name, "status": "absent", "looked_in": dataset_roots(),
                "reference_rows": os.path.relpath(head, os.path.dirname(_HERE))}
    reference = load_reference_head(head)
    n = len(reference[1]) if frames is None else min(frames, len(reference[1]))
    out = compare(run_plugin(path, n), reference)
    out.update(dataset=name, status="compared", path=path)
    out["within_1mm"] = out["max_position_delta_m"] is not None and out["max_position_delta_m"] <= 1e-3
    return out
