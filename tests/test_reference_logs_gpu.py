"""The hook that pins the plugin against the reference's published kfusion-cpp logs once a dataset file is on the
machine (slambench_amd/refcheck.py).  The datasets are not in this repository: the real comparisons skip here, and the
comparison code itself runs on a synthetic .slam file that stands in for one, with a log head written for it."""
import os

import numpy as np
import pytest

from slambench_amd import harness, refcheck, slamfile, synth
from slambench_amd.config import KFusionConfig
from slambench_amd.kfusion import KFusion

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(refcheck.REFERENCE_HEADS))
def test_dataset_against_the_reference_log(torch_cuda, name):
    """With the reference's defaults (256^3 over 8 m, compute-size-ratio 1, mu 0.1, pyramid 10/5/4) the plugin must
    put the camera where the reference's kfusion-cpp put it, frame by frame, and integrate and raycast on the same
    frames.  The bar is the 1 mm of the build's accuracy target; the reference's own CPU and CUDA builds agree to
    0.01-0.05 mm in ATE on these sequences."""
    if refcheck.find_dataset(name) is None:
        pytest.skip("%s is not on this machine (looked in %s)" % (name, refcheck.dataset_roots()))
    out = refcheck.check_dataset(name)
    assert out["frames_compared"] == 40 and out["timestamps_equal"]
    assert out["properties_differ"] == []
    assert out["integration_schedule_equal"] and out["raycast_schedule_equal"]
    assert out["max_position_delta_m"] <= 1e-3, out


def test_the_comparison_on_a_synthetic_file_standing_in_for_a_dataset(torch_cuda, tmp_path, monkeypatch):
    """A synthetic sequence written as a .slam file under KFH_DATASET_DIR with a dataset's name, and a "reference" log
    head made from the same frames through the Python mirror of the Kfusion class: the hook finds the file, runs it
    through the mini-harness and the plugin, and must report equality; a head that is off by 2 mm in one frame, or
    integrates on another frame, must be reported as such."""
    name = "living_room_traj2_loop.slam"
    n = 14
    intr = (481.2, 480.0, 319.5, 239.5)
    depth, gt, ts = synth.make_sequence(n, seed=3, intrinsics=intr)
    data_dir = tmp_path / "datasets" / "ICL_NUIM"
    os.makedirs(str(data_dir))
    slamfile.write_depth_sequence(str(data_dir / name), depth, ts, intr, gt_poses=gt)
    monkeypatch.setenv("KFH_DATASET_DIR", str(tmp_path / "datasets"))
    assert refcheck.find_dataset(name) == str(data_dir / name)
    # the stand-in for the reference's log: plugin defaults, positions of the direct pipeline, durations that show the schedule
    w, h = 640, 480
    intr_rt = tuple(float(np.float32(v / s) * np.float32(s)) for v, s in zip(intr, (w, h, w, h)))
    kf = KFusion(KFusionConfig(intrinsics=intr_rt))
    for i in range(n):
        kf.process_frame(depth[i])
    traj = kf.trajectory()
    flags = [kf.status(f) for f in range(n)]
    kf.close()
    props = {"compute-size-ratio": "1", "integration-rate": "2", "tracking-rate": "1", "rendering-rate": "4",
             "icp-threshold": "1e-05", "mu": "0.1", "volume-size": "8,8,8", "volume-direction": "4,4,4",
             "volume-resolution": "256,256,256", "pyramid-level1": "10", "pyramid-level2": "5", "pyramid-level3": "4"}
    phases = ["Preprocessing", "Tracking", "Integration", "Raycasting", "Render"]

    def head(positions, integrated):
        dur = np.zeros((n, 6))
        dur[:, 3] = np.where(integrated, 0.07, 1e-6)
        dur[:, 4] = np.where([f["raycasted"] for f in flags], 0.3, 2e-6)
        dur[:, 0] = dur[:, 1:].sum(1)
        return harness.format_log(props, ts, positions, dur, phases)

    integrated = np.array([f["integrated"] for f in flags])
    golden = tmp_path / "golden"
    os.makedirs(str(golden))
    monkeypatch.setattr(refcheck, "GOLDEN_DIR", str(golden))
    target = golden / refcheck.REFERENCE_HEADS[name]
    target.write_text(head(traj[:, :3, 3], integrated))
    out = refcheck.check_dataset(name)
    assert out["status"] == "compared" and out["frames_compared"] == n
    assert out["timestamps_equal"] and out["properties_differ"] == []
    assert out["integration_schedule_equal"] and out["raycast_schedule_equal"]
    assert out["max_position_delta_m"] < 1e-6 and out["within_1mm"]       # the log prints ten decimals
    assert integrated[:5].all() and not integrated[5] and integrated[6]    # the reference's start-up signature
    # a head that disagrees is reported as disagreeing
    moved = traj[:, :3, 3].copy()
    moved[9, 1] += 2e-3
    wrong = integrated.copy()
    wrong[7] = not wrong[7]
    target.write_text(head(moved, wrong))
    out = refcheck.check_dataset(name)
    assert not out["within_1mm"] and out["first_frame_beyond_1mm"] == 9
    assert not out["integration_schedule_equal"] and out["raycast_schedule_equal"]
    # and a dataset that is not there is reported as absent, with the places that were searched
    absent = refcheck.check_dataset("rgbd_dataset_freiburg1_xyz.slam")
    assert absent["status"] == "absent" and str(tmp_path / "datasets") in absent["looked_in"]
