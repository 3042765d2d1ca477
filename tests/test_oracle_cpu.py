"""CPU suite, part 1: the oracle against its committed golden vectors and against the reference's own data.

What the reference tree pins for this path (it holds no kfusion source and no per-kernel vectors):
  - the scheduling signature and the parameter defaults in its published kfusion-cpp logs
    (tests/golden/ref_kfusion_cpp_lr_kt2_head.tsv = the first 76 lines of
    icra2018_results/1080/violons_libkfusion-cpp_living_room_traj2_loop.log);
  - the harness-side ATE definition (framework/shared/src/metrics/ATEMetric.cpp:74-151).
Everything else is the oracle checked against vectors the oracle produced (regression protection only:
PARITY UNPINNED, see oracle/kf_ref.h).
"""
import os

import numpy as np
import pytest

from oracle import kf_ref
from slambench_amd import synth
from slambench_amd.config import KFusionConfig

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "kf_ref_golden.npz"))


@pytest.fixture(scope="module")
def fresh():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build()


def test_synthetic_stream_is_deterministic(gold, fresh):
    assert np.array_equal(gold["depth_u16"], fresh["depth_u16"])
    assert np.array_equal(gold["gt_poses"], fresh["gt_poses"])


@pytest.mark.parametrize("key", ["mm2meters", "mm2meters_half", "half_sample", "depth2vertex", "track_result_f7",
                                 "pipeline_flags", "render_depth_f7", "render_track_f7"])
def test_integer_and_exact_float_kernels(gold, fresh, key):
    """Index work and expressions without transcendental functions reproduce bit for bit."""
    assert np.array_equal(gold[key], fresh[key]), key


def test_shaded_volume_picture(gold, fresh):
    """Bytes of a float shading (square roots and divisions only, both correctly rounded): equal, except that a
    value within rounding of a byte boundary may fall on either side on another libm / compiler."""
    a, b = gold["render_volume_f7"].astype(np.int32), fresh["render_volume_f7"].astype(np.int32)
    assert (a[..., :3].max(axis=-1) > 0).mean() > 0.2                  # the picture shows something
    assert np.abs(a - b).max() <= 1 and (a != b).mean() < 1e-3


@pytest.mark.parametrize("key", ["bilateral", "vertex2normal", "volume_f7", "model_vertex_f7", "model_normal_f7",
                                 "reduction_f7", "track_error_f7", "reduce_f7", "solve_f7", "pipeline_poses",
                                 "se3_exp", "se3_exp_small"])
def test_float_kernels(gold, fresh, key):
    """Tolerance 1e-5 relative (expf / sin / cos come from the host libm), 1 LSB on the TSDF shorts."""
    a, b = gold[key], fresh[key]
    if a.dtype == np.int16:
        assert np.abs(a.astype(np.int32) - b.astype(np.int32)).max() <= 1
    else:
        np.testing.assert_allclose(b, a, rtol=1e-5, atol=1e-7)


def test_mm2meters_subsampling_indices():
    """out[x, y] = in[x*r, y*r] / 1000 for every ratio: index parity against a numpy gather."""
    d = (np.arange(480 * 640) % 60000).astype(np.uint16).reshape(480, 640)
    for r in (1, 2, 4, 8):
        got = kf_ref.mm2meters(d, 640 // r, 480 // r, 1000.0)
        want = d[::r, ::r].astype(np.float32) / np.float32(1000.0)
        assert np.array_equal(got, want)


def test_half_sample_robust_average():
    img = np.array([[1.0, 1.1, 5.0, 5.0], [1.2, 9.0, 5.0, 0.0]], np.float32)
    out = kf_ref.half_sample(img, np.float32(0.3), 1)
    # block 0: centre 1.0, neighbours within 0.3: 1.0, 1.1, 1.2 (9.0 rejected)
    assert out.shape == (1, 2)
    np.testing.assert_allclose(out[0, 0], (1.0 + 1.1 + 1.2) / 3, rtol=1e-6)
    # block 1: centre 5.0: 5, 5, 5 (0 rejected)
    np.testing.assert_allclose(out[0, 1], 5.0, rtol=1e-6)


def test_solve_truncates_small_singular_values():
    """GR_SVD backsub(b, 1e6): directions with sigma < sigma_max / 1e6 are dropped."""
    vals = np.zeros(27, np.float32)
    diag = [1e6, 1.0, 2.0, 3.0, 0.5, 4.0]          # sigma_max / 1e6 = 1: the entries 1.0 and 0.5 are cut
    idx = [6, 12, 17, 21, 24, 26]
    for i, d in zip(idx, diag):
        vals[i] = d
    vals[:6] = 1.0
    x = kf_ref.solve(vals)
    np.testing.assert_allclose(x, [1e-6, 0.0, 0.5, 1 / 3.0, 0.0, 0.25], rtol=1e-12, atol=1e-15)


def test_se3_exp_is_a_rigid_transform():
    T = kf_ref.se3_exp(np.array([0.1, 0.2, -0.3, 0.4, -0.2, 0.1])).astype(np.float64)
    R = T[:3, :3]
    np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-6)
    assert abs(np.linalg.det(R) - 1.0) < 1e-6
    assert np.array_equal(kf_ref.se3_exp(np.zeros(6)), np.eye(4, dtype=np.float32))


def test_check_pose_rejects_poor_tracking():
    pose = np.eye(4, dtype=np.float32)
    pose[0, 3] = 1.0
    old = np.eye(4, dtype=np.float32)
    sums = np.zeros(32, np.float32)
    sums[0], sums[28] = 1e-4 * 20000, 20000        # rms residual 1e-2 < 2e-2, 20000/(160*120) > 0.15
    p, ok = kf_ref.check_pose(pose, old, sums, 160, 120)
    assert ok and np.array_equal(p, pose)
    sums[28] = 2000                                  # too few pixels
    p, ok = kf_ref.check_pose(pose, old, sums, 160, 120)
    assert not ok and np.array_equal(p, old)
    sums[0], sums[28] = 1e-3 * 20000, 20000          # rms residual 3.2e-2 > 2e-2
    p, ok = kf_ref.check_pose(pose, old, sums, 160, 120)
    assert not ok and np.array_equal(p, old)


def _reference_log():
    path = os.path.join(GOLDEN, "ref_kfusion_cpp_lr_kt2_head.tsv")
    from slambench_amd.harness import parse_log
    return parse_log(open(path).read())


def test_parameter_defaults_match_reference_log():
    props, cols, rows = _reference_log()
    cfg = KFusionConfig()
    assert int(props["compute-size-ratio"]) == cfg.compute_size_ratio
    assert int(props["integration-rate"]) == cfg.integration_rate
    assert int(props["tracking-rate"]) == cfg.tracking_rate
    assert int(props["rendering-rate"]) == cfg.rendering_rate
    assert float(props["icp-threshold"]) == pytest.approx(cfg.icp_threshold)
    assert float(props["mu"]) == pytest.approx(cfg.mu)
    assert tuple(float(v) for v in props["volume-size"].split(",")) == cfg.volume_size
    assert tuple(float(v) for v in props["volume-direction"].split(",")) == cfg.volume_direction
    assert tuple(int(v) for v in props["volume-resolution"].split(",")) == cfg.volume_resolution
    assert (int(props["pyramid-level1"]), int(props["pyramid-level2"]), int(props["pyramid-level3"])) == cfg.pyramid
    assert props["Depth-disparity-params"] == "0.001,0"
    # the published intrinsics are normalised by the image size: 0.751875 * 640 = 481.2
    intr = [float(v) for v in props["Depth-intrinsics-parameters"].split(",")]
    assert intr[0] * 640 == pytest.approx(481.2) and intr[2] * 640 == pytest.approx(319.5, abs=1e-3)


def test_scheduling_signature_matches_reference_log():
    """Which phases run on which frame, read off the reference's kfusion-cpp log, against the oracle."""
    props, cols, rows = _reference_log()
    ci, cr = cols.index("Duration_Integration"), cols.index("Duration_Raycasting")
    cx = cols.index("X")
    integrated_ref = rows[:, ci] > 1e-4
    raycast_ref = rows[:, cr] > 1e-4
    n = rows.shape[0]
    cfg = KFusionConfig(compute_size_ratio=4, volume_resolution=(128, 128, 128))
    depth, gt, ts = synth.make_sequence(n, seed=1)
    ref = kf_ref.KFusionRef(cfg.to_c(kf_ref.Config))
    flags = [ref.process_frame(depth[i]) for i in range(n)]
    poses = []
    ref2 = kf_ref.KFusionRef(cfg.to_c(kf_ref.Config))
    for i in range(6):
        ref2.process_frame(depth[i])
        poses.append(ref2.pose[:3, 3].copy())
    assert [bool(f & 4) for f in flags] == list(raycast_ref)
    # integration runs on frames 0-3 unconditionally, afterwards only when tracking succeeded: the oracle
    # tracks the synthetic stream, the reference run tracked kt2, so the patterns must coincide
    assert all(f & 1 for f in flags[4:])
    assert [bool(f & 2) for f in flags] == list(integrated_ref)
    # the pose stays at volume-direction while there is no model to track against (rows 1-4 of the log)
    assert np.all(rows[:4, cx:cx + 3] == 4.0)
    assert all(np.array_equal(p, np.float32([4, 4, 4])) for p in poses[:4])
    assert not np.array_equal(poses[4], np.float32([4, 4, 4]))
    assert not np.all(rows[4, cx:cx + 3] == 4.0)


def test_oracle_tracks_the_synthetic_camera():
    cfg = KFusionConfig(compute_size_ratio=4, volume_resolution=(128, 128, 128))
    n = 30
    depth, gt, ts = synth.make_sequence(n, seed=2, noise=True)
    ref = kf_ref.KFusionRef(cfg.to_c(kf_ref.Config))
    est = []
    for i in range(n):
        ref.process_frame(depth[i])
        est.append(ref.pose.copy())
    from slambench_amd.metrics import ate
    a = ate(np.stack(est), gt, ts, ts)
    assert a["rmse"] < 0.01, a["rmse"]           # sub-centimetre on a 6 cm voxel grid
