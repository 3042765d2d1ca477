"""CPU suite, part 2: host logic, the C ABI surface and the plugin boundary (no compute without a GPU)."""
import ctypes as C
import os
import re
import subprocess

import collections

import numpy as np
import pytest

from slambench_amd import harness, native, synth
from slambench_amd.config import KFusionConfig, KfhConfig
from slambench_amd.metrics import align_umeyama, ate, umeyama

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    n = C.c_int(0)
    native.lib().kfh_device_count(C.byref(n))
    return n.value > 0


# ------------------------------------------------------------------ C ABI surface
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "kfusion_hip.h")).read()
    declared = set(re.findall(r"^(?:int|const char \*)\s*\*?(kfh_\w+)\s*\(", header, re.M))
    assert len(declared) >= 30
    assert declared == set(native.EXPORTED_SYMBOLS)
    L = C.CDLL(native.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert native.lib().kfh_version() >= 100


def test_config_struct_layout_matches_header():
    # 3 ints, 5 floats, 3+3+3 words, 2 floats, 3 ints, 2 ints, 1 int, 2 ints
    assert C.sizeof(KfhConfig) == 4 * (3 + 5 + 9 + 2 + 3 + 2 + 1 + 2)
    from oracle import kf_ref
    assert [f[0] for f in KfhConfig._fields_] == [f[0] for f in kf_ref.Config._fields_]
    assert C.sizeof(native.FrameStatus) == 4 * (4 + 16 + 32 + 4)      # ..., iterations, slow, examined, reserved


def test_invalid_arguments_are_rejected_before_any_device_work():
    L = native.lib()
    ctx = C.c_void_p()
    for mutate, text in [(lambda c: setattr(c, "compute_size_ratio", 3), b"compute_size_ratio"),
                         (lambda c: setattr(c, "mu", 0.0), b"mu must be positive"),
                         (lambda c: setattr(c, "slab_z1", 10 ** 6), b"slab"),
                         (lambda c: setattr(c, "input_width", 642), b"divisible")]:
        c = KFusionConfig().to_c()
        mutate(c)
        assert L.kfh_create(C.byref(c), C.byref(ctx)) == 1
        assert text in L.kfh_last_error()
        assert not ctx.value
    assert L.kfh_create(None, C.byref(ctx)) == 1
    assert L.kfh_sync(None) == 1 and L.kfh_get_pose(None, None) == 1
    assert L.kfh_k_half_sample(None, None, 8, 8, 0.3, None) == 1


def test_create_without_device_reports_no_device():
    if _has_gpu():
        pytest.skip("a HIP device is visible")
    c = KFusionConfig().to_c()
    ctx = C.c_void_p()
    assert native.lib().kfh_create(C.byref(c), C.byref(ctx)) == 4
    assert b"no HIP device" in native.lib().kfh_last_error()
    with pytest.raises(native.KfhError):
        from slambench_amd.kfusion import KFusion
        KFusion(KFusionConfig())


# ------------------------------------------------------------------ configuration
def test_config_validation_and_derived_values():
    cfg = KFusionConfig(compute_size_ratio=2)
    assert cfg.compute_size == (320, 240)
    assert cfg.camera == pytest.approx((240.6, 240.0, 159.75, 119.75))
    c = cfg.to_c()
    assert (c.slab_z0, c.slab_z1) == (0, 256) and c.depth_divisor == 1000.0
    for bad in [dict(compute_size_ratio=3), dict(volume_resolution=(0, 1, 1)), dict(integration_rate=0),
                dict(input_size=(642, 480)), dict(slab=(4, 2)), dict(pyramid=(1, 2))]:
        with pytest.raises(ValueError):
            KFusionConfig(**bad).to_c()
    assert KFusionConfig(volume_resolution=(64, 64, 64), slab=(16, 32)).to_c().slab_z1 == 32


# ------------------------------------------------------------------ synthetic stream
def test_synthetic_stream_properties():
    depth, gt, ts = synth.make_sequence(5, seed=3, size=(160, 120), intrinsics=(120.3, 120.0, 79.875, 59.875))
    depth2, gt2, _ = synth.make_sequence(5, seed=3, size=(160, 120), intrinsics=(120.3, 120.0, 79.875, 59.875))
    assert np.array_equal(depth, depth2) and np.array_equal(gt, gt2)
    assert depth.dtype == np.uint16 and depth.shape == (5, 120, 160)
    assert 500 < depth.min() and depth.max() < 4000          # the whole room is inside KinectFusion's range
    assert np.array_equal(gt[0], np.array([[1, 0, 0, 4], [0, 1, 0, 4], [0, 0, 1, 4], [0, 0, 0, 1]], float))
    assert np.linalg.norm(gt[3][:3, 3] - gt[0][:3, 3]) < 2e-3   # the camera starts at rest
    noisy, _, _ = synth.make_sequence(2, seed=3, size=(160, 120), noise=True)
    assert 0.005 < (noisy == 0).mean() < 0.05                   # about 2 % holes
    other, _, _ = synth.make_sequence(2, seed=4, size=(160, 120))
    assert not np.array_equal(other[:2], depth[:2])


# ------------------------------------------------------------------ ATE / alignment
def test_umeyama_recovers_a_similarity_transform():
    rng = np.random.default_rng(0)
    src = rng.normal(size=(3, 40))
    ang = 0.7
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    dst = 1.3 * R @ src + np.array([[0.5], [-2.0], [1.0]])
    T = umeyama(src, dst, True)
    np.testing.assert_allclose(T[:3, :3], 1.3 * R, atol=1e-9)
    np.testing.assert_allclose(T[:3, 3], [0.5, -2.0, 1.0], atol=1e-9)


def test_ate_definition():
    n = 20
    gt = synth.lissajous_trajectory(n, amp=0.5, period_s=1.0)
    ts = np.arange(n) / 30.0
    assert ate(gt, gt, ts, ts)["rmse"] < 1e-12
    # a rigidly displaced estimate aligns back to zero error (Umeyama), an unaligned one does not
    shifted = gt.copy()
    shifted[:, :3, 3] += [0.1, -0.2, 0.3]
    assert ate(shifted, gt, ts, ts)["rmse"] < 1e-9
    assert ate(shifted, gt, ts, ts, align=False)["max"] == pytest.approx(np.sqrt(0.14))
    # one outlier frame: mean / max / rmse as defined in the reference's ATEMetric
    est = gt.copy()
    est[7, 0, 3] += 0.3
    a = ate(est, gt, ts, ts, align=False)
    assert a["max"] == pytest.approx(0.3) and a["mean"] == pytest.approx(0.3 / n)
    assert a["rmse"] == pytest.approx(0.3 / np.sqrt(n)) and a["last"] == 0.0
    # ground truth at a different rate: each estimate meets the first ground-truth pose that is not earlier
    a2 = ate(gt[::2], gt, ts[::2], ts, align=False)
    assert a2["rmse"] < 1e-12
    assert align_umeyama(gt[:2], gt[:2], ts[:2], ts[:2]).shape == (4, 4)


# ------------------------------------------------------------------ plugin boundary
def test_plugin_exports_the_six_entry_points_and_no_relocalize():
    lib = C.CDLL(harness.PLUGIN_PATH)
    for name in harness.PLUGIN_SYMBOLS:
        assert hasattr(lib, name), name
    # the harness probes for the mangled C++ symbol and otherwise aliases relocalisation to process_once
    assert not hasattr(lib, "c_sb_relocalize")
    assert not hasattr(lib, "_Z13sb_relocalizeP22SLAMBenchLibraryHelper")
    out = subprocess.run(["nm", "-D", "--defined-only", harness.PLUGIN_PATH], capture_output=True, text=True).stdout
    assert "_Z15sb_process_onceP22SLAMBenchLibraryHelper" in out      # same mangled names as a reference plugin


def test_kfd_round_trip_and_log_parser(tmp_path):
    depth, gt, ts = synth.make_sequence(3, seed=1, size=(64, 48), intrinsics=(48.12, 48.0, 31.95, 23.95))
    path = str(tmp_path / "s.kfd")
    harness.write_kfd(path, depth, ts, (48.12, 48.0, 31.95, 23.95), 0.001)
    d2, t2, intr, scale = harness.read_kfd(path)
    assert np.array_equal(d2, depth) and np.allclose(t2, ts)
    assert intr == pytest.approx((48.12, 48.0, 31.95, 23.95), rel=1e-6) and scale == pytest.approx(0.001)
    props, cols, rows = harness.parse_log(
        "Properties:\n=================\n\nmu: 0.1\n\nStatistics:\n=================\n\n"
        "Frame Number\tTimestamp\tX\n1\t0.0000000000\t4.0000000000\n2\t0.0333333333\t4.0010000000\n")
    assert props == {"mu": "0.1"} and cols == ["Frame Number", "Timestamp", "X"] and rows.shape == (2, 3)


def test_mini_harness_exit_codes_without_a_gpu(tmp_path):
    depth, gt, ts = synth.make_sequence(2, seed=1, size=(64, 48), intrinsics=(48.12, 48.0, 31.95, 23.95))
    path = str(tmp_path / "s.kfd")
    harness.write_kfd(path, depth, ts, (48.12, 48.0, 31.95, 23.95))
    # a library without the plugin symbols is refused with exit code 1, like the reference loader
    r = harness.run_mini_harness(path, library=native.LIB_PATH)
    assert r.returncode == 1 and "Cannot load symbol" in r.stderr
    r = harness.run_mini_harness(path, ["--bogus"])
    assert r.returncode == 1 and "unknown option" in r.stderr
    if not _has_gpu():
        # sb_init_slam_system returns false when no device is present -> "exit(1)" path of the harness
        r = harness.run_mini_harness(path, ["-c", "1", "-v", "32"])
        assert r.returncode == 1 and "initialization failed" in r.stderr
        assert "compute-size-ratio: 1" in r.stdout and "volume-resolution: 32,32,32" in r.stdout


# ------------------------------------------------------------------ .slam container, RPE, log writer
def test_slam_file_layout_and_round_trip(tmp_path):
    """Byte layout of the header as the reference serialises it, then a full round trip."""
    import struct
    from slambench_amd import slamfile
    depth, gt, ts = synth.make_sequence(3, seed=4)
    path = str(tmp_path / "t.slam")
    sensors = slamfile.write_depth_sequence(path, depth, ts, (481.2, 480.0, 319.5, 239.5), gt_poses=gt)
    raw = open(path, "rb").read()
    assert raw[:5] == b"SLAM\0" and struct.unpack("<II", raw[5:13]) == (1, 4)
    # first sensor: u8 12 + "GroundTruth\0", u8 22 + "GroundTruthTrajectory\0", u8 index 0, u32 18 + description
    assert raw[13] == 12 and raw[14:26] == b"GroundTruth\0"
    assert raw[26] == 22 and raw[27:49] == b"GroundTruthTrajectory\0" and raw[49] == 0
    assert struct.unpack("<I", raw[50:54]) == (18,) and raw[54:72] == b"GroundTruthSensor\0"
    assert struct.unpack("<f", raw[72:76]) == (30.0,)                    # rate, then 64-byte pose, 4-byte delay
    header_gt = 13 + (1 + 12) + (1 + 22) + 1 + (4 + 18) + 4 + 64 + 4
    assert raw[header_gt] == 5 and raw[header_gt + 1:header_gt + 6] == b"Grey\0"
    # sizes: Camera block 36 + 60 bytes, Depth block 48 + 20 bytes; frames 9-byte header + payload
    common = lambda name, typ, desc: (1 + len(name) + 1) + (1 + len(typ) + 1) + 1 + (4 + len(desc) + 1) + 4 + 64 + 4
    header = 13 + common("GroundTruth", "GroundTruthTrajectory", "GroundTruthSensor") + common("Grey", "Camera", "Grey") + 96 \
        + common("RGB", "Camera", "RGB") + 96 + common("Depth", "Depth", "Depth") + 68
    per_step = (9 + 64) + (9 + 640 * 480) + (9 + 3 * 640 * 480) + (9 + 2 * 640 * 480)
    assert len(raw) == header + 3 * per_step
    sec, nsec, idx = struct.unpack("<IIB", raw[header:header + 9])
    assert (sec, nsec, idx) == (0, 0, 0)
    sensor, d, t, g, gts = slamfile.read_depth_sequence(path)
    assert sensor.name == "Depth" and (sensor.width, sensor.height) == (640, 480)
    assert sensor.disparity_params == (np.float32(0.001), 0.0) and sensor.pixel_format == 5     # D_I_16
    assert np.allclose(np.array(sensor.intrinsics) * [640, 480, 640, 480], (481.2, 480.0, 319.5, 239.5), atol=1e-4)
    assert np.array_equal(d, depth) and np.allclose(t, ts, atol=1e-9) and np.allclose(gts, ts, atol=1e-9)
    assert np.allclose(g, gt, atol=1e-6)
    with open(path, "rb") as f:
        ss = slamfile.read_slam_header(f)
        kinds = [s.type for _, s, _ in slamfile.iter_frames(f, ss)]
    assert kinds[:4] == ["GroundTruthTrajectory", "Camera", "Camera", "Depth"] and len(kinds) == 12


def test_slam_reader_refuses_what_it_cannot_size(tmp_path):
    from slambench_amd import slamfile
    bad = tmp_path / "bad.slam"
    bad.write_bytes(b"SLAN\0" + bytes(8))
    with pytest.raises(ValueError, match="magic"):
        slamfile.read_slam_header(open(bad, "rb"))
    odd = tmp_path / "odd.slam"
    with open(odd, "wb") as f:
        f.write(b"SLAM\0" + (1).to_bytes(4, "little") + (1).to_bytes(4, "little"))
        f.write(bytes([6]) + b"Sonar\0" + bytes([6]) + b"Sonar\0" + bytes([0]) + (1).to_bytes(4, "little") + b"\0" + bytes(72))
    with pytest.raises(ValueError, match="does not register"):
        slamfile.read_slam_header(open(odd, "rb"))


def _reference_shaped_slam(path, depth, ts, gt, intr, with_extras):
    """A .slam file in the shape the reference's generators give their files: a TUM file's default sensors (RGB,
    Grey, Depth, ground truth, Accelerometer: tools/dataset-tools/include/TUM.h:178-183) plus, with_extras, every
    other sensor type the reference registers, with frames of all of them mixed into the stream."""
    from slambench_amd import slamfile as sf
    n, h, w = depth.shape
    k = (intr[0] / w, intr[1] / h, intr[2] / w, intr[3] / h)
    sensors = [sf.Sensor("GroundTruth", sf.GROUND_TRUTH, description="GroundTruthSensor"),
               sf.Sensor("Grey", sf.CAMERA, rate=30, pixel_format=sf.PIXEL_FORMATS["G_I_8"], width=w, height=h, intrinsics=k),
               sf.Sensor("RGB", sf.CAMERA, rate=30, pixel_format=sf.PIXEL_FORMATS["RGB_III_888"], width=w, height=h, intrinsics=k),
               sf.Sensor("Depth", sf.DEPTH, rate=30, pixel_format=sf.PIXEL_FORMATS["D_I_16"], width=w, height=h, intrinsics=k,
                         disparity_params=(0.001, 0.0))]
    if with_extras:
        sensors += [sf.Sensor("Accelerometer", sf.ACCELEROMETER, block=tuple(range(18))),
                    sf.Sensor("PointCloud", sf.POINT_CLOUD, description="Ground truth point cloud"),
                    sf.Sensor("Gyro", sf.GYRO), sf.Sensor("IMU", sf.IMU, block=(1.0,) * 8), sf.Sensor("Odom", sf.ODOM),
                    sf.Sensor("Lidar", sf.LIDAR, block=(1e5, 1.7, -3.14, 3.14, -0.4, 0.3, 0.2, 0.4, 64)),
                    sf.Sensor("Events", sf.EVENT_CAMERA, width=346, height=260)]
    idx = {s.name: i for i, s in enumerate(sensors)}
    rng = np.random.default_rng(5)

    def frames():
        if with_extras:                                     # the whole-scene cloud comes first (ICLNUIM.cpp:394-406)
            pts = rng.normal(size=(1000, 3)).astype(np.float32)
            yield 0.0, idx["PointCloud"], (1000).to_bytes(4, "little") + pts.tobytes()
        for i in range(n):
            t = float(ts[i])
            yield t, idx["GroundTruth"], np.asarray(gt[i], np.float32).T.tobytes()
            if with_extras:                                 # inertial sensors tick several times per image
                for j in range(3):
                    yield t + 0.003 * j, idx["Accelerometer"], rng.normal(size=3).astype(np.float32).tobytes()
                yield t + 0.001, idx["Gyro"], bytes(12)
                yield t + 0.001, idx["IMU"], bytes(24)
                yield t + 0.002, idx["Odom"], bytes(52)
                yield t + 0.002, idx["Lidar"], bytes(16 * (i + 1))
                yield t + 0.002, idx["Events"], bytes(5)    # 346 x 260: two-byte coordinates
            yield t, idx["Grey"], bytes(w * h)
            yield t, idx["RGB"], bytes(3 * w * h)
            yield t, idx["Depth"], depth[i].tobytes()

    sf.write_slam(path, sensors, frames())
    return sensors


def test_slam_reader_steps_over_every_reference_sensor(tmp_path):
    """Files as the reference writes them by default: the Accelerometer of its TUM generator, the PointCloud of its
    ICL-NUIM generator, and the rest of its sensor database.  The depth stream read through them is the depth stream."""
    from slambench_amd import slamfile
    depth, gt, ts = synth.make_sequence(3, seed=3)
    intr = (481.2, 480.0, 319.5, 239.5)
    plain, full = str(tmp_path / "plain.slam"), str(tmp_path / "full.slam")
    _reference_shaped_slam(plain, depth, ts, gt, intr, with_extras=False)
    sensors = _reference_shaped_slam(full, depth, ts, gt, intr, with_extras=True)
    a = slamfile.read_depth_sequence(plain)
    b = slamfile.read_depth_sequence(full)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[1], depth)
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
    with open(full, "rb") as f:
        back = slamfile.read_slam_header(f)
        assert [s.type for s in back] == [s.type for s in sensors]
        by_name = {s.name: s for s in back}
        assert by_name["Accelerometer"].block == tuple(float(v) for v in range(18))
        assert by_name["Lidar"].block[-1] == 64 and by_name["Events"].frame_size == 5
        assert by_name["PointCloud"].variable_size and by_name["PointCloud"].is_ground_truth
        assert not by_name["Accelerometer"].variable_size and by_name["Accelerometer"].frame_size == 12
        kinds = collections.Counter(s.type for _, s, _ in slamfile.iter_frames(f, back))
    assert kinds == {"PointCloud": 1, "GroundTruthTrajectory": 3, "Accelerometer": 9, "Gyro": 3, "IMU": 3, "Odom": 3,
                     "Lidar": 3, "EventCamera": 3, "Camera": 6, "Depth": 3}
    # byte-level check of one sensor block against the reference's layout: name, type, index, description, rate,
    # pose, delay, then 12 + 3 + 3 floats (AccelerometerSensor.cpp:30-35)
    raw = open(full, "rb").read()
    at = raw.index(b"\x0eAccelerometer\0\x0eAccelerometer\0")
    tail = raw[at + 30:]
    assert tail[0] == 4 and tail[1:5] == (1).to_bytes(4, "little") and tail[5:6] == b"\0"
    block = np.frombuffer(tail[6 + 4 + 64 + 4:6 + 4 + 64 + 4 + 72], np.float32)
    assert np.array_equal(block, np.arange(18, dtype=np.float32))


def test_rpe_definition():
    """RPE_RMSE of the reference: consecutive estimates, translation of (E_i E_j^-1)(G_i G_j^-1)^-1."""
    from slambench_amd.metrics import rpe, frame_metrics
    depth, gt, ts = synth.make_sequence(8, seed=2)
    assert rpe(gt, gt, ts, ts) < 1e-12                      # a perfect trajectory has no relative error
    assert rpe(gt[:1], gt, ts[:1], ts) == 0.0               # fewer than two poses: 0 by definition
    est = gt.copy()
    est[5:, 0, 3] += 0.01                                   # one 1 cm jump between frames 4 and 5
    want = np.sqrt(sum(np.linalg.norm(((est[i - 1] @ np.linalg.inv(est[i])) @ np.linalg.inv(gt[i - 1] @ np.linalg.inv(gt[i])))[:3, 3]) ** 2
                       for i in range(1, 8)) / 7)
    assert abs(rpe(est, gt, ts, ts) - want) < 1e-12
    assert abs(want - 0.01 / np.sqrt(7)) < 2e-4             # the jump, seen through the rotation of the pair
    rows = frame_metrics(est, gt, ts, ts)
    assert len(rows) == 8 and set(rows[0]) == {"ATE_RMSE", "MaxATE", "MeanATE", "OrientationError", "RPE_RMSE"}
    assert rows[0]["RPE_RMSE"] == 0.0 and rows[-1]["MaxATE"] >= rows[-1]["MeanATE"] > 0


def test_log_writer_matches_the_reference_log_shape():
    """Same sections, separators and number formatting as the reference's log files (checked against the
    kept head of violons_libkfusion-cpp_living_room_traj2_loop.log)."""
    import os
    ref = open(os.path.join(os.path.dirname(__file__), "golden", "ref_kfusion_cpp_lr_kt2_head.tsv")).read()
    rprops, rcols, rrows = harness.parse_log(ref)
    phases = [c[len("Duration_"):] for c in rcols if c.startswith("Duration_") and c != "Duration_Frame"]
    assert phases == ["Preprocessing", "Tracking", "Integration", "Raycasting", "Render"]
    n = 5
    ts = rrows[:n, rcols.index("Timestamp")]
    pos = rrows[:n, -3:]
    dur = rrows[:n, rcols.index("Duration_Frame"):rcols.index("Duration_Frame") + 6]
    text = harness.format_log({"mu": "0.1", "volume-size": "8,8,8"}, ts, pos, dur, phases)
    props, cols, rows = harness.parse_log(text)
    assert props == {"mu": "0.1", "volume-size": "8,8,8"}
    assert cols == ["Frame Number", "Timestamp", "Duration_Frame"] + ["Duration_" + p for p in phases] + ["X", "Y", "Z"]
    assert np.array_equal(rows[:, 1], ts) and np.array_equal(rows[:, -3:], pos)
    # a data row of ours and the reference's row agree character for character on the shared columns
    ref_row = [l for l in ref.splitlines() if l.startswith("5\t")][0].split("\t")
    our_row = [l for l in text.splitlines() if l.startswith("5\t")][0].split("\t")
    assert our_row[:2] == ref_row[:2] and our_row[-3:] == ref_row[-3:]
    assert our_row[2:8] == ref_row[rcols.index("Duration_Frame"):rcols.index("Duration_Frame") + 6]
    assert "\nStatistics:\n=================\n\nFrame Number\t" in text and text.startswith("Properties:\n=================\n\n")
    # with ground truth the accuracy columns sit between the timestamp and the durations
    acc = [{c: 0.5 for c in harness.ACCURACY_COLUMNS}] * n
    cols2 = harness.parse_log(harness.format_log({}, ts, pos, dur, phases, acc, np.zeros((n, 2))))[1]
    assert cols2[2:7] == list(harness.ACCURACY_COLUMNS) and cols2[-5:-3] == ["CPU_Memory", "GPU_Memory"]


# ------------------------------------------------------------------------------------------ raw dataset importers
def _write_tum_folder(root, depth, stamps, gt_rows):
    import os
    from slambench_amd import datasets
    os.makedirs(os.path.join(root, "depth"))
    with open(os.path.join(root, "depth.txt"), "w") as f:
        f.write("# depth maps\n# timestamp filename\n")
        for img, ts in zip(depth, stamps):
            name = "depth/%s.png" % ts
            with open(os.path.join(root, name), "wb") as g:
                g.write(datasets.encode_png_grey16(img))
            f.write("%s %s\n" % (ts, name))
    with open(os.path.join(root, "groundtruth.txt"), "w") as f:
        f.write("# ground truth trajectory\n# timestamp tx ty tz qx qy qz qw\n")
        for row in gt_rows:
            f.write(" ".join(row) + "\n")


def test_tum_folder_import(tmp_path):
    from scipy.spatial.transform import Rotation
    from slambench_amd import datasets, slamfile
    rng = np.random.default_rng(0)
    depth = rng.integers(0, 40000, (3, 480, 640)).astype(np.uint16)
    stamps = ["1305031102.160407", "1305031102.194330", "1305031102.2265"]
    q = Rotation.from_euler("xyz", [[0.1, -0.2, 0.3], [0.0, 0.5, -0.1]]).as_quat()      # x y z w
    gt_rows = [["1305031098.6659", "1.3563", "0.6305", "1.6380"] + ["%.4f" % v for v in q[0]],
               ["1305031098.6758", "1.3543", "0.6306", "1.6360"] + ["%.4f" % v for v in q[1]]]
    root = str(tmp_path / "rgbd_dataset_freiburg2_xyz")
    _write_tum_folder(root, depth, stamps, gt_rows)
    seq = datasets.load_tum(root)
    assert np.array_equal(seq.depth, depth)
    assert seq.intrinsics_px == datasets.TUM_DEPTH_INTRINSICS["freiburg2"] and seq.disparity_scale == 0.0002
    # the fraction is scaled by its number of digits (reference TUM.cpp:82-83)
    assert seq.timestamps == [(1305031102, 160407000), (1305031102, 194330000), (1305031102, 226500000)]
    assert seq.gt_timestamps[0] == (1305031098, 665900000)
    for row, pose in zip(gt_rows, seq.gt_poses):
        vals = [float(v) for v in row[1:]]
        # the importer does not normalise the 4-decimal quaternion, scipy does: 1e-3
        np.testing.assert_allclose(pose[:3, :3], Rotation.from_quat(vals[3:]).as_matrix(), atol=1e-3)
        np.testing.assert_allclose(pose[:3, 3], vals[:3], rtol=1e-7)
    out = str(tmp_path / "fr2.slam")
    datasets.to_slam(seq, out)
    sensor, d, ts, g, gts = slamfile.read_depth_sequence(out)
    assert np.array_equal(d, depth) and sensor.disparity_params[0] == np.float32(0.0002)
    np.testing.assert_allclose(ts, seq.seconds(), atol=1e-9)
    np.testing.assert_allclose(g, seq.gt_poses, atol=0)
    assert abs(sensor.intrinsics[0] * 640 - 580.8) < 1e-4
    with pytest.raises(FileNotFoundError):
        datasets.load_tum(str(tmp_path))
    assert datasets.load_tum(root, frame_limit=2).depth.shape[0] == 2


def test_png_decoder_handles_every_filter():
    import io
    from PIL import Image
    from slambench_amd import datasets
    rng = np.random.default_rng(1)
    y, x = np.mgrid[0:48, 0:64]
    smooth = (1000 + 37 * x + 11 * y + rng.integers(0, 3, (48, 64))).astype(np.uint16)    # makes the encoder pick filters
    buf = io.BytesIO()
    Image.fromarray(smooth).save(buf, format="PNG", optimize=True)
    data = buf.getvalue()
    filters = {zlib_line[0] for zlib_line in _png_lines(data)}
    assert filters - {0}, filters                                        # not only "none"
    assert np.array_equal(datasets._decode_png_plain(data), smooth)
    assert np.array_equal(datasets.decode_png(data), smooth)
    rgb = rng.integers(0, 255, (20, 30, 3)).astype(np.uint8)
    rgb[5:15] = rgb[4]                                                   # repeated rows -> "up" filter
    buf = io.BytesIO()
    Image.fromarray(rgb).save(buf, format="PNG")
    assert np.array_equal(datasets._decode_png_plain(buf.getvalue()), rgb)
    assert np.array_equal(datasets._decode_png_plain(datasets.encode_png_grey16(smooth)), smooth)
    with pytest.raises(ValueError):
        datasets._decode_png_plain(b"not a png at all")


def _png_lines(data):
    import struct, zlib
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        (n,), kind = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", data[pos + 8:pos + 8 + n])
        if kind == b"IDAT":
            idat.append(data[pos + 8:pos + 8 + n])
        pos += 12 + n
    raw = zlib.decompress(b"".join(idat))
    stride = hdr[0] * {0: 1, 2: 3, 6: 4}[hdr[3]] * hdr[2] // 8 + 1
    return [raw[i:i + stride] for i in range(0, len(raw), stride)]


def test_icl_nuim_folder_import(tmp_path):
    import math
    from slambench_amd import datasets
    rng = np.random.default_rng(2)
    root = str(tmp_path / "living_room_traj2_loop")
    import os
    os.makedirs(root)
    rays = []
    for n in range(2):
        ray = (1.0 + 3.0 * rng.random((480, 640))).astype(np.float32)
        rays.append(ray)
        with open(os.path.join(root, "scene_00_%04d.depth" % n), "w") as f:
            f.write(" ".join("%.5f" % v for v in ray.reshape(-1)))
        with open(os.path.join(root, "scene_00_%04d.txt" % n), "w") as f:
            f.write("cam_pos      = [%g, 1.3, -1.5]';\n" % (0.5 + n))
            f.write("cam_dir      = [0, 0, 2]';\ncam_up       = [0, 3, 0]';\ncam_lookat   = [0, 1.3, 1]';\n")
            f.write("cam_sky      = [0, 1, 0]';\ncam_right    = [1.33, 0, 0]';\ncam_fpoint   = [0, 0, 10]';\n")
            f.write("cam_angle    = 90;\n")
    seq = datasets.load_icl_nuim(root)
    assert seq.depth.shape == (2, 480, 640) and seq.disparity_scale == 0.001
    assert seq.timestamps == [(0, 0), (0, 40000000)]
    # planar depth of a few pixels, restated per pixel
    for (v, u) in [(0, 0), (239, 320), (479, 639), (100, 500)]:
        stored = float(np.float32("%.5f" % rays[1][v, u]))
        mm = int(np.float32(stored) * np.float32(1000))
        a = float((np.float32(u) - np.float32(319.5)) / np.float32(481.2))
        b = float((np.float32(v) - np.float32(239.5)) / np.float32(-480.0))
        assert seq.depth[1, v, u] == int(mm / math.sqrt(a * a + b * b + 1))
    assert seq.depth[0, 239, 320] < rays[0][239, 320] * 1000 + 1             # centre pixel: almost unchanged
    assert seq.depth[0, 0, 0] < 0.8 * rays[0][0, 0] * 1000                    # corner: clearly shorter
    want = np.eye(4, dtype=np.float32)
    want[:3, 3] = [1.5, 1.3, -1.5]
    np.testing.assert_allclose(seq.gt_poses[1], want, atol=1e-7)
    assert seq.intrinsics_px[1] == -480.0
    flipped = datasets.load_icl_nuim(root, positive_focal=True)
    assert flipped.intrinsics_px[1] == 480.0
    assert flipped.gt_poses[0][0, 3] == np.float32(-0.5)                      # mirrored position
    assert np.allclose(flipped.gt_poses[0][:3, :3], np.diag([-1, -1, 1]), atol=1e-6)   # upside down, same z
    with pytest.raises(FileNotFoundError):
        datasets.load_icl_nuim(str(tmp_path))


# ------------------------------------------------------------------------------------------ design-space sweeps
def test_dse_dump_lists_the_plugin_parameters(tmp_path):
    """`--dse` needs no GPU: the harness stops after the plugin has declared its parameters."""
    from slambench_amd import dse, harness, synth
    depth, gt, ts = synth.make_sequence(1, seed=1)
    path = str(tmp_path / "s.kfd")
    harness.write_kfd(path, depth, ts, (481.2, 480.0, 319.5, 239.5))
    r = harness.run_mini_harness(path, ["--dse", "-c", "2"])
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("libs:")
    params = dse.parse_dse(r.stdout)
    assert list(params)[:4] == ["compute-size-ratio", "integration-rate", "tracking-rate", "rendering-rate"]
    assert params["compute-size-ratio"] == {"name": "compute-size-ratio", "short-option": "c",
                                            "description": "Compute ratio", "type": "i", "default": "1"}
    assert params["volume-resolution"]["default"] == "256,256,256"
    assert params["mu"]["default"] == "0.1" and params["pyramid-level1"]["default"] == "10"


def test_sweep_scheduler_keeps_one_process_per_gpu(tmp_path):
    import sys, time
    from slambench_amd import dse
    points = dse.grid_points({"mu": ["0.05", "0.1", "0.2"], "volume-resolution": ["64", "128"]})
    assert len(points) == 6 and points[0] == {"mu": "0.05", "volume-resolution": "64"} and points[1]["volume-resolution"] == "128"
    marker = str(tmp_path)
    code = ("import os,sys,time; g=os.environ['HIP_VISIBLE_DEVICES']; p=os.path.join(sys.argv[1], 'busy'+g); "
            "assert not os.path.exists(p), 'two processes on one GPU'; open(p,'w').close(); time.sleep(0.2); "
            "os.remove(p); print(g, sys.argv[2])")
    t0 = time.time()
    res = dse.run_sweep(points, lambda pt: [sys.executable, "-c", code, marker, pt["mu"] + "/" + pt["volume-resolution"]],
                        gpus=[0, 1, 2], summarise_fn=lambda text: {"echo": text.split()[1]})
    assert time.time() - t0 < 6 * 0.2 + 3                       # ran concurrently
    assert all(r.returncode == 0 for r in res), [r.stderr for r in res]
    assert [r.summary["echo"] for r in res] == [p["mu"] + "/" + p["volume-resolution"] for p in points]
    assert {r.gpu for r in res} == {0, 1, 2}
    assert all(r.stdout.split()[0] == str(r.gpu) for r in res)
    bad = dse.run_sweep([{}], lambda pt: [sys.executable, "-c", "import sys; sys.exit(3)"], gpus=[0])
    assert bad[0].returncode == 3 and bad[0].summary == {}


def test_plugin_options_behind_the_harness_own_dash_s_need_a_library_identifier(tmp_path):
    """The reference's option table stops registering a library's options at the first name that is already taken
    (framework/shared/src/ParameterManager.cpp:115-131), and the harness owns "-s" (start-frame,
    framework/shared/src/SLAMBenchConfiguration.cpp:68): loaded as a plain file name, the plugin keeps the options it
    declared before volume-size ("-s") and loses -d, -v, -y1..3 and --devices, short and long.  Loaded as
    file.so=kf every option exists as -kf-x / --kf-name.  (`--dse` stops before anything needs a GPU.)"""
    from slambench_amd import harness, synth
    depth, gt, ts = synth.make_sequence(1, seed=1)
    path = str(tmp_path / "s.kfd")
    harness.write_kfd(path, depth, ts, (481.2, 480.0, 319.5, 239.5))

    def run(args, library_id):
        return harness.run_mini_harness(path, ["--dse"] + args, library_id=library_id)

    # no identifier: declared before "-s" -> known; the long name of volume-size went in before its short name collided
    for args in (["-c", "2"], ["--mu", "0.2"], ["--volume-size", "4,4,4"]):
        r = run(args, None)
        assert r.returncode == 0, (args, r.stderr)
        assert "*** Duplicated short option replaced: 's'" in r.stderr
    # ... and everything declared after it is gone
    for args in (["-d", "2,2,2"], ["-v", "128"], ["--volume-resolution", "128"], ["-y1", "5"], ["--pyramid-level3", "2"],
                 ["--devices", "2"]):
        r = run(args, None)
        assert r.returncode == 1 and "unknown option " + args[0] in r.stderr, (args, r.stderr)
    # "-s" itself is the harness's start-frame, not the plugin's volume-size
    r = run(["-s", "4,4,4"], None)
    assert r.returncode == 0 and "default:8,8,8" in r.stdout
    # with an identifier nothing collides
    r = run(["-v", "128", "-d", "2,2,2", "-y1", "5", "--devices", "2", "-s", "4"], "kf")
    assert r.returncode == 0 and "Duplicated" not in r.stderr, r.stderr
    assert r.stdout.startswith("libs:kf\n") and "argument:kf-volume-resolution" in r.stdout
    # the plain names do not exist then
    r = harness.run_mini_harness(path, ["--dse", "-v", "128"], library_id=None, library=harness.PLUGIN_PATH + "=kf")
    assert r.returncode == 1 and "unknown option -v" in r.stderr
