"""Host-side helpers of the benchmark and of the profile tooling (no GPU): the frame order of the synthetic stream,
the roofline record of the tracking launches read from a kernel-statistics table, the frame count of a kernel trace
(one tracking-finish launch per frame) and the per-frame footer built on it, and the pixel shares of a split image."""
import csv
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _write_stats(path, rows):
    cols = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"]
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(cols)
        for name, calls, avg in rows:
            w.writerow([name, calls, int(calls * avg), avg, 0.0, int(avg * 0.8), int(avg * 1.5), 0.0])


def test_ping_pong_visits_neighbours_only():
    order = bench.ping_pong(5, 20)
    assert order[:9] == [0, 1, 2, 3, 4, 3, 2, 1, 0]
    assert all(abs(a - b) == 1 for a, b in zip(order, order[1:]))
    assert len(order) == 20 and max(order) == 4


def test_frames_of_a_trace_are_its_tracking_finish_launches(tmp_path):
    mk = _load(os.path.join(ROOT, "tools", "make_profiles.py"), "make_profiles")
    d = tmp_path / "kt"
    d.mkdir()
    rows = [("void kfh::k_icp_iteration<true, 7, 3>(kfh::IcpArgs)", 6100, 5700.0),
            ("kfh::k_icp_finish(kfh::FinishArgs)", 600, 6000.0),
            ("kfh::k_track_skip(kfh::Ctrl*)", 10, 2000.0),
            ("kfh::k_integrate(kfh::IntegrateArgs)", 305, 27000.0),
            ("kfh::k_stage_copy(void*, void const*, int)", 300, 16000.0)]
    _write_stats(str(d / "x_kernel_stats.csv"), rows)
    with open(str(d / "x_kernel_stats.csv")) as f:
        assert mk.frames_in_trace(list(csv.DictReader(f))) == 610
    _, text = mk.kernel_summary(str(d), 0)
    total_ns = sum(int(c * a) for _, c, a in rows)
    footer = text.strip().splitlines()[-1]
    assert "over 610 frames" in footer
    assert "%.1f us per frame" % (total_ns / 610 / 1e3) in footer
    assert "%.1f us without the upload kernels" % ((total_ns - 300 * 16000) / 610 / 1e3) in footer
    # an explicit frame count overrides the trace's
    _, text = mk.kernel_summary(str(d), 300)
    assert "over 300 frames" in text


def test_track_roofline_prices_each_level_with_its_own_pixels(tmp_path):
    path = str(tmp_path / "stats.csv")
    _write_stats(path, [("void kfh::k_icp_iteration<true, 7, 3>(kfh::IcpArgs)", 1000, 6000.0),
                        ("void kfh::k_icp_iteration<true, 5, 1>(kfh::IcpArgs)", 500, 6500.0),
                        ("void kfh::k_icp_iteration<true, 3, 1>(kfh::IcpArgs)", 400, 6400.0)])
    r = bench.track_roofline(path, (640, 480), 0.125)
    lv = r["levels"]
    assert lv["level0"]["pixels"] == 640 * 480 and lv["level1"]["pixels"] == 320 * 240 and lv["level2"]["pixels"] == 160 * 120
    want = 36.0 * 640 * 480 / 6000e-9 / 1e9
    assert abs(lv["level0"]["achieved"] - want) < 1e-6 * want
    assert abs(r["frac"] - want / 8000.0) < 1e-9 and r["tracking_ms_per_frame_this_run"] == 0.125
    # another compute size has other block shapes: nothing is claimed for it
    assert bench.track_roofline(path, (320, 240), 0.1)["levels"] is None


def test_shares_of_a_split_image_cover_every_pixel_once():
    """The rule kfh_group uses to split the image among its members (restated: a share is the pixel count divided by
    the members, rounded up to whole wavefronts; the last members may get less, or nothing)."""
    for npix, n in ((640 * 480, 8), (320 * 240, 3), (80 * 60, 7), (80 * 60, 16), (5 * 5, 4)):
        share = ((npix + n - 1) // n + 63) // 64 * 64
        firsts = [min(i * share, npix) for i in range(n)]
        counts = [min(share, npix - f) for f in firsts]
        assert all(c >= 0 for c in counts) and sum(counts) == npix
        covered = np.zeros(npix, int)
        for f, c in zip(firsts, counts):
            covered[f:f + c] += 1
        assert (covered == 1).all()
