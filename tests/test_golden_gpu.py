"""The HIP path against the committed golden vectors (tests/golden/kf_ref_golden.npz, written by the oracle at
64x48 pixels / 64^3 voxels): the same comparison as the oracle-vs-HIP tests, but against numbers that are in
the repository, so a change of the oracle and a change of the kernels cannot cancel out unnoticed."""
import ctypes as C
import os

import numpy as np
import pytest

from slambench_amd import native, synth
from slambench_amd.config import KFusionConfig
from slambench_amd.kfusion import KFusion
from tests.util import hp, vp

pytestmark = pytest.mark.gpu

INTR = (481.2, 480.0, 319.5, 239.5)
K64 = [v * 64 / 640 for v in INTR]


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "kf_ref_golden.npz"))


@pytest.fixture(scope="module")
def L():
    return native.lib()


def test_preprocessing_kernels(torch_cuda, L, gold):
    torch = torch_cuda
    d_in = torch.from_numpy(gold["depth_u16"][5].view(np.int16)).cuda()
    raw = torch.empty((48, 64), dtype=torch.float32, device="cuda")
    filt = torch.empty_like(raw)
    native.check(L.kfh_k_preprocess(vp(raw), vp(filt), 64, 48, vp(d_in), 64, 48, 1000.0, None))
    assert np.array_equal(raw.cpu().numpy(), gold["mm2meters"])
    np.testing.assert_allclose(filt.cpu().numpy(), gold["bilateral"], rtol=1e-5)
    half = torch.empty((24, 32), dtype=torch.float32, device="cuda")
    native.check(L.kfh_k_preprocess(vp(half), None, 32, 24, vp(d_in), 64, 48, 1000.0, None))
    assert np.array_equal(half.cpu().numpy(), gold["mm2meters_half"])
    # from here on the golden bilateral image is the input, so the results are bit-exact again
    g_filt = torch.from_numpy(gold["bilateral"]).cuda()
    hs = torch.empty((24, 32), dtype=torch.float32, device="cuda")
    native.check(L.kfh_k_half_sample(vp(hs), vp(g_filt), 64, 48, float(np.float32(0.1) * 3), None))
    assert np.array_equal(hs.cpu().numpy(), gold["half_sample"])
    v = torch.empty((48, 64, 4), dtype=torch.float32, device="cuda")
    n = torch.empty_like(v)
    native.check(L.kfh_k_vertex_normal(vp(v), vp(n), vp(g_filt), 64, 48, *[float(np.float32(x)) for x in K64], None))
    assert np.array_equal(v.cpu().numpy()[..., :3], gold["depth2vertex"])
    want_n = gold["vertex2normal"]
    ok = want_n[..., 0] != -2
    got_n = n.cpu().numpy()[..., :3]
    assert np.array_equal(got_n[..., 0] == -2, ~ok)
    np.testing.assert_allclose(got_n[ok], want_n[ok], rtol=1e-6, atol=1e-7)


def test_solve_and_pose_update(torch_cuda, L, gold):
    torch = torch_cuda
    sums = torch.from_numpy(gold["reduce_f7"].astype(np.float32)).cuda()
    pose = torch.eye(4, dtype=torch.float32, device="cuda")
    x = torch.zeros(6, dtype=torch.float64, device="cuda")
    conv = torch.zeros(2, dtype=torch.int32, device="cuda")
    native.check(L.kfh_k_update_pose(vp(pose), vp(x), vp(conv), vp(sums), 1e-5, None))
    torch.cuda.synchronize()
    want = gold["solve_f7"]
    np.testing.assert_allclose(x.cpu().numpy(), want, rtol=1e-6, atol=1e-12 + 1e-8 * np.abs(want).max())


def test_pipeline_and_pictures(torch_cuda, gold):
    depth, gt, ts = synth.make_sequence(30, seed=1, intrinsics=INTR, size=(64, 48), noise=True)
    assert np.array_equal(depth[:8], gold["depth_u16"])            # the stream the vectors were made from
    cfg = KFusionConfig(input_size=(64, 48), intrinsics=tuple(K64), volume_resolution=(64, 64, 64), mu=0.3)
    kf = KFusion(cfg)
    kf.enable_track_render(True)
    flags = []
    for i in range(30):
        kf.process_frame(depth[i])
        if i == 7:
            vol = kf.download(native.BUF_VOLUME)
            mv = kf.download(native.BUF_MODEL_VERTEX)[..., :3]
            pictures = [kf.render(native.RENDER_DEPTH), kf.render(native.RENDER_TRACK), kf.render(native.RENDER_VOLUME)]
    for f in range(30):
        st = kf.status(f)
        flags.append(int(st["tracked"]) | int(st["integrated"]) << 1 | int(st["raycasted"]) << 2)
        assert np.linalg.norm(st["pose"][:3, 3] - gold["pipeline_poses"][f][:3, 3]) < 2e-4, f
    assert flags == gold["pipeline_flags"].tolist()
    # frame 7: the pose already differs from the oracle's in the last bits, so a few voxels / pixels that sit on
    # a rounding boundary may differ by one step
    dv = np.abs(vol.astype(np.int32) - gold["volume_f7"].astype(np.int32))
    assert dv[..., 1].max() == 0 and dv[..., 0].max() <= 2 and (dv[..., 0] > 0).mean() < 0.02
    hit = gold["model_normal_f7"][..., 0] != -2
    assert hit.mean() > 0.9
    assert np.abs(mv - gold["model_vertex_f7"])[hit].max() < 1e-3
    assert np.array_equal(pictures[0], gold["render_depth_f7"])     # raw depth: bit-exact
    assert (pictures[1] != gold["render_track_f7"]).any(axis=-1).mean() < 0.01
    dp = np.abs(pictures[2].astype(np.int32) - gold["render_volume_f7"].astype(np.int32))
    assert (dp.max(axis=-1) > 2).mean() < 0.01
    kf.close()
