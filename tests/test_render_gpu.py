"""GUI images of the plugin (renderDepthKernel, renderTrackKernel, renderVolumeKernel): HIP against the oracle.

The images are bytes computed from maps that are themselves held bit-exact, so the bar is bit-exact RGBA.
The one exception is the tracking image of a running context: the ICP poses of the two implementations
differ in the last bits (summation order), so a handful of pixels that sit on a threshold may get another
result code; the test allows five of them.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import kf_ref
from slambench_amd import native, synth
from slambench_amd.config import KFusionConfig
from slambench_amd.kfusion import KFusion
from tests.util import hp, m4, make_depth, oracle_config, pad4, vp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    return native.lib()


@pytest.fixture(scope="module")
def scene():
    cfg = KFusionConfig(compute_size_ratio=4, volume_resolution=(96, 96, 96))
    depth, gt, _ = synth.make_sequence(8, seed=2, noise=True)
    k = kf_ref.KFusionRef(oracle_config(cfg))
    for i in range(8):
        k.process_frame(depth[i])
    return {"cfg": cfg, "depth": depth, "gt": gt, "ref": k}


def test_render_depth_bit_exact(torch_cuda, L):
    torch = torch_cuda
    h, w = 120, 160
    d = kf_ref.mm2meters(make_depth(480, 640, seed=4), w, h, 1000.0)
    # the branch boundaries: below near, exactly near, exactly far (sextant 6 -> black), beyond far, holes
    d[0, :8] = [0.0, 0.39999998, 0.4, 0.40000004, 3.9999998, 4.0, 4.0000005, 7.5]
    d[1] = np.linspace(0.4, 4.0, w, dtype=np.float32)
    want = kf_ref.render_depth(d, 0.4, 4.0)
    out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    native.check(L.kfh_k_render_depth(vp(out), vp(torch.from_numpy(d).cuda()), w, h, 0.4, 4.0, None))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert len(np.unique(want.reshape(-1, 4), axis=0)) > 50
    assert np.array_equal(got, want)


def test_render_track_bit_exact(torch_cuda, L, scene):
    torch = torch_cuda
    ref, cfg = scene["ref"], scene["cfg"]
    iv, inn, rv, rn = ref.input_vertex(0).copy(), ref.input_normal(0).copy(), ref.model_vertex().copy(), ref.model_normal().copy()
    h, w = iv.shape[:2]
    pose = kf_ref.mat_mul(kf_ref.se3_exp(np.array([0.03, -0.02, 0.02, 0.02, -0.015, 0.01])), ref.pose.copy())
    view = kf_ref.mat_mul(kf_ref.camera_matrix(cfg.camera), kf_ref.mat_inverse(ref.pose))
    td_want = kf_ref.track(iv, inn, rv, rn, pose, view)
    want = kf_ref.render_track(td_want)
    codes = set(np.unique(td_want["result"]).tolist())
    assert {1, -1, -4}.issubset(codes), codes                     # several colours are exercised
    t = [torch.from_numpy(pad4(a)).cuda() for a in (iv, inn, rv, rn)]
    d_l = torch.from_numpy(ref.scaled_depth(0).copy()).cuda()
    k_l = np.array(cfg.camera, np.float32)
    sums = torch.empty(32, dtype=torch.float32, device="cuda")
    td = torch.zeros((h, w, 8), dtype=torch.float32, device="cuda")
    native.check(L.kfh_k_track_reduce(vp(sums), vp(td), w, vp(d_l), hp(k_l), vp(t[1]), w, h, vp(t[2]), vp(t[3]),
                                      w, h, hp(m4(pose)), hp(m4(view)), 0.1, 0.8, None))
    out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    native.check(L.kfh_k_render_track(vp(out), vp(td), w, w, h, None))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)
    # every code, the unknown ones included, and a row stride wider than the image
    raw = np.zeros((4, 16), kf_ref.TRACK_DTYPE)
    raw["result"][:, :9] = [1, -1, -2, -3, -4, -5, 0, 7, -6]
    want = kf_ref.render_track(np.ascontiguousarray(raw[:, :12]))
    d_raw = torch.from_numpy(raw.view(np.float32).reshape(4, 16, 8).copy()).cuda()
    out = torch.zeros((4, 12, 4), dtype=torch.uint8, device="cuda")
    native.check(L.kfh_k_render_track(vp(out), vp(d_raw), 16, 12, 4, None))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)
    assert L.kfh_k_render_track(vp(out), vp(d_raw), 8, 12, 4, None) == 1     # stride below the width


@pytest.mark.parametrize("far", [4.0, 8.0])
def test_render_volume_bit_exact(torch_cuda, L, scene, far):
    torch = torch_cuda
    ref, cfg = scene["ref"], scene["cfg"]
    w, h = cfg.compute_size
    vol_np = ref.volume().copy()
    res, dim = cfg.volume_resolution, cfg.volume_size
    view = kf_ref.mat_mul(ref.pose, kf_ref.inverse_camera_matrix(cfg.camera))
    step = np.float32(min(dim)) / np.float32(max(res))
    large = float(np.float32(0.75) * np.float32(0.1))
    light, ambient = (1.0, 1.0, -1.0), (0.1, 0.1, 0.1)
    want = kf_ref.render_volume(vol_np, dim, w, h, view, 0.4, far, step, large, light, ambient)
    assert (want[..., :3].max(axis=-1) > 0).mean() > 0.5              # most rays hit something
    assert len(np.unique(want[..., 0])) > 30
    out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    native.check(L.kfh_k_render_volume(vp(out), w, h, vp(torch.from_numpy(vol_np).cuda()), (C.c_int32 * 3)(*res),
                                       (C.c_float * 3)(*dim), hp(m4(view)), 0.4, far, float(step), large,
                                       (C.c_float * 3)(*light), (C.c_float * 3)(*ambient), None))
    assert np.array_equal(out.cpu().numpy(), want)


def test_context_renders_match_oracle(torch_cuda):
    """Frames 0-3 run with the initial pose, so depth, volume and the (empty-model) tracking codes are bit-exact."""
    cfg = KFusionConfig(compute_size_ratio=4, volume_resolution=(128, 128, 128))
    depth, gt, ts = synth.make_sequence(6, seed=3, noise=True)
    ref = kf_ref.KFusionRef(oracle_config(cfg))
    kf = KFusion(cfg)
    with pytest.raises(native.KfhError):
        kf.render(native.RENDER_TRACK)                                # codes are not being kept yet
    kf.enable_track_render(True)
    unknown = np.array([255, 128, 128, 0], np.uint8)
    assert (kf.render(native.RENDER_TRACK) == unknown).all()          # nothing tracked so far
    for i in range(4):
        ref.process_frame(depth[i])
        kf.process_frame(depth[i])
    w, h = cfg.compute_size
    res, dim = cfg.volume_resolution, cfg.volume_size
    step = np.float32(min(dim)) / np.float32(max(res))
    large = float(np.float32(0.75) * np.float32(cfg.mu))
    assert np.array_equal(kf.render(native.RENDER_DEPTH), kf_ref.render_depth(ref.float_depth(), 0.4, 4.0))
    assert np.array_equal(kf.render(native.RENDER_TRACK), kf_ref.render_track(ref.tracking_result()))
    view = kf_ref.mat_mul(ref.pose, kf_ref.inverse_camera_matrix(cfg.camera))
    want = kf_ref.render_volume(ref.volume(), dim, w, h, view, 0.4, 8.0, step, large)
    assert (want[..., :3].max(axis=-1) > 0).mean() > 0.3
    assert np.array_equal(kf.render(native.RENDER_VOLUME), want)
    # a free viewpoint (Kfusion::setViewPose): half a metre to the side and turned a little
    free = kf_ref.mat_mul(ref.pose, kf_ref.se3_exp(np.array([0.5, -0.2, 0.1, 0.0, 0.15, 0.05])))
    view = kf_ref.mat_mul(free, kf_ref.inverse_camera_matrix(cfg.camera))
    want = kf_ref.render_volume(ref.volume(), dim, w, h, view, 0.4, 8.0, step, large)
    assert np.array_equal(kf.render(native.RENDER_VOLUME, free), want)
    # rendering does not disturb the pipeline: the next frames track against a real model
    for i in range(4, 6):
        ref.process_frame(depth[i])
        kf.process_frame(depth[i])
    assert kf.status(5)["tracked"]
    assert np.linalg.norm(kf.pose[:3, 3] - ref.pose[:3, 3]) < 1e-5
    got, want = kf.render(native.RENDER_TRACK), kf_ref.render_track(ref.tracking_result())
    assert (want == np.array([128, 128, 128, 0], np.uint8)).all(axis=-1).mean() > 0.5
    assert (got != want).any(axis=-1).sum() <= 5
    kf.reset()
    assert (kf.render(native.RENDER_TRACK) == unknown).all()
    kf.enable_track_render(False)
    with pytest.raises(native.KfhError):
        kf.render(native.RENDER_TRACK)
    kf.close()


def test_render_argument_errors(torch_cuda):
    cfg = KFusionConfig(compute_size_ratio=8, volume_resolution=(32, 32, 32))
    kf = KFusion(cfg)
    with pytest.raises(native.KfhError) as e:
        kf.render(7)
    assert e.value.code == 1 and "unknown image" in str(e.value)
    L = native.lib()
    small = np.zeros(16, np.uint8)
    rc = L.kfh_render(kf._ctx, native.RENDER_DEPTH, None, small.ctypes.data_as(C.c_void_p), C.c_size_t(small.nbytes))
    assert rc == 1 and b"bytes" in L.kfh_last_error()
    assert L.kfh_render(kf._ctx, native.RENDER_DEPTH, None, None, C.c_size_t(0)) == 1
    kf.close()
    # a z-slab context holds part of the volume only: no volume picture from it
    slab = KFusionConfig(compute_size_ratio=8, volume_resolution=(32, 32, 32), slab=(8, 24))
    ks = KFusion(slab)
    with pytest.raises(native.KfhError) as e:
        ks.render(native.RENDER_VOLUME)
    assert e.value.code == 3
    assert ks.render(native.RENDER_DEPTH).shape == (60, 80, 4)          # the other pictures do not need the volume
    ks.close()
