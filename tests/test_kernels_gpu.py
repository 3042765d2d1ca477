"""Per-kernel parity of the HIP path against the CPU oracle, through the C ABI.

Bars (stated in BASELINE.json's north_star): depth conversion and pyramid indices bit-exact; per-kernel
float outputs within 1e-5 relative.  Everything that involves no transcendental function is in fact
required to be bit-exact here, because both sides evaluate the same IEEE-754 binary32 expression
sequence (no FMA contraction); only the bilateral filter (expf) and the ICP sums (summation order)
use a tolerance, written next to the assertion.

PARITY UNPINNED: the oracle restates the kfusion use-case, whose source the reference tree does not hold.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import kf_ref
from slambench_amd import native, synth
from slambench_amd.config import KFusionConfig, ICL_NUIM_INTRINSICS
from tests.util import hp, m4, make_depth, oracle_config, pad4, vp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    return native.lib()


@pytest.fixture(scope="module")
def scene():
    """A few frames of the synthetic room run through the oracle: volume, model maps, pyramids, poses."""
    cfg = KFusionConfig(compute_size_ratio=4, volume_resolution=(96, 96, 96))
    depth, gt, _ = synth.make_sequence(8, seed=2, noise=True)
    k = kf_ref.KFusionRef(oracle_config(cfg))
    for i in range(8):
        k.process_frame(depth[i])
    return {"cfg": cfg, "depth": depth, "gt": gt, "ref": k}


# ------------------------------------------------------------------------------------------ A1 / A2
@pytest.mark.parametrize("ratio", [1, 2, 4, 8])
def test_mm2meters_bit_exact(torch_cuda, L, ratio):
    torch = torch_cuda
    ih, iw = 480, 640
    d = make_depth(ih, iw, seed=ratio)
    d[0, 0], d[-1, -1] = 65535, 1
    oh, ow = ih // ratio, iw // ratio
    want = kf_ref.mm2meters(d, ow, oh, 1000.0)
    d_in = torch.from_numpy(d.view(np.int16)).cuda()
    raw = torch.empty((oh, ow), dtype=torch.float32, device="cuda")
    native.check(L.kfh_k_preprocess(vp(raw), None, ow, oh, vp(d_in), iw, ih, 1000.0, None))
    torch.cuda.synchronize()
    got = raw.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_mm2meters_other_unit(torch_cuda, L):
    """TUM-style 5000 units per metre: the divisor comes from the sensor, not a hard-coded 1000."""
    torch = torch_cuda
    d = make_depth(120, 160, seed=9)
    want = kf_ref.mm2meters(d, 160, 120, 5000.0)
    d_in = torch.from_numpy(d.view(np.int16)).cuda()
    raw = torch.empty((120, 160), dtype=torch.float32, device="cuda")
    native.check(L.kfh_k_preprocess(vp(raw), None, 160, 120, vp(d_in), 160, 120, 5000.0, None))
    assert np.array_equal(raw.cpu().numpy().view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("shape,ratio", [((480, 640), 1), ((480, 640), 2), ((120, 160), 1), ((60, 80), 1)])
def test_preprocess_bilateral(torch_cuda, L, shape, ratio):
    torch = torch_cuda
    ih, iw = shape
    d = make_depth(ih, iw, seed=5)
    oh, ow = ih // ratio, iw // ratio
    raw_want = kf_ref.mm2meters(d, ow, oh, 1000.0)
    filt_want = kf_ref.bilateral(raw_want)
    d_in = torch.from_numpy(d.view(np.int16)).cuda()
    raw = torch.empty((oh, ow), dtype=torch.float32, device="cuda")
    filt = torch.empty((oh, ow), dtype=torch.float32, device="cuda")
    native.check(L.kfh_k_preprocess(vp(raw), vp(filt), ow, oh, vp(d_in), iw, ih, 1000.0, None))
    got = filt.cpu().numpy()
    assert np.array_equal(raw.cpu().numpy(), raw_want)
    assert np.array_equal(got == 0, filt_want == 0)           # holes stay holes
    # tolerance: 1e-5 relative (expf differs by <= 1 ulp between the device and the host libm)
    np.testing.assert_allclose(got, filt_want, rtol=1e-5, atol=0)
    # the float-input entry point runs the same kernel
    filt2 = torch.empty_like(filt)
    native.check(L.kfh_k_bilateral(vp(filt2), vp(raw), ow, oh, None))
    assert torch.equal(filt, filt2)


def test_bilateral_all_holes_and_single_pixel(torch_cuda, L):
    torch = torch_cuda
    img = np.zeros((16, 32), np.float32)
    img[7, 13] = 1.25
    d = torch.from_numpy(img).cuda()
    out = torch.empty_like(d)
    native.check(L.kfh_k_bilateral(vp(out), vp(d), 32, 16, None))
    want = kf_ref.bilateral(img)
    assert np.array_equal(out.cpu().numpy(), want)
    assert want[7, 13] == np.float32(1.25)


# ------------------------------------------------------------------------------------------ A3
@pytest.mark.parametrize("shape", [(480, 640), (240, 320), (30, 40), (6, 8)])
def test_half_sample_bit_exact(torch_cuda, L, shape):
    torch = torch_cuda
    h, w = shape
    raw = kf_ref.mm2meters(make_depth(h, w, seed=3, holes=0.1), w, h, 1000.0)
    want = kf_ref.half_sample(raw, np.float32(0.1) * 3, 1)
    d = torch.from_numpy(raw).cuda()
    out = torch.empty((h // 2, w // 2), dtype=torch.float32, device="cuda")
    native.check(L.kfh_k_half_sample(vp(out), vp(d), w, h, float(np.float32(0.1) * 3), None))
    assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))


# ------------------------------------------------------------------------------------------ A4 / A5
@pytest.mark.parametrize("level", [0, 1, 2])
def test_vertex_normal_bit_exact(torch_cuda, L, level):
    torch = torch_cuda
    h, w = 480 >> level, 640 >> level
    raw = kf_ref.mm2meters(make_depth(h, w, seed=7 + level), w, h, 1000.0)
    k = [np.float32(v) / np.float32(1 << level) for v in ICL_NUIM_INTRINSICS]
    v_want = kf_ref.depth2vertex(raw, kf_ref.inverse_camera_matrix(k))
    n_want = kf_ref.vertex2normal(v_want)
    d = torch.from_numpy(raw).cuda()
    v = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    n = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    native.check(L.kfh_k_vertex_normal(vp(v), vp(n), vp(d), w, h, *[float(x) for x in k], None))
    v_got, n_got = v.cpu().numpy(), n.cpu().numpy()
    assert np.array_equal(v_got[..., :3].view(np.uint32), v_want.view(np.uint32))
    invalid = n_want[..., 0] == -2
    assert np.array_equal(n_got[..., 0] == -2, invalid)
    assert np.array_equal(n_got[..., :3][~invalid].view(np.uint32), n_want[~invalid].view(np.uint32))
    assert invalid.any() and (~invalid).any()


# ------------------------------------------------------------------------------------------ A6 / A7
def _track_inputs(scene, level, perturb):
    ref = scene["ref"]
    cfg = scene["cfg"]
    k = cfg.camera
    pose = ref.pose.copy()
    if perturb:
        pose = kf_ref.mat_mul(kf_ref.se3_exp(np.array([0.004, -0.003, 0.002, 0.002, -0.0015, 0.001])), pose)
    view = kf_ref.mat_mul(kf_ref.camera_matrix(k), kf_ref.mat_inverse(ref.pose))
    k_level = np.array([np.float32(v) / np.float32(1 << level) for v in k], np.float32)
    return (ref.input_vertex(level).copy(), ref.input_normal(level).copy(), ref.model_vertex().copy(),
            ref.model_normal().copy(), pose, view, ref.scaled_depth(level).copy(), k_level)


@pytest.mark.parametrize("level,perturb", [(0, False), (0, True), (1, True), (2, True)])
def test_track_reduce(torch_cuda, L, scene, level, perturb):
    torch = torch_cuda
    iv, inn, rv, rn, pose, view, depth_l, k_l = _track_inputs(scene, level, perturb)
    ih, iw = iv.shape[:2]
    rh, rw = rv.shape[:2]
    td_want = kf_ref.track(iv, inn, rv, rn, pose, view)
    sums_want = kf_ref.reduce(td_want, iw, ih)
    t = [torch.from_numpy(pad4(a)).cuda() for a in (iv, inn, rv, rn)]
    d_l = torch.from_numpy(depth_l).cuda()
    sums = torch.empty(32, dtype=torch.float32, device="cuda")
    td = torch.zeros((rh, rw, 8), dtype=torch.float32, device="cuda")
    native.check(L.kfh_k_track_reduce(vp(sums), vp(td), rw, vp(d_l), hp(k_l), vp(t[1]), iw, ih, vp(t[2]), vp(t[3]),
                                      rw, rh, hp(m4(pose)), hp(m4(view)), 0.1, 0.8, None))
    td_got = td.cpu().numpy().view(kf_ref.TRACK_DTYPE).reshape(rh, rw)[:ih, :iw]
    want = td_want[:ih, :iw]
    # data association (result codes) is index work: exact
    assert np.array_equal(td_got["result"], want["result"])
    ok = want["result"] == 1
    assert ok.sum() > 0.3 * ok.size
    # residuals and Jacobians: same IEEE expression sequence -> bit-exact
    assert np.array_equal(td_got["error"][ok].view(np.uint32), want["error"][ok].view(np.uint32))
    assert np.array_equal(td_got["J"][ok].view(np.uint32), want["J"][ok].view(np.uint32))
    got = sums.cpu().numpy()
    # counters are exact; float sums differ only by summation order.  Tolerance: 1e-5 of the sum of
    # absolute terms (norm-wise relative), checked against a float64 sum of the oracle's per-pixel rows.
    assert np.array_equal(got[28:], sums_want[28:])
    e = want["error"][ok].astype(np.float64)
    J = want["J"][ok].astype(np.float64)
    exact = np.zeros(28)
    mag = np.zeros(28)
    exact[0], mag[0] = (e * e).sum(), (e * e).sum()
    exact[1:7], mag[1:7] = (e[:, None] * J).sum(0), np.abs(e[:, None] * J).sum(0)
    q = 7
    for i in range(6):
        for j in range(i, 6):
            exact[q], mag[q] = (J[:, i] * J[:, j]).sum(), np.abs(J[:, i] * J[:, j]).sum()
            q += 1
    assert np.all(np.abs(got[:28] - exact) <= 1e-5 * mag + 1e-12)
    assert np.all(np.abs(got[:28] - sums_want[:28]) <= 2e-5 * mag + 1e-12)


def test_track_reduce_no_model(torch_cuda, L):
    """Frames 0-2: the model maps are all zero, nothing is valid and the sums stay zero."""
    torch = torch_cuda
    h, w = 120, 160
    raw = kf_ref.mm2meters(make_depth(h, w, seed=11), w, h, 1000.0)
    k = [v / 4 for v in ICL_NUIM_INTRINSICS]
    iv = kf_ref.depth2vertex(raw, kf_ref.inverse_camera_matrix(k))
    inn = kf_ref.vertex2normal(iv)
    zeros = np.zeros((h, w, 3), np.float32)
    pose = np.eye(4, dtype=np.float32)
    pose[:3, 3] = 4
    view = kf_ref.mat_mul(kf_ref.camera_matrix(k), kf_ref.mat_inverse(pose))
    sums_want = kf_ref.reduce(kf_ref.track(iv, inn, zeros, zeros, pose, view), w, h)
    t = [torch.from_numpy(pad4(a)).cuda() for a in (iv, inn, zeros, zeros)]
    d_l = torch.from_numpy(raw).cuda()
    sums = torch.empty(32, dtype=torch.float32, device="cuda")
    native.check(L.kfh_k_track_reduce(vp(sums), None, 0, vp(d_l), hp(np.array(k, np.float32)), vp(t[1]), w, h,
                                      vp(t[2]), vp(t[3]), w, h, hp(m4(pose)), hp(m4(view)), 0.1, 0.8, None))
    got = sums.cpu().numpy()
    assert np.array_equal(got, sums_want)
    assert got[28] == 0 and got[:28].max() == 0


# ------------------------------------------------------------------------------------------ A8
def _update_pose_gpu(torch, L, pose, sums, thr=1e-5):
    p = torch.from_numpy(m4(pose).copy()).cuda()
    x = torch.zeros(6, dtype=torch.float64, device="cuda")
    conv = torch.zeros(2, dtype=torch.int32, device="cuda")
    s = torch.from_numpy(np.ascontiguousarray(sums, np.float32)).cuda()
    native.check(L.kfh_k_update_pose(vp(p), vp(x), vp(conv), vp(s), thr, None))
    torch.cuda.synchronize()
    conv = conv.cpu().numpy()
    return p.cpu().numpy().reshape(4, 4), x.cpu().numpy(), bool(conv[0]), int(conv[1])


def test_update_pose_well_conditioned(torch_cuda, L, scene):
    iv, inn, rv, rn, pose, view, _, _ = _track_inputs(scene, 0, True)
    sums = kf_ref.reduce(kf_ref.track(iv, inn, rv, rn, pose, view), iv.shape[1], iv.shape[0])
    x_want = kf_ref.solve(sums[1:28])
    pose_want, conv_want = kf_ref.update_pose(pose, sums)
    pose_got, x_got, conv_got, slow = _update_pose_gpu(torch_cuda, L, pose, sums)
    assert not slow                               # well conditioned: the elimination path is certified
    # double-precision solve: Gauss-Jordan on the device vs eigen-decomposition in the oracle
    np.testing.assert_allclose(x_got, x_want, rtol=1e-8, atol=1e-14)
    np.testing.assert_allclose(pose_got, pose_want, rtol=0, atol=1e-6)
    assert conv_got == conv_want


def test_update_pose_rank_deficient(torch_cuda, L):
    """A single plane seen head-on constrains 3 of 6 degrees of freedom: the singular-value cut-off
    (sigma < sigma_max / 1e6 -> dropped) must act, which the device does through its Jacobi path."""
    rng = np.random.default_rng(0)
    n = 2000
    nrm = np.tile(np.array([0.0, 0.0, -1.0]), (n, 1))
    pts = np.c_[rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), np.full(n, 2.0)]
    J = np.c_[nrm, np.cross(pts, nrm)]
    e = 0.01 + 0.002 * pts[:, 0]
    sums = np.zeros(32, np.float32)
    sums[0] = (e * e).sum()
    sums[1:7] = (e[:, None] * J).sum(0)
    q = 7
    for i in range(6):
        for j in range(i, 6):
            sums[q] = (J[:, i] * J[:, j]).sum()
            q += 1
    sums[28] = n
    pose = np.eye(4, dtype=np.float32)
    x_want = kf_ref.solve(sums[1:28])
    assert np.abs(x_want).max() < 1.0          # the cut-off keeps the update finite
    pose_want, conv_want = kf_ref.update_pose(pose, sums)
    pose_got, x_got, conv_got, slow = _update_pose_gpu(torch_cuda, L, pose, sums)
    assert slow == 1                            # full eigen-decomposition
    np.testing.assert_allclose(x_got, x_want, rtol=1e-6, atol=1e-10)
    np.testing.assert_allclose(pose_got, pose_want, rtol=0, atol=1e-6)
    assert conv_got == conv_want


def _system_with_spectrum(eigs, seed):
    """sums[1:28] of a normal-equation system C x = b with the given eigenvalues of C and b in its range."""
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((6, 6)))
    C = (q * np.asarray(eigs, np.float64)) @ q.T
    C = 0.5 * (C + C.T)
    b = C @ rng.standard_normal(6) + 1e-6 * np.abs(eigs).max() * q[:, 0]   # plus a little along the weakest axis
    sums = np.zeros(32, np.float32)
    sums[0] = 1.0
    sums[1:7] = b
    k = 7
    for i in range(6):
        for j in range(i, 6):
            sums[k] = C[i, j]
            k += 1
    sums[28] = 1000
    return sums


@pytest.mark.parametrize("case,eigs,paths", [
    ("well inside", [4.0, 40.0, 300.0, 5e3, 2e5, 1.8e6], (0,)),          # cond 0.45e6: certified by the norms
    ("just dropped", [1.5, 40.0, 300.0, 5e3, 2e5, 2.2e6], (2,)),         # cond 1.5e6: the smallest is dropped
    ("far dropped", [0.02, 40.0, 300.0, 5e3, 2e5, 2.2e6], (2,)),         # cond 1.1e8
    ("two dropped", [1.0, 1.4, 300.0, 5e3, 2e5, 2.2e6], (1,)),           # needs the full decomposition
    ("cluster kept", [2.6, 2.8, 3.0, 5e3, 2e5, 1.8e6], (0, 1)),          # cond 0.7e6, norms inconclusive
])
def test_update_pose_near_conditioning_limit(torch_cuda, L, case, eigs, paths):
    """Systems around the sigma_max / 1e6 cut-off: which path the device takes, and that every path
    reproduces the reference's truncated solution (the float-rounded sums decide the spectrum, so the
    expectation is taken from the oracle on the same 27 floats)."""
    sums = _system_with_spectrum(eigs, seed=3)
    pose = np.eye(4, dtype=np.float32)
    x_want = kf_ref.solve(sums[1:28])
    pose_want, conv_want = kf_ref.update_pose(pose, sums)
    pose_got, x_got, conv_got, slow = _update_pose_gpu(torch_cuda, L, pose, sums)
    assert slow in paths, (case, slow)
    np.testing.assert_allclose(x_got, x_want, rtol=1e-6, atol=1e-9 * np.abs(x_want).max())
    np.testing.assert_allclose(pose_got, pose_want, rtol=0, atol=1e-6)
    assert conv_got == conv_want


def test_update_pose_empty_system(torch_cuda, L):
    pose = np.eye(4, dtype=np.float32)
    pose[:3, 3] = [4, 4, 4]
    pose_got, x_got, conv, _ = _update_pose_gpu(torch_cuda, L, pose, np.zeros(32, np.float32))
    assert np.array_equal(x_got, np.zeros(6)) and conv
    assert np.array_equal(pose_got, pose)


# ------------------------------------------------------------------------------------------ A9
@pytest.mark.parametrize("res", [(64, 64, 64), (96, 80, 72)])
def test_integrate_bit_exact(torch_cuda, L, scene, res):
    torch = torch_cuda
    cfg = scene["cfg"]
    w, h = cfg.compute_size
    K = kf_ref.camera_matrix(cfg.camera)
    dim = (8.0, 8.0, 8.0)
    vol_ref = kf_ref.new_volume(res)
    vol = torch.from_numpy(vol_ref.copy()).cuda()
    res_c = (C.c_int32 * 3)(*res)
    dim_c = (C.c_float * 3)(*dim)
    for i in range(4):
        depth = kf_ref.mm2meters(scene["depth"][i], w, h, 1000.0)
        inv_pose = kf_ref.mat_inverse(scene["gt"][i].astype(np.float32))
        kf_ref.integrate(vol_ref, dim, depth, inv_pose, K, 0.1, 100.0)
        d = torch.from_numpy(depth).cuda()
        native.check(L.kfh_k_integrate(vp(vol), res_c, dim_c, 0, res[2], vp(d), w, h, hp(m4(inv_pose)), hp(m4(K)),
                                       0.1, 100.0, None))
    got = vol.cpu().numpy()
    assert (vol_ref[..., 1] > 0).sum() > 1000          # something was fused
    assert np.array_equal(got, vol_ref)


@pytest.mark.parametrize("maxweight", [1.0, 2.0, 2.5, 3.0])
def test_integrate_with_saturating_weights(torch_cuda, L, scene, maxweight):
    """Weights that reach the cap after one to three frames and stay there (a free-space voxel at the cap is no longer
    stored at all, a voxel near the surface keeps averaging with the capped weight); a cap that is no integer is
    truncated when the weight is stored, as the reference's conversion to short does."""
    torch = torch_cuda
    cfg = scene["cfg"]
    w, h = cfg.compute_size
    K = kf_ref.camera_matrix(cfg.camera)
    for res, dim in (((128, 64, 96), (8.0, 8.0, 8.0)), ((96, 96, 96), (1.5, 1.5, 1.5))):
        # (the small volume lies wholly inside the image in front of the surfaces: whole columns of free space, which
        # take the march with two columns per lane)
        vol_ref = kf_ref.new_volume(res)
        vol = torch.from_numpy(vol_ref.copy()).cuda()
        res_c, dim_c = (C.c_int32 * 3)(*res), (C.c_float * 3)(*dim)
        for i in range(6):
            depth = kf_ref.mm2meters(scene["depth"][i % 3], w, h, 1000.0)
            pose = scene["gt"][i % 3].astype(np.float32)
            if dim[0] < 8.0:                   # the small volume starts 0.4 m in front of the camera, centred on its axis
                pose = pose.copy()
                pose[:3, 3] = pose[:3, :3] @ np.array([0.75, 0.75, -0.4], np.float32)
            inv_pose = kf_ref.mat_inverse(pose)
            kf_ref.integrate(vol_ref, dim, depth, inv_pose, K, 0.1, maxweight)
            d = torch.from_numpy(depth).cuda()
            native.check(L.kfh_k_integrate(vp(vol), res_c, dim_c, 0, res[2], vp(d), w, h, hp(m4(inv_pose)), hp(m4(K)),
                                           0.1, maxweight, None))
            assert np.array_equal(vol.cpu().numpy(), vol_ref), (res, i)
        assert int(vol_ref[..., 1].max()) == int(maxweight) and (vol_ref[..., 1] == int(maxweight)).sum() > 1000


def test_integrate_slab_equals_full(torch_cuda, L, scene):
    """A z-slab [z0, z1) whose storage starts at plane z0 gets exactly the voxels of the full volume."""
    torch = torch_cuda
    cfg = scene["cfg"]
    w, h = cfg.compute_size
    res, dim = (64, 64, 64), (8.0, 8.0, 8.0)
    K = kf_ref.camera_matrix(cfg.camera)
    depth = kf_ref.mm2meters(scene["depth"][3], w, h, 1000.0)
    inv_pose = kf_ref.mat_inverse(scene["gt"][3].astype(np.float32))
    full = kf_ref.integrate(kf_ref.new_volume(res), dim, depth, inv_pose, K)
    d = torch.from_numpy(depth).cuda()
    res_c, dim_c = (C.c_int32 * 3)(*res), (C.c_float * 3)(*dim)
    for z0, z1 in [(0, 16), (16, 48), (40, 64)]:
        slab = torch.from_numpy(kf_ref.new_volume((res[0], res[1], z1 - z0))).cuda()
        native.check(L.kfh_k_integrate(vp(slab), res_c, dim_c, z0, z1, vp(d), w, h, hp(m4(inv_pose)), hp(m4(K)), 0.1,
                                       100.0, None))
        assert np.array_equal(slab.cpu().numpy(), full[z0:z1])


# ------------------------------------------------------------------------------------------ A10
def test_raycast_bit_exact(torch_cuda, L, scene):
    torch = torch_cuda
    ref, cfg = scene["ref"], scene["cfg"]
    w, h = cfg.compute_size
    vol_np = ref.volume().copy()
    res = cfg.volume_resolution
    dim = cfg.volume_size
    view = kf_ref.mat_mul(ref.pose, kf_ref.inverse_camera_matrix(cfg.camera))
    step = np.float32(min(dim)) / np.float32(max(res))
    v_want, n_want = kf_ref.raycast(vol_np, dim, w, h, view, 0.4, 4.0, step, np.float32(0.75) * np.float32(0.1))
    vol = torch.from_numpy(vol_np).cuda()
    v = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    n = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    native.check(L.kfh_k_raycast(vp(v), vp(n), w, h, vp(vol), (C.c_int32 * 3)(*res), (C.c_float * 3)(*dim),
                                 hp(m4(view)), 0.4, 4.0, float(step), float(np.float32(0.75) * np.float32(0.1)), None))
    v_got, n_got = v.cpu().numpy()[..., :3], n.cpu().numpy()[..., :3]
    hit = n_want[..., 0] != -2
    assert hit.mean() > 0.5
    assert np.array_equal(n_got[..., 0] == -2, ~hit)
    assert np.array_equal(v_got.view(np.uint32), v_want.view(np.uint32))
    assert np.array_equal(n_got[hit].view(np.uint32), n_want[hit].view(np.uint32))


def test_raycast_empty_volume_misses(torch_cuda, L):
    torch = torch_cuda
    res, dim = (32, 32, 32), (8.0, 8.0, 8.0)
    vol = torch.from_numpy(kf_ref.new_volume(res)).cuda()
    pose = np.eye(4, dtype=np.float32)
    pose[:3, 3] = 4
    view = kf_ref.mat_mul(pose, kf_ref.inverse_camera_matrix([v / 8 for v in ICL_NUIM_INTRINSICS]))
    v = torch.empty((60, 80, 4), dtype=torch.float32, device="cuda")
    n = torch.empty((60, 80, 4), dtype=torch.float32, device="cuda")
    native.check(L.kfh_k_raycast(vp(v), vp(n), 80, 60, vp(vol), (C.c_int32 * 3)(*res), (C.c_float * 3)(*dim),
                                 hp(m4(view)), 0.4, 4.0, 0.25, 0.075, None))
    assert float(v.abs().max()) == 0.0
    assert bool((n[..., 0] == -2).all())


def test_bad_arguments_are_rejected(torch_cuda, L):
    torch = torch_cuda
    d = torch.zeros((8, 8), dtype=torch.float32, device="cuda")
    rc = L.kfh_k_half_sample(vp(d), vp(d), 7, 8, 0.3, None)
    assert rc == 1 and b"bad argument" in L.kfh_last_error()
    rc = L.kfh_k_preprocess(vp(d), None, 6, 8, vp(d), 8, 8, 1000.0, None)
    assert rc == 1
    # a raycast fetch takes two x-adjacent voxels: a volume one voxel thin is refused, not read past its end
    vol = torch.zeros((4, 4, 1, 2), dtype=torch.int16, device="cuda")
    v = torch.empty((8, 8, 4), dtype=torch.float32, device="cuda")
    view = np.eye(4, dtype=np.float32)
    rc = L.kfh_k_raycast(vp(v), vp(v), 8, 8, vp(vol), (C.c_int32 * 3)(1, 4, 4), (C.c_float * 3)(1.0, 1.0, 1.0),
                         hp(m4(view)), 0.4, 4.0, 0.1, 0.1, None)
    assert rc == 1 and b"two voxels" in L.kfh_last_error()
