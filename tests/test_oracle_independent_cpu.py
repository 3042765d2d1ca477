"""CPU suite, part 3: a second opinion on the oracle.

The reference tree holds no kfusion source and no per-kernel vectors, so nothing of the reference can check the
oracle's kernels one by one.  What can be done without it: every kernel is restated here a second time, straight
from the published formulas, in vectorised float64 numpy that shares no code with oracle/kf_ref.cpp (different
language, different precision, different loop structure), and the oracle must agree with that restatement to
float32 accuracy; discrete outcomes (result codes, counters, which voxels are touched) may only differ on the few
elements that sit within rounding of a threshold.  An indexing slip, a swapped axis, a wrong constant or a wrong
sign in the oracle shows up here; a misremembered formula shared by both restatements does not (that is what
"parity unpinned" means).  The raycast is additionally checked against analytic geometry.
"""
import numpy as np
import pytest
import scipy.linalg

from oracle import kf_ref
from slambench_amd import synth

INVALID = -2.0
W, H = 160, 120
K4 = (481.2 / 4, 480.0 / 4, 319.5 / 4, 239.5 / 4)


@pytest.fixture(scope="module")
def frames():
    depth, gt, ts = synth.make_sequence(25, seed=11, intrinsics=K4, size=(W, H), noise=True)
    return depth, gt


def Kmat(k):
    fx, fy, cx, cy = k
    return np.array([[fx, 0, cx, 0], [0, fy, cy, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float64)


def Kinv(k):
    fx, fy, cx, cy = k
    return np.array([[1 / fx, 0, -cx / fx, 0], [0, 1 / fy, -cy / fy, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float64)


# ------------------------------------------------------------------ second restatement (float64, vectorised)
def np_bilateral(d, radius=2, delta=4.0, e_d=0.1):
    d = d.astype(np.float64)
    h, w = d.shape
    g = np.exp(-(np.arange(-radius, radius + 1) ** 2) / (2 * delta * delta))
    pad = np.pad(d, radius, mode="edge")
    num = np.zeros_like(d)
    den = np.zeros_like(d)
    for i in range(-radius, radius + 1):
        for j in range(-radius, radius + 1):
            p = pad[radius + i:radius + i + h, radius + j:radius + j + w]
            f = g[i + radius] * g[j + radius] * np.exp(-(p - d) ** 2 / (2 * e_d * e_d)) * (p > 0)
            num += f * p
            den += f
    return np.where(d == 0, 0.0, num / np.where(den == 0, 1, den))


def np_half_sample(d, e_d=0.3):
    d = d.astype(np.float64)
    h, w = d.shape
    centre = d[0::2, 0::2][:h // 2, :w // 2]
    num = np.zeros_like(centre)
    cnt = np.zeros_like(centre)
    for i in (0, 1):
        for j in (0, 1):
            p = d[i::2, j::2][:h // 2, :w // 2]
            ok = np.abs(p - centre) < e_d
            num += np.where(ok, p, 0)
            cnt += ok
    return num / cnt


def np_depth2vertex(d, k):
    h, w = d.shape
    ys, xs = np.mgrid[0:h, 0:w]
    rays = np.stack([xs, ys, np.ones_like(xs)], -1).astype(np.float64) @ Kinv(k)[:3, :3].T
    return np.where((d > 0)[..., None], d[..., None].astype(np.float64) * rays, 0.0)


def np_vertex2normal(v):
    v = v.astype(np.float64)
    left = np.concatenate([v[:, :1], v[:, :-1]], 1)
    right = np.concatenate([v[:, 1:], v[:, -1:]], 1)
    up = np.concatenate([v[:1], v[:-1]], 0)
    down = np.concatenate([v[1:], v[-1:]], 0)
    bad = (left[..., 2] == 0) | (right[..., 2] == 0) | (up[..., 2] == 0) | (down[..., 2] == 0)
    n = np.cross(down - up, right - left)
    length = np.linalg.norm(n, axis=-1, keepdims=True)
    n = n / np.where(length == 0, 1, length)
    n[bad] = (INVALID, 0, 0)
    return n, bad


def np_track(in_v, in_n, ref_v, ref_n, pose, view, dist=0.1, normal_thr=0.8):
    pose = pose.astype(np.float64)
    view = view.astype(np.float64)
    h, w, _ = in_v.shape
    rh, rw, _ = ref_v.shape
    v = in_v.astype(np.float64) @ pose[:3, :3].T + pose[:3, 3]
    q = v @ view[:3, :3].T + view[:3, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        ux = q[..., 0] / q[..., 2] + 0.5
        uy = q[..., 1] / q[..., 2] + 0.5
    result = np.ones((h, w), np.int32)
    no_input = in_n[..., 0] == INVALID
    outside = ~((ux >= 0) & (ux <= rw - 1) & (uy >= 0) & (uy <= rh - 1))
    rx = np.where(outside, 0, ux).astype(np.int64)
    ry = np.where(outside, 0, uy).astype(np.int64)
    rn = ref_n[ry, rx].astype(np.float64)
    rv = ref_v[ry, rx].astype(np.float64)
    no_ref = rn[..., 0] == INVALID
    diff = rv - v
    n = in_n.astype(np.float64) @ pose[:3, :3].T
    far = np.linalg.norm(diff, axis=-1) > dist
    turned = np.einsum("...i,...i", n, rn) < normal_thr
    # the first failing test decides, in this order
    for code, mask in ((-5, turned), (-4, far), (-3, no_ref), (-2, outside), (-1, no_input)):
        result[mask] = code
    error = np.einsum("...i,...i", rn, diff)
    J = np.concatenate([rn, np.cross(v, rn)], -1)
    # margins: how close each pixel is to flipping one of its tests (for the borderline allowance)
    margin = np.minimum.reduce([np.abs(np.linalg.norm(diff, axis=-1) - dist),
                                np.abs(np.einsum("...i,...i", n, rn) - normal_thr),
                                np.abs(ux - np.round(ux)), np.abs(uy - np.round(uy))])
    return result, error, J, margin


def np_se3_exp(x):
    t, w = x[:3], x[3:]
    m = np.zeros((4, 4))
    m[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    m[:3, 3] = t
    return scipy.linalg.expm(m)


def np_integrate(vol, dim, depth, inv_track, k, mu=0.1, maxweight=100.0):
    """vol: int16 [z, y, x, 2]; returns the new (tsdf float, weight float, touched mask), float64 arithmetic."""
    sz, sy, sx, _ = vol.shape
    h, w = depth.shape
    zs, ys, xs = np.mgrid[0:sz, 0:sy, 0:sx]
    p = np.stack([(xs + 0.5) * dim[0] / sx, (ys + 0.5) * dim[1] / sy, (zs + 0.5) * dim[2] / sz], -1)
    pos = p @ inv_track[:3, :3].astype(np.float64).T + inv_track[:3, 3].astype(np.float64)
    cam = pos @ Kmat(k)[:3, :3].T
    front = pos[..., 2] >= 0.0001
    with np.errstate(divide="ignore", invalid="ignore"):
        px = cam[..., 0] / cam[..., 2] + 0.5
        py = cam[..., 1] / cam[..., 2] + 0.5
    inside = front & (px >= 0) & (px <= w - 1) & (py >= 0) & (py <= h - 1)
    ix = np.where(inside, px, 0).astype(np.int64)
    iy = np.where(inside, py, 0).astype(np.int64)
    d = depth[iy, ix].astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        diff = (d - cam[..., 2]) * np.sqrt(1 + (pos[..., 0] / pos[..., 2]) ** 2 + (pos[..., 1] / pos[..., 2]) ** 2)
    touched = inside & (d != 0) & (diff > -mu)
    sdf = np.minimum(1.0, diff / mu)
    t_old = vol[..., 0].astype(np.float64) * 0.00003051944088
    w_old = vol[..., 1].astype(np.float64)
    t_new = np.clip((w_old * t_old + sdf) / (w_old + 1), -1, 1)
    w_new = np.minimum(w_old + 1, maxweight)
    margin = np.minimum(np.abs(diff + mu), np.minimum(np.abs(px - np.round(px)), np.abs(py - np.round(py))))
    return np.where(touched, t_new, t_old), np.where(touched, w_new, w_old), touched, margin


# ------------------------------------------------------------------ the comparisons
def test_bilateral_filter(frames):
    depth, _ = frames
    d = kf_ref.mm2meters(depth[0], W, H)
    assert np.array_equal(d, (depth[0] / np.float32(1000.0)).astype(np.float32))
    assert np.array_equal(kf_ref.mm2meters(depth[0], W // 4, H // 4), d[::4, ::4])
    got = kf_ref.bilateral(d)
    want = np_bilateral(d)
    assert (d == 0).any() and ((got == 0) == (d == 0)).all()
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=0)


def test_half_sample(frames):
    depth, _ = frames
    d = kf_ref.bilateral(kf_ref.mm2meters(depth[0], W, H))
    got = kf_ref.half_sample(d)
    want = np_half_sample(d)
    assert got.shape == (H // 2, W // 2)
    np.testing.assert_allclose(got, want, rtol=1e-6)
    got2 = kf_ref.half_sample(got)
    np.testing.assert_allclose(got2, np_half_sample(got), rtol=1e-6)


def test_vertex_and_normal_maps(frames):
    depth, _ = frames
    d = kf_ref.bilateral(kf_ref.mm2meters(depth[0], W, H))
    v = kf_ref.depth2vertex(d, kf_ref.inverse_camera_matrix(K4))
    np.testing.assert_allclose(v, np_depth2vertex(d, K4), rtol=2e-6, atol=5e-7)     # x - cx cancels near the axis
    n = kf_ref.vertex2normal(v)
    want, bad = np_vertex2normal(v)
    assert bad.any() and not bad.all()
    assert np.array_equal(n[..., 0] == INVALID, bad)
    np.testing.assert_allclose(n[~bad], want[~bad], atol=2e-4)      # cross products of small differences
    np.testing.assert_allclose(np.linalg.norm(n[~bad], axis=-1), 1, atol=1e-5)


@pytest.fixture(scope="module")
def tracking_inputs(frames):
    """Model maps of frame 0 seen from its true pose, input maps of frame 8, pose guess = frame 0's pose."""
    depth, gt = frames
    invK = kf_ref.inverse_camera_matrix(K4)
    maps = []
    for i in (0, 8):
        d = kf_ref.bilateral(kf_ref.mm2meters(depth[i], W, H))
        v = kf_ref.depth2vertex(d, invK)
        maps.append((v, kf_ref.vertex2normal(v)))
    pose0 = gt[0].astype(np.float32)
    # model maps live in world coordinates
    v0, n0 = maps[0]
    valid = n0[..., 0] != INVALID
    ref_v = np.where(valid[..., None], v0 @ pose0[:3, :3].T + pose0[:3, 3], 0).astype(np.float32)
    ref_n = np.where(valid[..., None], n0 @ pose0[:3, :3].T, (INVALID, 0, 0)).astype(np.float32)
    view = kf_ref.mat_mul(kf_ref.camera_matrix(K4), kf_ref.mat_inverse(pose0))
    return maps[1][0], maps[1][1], ref_v, ref_n, pose0, view


def test_track_and_reduce(tracking_inputs):
    in_v, in_n, ref_v, ref_n, pose, view = tracking_inputs
    td = kf_ref.track(in_v, in_n, ref_v, ref_n, pose, view)
    result, error, J, margin = np_track(in_v, in_n, ref_v, ref_n, pose, view)
    codes = set(np.unique(td["result"]))
    assert {1, -1, -4}.issubset(codes), codes                  # the scene exercises several outcomes
    # A hole of the depth map whose four neighbours are valid has a vertex of (0, 0, 0) with a valid normal (the
    # normal map only looks at the neighbours).  It transforms to the camera centre, whose projection is 0 / 0 in
    # exact arithmetic and rounding noise in float32: such a pixel must be rejected, by whichever test.
    hole = (in_v[..., 2] == 0) & (in_n[..., 0] != INVALID)
    assert hole.any() and (td["result"][hole] < 1).all()
    differ = (td["result"] != result) & ~hole
    assert differ.mean() < 2e-3                                # only pixels within rounding of a threshold
    assert (margin[differ] < 1e-4).all(), margin[differ].max()
    ok = (td["result"] == 1) & (result == 1)
    assert ok.mean() > 0.5
    np.testing.assert_allclose(td["error"][ok], error[ok], atol=2e-6)
    np.testing.assert_allclose(td["J"][ok], J[ok], atol=5e-6)

    sums = kf_ref.reduce(td, W, H)
    e = td["error"][td["result"] == 1].astype(np.float64)
    Jv = td["J"][td["result"] == 1].astype(np.float64)
    want = np.zeros(32)
    want[0] = (e * e).sum()
    want[1:7] = (e[:, None] * Jv).sum(0)
    want[7:28] = (Jv[:, :, None] * Jv[:, None, :]).sum(0)[np.triu_indices(6)]
    want[28] = (td["result"] == 1).sum()
    want[29] = (td["result"] == -4).sum()
    want[30] = (td["result"] == -5).sum()
    want[31] = ((td["result"] > -4) & (td["result"] < 1)).sum()
    assert np.array_equal(sums[28:], want[28:])
    scale = np.zeros(32)
    scale[0] = (e * e).sum()
    scale[1:7] = np.abs(e[:, None] * Jv).sum(0)
    scale[7:28] = np.abs(Jv[:, :, None] * Jv[:, None, :]).sum(0)[np.triu_indices(6)]
    assert (np.abs(sums[:28] - want[:28]) <= 1e-5 * scale[:28] + 1e-12).all()

    # solve: the normal equations of these sums, against a general solver
    x = kf_ref.solve(sums[1:28])
    C = np.zeros((6, 6))
    C[np.triu_indices(6)] = sums[7:28].astype(np.float64)
    C = C + np.triu(C, 1).T
    np.testing.assert_allclose(x, np.linalg.solve(C, sums[1:7].astype(np.float64)), rtol=1e-8)
    # one Gauss-Newton step moves the guess towards the true pose of the input frame
    new_pose, converged = kf_ref.update_pose(pose, sums)
    assert not converged
    np.testing.assert_allclose(new_pose, (np_se3_exp(x) @ pose.astype(np.float64)), atol=2e-6)


def test_gauss_newton_converges_to_the_true_motion(frames, tracking_inputs):
    depth, gt = frames
    in_v, in_n, ref_v, ref_n, pose, view = tracking_inputs
    start = np.linalg.norm(pose[:3, 3] - gt[8][:3, 3])
    for _ in range(12):
        td = kf_ref.track(in_v, in_n, ref_v, ref_n, pose, view)
        pose, converged = kf_ref.update_pose(pose, kf_ref.reduce(td, W, H))
    end = np.linalg.norm(pose[:3, 3] - gt[8][:3, 3])
    assert start > 5e-3 and end < 0.2 * start, (start, end)     # noisy single-view model: a fifth is plenty


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_se3_exponential(seed):
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 0.2 * (seed + 1), 6)
    np.testing.assert_allclose(kf_ref.se3_exp(x), np_se3_exp(x), atol=1e-6)
    tiny = x * 1e-6                                             # the series branch near zero rotation
    np.testing.assert_allclose(kf_ref.se3_exp(tiny), np_se3_exp(tiny), atol=1e-7)


def test_integration(frames):
    depth, gt = frames
    d = kf_ref.mm2meters(depth[0], W, H)
    dim = (8.0, 8.0, 8.0)
    vol = kf_ref.new_volume((40, 48, 56))
    inv0 = kf_ref.mat_inverse(gt[0].astype(np.float32))
    K = kf_ref.camera_matrix(K4)
    for rep, (dd, inv) in enumerate([(d, inv0), (kf_ref.mm2meters(depth[8], W, H), kf_ref.mat_inverse(gt[8].astype(np.float32)))]):
        before = vol.copy()
        t_want, w_want, touched, margin = np_integrate(before, dim, dd, inv, K4, mu=0.3)
        kf_ref.integrate(vol, dim, dd, inv, K, mu=0.3)
        changed = (vol != before).any(-1)
        assert 0.005 < touched.mean() < 0.9
        # weights: equal except where a voxel sits within rounding of a test
        wdiff = vol[..., 1] != w_want
        assert wdiff.mean() < 1e-3 and (margin[wdiff] < 1e-3).all()
        assert not (changed & ~touched & ~wdiff).any()
        same = ~wdiff
        # tsdf: the stored short is the float value times 32766, truncated: one step of slack, plus the float32
        # rounding of the distance (1e-6 m over mu, times 32766)
        assert np.abs(vol[..., 0][same] - t_want[same] * 32766.0).max() <= 1.25
    assert vol[..., 1].max() == 2


def test_raycast_against_analytic_planes():
    """A wall seen at an angle, fused four times from the same pose: every ray that hits must end on the wall
    (to a fraction of a voxel) and carry the wall's normal."""
    res, dim = (96, 96, 96), (4.0, 4.0, 4.0)
    pose = np.eye(4, dtype=np.float32)
    pose[:3, 3] = (2.0, 2.0, 0.2)                                # looking along +z from near the front face
    n_wall = np.array([0.3, -0.2, -1.0])
    n_wall /= np.linalg.norm(n_wall)
    p_wall = np.array([2.0, 2.0, 2.2])
    ys, xs = np.mgrid[0:H, 0:W]
    rays = np.stack([xs, ys, np.ones_like(xs)], -1) @ Kinv(K4)[:3, :3].T
    origin = pose[:3, 3].astype(np.float64)
    t_true = ((p_wall - origin) @ n_wall) / (rays @ n_wall)     # planar depth = z of the hit = t (rays have z = 1)
    depth = t_true.astype(np.float32)
    vol = kf_ref.new_volume(res)
    for _ in range(4):
        kf_ref.integrate(vol, dim, depth, kf_ref.mat_inverse(pose), kf_ref.camera_matrix(K4), mu=0.1)
    view = kf_ref.mat_mul(pose, kf_ref.inverse_camera_matrix(K4))
    vertex, normal = kf_ref.raycast(vol, dim, W, H, view)
    hit = normal[..., 0] != INVALID
    inside = np.all((origin + rays * t_true[..., None] > 0.1) & (origin + rays * t_true[..., None] < 3.9), -1)
    # at the rim of the image the voxels on the far side of a sample lie outside the frustum and were never fused
    inside &= (xs >= 6) & (xs < W - 6) & (ys >= 6) & (ys < H - 6)
    assert hit[inside].mean() > 0.99
    both = hit & inside
    voxel = dim[0] / res[0]
    off_plane = (vertex[both].astype(np.float64) - p_wall) @ n_wall
    assert np.abs(off_plane).max() < 0.25 * voxel, np.abs(off_plane).max() / voxel
    on_ray = np.linalg.norm(np.cross(vertex[both] - origin, rays[both]), axis=-1) / np.linalg.norm(rays[both], axis=-1)
    assert on_ray.max() < 1e-5
    # the gradient reads two voxels to either side (8 cm, five pixels at this distance): keep clear of the rim
    core = both & (xs >= 16) & (xs < W - 16) & (ys >= 16) & (ys < H - 16)
    cosang = np.abs(normal[core].astype(np.float64) @ n_wall)
    # voxels of 4 cm under a band of 10 cm: behind the wall the band ends after two voxels and the untouched
    # voxels past it still hold +1, so a few gradients that reach them are off by tens of degrees
    assert np.median(cosang) > 0.9995 and np.percentile(cosang, 1) > 0.99 and (cosang < 0.95).mean() < 0.005, \
        (np.median(cosang), np.percentile(cosang, 1), (cosang < 0.95).mean())
    # the gradient points from inside the wall (negative values) to free space (positive): towards the camera
    assert ((normal[core].astype(np.float64) @ n_wall) > 0).all()
    assert (vertex[~hit] == 0).all()
