"""kfh_k_integrate_noalloc (the entry point that allocates nothing: its work list lives in a fixed array in device
memory) against kfh_k_integrate, through the C ABI: the same voxels bit for bit, on two calls in a row of which
the second has the smaller input, then a stream-ordered run of calls without a wait in between, the size limit,
and a capture of the call in the mode that forbids allocation."""
import ctypes as C

import numpy as np
import pytest

from oracle import kf_ref
from slambench_amd import native, synth
from slambench_amd.config import KFusionConfig
from tests.util import hp, m4, vp

pytestmark = pytest.mark.gpu


def _case(torch, L, res, csr, frame, entry, vol=None, stream=None, wait=True):
    """One integration of frame `frame` of the synthetic room into a volume of `res` voxels through `entry`."""
    cfg = KFusionConfig(compute_size_ratio=csr)
    w, h = cfg.compute_size
    depth_u16, gt, _ = synth.make_sequence(frame + 1, seed=2, noise=True)
    depth = kf_ref.mm2meters(depth_u16[frame], w, h, 1000.0)
    inv_pose = kf_ref.mat_inverse(gt[frame].astype(np.float32))
    K = kf_ref.camera_matrix(cfg.camera)
    if vol is None:
        vol = torch.from_numpy(kf_ref.new_volume(res)).cuda()
    d = torch.from_numpy(depth).cuda()
    res_c, dim_c = (C.c_int32 * 3)(*res), (C.c_float * 3)(8.0, 8.0, 8.0)
    native.check(entry(vp(vol), res_c, dim_c, 0, res[2], vp(d), w, h, hp(m4(inv_pose)), hp(m4(K)), 0.1, 100.0, stream))
    if wait:
        torch.cuda.synchronize()
    return vol, d


def test_no_alloc_integrate_matches_the_allocating_entry_point(torch_cuda):
    torch = torch_cuda
    L = native.lib()
    # two calls in a row, the second with the smaller input: the array keeps the first call's (longer) list
    for res, csr, frame in (((160, 144, 128), 2, 3), ((64, 48, 40), 4, 1)):
        want, _ = _case(torch, L, res, csr, frame, L.kfh_k_integrate)
        got, _ = _case(torch, L, res, csr, frame, L.kfh_k_integrate_noalloc)
        want, got = want.cpu().numpy(), got.cpu().numpy()
        assert (want[..., 1] > 0).sum() > 1000          # something was fused
        assert np.array_equal(got, want)


def test_no_alloc_integrate_back_to_back_on_one_stream(torch_cuda):
    """Nothing waits between the calls: each launch group reads the pose and the list of its own call."""
    torch = torch_cuda
    L = native.lib()
    res = (96, 96, 96)
    want = None
    for frame in range(4):
        want, _ = _case(torch, L, res, 4, frame, L.kfh_k_integrate, vol=want)
    s = torch.cuda.Stream()
    got, keep = None, []
    for frame in range(4):
        got, d = _case(torch, L, res, 4, frame, L.kfh_k_integrate_noalloc, vol=got, stream=C.c_void_p(s.cuda_stream),
                       wait=False)
        keep.append(d)                                   # the depth maps stay alive until the stream has drained
    s.synchronize()
    assert np.array_equal(got.cpu().numpy(), want.cpu().numpy())


def test_no_alloc_integrate_under_stream_capture(torch_cuda):
    """The callers it is for: a stream capture in the strictest mode turns any hipMalloc / hipFree / wait made by a
    captured call into an error.  The captured call must survive it, and replaying the graph must fuse the frame
    once per replay exactly as the allocating entry point does per call."""
    torch = torch_cuda
    L = native.lib()
    res, csr, frame = (96, 96, 96), 4, 2
    cfg = KFusionConfig(compute_size_ratio=csr)
    w, h = cfg.compute_size
    depth_u16, gt, _ = synth.make_sequence(frame + 1, seed=2, noise=True)
    d = torch.from_numpy(kf_ref.mm2meters(depth_u16[frame], w, h, 1000.0)).cuda()
    inv_pose = m4(kf_ref.mat_inverse(gt[frame].astype(np.float32)))
    K = m4(kf_ref.camera_matrix(cfg.camera))
    res_c, dim_c = (C.c_int32 * 3)(*res), (C.c_float * 3)(8.0, 8.0, 8.0)
    want = torch.from_numpy(kf_ref.new_volume(res)).cuda()
    for _ in range(3):
        native.check(L.kfh_k_integrate(vp(want), res_c, dim_c, 0, res[2], vp(d), w, h, hp(inv_pose), hp(K), 0.1, 100.0, None))
    # the first call on a device looks the array's address up: outside the capture, into a volume of its own
    warm = torch.from_numpy(kf_ref.new_volume(res)).cuda()
    native.check(L.kfh_k_integrate_noalloc(vp(warm), res_c, dim_c, 0, res[2], vp(d), w, h, hp(inv_pose), hp(K), 0.1, 100.0, None))
    torch.cuda.synchronize()
    got = torch.from_numpy(kf_ref.new_volume(res)).cuda()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="global"):
        native.check(L.kfh_k_integrate_noalloc(vp(got), res_c, dim_c, 0, res[2], vp(d), w, h, hp(inv_pose), hp(K), 0.1,
                                               100.0, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), kf_ref.new_volume(res))      # captured, not executed
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), want.cpu().numpy())


def test_no_alloc_integrate_refuses_what_does_not_fit(torch_cuda):
    torch = torch_cuda
    L = native.lib()
    tiny = torch.zeros(16, dtype=torch.int32, device="cuda")       # never touched: the size check comes first
    d = torch.zeros((120, 160), dtype=torch.float32, device="cuda")
    res_c, dim_c = (C.c_int32 * 3)(4096, 4096, 4096), (C.c_float * 3)(8.0, 8.0, 8.0)
    eye = np.eye(4, dtype=np.float32)
    rc = L.kfh_k_integrate_noalloc(vp(tiny), res_c, dim_c, 0, 4096, vp(d), 160, 120, hp(m4(eye)), hp(m4(eye)), 0.1, 100.0,
                                   None)
    assert rc == 1                                                  # KFH_ERR_INVALID_ARG
    assert b"kfh_k_integrate_noalloc holds" in L.kfh_last_error()
    torch.cuda.synchronize()
