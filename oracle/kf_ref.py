"""ctypes binding of the CPU oracle (oracle/build/libkf_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py.  The product package (slambench_amd) never imports this module.

PARITY UNPINNED: see oracle/kf_ref.h — the kfusion arithmetic is not vendored in the reference
tree, so the oracle is a restatement of the published algorithm, not a build of the reference.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "libkf_ref.so")


class Config(C.Structure):
    """Mirror of kfr_config (oracle/kf_ref.h); same layout as kfh_config."""
    _fields_ = [
        ("input_width", C.c_int32), ("input_height", C.c_int32),
        ("compute_size_ratio", C.c_int32),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("depth_divisor", C.c_float),
        ("vol_res", C.c_int32 * 3), ("vol_dim", C.c_float * 3), ("init_pos", C.c_float * 3),
        ("mu", C.c_float), ("icp_threshold", C.c_float),
        ("pyramid", C.c_int32 * 3),
        ("integration_rate", C.c_int32), ("tracking_rate", C.c_int32),
        ("integrate_filtered", C.c_int32),
        ("slab_z0", C.c_int32), ("slab_z1", C.c_int32),
    ]


class TrackData(C.Structure):
    _fields_ = [("result", C.c_int32), ("error", C.c_float), ("J", C.c_float * 6)]


TRACK_DTYPE = np.dtype([("result", np.int32), ("error", np.float32), ("J", np.float32, (6,))])


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "kf_ref.cpp"))):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.kfr_create.restype = C.c_void_p
        L.kfr_create.argtypes = [C.POINTER(Config)]
        L.kfr_destroy.argtypes = [C.c_void_p]
        L.kfr_process_frame.restype = C.c_int
        L.kfr_process_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.kfr_get_pose.argtypes = [C.c_void_p, C.c_void_p]
        L.kfr_set_pose.argtypes = [C.c_void_p, C.c_void_p]
        L.kfr_frame_index.restype = C.c_int
        L.kfr_frame_index.argtypes = [C.c_void_p]
        for name in ("kfr_float_depth", "kfr_model_vertex", "kfr_model_normal", "kfr_volume",
                     "kfr_reduction", "kfr_tracking_result"):
            getattr(L, name).restype = C.c_void_p
            getattr(L, name).argtypes = [C.c_void_p]
        for name in ("kfr_scaled_depth", "kfr_input_vertex", "kfr_input_normal"):
            getattr(L, name).restype = C.c_void_p
            getattr(L, name).argtypes = [C.c_void_p, C.c_int]
        L.kfr_icp_iterations.restype = C.c_int
        L.kfr_icp_iterations.argtypes = [C.c_void_p]
        L.kfr_cutoff_margin.restype = C.c_double
        L.kfr_cutoff_margin.argtypes = [C.c_void_p]
        L.kfr_set_threads.restype = C.c_int
        L.kfr_set_threads.argtypes = [C.c_int]
        L.kfr_update_pose.restype = C.c_int
        L.kfr_check_pose.restype = C.c_int
        _lib = L
    return _lib


def set_reduce_rows(rows=8):
    """Test aid: the ICP sums over `rows` strided partial rows instead of the reference's 8 (another summation
    order of the same terms).  Process-wide; restore 8 afterwards."""
    lib().kfr_set_reduce_rows(C.c_int(rows))


def set_threads(n=0):
    """OpenMP worker threads of the oracle (0: query only); returns the count in force."""
    return int(lib().kfr_set_threads(int(n)))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _m4(a):
    a = _f32(a).reshape(16)
    return a


# ---------------------------------------------------------------- stateless kernels
def mm2meters(depth_u16, out_w, out_h, divisor=1000.0):
    d = np.ascontiguousarray(depth_u16, dtype=np.uint16)
    ih, iw = d.shape
    out = np.empty((out_h, out_w), np.float32)
    lib().kfr_mm2meters(_p(out), C.c_int(out_w), C.c_int(out_h), _p(d), C.c_int(iw), C.c_int(ih),
                        C.c_float(divisor))
    return out


def bilateral(depth, radius=2, delta=4.0, e_d=0.1):
    d = _f32(depth)
    h, w = d.shape
    out = np.empty_like(d)
    lib().kfr_bilateral(_p(out), _p(d), C.c_int(w), C.c_int(h), C.c_int(radius), C.c_float(delta),
                        C.c_float(e_d))
    return out


def half_sample(depth, e_d=0.1 * 3, r=1):
    d = _f32(depth)
    h, w = d.shape
    out = np.empty((h // 2, w // 2), np.float32)
    lib().kfr_half_sample(_p(out), _p(d), C.c_int(w), C.c_int(h), C.c_float(np.float32(e_d)), C.c_int(r))
    return out


def inverse_camera_matrix(k):
    out = np.empty(16, np.float32)
    lib().kfr_inverse_camera_matrix(_p(out), *[C.c_float(float(v)) for v in k])
    return out.reshape(4, 4)


def camera_matrix(k):
    out = np.empty(16, np.float32)
    lib().kfr_camera_matrix(_p(out), *[C.c_float(float(v)) for v in k])
    return out.reshape(4, 4)


def mat_inverse(m):
    out = np.empty(16, np.float32)
    lib().kfr_mat_inverse(_p(out), _p(_m4(m)))
    return out.reshape(4, 4)


def mat_mul(a, b):
    out = np.empty(16, np.float32)
    lib().kfr_mat_mul(_p(out), _p(_m4(a)), _p(_m4(b)))
    return out.reshape(4, 4)


def depth2vertex(depth, invK):
    d = _f32(depth)
    h, w = d.shape
    out = np.empty((h, w, 3), np.float32)
    lib().kfr_depth2vertex(_p(out), _p(d), C.c_int(w), C.c_int(h), _p(_m4(invK)))
    return out


def vertex2normal(vertex):
    v = _f32(vertex)
    h, w, _ = v.shape
    out = np.empty_like(v)
    lib().kfr_vertex2normal(_p(out), _p(v), C.c_int(w), C.c_int(h))
    return out


def track(in_vertex, in_normal, ref_vertex, ref_normal, Ttrack, view, dist_threshold=0.1,
          normal_threshold=0.8):
    iv, inn = _f32(in_vertex), _f32(in_normal)
    rv, rn = _f32(ref_vertex), _f32(ref_normal)
    ih, iw, _ = iv.shape
    rh, rw, _ = rv.shape
    out = np.zeros((rh, rw), TRACK_DTYPE)
    lib().kfr_track(_p(out), C.c_int(rw), _p(iv), _p(inn), C.c_int(iw), C.c_int(ih), _p(rv), _p(rn),
                    C.c_int(rw), C.c_int(rh), _p(_m4(Ttrack)), _p(_m4(view)),
                    C.c_float(dist_threshold), C.c_float(normal_threshold))
    return out


def reduce(track_data, w, h):
    td = np.ascontiguousarray(track_data)
    assert td.dtype == TRACK_DTYPE
    out = np.empty(8 * 32, np.float32)
    lib().kfr_reduce(_p(out), _p(td), C.c_int(td.shape[1]), C.c_int(w), C.c_int(h))
    return out[:32].copy()


def solve(vals27):
    v = _f32(vals27)
    x = np.empty(6, np.float64)
    lib().kfr_solve(_p(x), _p(v))
    return x


def se3_exp(x6):
    x = np.ascontiguousarray(x6, dtype=np.float64)
    out = np.empty(16, np.float32)
    lib().kfr_se3_exp(_p(out), _p(x))
    return out.reshape(4, 4)


def update_pose(pose, sums, icp_threshold=1e-5):
    p = _m4(pose).copy()
    s = _f32(sums)
    conv = lib().kfr_update_pose(_p(p), _p(s), C.c_float(icp_threshold))
    return p.reshape(4, 4), bool(conv)


def check_pose(pose, old_pose, sums, w, h, track_threshold=0.15):
    p = _m4(pose).copy()
    ok = lib().kfr_check_pose(_p(p), _p(_m4(old_pose)), _p(_f32(sums)), C.c_int(w), C.c_int(h),
                              C.c_float(track_threshold))
    return p.reshape(4, 4), bool(ok)


def new_volume(res):
    vol = np.empty((res[2], res[1], res[0], 2), np.int16)
    vol[..., 0] = 32766
    vol[..., 1] = 0
    return vol


def integrate(vol, dim, depth, inv_track, K, mu=0.1, maxweight=100.0):
    """Updates vol (int16 array [z, y, x, 2]) in place."""
    assert vol.dtype == np.int16 and vol.flags.c_contiguous
    d = _f32(depth)
    h, w = d.shape
    res = (C.c_int32 * 3)(vol.shape[2], vol.shape[1], vol.shape[0])
    dm = (C.c_float * 3)(*[float(v) for v in dim])
    lib().kfr_integrate(_p(vol), res, dm, _p(d), C.c_int(w), C.c_int(h), _p(_m4(inv_track)), _p(_m4(K)),
                        C.c_float(mu), C.c_float(maxweight))
    return vol


def raycast(vol, dim, w, h, view, near=0.4, far=4.0, step=None, largestep=0.075):
    assert vol.dtype == np.int16 and vol.flags.c_contiguous
    res = (C.c_int32 * 3)(vol.shape[2], vol.shape[1], vol.shape[0])
    dm = (C.c_float * 3)(*[float(v) for v in dim])
    if step is None:
        step = np.float32(min(dim)) / np.float32(max(vol.shape[:3]))
    vertex = np.empty((h, w, 3), np.float32)
    normal = np.empty((h, w, 3), np.float32)
    lib().kfr_raycast(_p(vertex), _p(normal), C.c_int(w), C.c_int(h), _p(vol), res, dm, _p(_m4(view)),
                      C.c_float(near), C.c_float(far), C.c_float(float(step)), C.c_float(largestep))
    return vertex, normal


def render_depth(depth, near=0.4, far=4.0):
    d = _f32(depth)
    h, w = d.shape
    out = np.empty((h, w, 4), np.uint8)
    lib().kfr_render_depth(_p(out), _p(d), C.c_int(w), C.c_int(h), C.c_float(near), C.c_float(far))
    return out


def render_track(track_data):
    td = np.ascontiguousarray(track_data)
    assert td.dtype == TRACK_DTYPE
    h, w = td.shape
    out = np.empty((h, w, 4), np.uint8)
    lib().kfr_render_track(_p(out), _p(td), C.c_int(w), C.c_int(h))
    return out


def render_volume(vol, dim, w, h, view, near=0.4, far=8.0, step=None, largestep=0.075,
                  light=(1.0, 1.0, -1.0), ambient=(0.1, 0.1, 0.1)):
    assert vol.dtype == np.int16 and vol.flags.c_contiguous
    res = (C.c_int32 * 3)(vol.shape[2], vol.shape[1], vol.shape[0])
    dm = (C.c_float * 3)(*[float(v) for v in dim])
    if step is None:
        step = np.float32(min(dim)) / np.float32(max(vol.shape[:3]))
    out = np.empty((h, w, 4), np.uint8)
    lib().kfr_render_volume(_p(out), C.c_int(w), C.c_int(h), _p(vol), res, dm, _p(_m4(view)),
                            C.c_float(near), C.c_float(far), C.c_float(float(step)), C.c_float(largestep),
                            (C.c_float * 3)(*light), (C.c_float * 3)(*ambient))
    return out


# ---------------------------------------------------------------- whole pipeline
class KFusionRef:
    """Drives kfr_process_frame; mirrors the upstream Kfusion class + sb_process_once order."""

    def __init__(self, cfg):
        self.cfg = cfg
        self._c = lib().kfr_create(C.byref(cfg))
        r = cfg.compute_size_ratio
        self.w, self.h = cfg.input_width // r, cfg.input_height // r
        self.phase_seconds = np.zeros(4)

    def close(self):
        if self._c:
            lib().kfr_destroy(self._c)
            self._c = None

    def __del__(self):
        self.close()

    def process_frame(self, depth_u16):
        d = np.ascontiguousarray(depth_u16, dtype=np.uint16)
        assert d.shape == (self.cfg.input_height, self.cfg.input_width)
        return lib().kfr_process_frame(self._c, _p(d), _p(self.phase_seconds))

    @property
    def pose(self):
        out = np.empty(16, np.float32)
        lib().kfr_get_pose(self._c, _p(out))
        return out.reshape(4, 4)

    def set_pose(self, pose):
        lib().kfr_set_pose(self._c, _p(_m4(pose)))

    def _view(self, ptr, shape, dtype):
        n = int(np.prod(shape))
        buf = (C.c_byte * (n * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def float_depth(self):
        return self._view(lib().kfr_float_depth(self._c), (self.h, self.w), np.float32)

    def scaled_depth(self, level):
        return self._view(lib().kfr_scaled_depth(self._c, level), (self.h >> level, self.w >> level),
                          np.float32)

    def input_vertex(self, level):
        return self._view(lib().kfr_input_vertex(self._c, level), (self.h >> level, self.w >> level, 3),
                          np.float32)

    def input_normal(self, level):
        return self._view(lib().kfr_input_normal(self._c, level), (self.h >> level, self.w >> level, 3),
                          np.float32)

    def model_vertex(self):
        return self._view(lib().kfr_model_vertex(self._c), (self.h, self.w, 3), np.float32)

    def model_normal(self):
        return self._view(lib().kfr_model_normal(self._c), (self.h, self.w, 3), np.float32)

    def volume(self):
        r = self.cfg.vol_res
        return self._view(lib().kfr_volume(self._c), (r[2], r[1], r[0], 2), np.int16)

    def reduction(self):
        return self._view(lib().kfr_reduction(self._c), (8, 32), np.float32)[0]

    def icp_iterations(self):
        """Track + reduce passes of the last frame on which tracking ran."""
        return lib().kfr_icp_iterations(self._c)

    def cutoff_margin(self):
        """Smallest |ln(|w_i| * 1e6 / max|w|)| over the solves of the last tracked frame: below a few percent a
        keep/drop decision of the singular-value cut-off sat on its limit (float32 sums decide it by rounding)."""
        return float(lib().kfr_cutoff_margin(self._c))

    def tracking_result(self):
        return self._view(lib().kfr_tracking_result(self._c), (self.h, self.w), TRACK_DTYPE)
