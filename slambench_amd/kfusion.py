"""Host-side mirror of the kfusion use-case's `Kfusion` class on top of the C ABI.

Method names follow the use-case (preprocessing / tracking / integration / raycasting are the four
phases the plugin brackets in sb_process_once; the reference's log header lists them as
Duration_Preprocessing .. Duration_Raycasting,
icra2018_results/1080/violons_libkfusion-cpp_living_room_traj0_loop.log:36).  All work is queued on
the context's HIP stream; only the accessors synchronise.
"""
import ctypes as C

import numpy as np

from . import native
from .config import KFusionConfig


def _device_pointer(obj):
    """Device address of a caller-owned buffer: an int, or anything with .data_ptr() (torch tensors)."""
    if isinstance(obj, int):
        return obj
    if hasattr(obj, "data_ptr"):
        return int(obj.data_ptr())
    raise TypeError("expected a device pointer (int) or an object with data_ptr()")


class KFusion:
    def __init__(self, cfg: KFusionConfig, _borrowed_ctx=None):
        self.cfg = cfg
        self._L = native.lib()
        self._owned = _borrowed_ctx is None
        if self._owned:
            self._ctx = C.c_void_p()
            ccfg = cfg.to_c()
            native.check(self._L.kfh_create(C.byref(ccfg), C.byref(self._ctx)))
        else:                           # a member of a group: the group destroys it
            self._ctx = _borrowed_ctx
        self.width, self.height = cfg.compute_size

    # ------------------------------------------------------------------ lifecycle
    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            if self._owned:
                self._L.kfh_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def reset(self):
        native.check(self._L.kfh_reset(self._ctx))

    # ------------------------------------------------------------------ frame input
    def set_depth(self, depth):
        """Host uint16 array [H, W] (copied through pinned staging) or a device buffer already in HBM."""
        if isinstance(depth, np.ndarray):
            w, h = self.cfg.input_size
            if depth.dtype != np.uint16 or depth.shape != (h, w):
                raise ValueError("depth must be uint16 [%d, %d]" % (h, w))
            d = np.ascontiguousarray(depth)
            native.check(self._L.kfh_upload_depth(self._ctx, d.ctypes.data_as(C.c_void_p)))
        else:
            native.check(self._L.kfh_set_depth_device(self._ctx, C.c_void_p(_device_pointer(depth))))

    # ------------------------------------------------------------------ phases
    def preprocessing(self):
        native.check(self._L.kfh_preprocess(self._ctx))

    def tracking(self):
        native.check(self._L.kfh_track(self._ctx))

    def integration(self):
        native.check(self._L.kfh_integrate(self._ctx))

    def raycasting(self):
        native.check(self._L.kfh_raycast(self._ctx))

    def end_frame(self):
        native.check(self._L.kfh_end_frame(self._ctx))

    def process_frame(self, depth=None):
        """One sb_process_once: preprocessing -> tracking -> integration -> raycasting."""
        if depth is not None:
            self.set_depth(depth)
        native.check(self._L.kfh_process_frame(self._ctx))

    def sync(self):
        native.check(self._L.kfh_sync(self._ctx))

    # ------------------------------------------------------------------ results
    @property
    def frame(self):
        f = C.c_int()
        native.check(self._L.kfh_frame_index(self._ctx, C.byref(f)))
        return f.value

    @property
    def pose(self):
        out = np.empty(16, np.float32)
        native.check(self._L.kfh_get_pose(self._ctx, out.ctypes.data_as(C.c_void_p)))
        return out.reshape(4, 4)

    def set_pose(self, pose):
        p = np.ascontiguousarray(pose, dtype=np.float32).reshape(16)
        native.check(self._L.kfh_set_pose(self._ctx, p.ctypes.data_as(C.c_void_p)))

    def status(self, frame):
        st = native.FrameStatus()
        native.check(self._L.kfh_get_status(self._ctx, int(frame), C.byref(st)))
        return {
            "frame": st.frame, "tracked": bool(st.tracked), "integrated": bool(st.integrated),
            "raycasted": bool(st.raycasted),
            "pose": np.array(st.pose, np.float32).reshape(4, 4),
            "sums": np.array(st.sums, np.float32),
            "icp_iterations": st.icp_iterations, "slow_solves": st.slow_solves,
            "examined_solves": st.examined_solves,
        }

    def integrate_totals(self):
        """(boxes kept by the integration plans, integrations run, voxels per box) since creation / reset."""
        b, n, v = C.c_uint64(), C.c_uint64(), C.c_int32()
        native.check(self._L.kfh_integrate_totals(self._ctx, C.byref(b), C.byref(n), C.byref(v)))
        return b.value, n.value, v.value

    def trajectory(self, first=0, count=None):
        n = self.frame
        if count is None:
            count = n - first
        return np.stack([self.status(f)["pose"] for f in range(first, first + count)]) if count > 0 else \
            np.zeros((0, 4, 4), np.float32)

    def device_ptr(self, buffer, level=0):
        p, n = C.c_void_p(), C.c_size_t()
        native.check(self._L.kfh_device_ptr(self._ctx, buffer, level, C.byref(p), C.byref(n)))
        return p.value, n.value

    def stream(self):
        s = C.c_void_p()
        native.check(self._L.kfh_stream(self._ctx, C.byref(s)))
        return s.value

    def download(self, buffer, level=0):
        """Copy one of the context's buffers to a numpy array (synchronises)."""
        _, nbytes = self.device_ptr(buffer, level)
        w, h = self.width >> level, self.height >> level
        if buffer in (native.BUF_FLOAT_DEPTH, native.BUF_SCALED_DEPTH, native.BUF_RAY_T):
            out = np.empty((h, w), np.float32)
        elif buffer in (native.BUF_INPUT_VERTEX, native.BUF_INPUT_NORMAL, native.BUF_MODEL_VERTEX,
                        native.BUF_MODEL_NORMAL, native.BUF_MODEL_TN):
            out = np.empty((h, w, 4), np.float32)
        elif buffer == native.BUF_VOLUME:
            rx, ry, _ = self.cfg.volume_resolution
            out = np.empty((nbytes // (4 * rx * ry), ry, rx, 2), np.int16)
        elif buffer == native.BUF_BRICK_MARKS:
            out = np.empty(tuple((r + 7) // 8 for r in reversed(self.cfg.volume_resolution)), np.uint8)
        else:
            raise ValueError("unknown buffer %r" % (buffer,))
        assert out.nbytes == nbytes, (out.nbytes, nbytes)
        native.check(self._L.kfh_download(self._ctx, buffer, level, out.ctypes.data_as(C.c_void_p),
                                          C.c_size_t(nbytes)))
        return out

    def enable_track_render(self, enabled=True):
        """Tracking stores its per-pixel result codes from now on (needed by render(RENDER_TRACK))."""
        native.check(self._L.kfh_enable_track_render(self._ctx, int(bool(enabled))))

    def render(self, image, view_pose=None):
        """One of the plugin's GUI images as uint8 [h, w, 4] (RGBA, alpha 0); synchronises.
        view_pose (4x4 camera-to-world, RENDER_VOLUME only): None renders from the current pose."""
        out = np.empty((self.height, self.width, 4), np.uint8)
        vp = None
        if view_pose is not None:
            vp = np.ascontiguousarray(view_pose, dtype=np.float32).reshape(16)
        native.check(self._L.kfh_render(self._ctx, int(image), None if vp is None else vp.ctypes.data_as(C.c_void_p),
                                        out.ctypes.data_as(C.c_void_p), C.c_size_t(out.nbytes)))
        return out

    def set_timing(self, mode):
        """0 off, 1 (or True) all four phases, 2 integration only."""
        native.check(self._L.kfh_set_timing(self._ctx, int(mode)))

    def phase_ms(self, frame=None):
        """HIP-event milliseconds {preprocessing, tracking, integration, raycasting} of a frame."""
        ms = np.empty(4, np.float32)
        if frame is None:
            native.check(self._L.kfh_phase_ms(self._ctx, ms.ctypes.data_as(C.c_void_p)))
        else:
            native.check(self._L.kfh_phase_ms_frame(self._ctx, int(frame), ms.ctypes.data_as(C.c_void_p)))
        return ms


class GroupKFusion:
    """kfh_group of include/kfusion_hip.h: ONE sequence whose volume is split into z-slabs over several contexts of
    this process (one per entry of `devices`; an index may repeat), driven from the calling thread.  `members` are
    KFusion views of the contexts (status records, downloads); stepping a member on its own breaks the group."""

    def __init__(self, cfg: KFusionConfig, devices):
        from .sharded import slab_bounds
        self.cfg = cfg
        self._L = native.lib()
        self._g = C.c_void_p()
        devices = [int(d) for d in devices]
        n = len(devices)
        ccfg = cfg.to_c()
        arr = (C.c_int * n)(*devices)
        native.check(self._L.kfh_group_create(C.byref(ccfg), arr, n, C.byref(self._g)))
        self.members = []
        for i in range(n):
            ctx = C.c_void_p()
            native.check(self._L.kfh_group_member(self._g, i, C.byref(ctx)))
            slab = slab_bounds(cfg.volume_resolution[2], n, i) if n > 1 else (0, 0)
            self.members.append(KFusion(KFusionConfig(**{**cfg.__dict__, "slab": slab}), _borrowed_ctx=ctx))

    def close(self):
        if getattr(self, "_g", None) and self._g.value:
            for m in self.members:
                m.close()
            self._L.kfh_group_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        native.check(self._L.kfh_group_reset(self._g))

    def process_frame(self, depth=None):
        """depth: a host uint16 frame (uploaded to every member), or one device pointer per member."""
        if depth is not None:
            if isinstance(depth, np.ndarray):
                d = np.ascontiguousarray(depth)
                native.check(self._L.kfh_group_upload_depth(self._g, d.ctypes.data_as(C.c_void_p)))
            else:
                ptrs = (C.c_void_p * len(self.members))(*[_device_pointer(p) for p in depth])
                native.check(self._L.kfh_group_set_depth_device(self._g, ptrs))
        native.check(self._L.kfh_group_process_frame(self._g))

    def sync(self):
        native.check(self._L.kfh_group_sync(self._g))

    @property
    def pose(self):
        out = np.empty(16, np.float32)
        native.check(self._L.kfh_group_get_pose(self._g, out.ctypes.data_as(C.c_void_p)))
        return out.reshape(4, 4)
