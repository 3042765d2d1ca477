"""ctypes binding of libkfusion_hip_core.so (include/kfusion_hip.h).

There is no fallback: if the shared object is missing the import of this module's `lib()` raises,
and every compute entry point needs a HIP device.
"""
import ctypes as C
import os

from .config import KfhConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
# KFH_CORE_LIB: another build of the same library (the diagnostic one: make -C slambench_amd/csrc STAMPS=1)
LIB_PATH = os.environ.get("KFH_CORE_LIB") or os.path.join(_HERE, "lib", "libkfusion_hip_core.so")

KFH_OK = 0
BUF_FLOAT_DEPTH, BUF_SCALED_DEPTH, BUF_INPUT_VERTEX, BUF_INPUT_NORMAL = 0, 1, 2, 3
BUF_MODEL_VERTEX, BUF_MODEL_NORMAL, BUF_VOLUME, BUF_RAY_T, BUF_BRICK_MARKS, BUF_MODEL_TN = 4, 5, 6, 7, 8, 9
RENDER_DEPTH, RENDER_TRACK, RENDER_VOLUME = 0, 1, 2

# every symbol include/kfusion_hip.h declares
EXPORTED_SYMBOLS = (
    "kfh_last_error", "kfh_version", "kfh_device_count",
    "kfh_create", "kfh_destroy", "kfh_reset",
    "kfh_upload_depth", "kfh_set_depth_device",
    "kfh_preprocess", "kfh_track", "kfh_integrate", "kfh_raycast", "kfh_end_frame", "kfh_process_frame",
    "kfh_sync",
    "kfh_get_pose", "kfh_set_pose", "kfh_get_status", "kfh_frame_index", "kfh_integrate_totals", "kfh_compute_size",
    "kfh_download", "kfh_device_ptr", "kfh_stream", "kfh_set_timing", "kfh_phase_ms", "kfh_phase_ms_frame",
    "kfh_raycast_slab", "kfh_composite_slabs", "kfh_composite_tile", "kfh_track_rows",
    "kfh_group_create", "kfh_group_destroy", "kfh_group_reset", "kfh_group_size", "kfh_group_member",
    "kfh_group_upload_depth", "kfh_group_set_depth_device", "kfh_group_process_frame", "kfh_group_sync",
    "kfh_group_get_pose",
    "kfh_render", "kfh_render_device", "kfh_enable_track_render",
    "kfh_k_render_depth", "kfh_k_render_track", "kfh_k_render_volume",
    "kfh_k_preprocess", "kfh_k_bilateral", "kfh_k_half_sample", "kfh_k_vertex_normal",
    "kfh_k_track_reduce", "kfh_k_update_pose", "kfh_k_integrate", "kfh_k_integrate_noalloc", "kfh_k_raycast",
)


# int (*kfh_allreduce_fn)(void *user, float *d_sums32, void *stream)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)


class FrameStatus(C.Structure):
    _fields_ = [("frame", C.c_int32), ("tracked", C.c_int32), ("integrated", C.c_int32),
                ("raycasted", C.c_int32), ("pose", C.c_float * 16), ("sums", C.c_float * 32),
                ("icp_iterations", C.c_int32), ("slow_solves", C.c_int32),
                ("examined_solves", C.c_int32), ("reserved_", C.c_int32)]


class KfhError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("kfusion_hip error %d: %s" % (code, message))
        self.code = code


_lib = None


def lib():
    """Load the core library; raises if it has not been built (there is no CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `make -C slambench_amd/csrc` (or __graft_entry__.build())" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
    L.kfh_last_error.restype = C.c_char_p
    L.kfh_last_error.argtypes = []
    L.kfh_version.restype = C.c_int
    sig = {
        "kfh_device_count": [C.POINTER(C.c_int)],
        "kfh_create": [C.POINTER(KfhConfig), C.POINTER(vp)],
        "kfh_destroy": [vp], "kfh_reset": [vp],
        "kfh_upload_depth": [vp, vp], "kfh_set_depth_device": [vp, vp],
        "kfh_preprocess": [vp], "kfh_track": [vp], "kfh_integrate": [vp], "kfh_raycast": [vp],
        "kfh_end_frame": [vp], "kfh_process_frame": [vp], "kfh_sync": [vp],
        "kfh_get_pose": [vp, vp], "kfh_set_pose": [vp, vp],
        "kfh_get_status": [vp, i32, C.POINTER(FrameStatus)],
        "kfh_frame_index": [vp, C.POINTER(C.c_int)],
        "kfh_integrate_totals": [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)],
        "kfh_compute_size": [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)],
        "kfh_download": [vp, i32, i32, vp, C.c_size_t],
        "kfh_device_ptr": [vp, i32, i32, C.POINTER(vp), C.POINTER(C.c_size_t)],
        "kfh_stream": [vp, C.POINTER(vp)],
        "kfh_set_timing": [vp, i32], "kfh_phase_ms": [vp, vp], "kfh_phase_ms_frame": [vp, i32, vp],
        "kfh_raycast_slab": [vp], "kfh_composite_slabs": [vp, vp, i32],
        "kfh_composite_tile": [vp, vp, i32, i32, i32, vp],
        "kfh_track_rows": [vp, i32, i32, ALLREDUCE_FN, vp],
        "kfh_group_create": [C.POINTER(KfhConfig), C.POINTER(C.c_int), i32, C.POINTER(vp)],
        "kfh_group_destroy": [vp], "kfh_group_reset": [vp], "kfh_group_size": [vp, C.POINTER(C.c_int)],
        "kfh_group_member": [vp, i32, C.POINTER(vp)],
        "kfh_group_upload_depth": [vp, vp], "kfh_group_set_depth_device": [vp, C.POINTER(vp)],
        "kfh_group_process_frame": [vp], "kfh_group_sync": [vp], "kfh_group_get_pose": [vp, vp],
        "kfh_k_preprocess": [vp, vp, i32, i32, vp, i32, i32, f32, vp],
        "kfh_k_bilateral": [vp, vp, i32, i32, vp],
        "kfh_k_half_sample": [vp, vp, i32, i32, f32, vp],
        "kfh_k_vertex_normal": [vp, vp, vp, i32, i32, f32, f32, f32, f32, vp],
        "kfh_k_track_reduce": [vp, vp, i32, vp, vp, vp, i32, i32, vp, vp, i32, i32, vp, vp, f32, f32, vp],
        "kfh_k_update_pose": [vp, vp, vp, vp, f32, vp],
        "kfh_k_integrate": [vp, vp, vp, i32, i32, vp, i32, i32, vp, vp, f32, f32, vp],
        "kfh_k_integrate_noalloc": [vp, vp, vp, i32, i32, vp, i32, i32, vp, vp, f32, f32, vp],
        "kfh_k_raycast": [vp, vp, i32, i32, vp, vp, vp, vp, f32, f32, f32, f32, vp],
        "kfh_render": [vp, i32, vp, vp, C.c_size_t], "kfh_render_device": [vp, i32, vp, vp],
        "kfh_enable_track_render": [vp, i32],
        "kfh_k_render_depth": [vp, vp, i32, i32, f32, f32, vp],
        "kfh_k_render_track": [vp, vp, i32, i32, i32, vp],
        "kfh_k_render_volume": [vp, i32, i32, vp, vp, vp, vp, f32, f32, f32, f32, vp, vp, vp],
    }
    for name, argtypes in sig.items():
        fn = getattr(L, name)
        fn.restype = C.c_int
        fn.argtypes = argtypes
    _lib = L
    return L


def check(rc):
    if rc != KFH_OK:
        raise KfhError(rc, lib().kfh_last_error().decode("utf-8", "replace"))
