"""Z-slab sharding of one KinectFusion volume across the GPUs of a node (one process per GPU).

Rank g owns the planes z in [g*Z/N, (g+1)*Z/N) of the short2 volume, plus two halo planes per side that it
integrates redundantly, so integration needs no exchange at all.  Every rank preprocesses the (broadcast) depth
frame and runs the full ICP on its own copy of the model maps — the 32-float ICP sums are therefore identical
on all ranks without an all-reduce.  The only exchange step of a frame is the raycast: each rank marches the
part of every ray that lies inside its slab, the per-slab results (hit distance, vertex, normal: 36 bytes per
pixel) are all-gathered, and every rank keeps the nearest hit per pixel (kfh_composite_slabs).

The reference has no multi-device path at all (no NCCL/MPI anywhere in its tree); this module is the new
component the volume-size axis of the reference's design space (volume-resolution) needs on an 8-GPU node.

Marching per slab restarts the coarse/fine step state where a ray enters a slab, so the composited maps are
not bit-identical to a single-device raycast (hit positions differ by less than one fine step before the
zero-crossing interpolation); parity for the sharded mode is defined against a single-device run of the same
slab-wise algorithm.
"""
import ctypes as C

import numpy as np

from . import native
from .config import KFusionConfig
from .kfusion import KFusion


def slab_bounds(z_res, world_size, rank):
    """Contiguous, nearly equal z ranges; the first z_res % world_size ranks get one extra plane."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    if world_size > z_res:
        raise ValueError("more ranks (%d) than z planes (%d)" % (world_size, z_res))
    base, extra = divmod(z_res, world_size)
    z0 = rank * base + min(rank, extra)
    return z0, z0 + base + (1 if rank < extra else 0)


def composite_numpy(t_parts, v_parts, n_parts):
    """Host restatement of kfh_composite_slabs: per pixel, the candidate with the smallest positive t."""
    t = np.asarray(t_parts, np.float32)
    masked = np.where(t > 0, t, np.inf)
    arg = masked.argmin(axis=0)
    hit = np.isfinite(masked.min(axis=0))
    idx = np.indices(arg.shape)
    v = np.asarray(v_parts)[(arg,) + tuple(idx)]
    n = np.asarray(n_parts)[(arg,) + tuple(idx)]
    v = np.where(hit[..., None], v, 0.0).astype(np.float32)
    miss_normal = np.zeros(4, np.float32)
    miss_normal[0] = -2.0
    n = np.where(hit[..., None], n, miss_normal).astype(np.float32)
    return np.where(hit, masked.min(axis=0), 0.0).astype(np.float32), v, n


class SlabKFusion:
    """One rank's share of a sharded KinectFusion run.  `dist` is torch.distributed (already initialised)
    or None for a single process that owns the whole volume."""

    def __init__(self, cfg: KFusionConfig, rank=0, world_size=1, dist=None, torch=None, icp="replicated"):
        if icp not in ("replicated", "allreduce"):
            raise ValueError("icp must be 'replicated' or 'allreduce'")
        self.rank, self.world = rank, world_size
        self.dist, self.torch = dist, torch
        self.icp = icp if world_size > 1 else "replicated"
        z0, z1 = slab_bounds(cfg.volume_resolution[2], world_size, rank)
        self.cfg = KFusionConfig(**{**cfg.__dict__, "slab": (z0, z1) if world_size > 1 else (0, 0)})
        self.kf = KFusion(self.cfg)
        w, h = self.cfg.compute_size
        self.npix = w * h
        self._bufs = None

    def close(self):
        self.kf.close()

    # -- exchange -----------------------------------------------------------------------------------------
    def _gather(self):
        """All-gather (t, vertex, normal) of this rank's slab march; returns device pointers of the gathered
        [world][...] arrays, kept alive in self._bufs."""
        torch, dist = self.torch, self.dist
        kf = self.kf
        w, h = self.cfg.compute_size
        if self._bufs is None:
            dev = "cuda"
            self._bufs = {
                "t": torch.empty((self.world, h, w), dtype=torch.float32, device=dev),
                "v": torch.empty((self.world, h, w, 4), dtype=torch.float32, device=dev),
                "n": torch.empty((self.world, h, w, 4), dtype=torch.float32, device=dev),
            }
        kf.sync()        # the slab march ran on the context's stream; the collectives use torch's
        parts = {}
        for key, buf_id, shape in (("t", native.BUF_RAY_T, (h, w)), ("v", native.BUF_MODEL_VERTEX, (h, w, 4)),
                                   ("n", native.BUF_MODEL_NORMAL, (h, w, 4))):
            ptr, nbytes = kf.device_ptr(buf_id)
            mine = _wrap_device_buffer(torch, ptr, shape)
            if dist.get_backend() == "nccl":
                dist.all_gather_into_tensor(self._bufs[key], mine)
            else:                     # gloo: stage through the host (used by the CPU / single-GPU rehearsal)
                host = mine.cpu()
                out = [torch.empty_like(host) for _ in range(self.world)]
                dist.all_gather(out, host)
                self._bufs[key].copy_(torch.stack(out))
            parts[key] = self._bufs[key]
        torch.cuda.synchronize()
        return parts

    # -- one frame ----------------------------------------------------------------------------------------
    def process_frame(self, depth):
        kf = self.kf
        kf.set_depth(depth)
        kf.preprocessing()
        kf.tracking()
        kf.integration()
        if self.world == 1:
            kf.raycasting()
        else:
            native.check(kf._L.kfh_raycast_slab(kf._ctx))
            if kf.frame > 2:
                p = self._gather()
                native.check(kf._L.kfh_composite_slabs(kf._ctx, C.c_void_p(p["t"].data_ptr()),
                                                       C.c_void_p(p["v"].data_ptr()),
                                                       C.c_void_p(p["n"].data_ptr()), self.world))
        kf.end_frame()


def _wrap_device_buffer(torch, ptr, shape):
    """A torch view of memory owned by the C library (no copy), through the CUDA array interface."""

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(h, device="cuda")
