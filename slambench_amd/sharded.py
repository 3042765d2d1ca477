"""Z-slab sharding of one KinectFusion volume across the GPUs of a node (one process per GPU).

Rank g owns the planes z in [g*Z/N, (g+1)*Z/N) of the short2 volume, plus two halo planes per side that it
integrates redundantly, so integration needs no exchange at all.  Every rank preprocesses the (broadcast) depth
frame and runs the full ICP on its own copy of the model maps — the 32-float ICP sums are therefore identical
on all ranks without an all-reduce (icp="allreduce" splits the pixels by rows instead and all-reduces the sums
after every iteration: kfh_track_rows).  The only other exchange step of a frame is the raycast: each rank marches
the part of every ray that lies inside its slab, the per-slab results (hit distance and normal: 16 bytes per pixel;
the vertex is a function of the pixel and the distance) are all-gathered in one collective, and every rank keeps the
nearest hit per pixel (kfh_composite_slabs).  exchange="tiles" splits the image among the ranks instead: an all-to-all
hands every rank the N candidates of its 1/N of the pixels, the rank selects (kfh_composite_tile), and an all-gather of
the selected shares completes the map everywhere: two collectives that move 2/N of the bytes of the single all-gather,
which is what counts on point-to-point links once N is 4 or 8.

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


def composite_numpy(tn_parts):
    """Host restatement of kfh_composite_slabs on packed maps [parts][h][w][4] = {t, normal}: per pixel, the
    candidate with the smallest positive t; a pixel no slab hit gets {0, (-2, 0, 0)}."""
    tn = np.asarray(tn_parts, np.float32)
    t = tn[..., 0]
    masked = np.where(t > 0, t, np.inf)
    arg = masked.argmin(axis=0)
    hit = np.isfinite(masked.min(axis=0))
    idx = np.indices(arg.shape)
    best = tn[(arg,) + tuple(idx)]
    miss = np.array([0.0, -2.0, 0.0, 0.0], np.float32)
    return np.where(hit[..., None], best, miss).astype(np.float32)


class SlabKFusion:
    """One rank's share of a sharded KinectFusion run.  `dist` is torch.distributed (already initialised)
    or None for a single process that owns the whole volume.

    icp="replicated": every rank tracks the whole image against its own copy of the model map (identical sums on
    all ranks, no collective).  icp="allreduce": every rank tracks its share of the rows of each pyramid level and
    the 32 ICP sums are all-reduced after every iteration (kfh_track_rows); 19 small collectives per frame."""

    def __init__(self, cfg: KFusionConfig, rank=0, world_size=1, dist=None, torch=None, icp="replicated",
                 force_exchange=False, exchange="gather"):
        """force_exchange: take the exchange path (slab march, all-gather, composite, and the all-reduce when
        icp="allreduce") even in a world of one rank: a one-GPU box can then run the device-side collectives."""
        if icp not in ("replicated", "allreduce"):
            raise ValueError("icp must be 'replicated' or 'allreduce'")
        if exchange not in ("gather", "tiles"):
            raise ValueError("exchange must be 'gather' or 'tiles'")
        self.rank, self.world = rank, world_size
        self.dist, self.torch = dist, torch
        self.exchange = world_size > 1 or (force_exchange and dist is not None)
        self.icp = icp if self.exchange else "replicated"
        z0, z1 = slab_bounds(cfg.volume_resolution[2], world_size, rank)
        self.cfg = KFusionConfig(**{**cfg.__dict__, "slab": (z0, z1) if world_size > 1 else (0, 0)})
        self.kf = KFusion(self.cfg)
        w, h = self.cfg.compute_size
        self.npix = w * h
        if exchange == "tiles" and self.npix % world_size != 0:
            raise ValueError("exchange='tiles' needs a pixel count (%d) divisible by the number of ranks (%d)"
                             % (self.npix, world_size))
        self.exchange_mode = exchange
        self._recv = self._tile = None
        self._gathered = None
        self._mine = None
        self._sums = {}
        self.allreduce_calls = 0
        self._device_collectives = self.exchange and dist.get_backend() == "nccl"
        # RCCL work is ordered against torch's current stream: make the context's own stream that stream around
        # the collectives, so that neither side has to wait for the host
        self._stream = torch.cuda.ExternalStream(self.kf.stream()) if self._device_collectives else None
        self._callback = native.ALLREDUCE_FN(self._allreduce_sums)      # kept alive with the object

    def close(self):
        self.kf.close()

    # -- exchange -----------------------------------------------------------------------------------------
    def _allreduce_sums(self, user, d_sums32, stream):
        """kfh_allreduce_fn: in-place sum of one row of 32 ICP sums over all ranks, in stream order."""
        try:
            torch, dist = self.torch, self.dist
            t = self._sums.get(d_sums32)
            if t is None:
                t = self._sums[d_sums32] = _wrap_device_buffer(torch, d_sums32, (32,))
            self.allreduce_calls += 1
            if self._device_collectives:
                with torch.cuda.stream(self._stream):
                    dist.all_reduce(t)
            else:                         # gloo: the row takes a round trip through host memory
                self.kf.sync()
                host = t.cpu()
                dist.all_reduce(host)
                t.copy_(host)
                torch.cuda.synchronize()
            return 0
        except Exception as exc:          # an exception must not unwind through the C caller
            self._callback_error = exc
            return 1

    def _gather(self):
        """All-gather the packed slab segments {t, normal} (16 bytes per pixel) of all ranks; returns the device
        pointer of the gathered [world][h][w][4] array."""
        torch, dist = self.torch, self.dist
        kf = self.kf
        w, h = self.cfg.compute_size
        if self._gathered is None:
            self._gathered = torch.empty((self.world, h, w, 4), dtype=torch.float32, device="cuda")
            ptr, _ = kf.device_ptr(native.BUF_MODEL_TN)
            self._mine = _wrap_device_buffer(torch, ptr, (h, w, 4))
        if self._device_collectives:
            with torch.cuda.stream(self._stream):
                dist.all_gather_into_tensor(self._gathered, self._mine)
        else:                             # gloo: stage through the host (CPU / single-GPU rehearsal)
            kf.sync()
            host = self._mine.cpu()
            out = [torch.empty_like(host) for _ in range(self.world)]
            dist.all_gather(out, host)
            self._gathered.copy_(torch.stack(out))
            torch.cuda.synchronize()
        return self._gathered.data_ptr()

    def _exchange_tiles(self):
        """The raycast exchange with the image split among the ranks: all-to-all of the candidates, selection of this
        rank's share, all-gather of the selected shares into the packed model map."""
        torch, dist, kf = self.torch, self.dist, self.kf
        n, share = self.world, self.npix // self.world
        if self._mine is None:
            ptr, _ = kf.device_ptr(native.BUF_MODEL_TN)
            self._mine = _wrap_device_buffer(torch, ptr, self.cfg.compute_size[::-1] + (4,))
        if self._recv is None:
            self._recv = torch.empty((n, share, 4), dtype=torch.float32, device="cuda")
            self._tile = torch.empty((share, 4), dtype=torch.float32, device="cuda")
        mine = self._mine.view(-1)
        if self._device_collectives:
            with torch.cuda.stream(self._stream):
                dist.all_to_all_single(self._recv.view(-1), mine)
        else:                             # gloo: through the host
            kf.sync()
            host = mine.cpu()
            got = torch.empty_like(host)
            dist.all_to_all_single(got, host)
            self._recv.view(-1).copy_(got)
            torch.cuda.synchronize()
        native.check(kf._L.kfh_composite_tile(kf._ctx, C.c_void_p(self._recv.data_ptr()), n, self.rank * share, share,
                                              C.c_void_p(self._tile.data_ptr())))
        if self._device_collectives:
            with torch.cuda.stream(self._stream):
                dist.all_gather_into_tensor(mine, self._tile.view(-1))
        else:
            kf.sync()
            host = self._tile.view(-1).cpu()
            out = [torch.empty_like(host) for _ in range(n)]
            dist.all_gather(out, host)
            mine.copy_(torch.cat(out))
            torch.cuda.synchronize()

    # -- one frame ----------------------------------------------------------------------------------------
    def process_frame(self, depth):
        kf = self.kf
        kf.set_depth(depth)
        kf.preprocessing()
        if self.icp == "allreduce":
            self._callback_error = None
            rc = kf._L.kfh_track_rows(kf._ctx, self.rank, self.world, self._callback, None)
            if self._callback_error is not None:
                raise self._callback_error
            native.check(rc)
        else:
            kf.tracking()
        kf.integration()
        if not self.exchange:
            kf.raycasting()
        else:
            native.check(kf._L.kfh_raycast_slab(kf._ctx))
            if kf.frame > 2:
                if self.exchange_mode == "tiles":
                    self._exchange_tiles()
                else:
                    native.check(kf._L.kfh_composite_slabs(kf._ctx, C.c_void_p(self._gather()), self.world))
        kf.end_frame()


def _wrap_device_buffer(torch, ptr, shape):
    """A torch view of memory owned by the C library (no copy), through the CUDA array interface."""

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(h, device="cuda")
