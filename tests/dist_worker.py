"""Worker of tests/test_distributed_cpu.py: one rank of a gloo group rehearsing the sharded exchange on CPU."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slambench_amd.sharded import composite_numpy, slab_bounds  # noqa: E402


def fake_slab_result(rank, world, h, w, seed=7):
    """Per-slab ray segments in the packed layout {t, normal}: each pixel is hit in some slabs, missed in others."""
    rng = np.random.default_rng(seed)                  # same stream on every rank
    tn_all = rng.normal(size=(world, h, w, 4)).astype(np.float32)
    t_all = rng.uniform(0.4, 4.0, size=(world, h, w)).astype(np.float32)
    t_all[rng.random((world, h, w)) < 0.5] = 0.0       # miss
    tn_all[..., 0] = t_all
    return tn_all


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    h, w, z = 12, 16, 50
    # 1. slab partition: contiguous cover of [0, z)
    z0, z1 = slab_bounds(z, world, rank)
    bounds = [None] * world
    dist.all_gather_object(bounds, (z0, z1))
    assert bounds[0][0] == 0 and bounds[-1][1] == z
    assert all(bounds[i][1] == bounds[i + 1][0] for i in range(world - 1))
    assert max(b[1] - b[0] for b in bounds) - min(b[1] - b[0] for b in bounds) <= 1
    # 2. exchange: ONE all-gather of the packed segments, then the nearest-hit composite on every rank
    tn_all = fake_slab_result(rank, world, h, w)
    mine = torch.from_numpy(tn_all[rank].copy())
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    gathered = torch.stack(out).numpy()
    assert np.array_equal(gathered, tn_all)
    comp = composite_numpy(gathered)
    # 3. every rank ends with the same map (checksum all-reduced with MIN and MAX)
    chk = torch.tensor([float(comp[..., 0].sum()), float(comp[..., 1:].sum())], dtype=torch.float64)
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert torch.equal(lo, hi)
    # 4. the composite rule itself
    t_all = tn_all[..., 0]
    masked = np.where(t_all > 0, t_all, np.inf)
    assert np.array_equal(comp[..., 0], np.where(np.isfinite(masked.min(0)), masked.min(0), 0).astype(np.float32))
    miss = ~np.isfinite(masked.min(0))
    assert np.all(comp[miss] == np.array([0, -2, 0, 0], np.float32))
    arg = masked.argmin(0)
    yy, xx = np.nonzero(~miss)
    assert np.array_equal(comp[yy, xx], tn_all[arg[yy, xx], yy, xx])
    # 4b. the row-sharded ICP's exchange: every rank contributes one row of 32 sums, all end with the same total
    row = torch.arange(32, dtype=torch.float32) * (rank + 1)
    dist.all_reduce(row)
    assert torch.equal(row, torch.arange(32, dtype=torch.float32) * (world * (world + 1) // 2))
    # 5. frames/s aggregation of the benchmark: MAX of the per-rank elapsed time
    el = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    assert el.item() == float(world)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("DIST_OK")


if __name__ == "__main__":
    main()
