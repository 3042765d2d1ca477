"""Worker of tests/test_sharded_procs_gpu.py: one rank of a z-slab run, one process per rank, all ranks on GPU 0.

Runs SlabKFusion.process_frame (the code path `bench.py --mode zslab` runs) over a synthetic sequence and writes
this rank's trajectory, tracked flags and final model maps to OUT_DIR/rank<r>.npz.
usage (under torch.distributed.run): sharded_worker.py OUT_DIR FRAMES VOLUME ICP_MODE [BACKEND [EXCHANGE]]
BACKEND nccl is only possible with one rank per GPU; with a single rank the exchange path is forced, so that the
device-side collectives (RCCL on the context's own stream) run on a one-GPU box.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slambench_amd import native, synth  # noqa: E402
from slambench_amd.config import KFusionConfig  # noqa: E402
from slambench_amd.sharded import SlabKFusion  # noqa: E402


def main():
    out_dir, n, V, icp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    backend = sys.argv[5] if len(sys.argv) > 5 else "gloo"
    exchange = sys.argv[6] if len(sys.argv) > 6 else "gather"
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    cfg = KFusionConfig(compute_size_ratio=2, volume_resolution=(V, V, V))
    depth, _, _ = synth.make_sequence(n, seed=1)
    sk = SlabKFusion(cfg, rank, world, dist, torch, icp=icp, force_exchange=True, exchange=exchange)
    for i in range(n):
        sk.process_frame(depth[i])
    kf = sk.kf
    kf.sync()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank),
             trajectory=kf.trajectory(), tracked=np.array([kf.status(f)["tracked"] for f in range(n)]),
             sums=np.stack([kf.status(f)["sums"] for f in range(n)]),
             vertex=kf.download(native.BUF_MODEL_VERTEX), normal=kf.download(native.BUF_MODEL_NORMAL),
             allreduce_calls=np.array(sk.allreduce_calls))
    sk.close()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("SHARDED_OK")


if __name__ == "__main__":
    main()
