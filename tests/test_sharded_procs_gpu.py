"""The z-slab path as shipped: one process per rank (torch.distributed.run, gloo, both ranks on GPU 0) running
SlabKFusion.process_frame, i.e. kfh_raycast_slab -> all-gather of the per-slab ray segments -> kfh_composite_slabs
inside every frame, against the in-process two-context construction of tests/test_sharded_gpu.py (the same slab
algorithm with the all-gather replaced by a local stack)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from slambench_amd import native, synth
from slambench_amd.config import KFusionConfig
from slambench_amd.kfusion import KFusion
from slambench_amd.sharded import slab_bounds

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def run_ranks(tmp_path, world, frames, volume, icp):
    port = 29700 + (os.getpid() % 200) + (7 if icp == "allreduce" else 0)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(HERE, "sharded_worker.py"),
           str(tmp_path), str(frames), str(volume), icp]
    env = dict(os.environ, OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "SHARDED_OK" in r.stdout
    return [np.load(os.path.join(str(tmp_path), "rank%d.npz" % k)) for k in range(world)]


def in_process_slabs(torch, frames, V):
    """Two slab contexts in this process, the exchange done by hand (as tests/test_sharded_gpu.py does)."""
    base = dict(compute_size_ratio=2, volume_resolution=(V, V, V))
    slabs = [KFusion(KFusionConfig(**base, slab=slab_bounds(V, 2, r))) for r in range(2)]
    depth, _, _ = synth.make_sequence(frames, seed=1)
    L = native.lib()
    for i in range(frames):
        for kf in slabs:
            kf.set_depth(depth[i])
            kf.preprocessing()
            kf.tracking()
            kf.integration()
            native.check(L.kfh_raycast_slab(kf._ctx))
        if i > 2:
            t = torch.from_numpy(np.stack([kf.download(native.BUF_RAY_T) for kf in slabs])).cuda()
            v = torch.from_numpy(np.stack([kf.download(native.BUF_MODEL_VERTEX) for kf in slabs])).cuda()
            nm = torch.from_numpy(np.stack([kf.download(native.BUF_MODEL_NORMAL) for kf in slabs])).cuda()
            for kf in slabs:
                native.check(L.kfh_composite_slabs(kf._ctx, C.c_void_p(t.data_ptr()), C.c_void_p(v.data_ptr()),
                                                   C.c_void_p(nm.data_ptr()), 2))
        for kf in slabs:
            kf.end_frame()
    out = (slabs[0].trajectory(), np.array([slabs[0].status(f)["tracked"] for f in range(frames)]),
           slabs[0].download(native.BUF_MODEL_VERTEX), slabs[0].download(native.BUF_MODEL_NORMAL))
    for kf in slabs:
        kf.close()
    return out


def test_two_rank_processes_match_the_in_process_slabs(torch_cuda, tmp_path):
    frames, V = 12, 128
    ranks = run_ranks(tmp_path, 2, frames, V, "replicated")
    traj, tracked, vertex, normal = in_process_slabs(torch_cuda, frames, V)
    assert tracked[4:].all(), "the sequence must track once the model exists"
    for r in ranks:
        # bit for bit: same kernels, same inputs, the all-gather moves bytes
        assert np.array_equal(r["trajectory"], traj)
        assert np.array_equal(r["tracked"], tracked)
        assert np.array_equal(r["vertex"], vertex)
        assert np.array_equal(r["normal"], normal)
