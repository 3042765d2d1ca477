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


def run_ranks(tmp_path, world, frames, volume, icp, backend="gloo", exchange="gather"):
    port = (29700 + (os.getpid() % 200) + (7 if icp == "allreduce" else 0) + (13 if backend == "nccl" else 0)
            + (29 if exchange == "tiles" else 0))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(HERE, "sharded_worker.py"),
           str(tmp_path), str(frames), str(volume), icp, backend, exchange]
    env = dict(os.environ, OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.makedirs(str(tmp_path), exist_ok=True)
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
            tn = torch.from_numpy(np.stack([kf.download(native.BUF_MODEL_TN) for kf in slabs])).cuda()
            for kf in slabs:
                native.check(L.kfh_composite_slabs(kf._ctx, C.c_void_p(tn.data_ptr()), 2))
        for kf in slabs:
            kf.end_frame()
    out = (slabs[0].trajectory(), np.array([slabs[0].status(f)["tracked"] for f in range(frames)]),
           slabs[0].download(native.BUF_MODEL_VERTEX), slabs[0].download(native.BUF_MODEL_NORMAL),
           np.stack([slabs[0].status(f)["sums"] for f in range(frames)]))
    for kf in slabs:
        kf.close()
    return out


def test_two_rank_processes_match_the_in_process_slabs(torch_cuda, tmp_path):
    frames, V = 12, 128
    ranks = run_ranks(tmp_path, 2, frames, V, "replicated")
    traj, tracked, vertex, normal, _ = in_process_slabs(torch_cuda, frames, V)
    assert tracked[4:].all(), "the sequence must track once the model exists"
    for r in ranks:
        # bit for bit: same kernels, same inputs, the all-gather moves bytes
        assert np.array_equal(r["trajectory"], traj)
        assert np.array_equal(r["tracked"], tracked)
        assert np.array_equal(r["vertex"], vertex)
        assert np.array_equal(r["normal"], normal)


@pytest.mark.parametrize("world,backend", [(2, "gloo"), (3, "gloo"), (1, "nccl")])
def test_tile_exchange_gives_the_maps_of_the_gather_exchange(torch_cuda, tmp_path, world, backend):
    """exchange="tiles" (all-to-all of the candidates, kfh_composite_tile on this rank's share of the pixels, all-gather
    of the selected shares) must leave every rank with the bits the single all-gather + kfh_composite_slabs leaves:
    two and three ranks over gloo, and one rank through the device-side RCCL collectives."""
    frames, V = 8, 126 if world == 3 else 128
    tiles = run_ranks(tmp_path / "tiles", world, frames, V, "replicated", backend=backend, exchange="tiles")
    gather = run_ranks(tmp_path / "gather", world, frames, V, "replicated", backend=backend, exchange="gather")
    for a, b in zip(tiles, gather):
        assert np.array_equal(a["trajectory"], b["trajectory"])
        assert np.array_equal(a["vertex"], b["vertex"]) and np.array_equal(a["normal"], b["normal"])
        assert a["tracked"][4:].all()
    assert (tiles[0]["normal"][..., 0] != -2).mean() > 0.5


def test_row_sharded_icp_with_allreduced_sums_matches_the_replicated_mode(torch_cuda, tmp_path):
    """icp="allreduce": every rank tracks half of the rows of each pyramid level and the 32 sums are all-reduced
    after every iteration.  Against the replicated mode: the counters of the last reduction are equal, its float
    sums agree to 1e-4 of their magnitude (nineteen iterations with another summation order move a few pixels
    across a threshold), the trajectory stays within a millimetre, and
    both ranks hold the same bits."""
    frames, V = 12, 128
    ranks = run_ranks(tmp_path, 2, frames, V, "allreduce")
    traj, tracked, vertex, normal, sums = in_process_slabs(torch_cuda, frames, V)
    a, b = ranks
    assert np.array_equal(a["trajectory"], b["trajectory"]) and np.array_equal(a["sums"], b["sums"])
    assert np.array_equal(a["vertex"], b["vertex"])
    assert np.array_equal(a["tracked"], tracked) and tracked[4:].all()
    assert np.abs(a["trajectory"][:, :3, 3] - traj[:, :3, 3]).max() < 1e-3
    # frame 4 is the first tracked frame: both modes start it from identical state, so its iterations see the same
    # pixels until the poses part by rounding
    w, h = 320, 240
    for f in range(4, frames):
        got, want = a["sums"][f], sums[f]
        # every pixel of the finest level is counted exactly once over the two shares
        assert got[28:].sum() == w * h == want[28:].sum(), (f, got[28:], want[28:])
        # after 19 iterations whose sums were rounded in another order a pixel on a threshold may change class
        assert np.abs(got[28:] - want[28:]).max() <= 8, (f, got[28:], want[28:])
        scale = np.abs(want[:28]).max()
        assert np.abs(got[:28] - want[:28]).max() <= 1e-4 * scale, f
    assert int(a["allreduce_calls"]) == 19 * frames


def test_device_side_collectives_on_a_single_rank_rccl_group(torch_cuda, tmp_path):
    """The RCCL path of the exchange (all_gather_into_tensor and all_reduce on views of the library's own buffers,
    queued against the context's stream through torch.cuda.ExternalStream, no host wait) on the one GPU of this
    box: a world of one rank with the exchange forced.  One slab is the whole volume and one share is the whole
    image, so the run must follow a plain single-context run (the sums are folded in another order)."""
    frames, V = 12, 128
    (r,) = run_ranks(tmp_path, 1, frames, V, "allreduce", backend="nccl")
    cfg = KFusionConfig(compute_size_ratio=2, volume_resolution=(V, V, V))
    depth, _, _ = synth.make_sequence(frames, seed=1)
    kf = KFusion(cfg)
    for i in range(frames):
        kf.process_frame(depth[i])
    traj = kf.trajectory()
    assert np.array_equal(r["tracked"], np.array([kf.status(f)["tracked"] for f in range(frames)]))
    assert r["tracked"][4:].all()
    assert np.abs(r["trajectory"][:, :3, 3] - traj[:, :3, 3]).max() < 1e-4
    # the maps of the last raycast: same volume up to the pose difference, same pixels hit
    hit = kf.download(native.BUF_MODEL_NORMAL)[..., 0] != -2
    assert ((r["normal"][..., 0] != -2) == hit).mean() > 0.999
    assert int(r["allreduce_calls"]) == 19 * frames
    kf.close()


def test_bench_with_two_ranks_reports_the_replica_leg_and_the_zslab_leg(torch_cuda):
    """`python bench.py --gpus 2` (the driver's multi-GPU command; two ranks on GPU 0 over gloo here): one JSON line
    with the replica leg as `value` and the sharded leg of BASELINE.json's 8-GPU configuration under zslab_<volume>
    (a small volume here), its boxes summed and its integrate time maximised over the ranks."""
    import json
    root = os.path.dirname(HERE)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "24", "--warmup", "8",
           "--no-cpu-baseline", "--no-host-frames", "--no-dense-leg", "--zslab-volume", "256", "--zslab-steps", "16",
           "--unique-frames", "16"]
    env = dict(os.environ, OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line: %r" % r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    leg = d["zslab_256"]
    assert "error" not in leg, leg
    assert leg["ranks"] == 2 and leg["volume"] == 256 and leg["scaling"] == "strong"
    assert leg["frames_per_s"] > 0 and leg["voxel_updates_per_s"] > 0
    assert leg["tracked_fraction"] == 1.0 and leg["ate_rmse_m"] < 0.01
    roof = leg["roofline"]
    assert roof["peak"] == 16000.0 and roof["achieved"] > 0 and 0 < roof["frac"] < 1
    # both ranks' boxes are in the count: more voxels than one slab of the volume's frustum share alone could give
    assert leg["integrate_voxels_per_launch_all_ranks"] > 0 and leg["integrate_ms_slowest_rank"] > 0
    assert leg["integrate_voxels_per_s_all_ranks"] > 0
    # the same configuration through the in-process group, run by rank 0 on the same device(s)
    grp = d["zslab_256_in_process"]
    assert "error" not in grp, grp
    assert grp["members"] == 2 and grp["frames_per_s"] > 0 and grp["tracked_fraction"] == 1.0
    assert abs(grp["ate_rmse_m"] - leg["ate_rmse_m"]) < 1e-3
    # a few more frames with the ICP sums all-reduced (what the configuration names; replicated tracking is the default)
    other = leg["other_icp_mode"]
    assert other["icp"] == "allreduce" and other["frames_per_s"] > 0 and other["tracked_fraction"] == 1.0
    # both exchanges were timed on a few frames and the faster one ran the leg
    probe = leg["exchange_probe_ms_per_frame"]
    assert set(probe) == {"gather", "tiles"} and all(v > 0 for v in probe.values())
    assert leg["exchange"].startswith("all-gather" if probe["gather"] <= probe["tiles"] else "all-to-all")
