"""Z-slab sharding on one GPU: two slab contexts in one process against the single full-volume context.

The exchange step (all-gather of the per-slab ray segments) is replaced by a local stack, so this exercises
kfh_raycast_slab / kfh_composite_slabs and the redundant-halo integration through the C ABI; the collective
itself is covered by the gloo test in tests/test_distributed_cpu.py.
"""
import ctypes as C

import numpy as np
import pytest

from slambench_amd import native, synth
from slambench_amd.config import KFusionConfig
from slambench_amd.kfusion import KFusion
from slambench_amd.sharded import composite_numpy, slab_bounds

pytestmark = pytest.mark.gpu


def test_two_slabs_reproduce_the_full_volume(torch_cuda):
    torch = torch_cuda
    V, n = 128, 16
    base = dict(compute_size_ratio=2, volume_resolution=(V, V, V))
    full = KFusion(KFusionConfig(**base))
    slabs = [KFusion(KFusionConfig(**base, slab=slab_bounds(V, 2, r))) for r in range(2)]
    depth, gt, ts = synth.make_sequence(n, seed=1)
    L = native.lib()
    w, h = full.width, full.height
    for i in range(n):
        full.process_frame(depth[i])
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
            if i == 3:
                # device composite == host restatement of the min-t rule
                want = composite_numpy(tn.cpu().numpy())
                assert np.array_equal(slabs[0].download(native.BUF_MODEL_TN), want)
                assert np.array_equal(slabs[0].download(native.BUF_RAY_T), want[..., 0])
                got_n = slabs[1].download(native.BUF_MODEL_NORMAL)
                assert np.array_equal(got_n[..., :3], want[..., 1:])
                nc = got_n
                vc = slabs[0].download(native.BUF_MODEL_VERTEX)
                # the vertex map is rebuilt from the hit distance: zero where nothing was hit, and elsewhere
                # origin + direction * t of the view the raycast used (pose * inverse camera matrix), restated here
                # in float32 operation by operation
                assert np.array_equal((vc[..., :3] != 0).any(-1) | (want[..., 0] == 0), np.ones(want.shape[:2], bool))
                f32 = np.float32
                pose = slabs[0].pose.astype(f32)            # the pose frame 3 was raycast from
                fx, fy, cx, cy = [f32(v) / f32(slabs[0].cfg.compute_size_ratio) for v in slabs[0].cfg.intrinsics]
                kinv = np.array([[f32(1) / fx, 0, -cx / fx, 0], [0, f32(1) / fy, -cy / fy, 0], [0, 0, 1, 0], [0, 0, 0, 1]], f32)
                view = np.zeros((4, 4), f32)
                for r in range(4):
                    for col in range(4):
                        acc = pose[r, 0] * kinv[0, col]
                        for k in range(1, 4):
                            acc = f32(acc + f32(pose[r, k] * kinv[k, col]))
                        view[r, col] = acc
                ys, xs = np.mgrid[0:h, 0:w].astype(f32)
                t = want[..., 0]
                want_v = np.zeros((h, w, 3), f32)
                for r in range(3):
                    direction = (view[r, 0] * xs + view[r, 1] * ys) + view[r, 2] * f32(1)
                    want_v[..., r] = view[r, 3] + direction * t
                hit = t > 0
                assert hit.sum() > 1000
                assert np.array_equal(vc[..., :3][hit].view(np.uint32), want_v[hit].view(np.uint32))
                # first raycast: same volume, same pose -> the composited map is the single-device map except
                # where a ray's step state restarts at the slab boundary
                v_full = full.download(native.BUF_MODEL_VERTEX)[..., :3]
                hit_full = full.download(native.BUF_MODEL_NORMAL)[..., 0] != -2
                hit_sh = nc[..., 0] != -2
                assert (hit_full == hit_sh).mean() > 0.995
                both = hit_full & hit_sh
                d = np.linalg.norm(v_full[both] - vc[..., :3][both], axis=1)
                assert (d == 0).mean() > 0.85          # most rays never change slab before they hit
                assert d.max() < 2 * 8.0 / V           # the rest moves by less than two voxels
        for kf in slabs:
            kf.end_frame()
        if i == 3:
            # pose still the initial one: the owned planes of each slab are bit-identical to the full volume
            vol = full.download(native.BUF_VOLUME)
            for r, kf in enumerate(slabs):
                z0, z1 = slab_bounds(V, 2, r)
                zs0 = max(z0 - 2, 0)
                part = kf.download(native.BUF_VOLUME)
                assert part.shape[0] == min(z1 + 2, V) - zs0
                assert np.array_equal(part[z0 - zs0:z1 - zs0], vol[z0:z1])
    # both slab ranks hold the same trajectory, within 1 mm of the single-device run
    tr_full = full.trajectory()
    tr = [kf.trajectory() for kf in slabs]
    assert np.array_equal(tr[0], tr[1])
    assert np.abs(tr[0][:, :3, 3] - tr_full[:, :3, 3]).max() < 1e-3
    assert all(full.status(f)["tracked"] == slabs[0].status(f)["tracked"] for f in range(n))
    for kf in slabs + [full]:
        kf.close()


def _run_slabs(torch, V, n, world):
    base = dict(compute_size_ratio=2, volume_resolution=(V, V, V))
    slabs = [KFusion(KFusionConfig(**base, slab=slab_bounds(V, world, r))) for r in range(world)]
    depth, _, _ = synth.make_sequence(n, seed=1)
    L = native.lib()
    maps = []
    for i in range(n):
        for kf in slabs:
            kf.set_depth(depth[i])
            kf.preprocessing()
            kf.tracking()
            kf.integration()
            native.check(L.kfh_raycast_slab(kf._ctx))
        if i > 2:
            parts = np.stack([kf.download(native.BUF_MODEL_TN) for kf in slabs])
            maps.append(parts)
            tn = torch.from_numpy(parts).cuda()
            for kf in slabs:
                native.check(L.kfh_composite_slabs(kf._ctx, C.c_void_p(tn.data_ptr()), world))
        for kf in slabs:
            kf.end_frame()
    return slabs, maps


def test_slab_raycast_with_brick_marks_equals_the_one_without(torch_cuda, monkeypatch):
    """A slab context keeps brick marks for the bricks it can vouch for (all their planes and the plane above are
    stored) and none for those that reach outside its planes; the slab march skips marked bricks.  Its per-slab ray
    segments must be the bits of a march that fetches every sample's voxels."""
    V, n, world = 128, 10, 3          # slabs of 43 / 43 / 42 planes: boundaries that are no multiples of the brick size
    slabs, maps = _run_slabs(torch_cuda, V, n, world)
    marks = [kf.download(native.BUF_BRICK_MARKS) for kf in slabs]
    traj = [kf.trajectory() for kf in slabs]
    for kf in slabs:
        kf.close()
    cleared_somewhere = False
    for r, m in enumerate(marks):
        z0, z1 = slab_bounds(V, world, r)
        zs0, zs1 = max(z0 - 2, 0), min(z1 + 2, V)
        bz = np.arange(V // 8)
        inside = ((8 * bz >= zs0) | (zs0 == 0)) & ((8 * bz + 8 <= zs1 - 1) | (zs1 == V))
        assert not m[~inside].any(), "bricks that reach outside the stored planes must never be marked"
        assert m[inside].mean() > 0.3, "most bricks of a room are free space"
        cleared_somewhere = cleared_somewhere or not m[inside].all()
    assert cleared_somewhere, "the surface must have cleared marks in the slab that holds it"
    monkeypatch.setenv("KFH_NO_BRICK_MARKS", "1")
    plain, maps_plain = _run_slabs(torch_cuda, V, n, world)
    traj_plain = [kf.trajectory() for kf in plain]
    for kf in plain:
        kf.close()
    assert len(maps) == len(maps_plain) == n - 3
    for a, b in zip(maps, maps_plain):
        assert np.array_equal(a, b)
    assert all(np.array_equal(a, b) for a, b in zip(traj, traj_plain))
    assert (maps[-1][..., 0] > 0).any(axis=0).mean() > 0.5          # the composite hits most pixels


@pytest.mark.parametrize("members,threads,exchange", [(2, "0", "gather"), (3, "0", "gather"), (3, "1", "gather"),
                                                      (5, "1", "share"), (2, "0", "share"), (3, "1", "share"),
                                                      (4, "0", "")])
def test_group_inside_one_process_gives_the_bits_of_the_gather_exchange(torch_cuda, monkeypatch, members, threads,
                                                                        exchange):
    """kfh_group (the members' composites reading each other's segment maps in place, ordered by events - every member
    the whole image, or each its share of the pixels stored into every member's map; launches queued by the calling
    thread or, as for members on several devices, by one host thread per member)
    against the construction above: every slab marched by kfh_raycast_slab, the segment maps stacked and every context
    compositing with kfh_composite_slabs - what the one-process-per-rank path does with an all-gather in between."""
    from slambench_amd.kfusion import GroupKFusion
    torch = torch_cuda
    V, frames = 96, 14                    # 96 planes: the slab boundaries of three members are off the brick grid
    base = dict(compute_size_ratio=2, volume_resolution=(V, V, V))
    depth, _, _ = synth.make_sequence(frames, seed=1)
    L = native.lib()
    slabs = [KFusion(KFusionConfig(**base, slab=slab_bounds(V, members, r))) for r in range(members)]
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
                native.check(L.kfh_composite_slabs(kf._ctx, C.c_void_p(tn.data_ptr()), members))
        for kf in slabs:
            kf.end_frame()
    monkeypatch.setenv("KFH_GROUP_THREADS", threads)           # both read when the group is created
    if exchange:                                               # "": the default (the image split from four members on)
        monkeypatch.setenv("KFH_GROUP_EXCHANGE", exchange)
    else:
        monkeypatch.delenv("KFH_GROUP_EXCHANGE", raising=False)
    group = GroupKFusion(KFusionConfig(**base), [0] * members)
    for i in range(frames):
        group.process_frame(depth[i])
    pose = group.pose
    assert np.array_equal(pose, slabs[0].pose)
    tracked = [slabs[0].status(f)["tracked"] for f in range(frames)]
    assert all(tracked[4:]), "the sequence must track once the model exists"
    for r in range(members):
        m = group.members[r]
        assert np.array_equal(m.trajectory(), slabs[r].trajectory())
        assert [m.status(f)["tracked"] for f in range(frames)] == tracked
        assert np.array_equal(m.download(native.BUF_MODEL_TN), slabs[r].download(native.BUF_MODEL_TN))
        assert np.array_equal(m.download(native.BUF_VOLUME), slabs[r].download(native.BUF_VOLUME))
    # a reset group runs the same sequence to the same bits
    group.reset()
    for i in range(frames):
        group.process_frame(depth[i])
    assert np.array_equal(group.pose, pose)
    group.close()
    for kf in slabs:
        kf.close()


def test_group_rejects_what_it_cannot_run(torch_cuda):
    from slambench_amd.kfusion import GroupKFusion
    cfg = KFusionConfig(compute_size_ratio=4, volume_resolution=(32, 32, 32))
    with pytest.raises(native.KfhError):
        GroupKFusion(cfg, [0] * 17)           # at most 16 members
    with pytest.raises(native.KfhError):
        GroupKFusion(cfg, [0, 99])            # no such device
    with pytest.raises(native.KfhError):
        GroupKFusion(cfg, [])
