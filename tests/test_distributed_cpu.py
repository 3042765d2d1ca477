"""World-size-2 (and 3) rehearsal of the multi-GPU host logic on CPU with the gloo backend."""
import os
import subprocess
import sys

import pytest

from slambench_amd.sharded import slab_bounds

HERE = os.path.dirname(os.path.abspath(__file__))


def test_slab_bounds():
    assert slab_bounds(512, 1, 0) == (0, 512)
    assert [slab_bounds(2048, 8, r) for r in range(8)] == [(256 * r, 256 * (r + 1)) for r in range(8)]
    assert [slab_bounds(10, 3, r) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    with pytest.raises(ValueError):
        slab_bounds(4, 8, 0)
    with pytest.raises(ValueError):
        slab_bounds(8, 2, 2)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_exchange_with_gloo(world):
    port = 29500 + world + (os.getpid() % 200)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(HERE, "dist_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "DIST_OK" in r.stdout
