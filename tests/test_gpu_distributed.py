"""GPU: the sharded sampler (2 ranks, both on cuda:0, gloo as the transport for this one-card rehearsal; the
bench uses RCCL) must reproduce the single-rank particles.  Exercises the real HIP stages with row0 > 0, the
level-0 histogram taken in the distance epilogue of a NON-symmetric block, and the host collective protocol."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, d, steps, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stein_amd.samplers import SteinSampler
        from stein_amd.optimizers import AdamGradientDescent
        rng = np.random.default_rng(7)
        T, Gs = rng.normal(size=(n, d)), rng.normal(size=(steps, n, d)) * 30.0   # |phi| > 10: clip needs the GLOBAL norm
        nl = n // world
        sl = slice(rank * nl, (rank + 1) * nl)
        s = SteinSampler(n, None, AdamGradientDescent(0.05, decay=0.9), theta=T[sl].copy(), device="cuda:0",
                         group=dist.group.WORLD)
        norms = []
        for G in Gs:
            s.update_particles(G[sl])
            norms.append(float(s.engine.sqnorm.item()))
        np.savez(os.path.join(out_dir, "r%d.npz" % rank), theta=s.samples, h2=float(s.engine.h2.item()), norms=norms)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,d", [(512, 24), (1280, 130)])
def test_two_ranks_match_one(cuda, tmp_path, n, d):
    world, steps = 2, 3
    mp.spawn(_worker, args=(world, _free_port(), n, d, steps, str(tmp_path)), nprocs=world, join=True)
    from stein_amd.samplers import SteinSampler
    from stein_amd.optimizers import AdamGradientDescent
    rng = np.random.default_rng(7)
    T, Gs = rng.normal(size=(n, d)), rng.normal(size=(steps, n, d)) * 30.0
    one = SteinSampler(n, None, AdamGradientDescent(0.05, decay=0.9), theta=T.copy(), device=cuda)
    norms = []
    for G in Gs:
        one.update_particles(G)
        norms.append(float(one.engine.sqnorm.item()))
    assert np.sqrt(norms[0]) > 10
    parts = [np.load(os.path.join(str(tmp_path), "r%d.npz" % r)) for r in range(world)]
    assert parts[0]["h2"] == parts[1]["h2"]
    np.testing.assert_allclose(parts[0]["norms"], parts[1]["norms"], rtol=0)       # same global norm on both ranks
    np.testing.assert_allclose(parts[0]["norms"], norms, rtol=1e-5)
    sharded = np.concatenate([p["theta"] for p in parts], axis=0)
    assert np.abs(sharded - one.samples).max() <= 2e-6 * np.abs(one.samples).max()
