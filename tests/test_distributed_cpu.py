"""CPU, gloo, world_size 2 (and once 8): the multi-rank protocol of stein_amd.engine.SvgdEngine.

The HIP stages cannot run without a GPU, so the engine is given the NumPy stage model
(oracle/staged_model.py, test infrastructure).  What is under test is the PRODUCT host logic: row
ownership, the all-gather of theta/score rows, the per-level histogram all-reduce that makes every rank
resolve the same median, the |phi|^2 all-reduce, and that the sharded result equals the single-rank one.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, d, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stein_amd import _lib
        from stein_amd.engine import SvgdEngine
        from oracle.staged_model import NumpyStages
        rng = np.random.default_rng(42)
        T, G = rng.normal(size=(n, d)), rng.normal(size=(n, d))
        eng = SvgdEngine(n, d, device="cpu", group=dist.group.WORLD, stages=NumpyStages(_lib.workspace_layout))
        assert eng.n_local == n // world and eng.row0 == rank * eng.n_local
        sl = slice(eng.row0, eng.row0 + eng.n_local)
        th = torch.tensor(T[sl], dtype=torch.float32).contiguous()
        sc = torch.tensor(G[sl], dtype=torch.float32).contiguous()
        phi = eng.compute_phi(th, sc).clone()
        # second call reuses every buffer (histograms must be re-zeroed)
        phi2 = eng.compute_phi(th, sc).clone()
        assert torch.equal(phi, phi2)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), phi=phi.numpy(), h2=eng.h2.numpy(),
                 median=eng.median.numpy(), sqnorm=eng.sqnorm.numpy(), row0=eng.row0,
                 gathered_ok=np.array(np.allclose(eng.T_all.numpy(), T.astype(np.float32))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,d", [(2, 64, 6), (2, 130, 9), (8, 1024, 5)])
def test_sharded_protocol_matches_single_rank(tmp_path, world, n, d):
    mp.spawn(_worker, args=(world, _free_port(), n, d, str(tmp_path)), nprocs=world, join=True)
    from oracle import svgd_oracle as orc
    rng = np.random.default_rng(42)
    T, G = rng.normal(size=(n, d)), rng.normal(size=(n, d))
    T32, G32 = T.astype(np.float32).astype(np.float64), G.astype(np.float32).astype(np.float64)
    ref = orc.svgd_step(T32, G32, orc.AdagradState(), np.float32)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    assert all(bool(p["gathered_ok"]) for p in parts)
    # identical bandwidth / median / global norm on every rank
    assert all(p["h2"][0] == ref["h2"] and p["median"][0] == ref["median"] for p in parts)
    assert all(p["sqnorm"][0] == parts[0]["sqnorm"][0] for p in parts)
    phi = np.concatenate([p["phi"] for p in parts], axis=0)
    np.testing.assert_allclose(phi, ref["phi"], rtol=0, atol=2e-6 * np.abs(ref["phi"]).max())
    np.testing.assert_allclose(parts[0]["sqnorm"][0], ref["sqnorm"], rtol=1e-5)


def _worker_window(rank, world, port, n, d, steps, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stein_amd import _lib
        from stein_amd.engine import SvgdEngine
        from oracle.staged_model import NumpyStages
        rng = np.random.default_rng(5)
        T, G, V = rng.normal(size=(n, d)), rng.normal(size=(n, d)), 1e-3 * rng.normal(size=(n, d))
        eng = SvgdEngine(n, d, device="cpu", group=dist.group.WORLD, stages=NumpyStages(_lib.workspace_layout), dist_window=True)
        assert eng.dist_window
        sl = slice(eng.row0, eng.row0 + eng.n_local)
        sc = torch.tensor(G[sl], dtype=torch.float32).contiguous()
        h2s, hits = [], []
        for step in range(steps):
            if step == steps - 2:
                T = T * 2.0                                   # a jump: the window misses, the radix protocol takes over
            th = torch.tensor((T + step * V)[sl], dtype=torch.float32).contiguous()
            eng.compute_phi(th, sc)
            h2s.append(float(eng.h2[0]))
            hits.append(int(eng.window_hit))
        np.savez(os.path.join(out_dir, "win%d.npz" % rank), h2=h2s, hits=hits)
    finally:
        dist.destroy_process_group()


def test_two_rank_window_protocol(tmp_path):
    """The engine's window protocol (tally, table all-reduce, pick, hit read-back, fall-back to the radix
    protocol) with the NumPy stage model: the bandwidth of every step equals the oracle's, on both ranks."""
    world, n, d, steps = 2, 96, 5, 8
    mp.spawn(_worker_window, args=(world, _free_port(), n, d, steps, str(tmp_path)), nprocs=world, join=True)
    from oracle import svgd_oracle as orc
    rng = np.random.default_rng(5)
    T, G, V = rng.normal(size=(n, d)), rng.normal(size=(n, d)), 1e-3 * rng.normal(size=(n, d))
    parts = [np.load(os.path.join(str(tmp_path), "win%d.npz" % r)) for r in range(world)]
    for step in range(steps):
        if step == steps - 2:
            T = T * 2.0
        T32 = (T + step * V).astype(np.float32).astype(np.float64)
        ref = orc.svgd_step(T32, G.astype(np.float32).astype(np.float64), orc.AdagradState(), np.float32)
        assert parts[0]["h2"][step] == parts[1]["h2"][step] == ref["h2"], step
    hits = list(parts[0]["hits"])
    assert hits == list(parts[1]["hits"])
    assert hits[0] == 0 and sum(hits[2:steps - 2]) >= 2 and hits[steps - 2] == 0, hits


def test_eight_rank_window_protocol_at_c5_shard_geometry(tmp_path):
    """World size 8, n_local = n / 8 with 128-row-aligned shards (BASELINE config 5's geometry, 131072 / 8 = 16384 rows
    per rank, scaled down 128x so that NumPy can stand in for the kernels), window form: the exact 8-way protocol --
    all-gathers of 8 row blocks, the table all-reduce of the window tally over 8 ranks, the radix fall-back after a jump,
    the |phi|^2 all-reduce -- gives the oracle's bandwidth at every step on every rank."""
    world, n, d, steps = 8, 1024, 6, 6
    mp.spawn(_worker_window, args=(world, _free_port(), n, d, steps, str(tmp_path)), nprocs=world, join=True)
    from oracle import svgd_oracle as orc
    rng = np.random.default_rng(5)
    T, G, V = rng.normal(size=(n, d)), rng.normal(size=(n, d)), 1e-3 * rng.normal(size=(n, d))
    parts = [np.load(os.path.join(str(tmp_path), "win%d.npz" % r)) for r in range(world)]
    for step in range(steps):
        if step == steps - 2:
            T = T * 2.0
        T32 = (T + step * V).astype(np.float32).astype(np.float64)
        ref = orc.svgd_step(T32, G.astype(np.float32).astype(np.float64), orc.AdagradState(), np.float32)
        assert all(p["h2"][step] == ref["h2"] for p in parts), step
    hits = list(parts[0]["hits"])
    assert all(list(p["hits"]) == hits for p in parts)
    assert hits[0] == 0 and sum(hits[2:steps - 2]) >= 1 and hits[steps - 2] == 0, hits


def test_uneven_sharding_is_refused():
    from stein_amd import _lib
    from stein_amd.engine import SvgdEngine
    from oracle.staged_model import NumpyStages

    class FakeGroup:
        pass
    import torch.distributed as d2
    orig = (d2.get_world_size, d2.get_rank)
    d2.get_world_size, d2.get_rank = (lambda g=None: 3), (lambda g=None: 0)
    try:
        with pytest.raises(ValueError, match="divisible"):
            SvgdEngine(64, 4, device="cpu", group=FakeGroup(), stages=NumpyStages(_lib.workspace_layout))
    finally:
        d2.get_world_size, d2.get_rank = orig


def _worker_seed(rank, world, port, n, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stein_amd.samplers import SteinSampler
        from stein_amd.optimizers import AdamGradientDescent
        s = SteinSampler(n, None, AdamGradientDescent(0.1), theta=None, model_vars={"model/w:0": [3, 1], "model/b:0": []},
                         device="cpu", group=dist.group.WORLD, seed=11, dtype=torch.float64)
        np.save(os.path.join(out_dir, "seed%d.npy" % rank), s.theta_matrix.numpy())
    finally:
        dist.destroy_process_group()


def test_seeded_initial_particles_are_distinct_across_ranks_and_match_one_rank(tmp_path):
    """theta=None with a seed: every rank keeps ITS rows of the one seeded [n, d] draw (identical particles on two
    ranks would stay identical forever -- SVGD is deterministic -- and halve the effective particle count)."""
    world, n = 2, 12
    mp.spawn(_worker_seed, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(os.path.join(str(tmp_path), "seed%d.npy" % r)) for r in range(world)]
    both = np.concatenate(parts, axis=0)
    assert both.shape == (n, 4)
    assert len({tuple(row) for row in both}) == n                       # no particle appears twice
    from stein_amd.samplers import SteinSampler
    from stein_amd.optimizers import AdamGradientDescent
    one = SteinSampler(n, None, AdamGradientDescent(0.1), theta=None, model_vars={"model/w:0": [3, 1], "model/b:0": []},
                       device="cpu", seed=11, dtype=torch.float64)
    assert np.array_equal(both, one.theta_matrix.numpy())                # the sharded draw IS the single-rank draw
    assert abs(both.std() - 0.01) < 0.004                                # N(0, 0.01^2), abstract_stein_sampler.py:69-74


def _worker_comm_arg(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stein_amd import _lib
        from stein_amd.engine import SvgdEngine
        from oracle.staged_model import NumpyStages
        st = NumpyStages(_lib.workspace_layout)
        eng = SvgdEngine(64, 4, device="cpu", group=dist.group.WORLD, stages=st)          # auto -> torch on gloo / CPU
        msg = ""
        try:
            SvgdEngine(64, 4, device="cpu", group=dist.group.WORLD, stages=st, comm="native")
        except ValueError as e:
            msg = str(e)
        bad = ""
        try:
            SvgdEngine(64, 4, device="cpu", group=dist.group.WORLD, stages=st, comm="mpi")
        except ValueError as e:
            bad = str(e)
        eng.close(); eng.close()                                                            # idempotent, no communicator
        np.save(os.path.join(out_dir, "comm%d.npy" % rank), np.array([eng.comm, msg, bad]))
    finally:
        dist.destroy_process_group()


def test_comm_argument_is_validated(tmp_path):
    """comm='auto' falls back to torch.distributed collectives off the GPU / off RCCL; comm='native' there is an error
    (the library's communicator needs HIP tensors, the HIP stages and an nccl group), as is an unknown value."""
    mp.spawn(_worker_comm_arg, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        comm, msg, bad = np.load(os.path.join(str(tmp_path), "comm%d.npy" % r))
        assert comm == "torch"
        assert "comm='native' needs" in msg
        assert "comm must be" in bad


def _posterior_func(theta, feed):
    """batched over particles: per particle the vector of sigmoid(x . w + b) over the feed's points (the logistic example's
    `evaluate`-style readout, examples/logistic_regression/main.py:52-61)"""
    w, b = theta["model/w:0"], theta["model/b:0"]                 # [m, 3, 1], [m]
    X = torch.as_tensor(feed["X"], dtype=w.dtype)
    return torch.sigmoid(torch.einsum("pf,mfo->mp", X, w) + b[:, None])


def _worker_posterior(rank, world, port, n, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stein_amd.samplers import SteinSampler
        from stein_amd.optimizers import AdamGradientDescent
        s = SteinSampler(n, None, AdamGradientDescent(0.1), theta=None, model_vars={"model/w:0": [3, 1], "model/b:0": []},
                         device="cpu", group=dist.group.WORLD, seed=11, dtype=torch.float64)
        X = np.random.default_rng(3).normal(size=(5, 3))
        full = s.function_posterior(_posterior_func, {"X": X})
        mean0 = s.function_posterior(_posterior_func, {"X": X}, axis=0)
        local = s.function_posterior(_posterior_func, {"X": X}, gather=False)
        np.savez(os.path.join(out_dir, "post%d.npz" % rank), full=full, mean0=mean0, local=local, samples=s.samples_all())
    finally:
        dist.destroy_process_group()


def test_function_posterior_gathers_all_particles_on_a_sharded_sampler(tmp_path):
    """abstract_stein_sampler.py:157-168 returns one row per particle for ALL n particles (and the mean over them with
    `axis`); a sharded sampler evaluates its own rows and all-gathers them in rank order.  Checked against the reference's
    per-particle loop (NumPy, one particle at a time, np.ravel of each output) over the single-rank draw of the same seed."""
    world, n = 2, 12
    mp.spawn(_worker_posterior, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    from stein_amd.samplers import SteinSampler
    from stein_amd.optimizers import AdamGradientDescent
    one = SteinSampler(n, None, AdamGradientDescent(0.1), theta=None, model_vars={"model/w:0": [3, 1], "model/b:0": []},
                       device="cpu", seed=11, dtype=torch.float64)
    X = np.random.default_rng(3).normal(size=(5, 3))
    th = {v: t.numpy() for v, t in one.theta.items()}
    rows = []
    for i in range(n):                                            # the reference's loop, :160-162
        w, b = th["model/w:0"][i], th["model/b:0"][i]
        rows.append(np.ravel(1.0 / (1.0 + np.exp(-(X @ w + b)))))
    want = np.array(rows)
    assert want.shape == (n, 5)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "post%d.npz" % r))
        assert got["full"].shape == (n, 5) and np.allclose(got["full"], want, rtol=1e-12, atol=0)
        assert np.allclose(got["mean0"], want.mean(axis=0), rtol=1e-12, atol=0)
        assert np.allclose(got["local"], want[r * (n // world):(r + 1) * (n // world)], rtol=1e-12, atol=0)
        assert np.array_equal(got["samples"], one.theta_matrix.numpy())
