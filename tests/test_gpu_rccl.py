"""GPU: the multi-rank protocol of SvgdEngine driven through RCCL on ONE card -- by the library's own communicator
(comm="native": stein_comm_init + stein_rank_step, the whole step one C call) and by torch.distributed's "nccl" backend
between the rank segments (comm="torch").

An 8-GPU node is not available to the builder, and every other multi-rank test uses gloo as the transport.  A
one-rank "nccl" process group still sends every collective of the step through RCCL's API on the device's stream:

    all_gather_into_tensor(theta)                       synchronous, then the stream continues
    all_gather_into_tensor(score, async_op=True)        + Work.wait() before the contraction
    all_reduce(int64 histogram [2][2048]) x 3           radix-select form of the median
    all_reduce(int64 window table, 65544 words)         speculative-window form of the median
    all_reduce(fp64 |phi|^2 scalar)

`force_collectives=True` makes the engine run that protocol (row-block kernels, row0 / n_local arguments, staged calls)
although the group has a single rank.  The results must equal the single-rank staged path: bit for bit on the
fp32-MFMA kernels (whose symmetric and row-block distance passes agree exactly), to rounding on the split path (a row
block forms the lo*hi / hi*lo products of an entry below the diagonal in the other order).
The 1 -> 8 GPU scaling curve itself remains unmeasured by this test.
"""
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


def _worker(rank, port, n, d, steps, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        from stein_amd.engine import SvgdEngine
        from stein_amd.optimizers import AdagradGradientDescent
        rng = np.random.default_rng(21)
        T0, G0 = rng.normal(size=(n, d)), rng.normal(size=(n, d))
        out = {}
        import itertools
        for comm, x3 in itertools.product(("torch", "native"), (False, True)):
            for form in ("radix", "window"):
                theta = torch.tensor(T0, dtype=torch.float32, device=dev)
                score = torch.tensor(G0, dtype=torch.float32, device=dev)
                eng = SvgdEngine(n, d, device=dev, group=dist.group.WORLD, force_collectives=True, x3=x3, small=False,
                                 comm=comm, dist_window=(form == "window"))
                assert eng.sharded and eng.dist_window == (form == "window") and eng.comm == comm
                gd = AdagradGradientDescent(learning_rate=1e-3)
                ref = SvgdEngine(n, d, device=dev, x3=x3, small=False)       # single-rank staged path, same inputs
                theta_r, gd_r = theta.clone(), AdagradGradientDescent(learning_rate=1e-3)
                h2s, hits, errs = [], [], []
                for _ in range(steps):
                    phi = eng.compute_phi(theta, score)
                    phi_r = ref.compute_phi(theta_r, score, mark=lambda label: None)
                    torch.cuda.synchronize()
                    h2s.append((float(eng.h2.item()), float(ref.h2.item())))
                    hits.append(-1 if eng.window_hit is None else int(eng.window_hit))
                    errs.append(float((phi - phi_r).norm() / phi_r.norm()))
                    assert abs(float(eng.sqnorm.item()) - float(ref.sqnorm.item())) <= 1e-6 * float(ref.sqnorm.item())
                    gd.apply_(theta, phi, eng.sqnorm)
                    gd_r.apply_(theta_r, phi_r, ref.sqnorm)
                out["%s_%d_%s" % (form, int(x3), comm)] = dict(
                    h2=h2s, hits=hits, errs=errs, theta_equal=bool(torch.equal(theta, theta_r)),
                    theta_err=float((theta - theta_r).abs().max() / theta_r.abs().max()))
                eng.close()
        # bf16 particles (BASELINE config 2's input format) and the optional dK output through stein_rank_step
        tb = torch.tensor(T0, dtype=torch.float32, device=dev).bfloat16()
        gb = torch.tensor(G0, dtype=torch.float32, device=dev).bfloat16()
        eng = SvgdEngine(n, d, device=dev, group=dist.group.WORLD, force_collectives=True, dtype=torch.bfloat16, small=False,
                         comm="native", dist_window=True)
        ref = SvgdEngine(n, d, device=dev, dtype=torch.bfloat16, small=False)
        dK, dK_r = torch.empty(n, d, device=dev), torch.empty(n, d, device=dev)
        errs = []
        for _ in range(3):
            phi = eng.compute_phi(tb, gb, dK_out=dK)
            phi_r = ref.compute_phi(tb, gb, dK_out=dK_r, mark=lambda label: None)
            torch.cuda.synchronize()
            errs.append((float((phi - phi_r).norm() / phi_r.norm()), float((dK - dK_r).norm() / dK_r.norm()),
                         float(eng.h2.item()), float(ref.h2.item())))
        out["bf16"] = errs
        eng.close()
        # "auto" keeps torch.distributed's collectives (the library's communicator is opt-in until it has run on >= 2 GPUs)
        eng = SvgdEngine(n, d, device=dev, group=dist.group.WORLD, force_collectives=True)
        out["auto"] = eng.comm
        eng.close()
        np.save(os.path.join(out_dir, "rccl.npy"), out, allow_pickle=True)
    finally:
        dist.destroy_process_group()


def test_protocol_through_rccl_on_one_rank(cuda, tmp_path):
    n, d, steps = 1280, 130, 6
    mp.spawn(_worker, args=(_free_port(), n, d, steps, str(tmp_path)), nprocs=1, join=True)
    out = np.load(os.path.join(str(tmp_path), "rccl.npy"), allow_pickle=True).item()
    assert out["auto"] == "torch"
    for e_phi, e_dk, h2a, h2b in out["bf16"]:          # bf16: K is rounded to bf16 in both runs; row block vs symmetric order
        assert e_phi <= 4e-3 and e_dk <= 4e-3 and abs(h2a - h2b) <= 1e-6 * h2b, out["bf16"]
    for comm in ("torch", "native"):
        for form in ("radix", "window"):
            exact = out["%s_0_%s" % (form, comm)]          # fp32-MFMA kernels: the sharded protocol is bit-identical
            assert all(a == b for a, b in exact["h2"]), exact["h2"]
            assert all(e == 0.0 for e in exact["errs"]) and exact["theta_equal"], exact
            split = out["%s_1_%s" % (form, comm)]          # split path: to rounding
            assert all(abs(a - b) <= 2e-6 * b for a, b in split["h2"]), split["h2"]
            assert max(split["errs"]) <= 5e-6 and split["theta_err"] <= 5e-6, split
        for x3 in (0, 1):
            hits = out["window_%d_%s" % (x3, comm)]["hits"]
            assert hits[0] == 0 and sum(hits[2:]) >= 3, hits      # the window needs two medians of history, then hits
            assert all(h == -1 for h in out["radix_%d_%s" % (x3, comm)]["hits"])
    # the two ways of issuing the collectives run the same kernels on the same data: identical trajectories
    for form in ("radix", "window"):
        for x3 in (0, 1):
            a, b = out["%s_%d_torch" % (form, x3)], out["%s_%d_native" % (form, x3)]
            assert a["h2"] == b["h2"] and a["errs"] == b["errs"] and a["hits"] == b["hits"], (form, x3)
