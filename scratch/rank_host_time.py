"""Host issue time of one rank's step through the stein_rank_* segments (one-rank RCCL group, every collective issued)
against the staged calls and the fused single-rank call.  One GPU.  For an honest issue-time figure the host must not be
blocked by the stream: small shapes where the GPU finishes a step faster than the host issues it are reported as wall."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
import torch.distributed as dist
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from stein_amd.engine import SvgdEngine
for n, d in ((2048, 256), (16384, 256)):
    T = torch.randn(n, d, device=dev); G = torch.randn(n, d, device=dev)
    for name, env, kw, ekw in (("fused (single rank)", "0", {}, {}),
                               ("stein_rank_step (library RCCL), radix", "0", {}, dict(group=dist.group.WORLD, force_collectives=True, comm="native")),
                               ("stein_rank_step (library RCCL), window", "1", {}, dict(group=dist.group.WORLD, force_collectives=True, comm="native")),
                               ("segments + torch collectives, radix", "0", {}, dict(group=dist.group.WORLD, force_collectives=True, comm="torch")),
                               ("segments + torch collectives, window", "1", {}, dict(group=dist.group.WORLD, force_collectives=True, comm="torch")),
                               ("staged calls, radix form (round 1)", "0", dict(mark=lambda s: None), dict(group=dist.group.WORLD, force_collectives=True, comm="torch"))):
        eng = SvgdEngine(n, d, device=dev, small=False, dist_window=(env == "1") if ekw else None, **ekw)
        for _ in range(8): eng.compute_phi(T, G, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): eng.compute_phi(T, G, **kw)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        if hasattr(eng, "close"): eng.close()
        print("n=%d d=%d %-40s host issue %.1f us/step, wall %.1f us/step" % (n, d, name, (t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6), flush=True)
dist.destroy_process_group()
