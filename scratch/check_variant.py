"""Correctness of a diagnostic / experiment build against a torch fp64 evaluation on the device.
usage: python scratch/check_variant.py lib_<name>.so   ('' = the shipped library)"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
if len(sys.argv) > 1 and sys.argv[1]:
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), sys.argv[1])
from stein_amd.engine import SvgdEngine
def ref(T, G, h2):
    T, G = T.double(), G.double(); n = T.shape[0]
    r = (T * T).sum(1); D = r[:, None] + r[None, :] - 2.0 * (T @ T.T)
    K = torch.exp(-D / h2 / 2.0)
    return (K @ G + (K.sum(1)[:, None] * T - K @ T) / h2) / n
worst = 0.0
for n, d, dt in [(257, 33, torch.float32), (1000, 130, torch.float32), (1536, 256, torch.float32), (640, 2001, torch.float32),
                 (4096, 128, torch.float32), (2048, 128, torch.float32), (4096, 256, torch.float32),
                 (1024, 128, torch.bfloat16), (4096, 128, torch.bfloat16), (4096, 256, torch.bfloat16)]:
    g = torch.Generator().manual_seed(n + d)
    T = torch.randn(n, d, generator=g).cuda(); G = torch.randn(n, d, generator=g).cuda()
    if dt == torch.bfloat16: T, G = T.bfloat16().float(), G.bfloat16().float()
    eng = SvgdEngine(n, d, device="cuda", small=False, dtype=dt)
    phi = eng.compute_phi(T.to(dt), G.to(dt)).double(); torch.cuda.synchronize()
    r = ref(T, G, eng.h2.item())
    err = ((phi - r).norm() / r.norm()).item()
    tol = 1e-5 if dt == torch.float32 else 4e-3
    worst = max(worst, err / tol)
    rowerr = ((phi - r).norm(dim=1) / r.norm(dim=1))
    print((n, d), str(dt).split(".")[-1], "split", eng.split, "rel err %.3e" % err, "worst rows", rowerr.topk(3).indices.tolist(), flush=True)
print("OK" if worst < 1 else "FAILED", "worst/tol %.3f" % worst)
