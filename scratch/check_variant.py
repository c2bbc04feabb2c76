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
for n, d in [(257, 33), (1000, 130), (1536, 256), (640, 2001), (4096, 128)]:
    g = torch.Generator().manual_seed(n + d)
    T = torch.randn(n, d, generator=g).cuda(); G = torch.randn(n, d, generator=g).cuda()
    eng = SvgdEngine(n, d, device="cuda", small=False)
    phi = eng.compute_phi(T, G).double(); torch.cuda.synchronize()
    r = ref(T, G, eng.h2.item())
    err = ((phi - r).norm() / r.norm()).item(); worst = max(worst, err)
    print((n, d), "rel err %.3e" % err, flush=True)
print("OK" if worst < 1e-5 else "FAILED", "worst %.3e" % worst)
