"""Crossover of the one-kernel path against the tiled kernels in n^2 d (both with their current launch counts)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd.engine import SvgdEngine
def t(eng, T, G):
    for _ in range(10): eng.compute_phi(T, G)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(100): eng.compute_phi(T, G)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 10
for n in (20, 50, 100, 128, 144, 160):
    for d in (32, 64, 128, 200, 303, 512, 1024, 2001):
        if n * n * d > 4_200_000: continue
        T = torch.randn(n, d, device="cuda"); G = torch.randn(n, d, device="cuda")
        a = t(SvgdEngine(n, d, device="cuda"), T, G); b = t(SvgdEngine(n, d, device="cuda", small=False), T, G)
        print("n=%3d d=%4d  n2d=%8d  one-kernel %6.1f us   tiled %6.1f us   %s" % (n, d, n * n * d, a, b, "<-- tiled wins" if b < a else ""), flush=True)
