"""Tiled path at latency-bound sizes: split-precision kernels (need the prepare stage) against the fp32-MFMA kernels."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd.engine import SvgdEngine
from stein_amd.optimizers import AdagradGradientDescent
def t(eng, T, G):
    gd = AdagradGradientDescent(learning_rate=1e-3)
    th = T.clone()
    for _ in range(10): gd.apply_(th, eng.compute_phi(th, G), eng.sqnorm)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(100): gd.apply_(th, eng.compute_phi(th, G), eng.sqnorm)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 10
for n, d in ((130, 10), (200, 64), (256, 64), (256, 256), (512, 128), (512, 1024), (1024, 128), (1024, 512), (2048, 128), (2048, 256), (4096, 128)):
    T = torch.randn(n, d, device="cuda"); G = torch.randn(n, d, device="cuda")
    a = t(SvgdEngine(n, d, device="cuda", x3=True, small=False), T, G)
    b = t(SvgdEngine(n, d, device="cuda", x3=False, small=False), T, G)
    print("n=%4d d=%4d  split %6.1f us   fp32 %6.1f us  %s" % (n, d, a, b, "<-- fp32 wins" if b < a else ""), flush=True)
