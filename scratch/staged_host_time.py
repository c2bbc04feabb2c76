"""Host cost of the staged (multi-rank style) step against the fused call, one GPU, no collectives."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd.engine import SvgdEngine
for n, d in ((2048, 256), (16384, 256)):
    T = torch.randn(n, d, device="cuda"); G = torch.randn(n, d, device="cuda")
    eng = SvgdEngine(n, d, device="cuda", small=False)
    for name, kw in (("fused", {}), ("staged", dict(mark=lambda s: None))):
        for _ in range(5): eng.compute_phi(T, G, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): eng.compute_phi(T, G, **kw)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("n=%d d=%d %-6s host issue %.1f us/step, wall %.1f us/step" % (n, d, name, (t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6))
