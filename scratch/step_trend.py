"""Wall time per step of the plain fused loop in consecutive blocks of 20 steps (does the step get faster over a run?)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
from stein_amd.engine import SvgdEngine
from stein_amd.optimizers import AdagradGradientDescent
n, d = 16384, 256
torch.manual_seed(0)
T = torch.randn(n, d, device="cuda"); G = torch.randn(n, d, device="cuda")
eng = SvgdEngine(n, d, device="cuda"); gd = AdagradGradientDescent(learning_rate=1e-3, alpha=0.9)
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
prewarm_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
if prewarm_ms:      # a generic load (fp32 matmuls) before anything of the workload runs
    A = torch.randn(4096, 4096, device="cuda"); B = torch.randn(4096, 4096, device="cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < prewarm_ms:
        for _ in range(10): C = A @ B
        torch.cuda.synchronize()
for _ in range(5):
    phi = eng.compute_phi(T, G); gd.apply_(T, phi, eng.sqnorm)
torch.cuda.synchronize()
out = []
for blk in range(12):
    if mode != "plain": _lib.timing_reserve(20)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        phi = eng.compute_phi(T, G, timing={"plain": False, "contract": "contract", "full": True}[mode]); gd.apply_(T, phi, eng.sqnorm)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
    extra = ""
    if mode != "plain":
        pc = _lib.timing_read(20); extra = " contract %.4f" % (sum(c["contract"] for c in pc) / len(pc))
    out.append("%.4f%s" % (ms, extra))
print(mode, "prewarm %g ms" % prewarm_ms, "| ms/step per block of 20:", "  ".join(out))
