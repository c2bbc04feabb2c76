"""Does replaying the fused step from a captured HIP graph shorten it?  (launch gaps between its 7-8 dependent kernels)
usage: python scratch/graph_try.py <c2|c3>"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd.engine import SvgdEngine
from stein_amd.optimizers import AdagradGradientDescent
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
n, d, dt = (4096, 128, torch.bfloat16) if which == "c2" else (16384, 256, torch.float32)
torch.manual_seed(0)
T = torch.randn(n, d, device="cuda"); G = torch.randn(n, d, device="cuda").to(dt)
Tq = torch.empty(n, d, device="cuda", dtype=dt)
eng = SvgdEngine(n, d, device="cuda", dtype=dt); gd = AdagradGradientDescent(learning_rate=1e-3)
def step():
    if dt != torch.float32: Tq.copy_(T)
    phi = eng.compute_phi(Tq if dt != torch.float32 else T, G)
    gd.apply_(T, phi, eng.sqnorm)
for _ in range(6): step()
torch.cuda.synchronize()
def timed(fn, steps=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
print("%s eager: %.4f ms/step" % (which, timed(step)))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
try:
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g, stream=s):
        step()
    torch.cuda.synchronize()
    print("%s graph replay: %.4f ms/step" % (which, timed(g.replay)))
    print("finite:", bool(torch.isfinite(T).all()), "window:", eng.window_stats() if hasattr(eng, "window_stats") else None)
except Exception as exc:
    print("capture failed: %s: %s" % (type(exc).__name__, str(exc)[:300]))
