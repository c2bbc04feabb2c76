"""Does a HIP graph of one fused step (stein_svgd_phi + apply) beat eager launches at latency-bound sizes?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd.engine import SvgdEngine
from stein_amd.optimizers import AdagradGradientDescent
for n, d in ((256, 64), (1024, 128), (4096, 256), (100, 10)):
    theta = torch.randn(n, d, device="cuda"); G = torch.randn(n, d, device="cuda")
    eng = SvgdEngine(n, d, device="cuda"); gd = AdagradGradientDescent(learning_rate=1e-3)
    def step():
        phi = eng.compute_phi(theta, G); gd.apply_(theta, phi, eng.sqnorm)
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): step()
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 200
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            step()
    except Exception as e:
        print("n=%d d=%d capture failed: %s" % (n, d, str(e)[:200])); continue
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): g.replay()
    torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 200
    print("n=%d d=%d  eager %.1f us/step   graph %.1f us/step" % (n, d, eager * 1e6, graph * 1e6), flush=True)
