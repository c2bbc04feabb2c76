"""Step time (fused call + Adagrad apply) and the library's stage times over mid sizes: how launch-bound is the path?"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
from stein_amd.engine import SvgdEngine
from stein_amd.optimizers import AdagradGradientDescent
sizes = [(int(a), int(b)) for a, b in (s.split("x") for s in sys.argv[1:])] or [(256, 64), (512, 128), (1024, 128), (2048, 128), (4096, 128), (4096, 256), (8192, 256)]
for n, d in sizes:
    torch.manual_seed(0)
    theta = torch.randn(n, d, device="cuda"); G = torch.randn(n, d, device="cuda")
    eng = SvgdEngine(n, d, device="cuda"); gd = AdagradGradientDescent(learning_rate=1e-3, alpha=0.9)
    for i in range(5):
        phi = eng.compute_phi(theta, G); gd.apply_(theta, phi, eng.sqnorm)
    torch.cuda.synchronize()
    steps = 50
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps):
        phi = eng.compute_phi(theta, G); gd.apply_(theta, phi, eng.sqnorm)
    e1.record(); torch.cuda.synchronize()
    wall = e0.elapsed_time(e1) / steps
    _lib.timing_reserve(20)
    for i in range(20):
        phi = eng.compute_phi(theta, G, timing=True); gd.apply_(theta, phi, eng.sqnorm)
    torch.cuda.synchronize()
    per = _lib.timing_read(20)
    st = {k: float(np.mean([p[k] for p in per])) for k in per[0]}
    print("n=%d d=%d  ms/step %.4f  stages %s  sum %.4f" % (n, d, wall, {k: round(v, 4) for k, v in st.items()}, sum(st.values())), flush=True)
