"""Same-box A/B of the contraction's workgroup order on a wide [G | theta] (C4 shape): fused steps, library stage events.
usage: STAMPLIB=lib_x.so python scratch/phimap_ab.py n d"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
if os.environ.get("STAMPLIB"): _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ["STAMPLIB"])
from stein_amd.engine import SvgdEngine
from stein_amd.optimizers import AdagradGradientDescent
n, d = int(sys.argv[1]), int(sys.argv[2])
torch.manual_seed(0)
th = torch.randn(n, d, device="cuda"); G = torch.randn(n, d, device="cuda")
e = SvgdEngine(n, d, device="cuda"); gd = AdagradGradientDescent(learning_rate=1e-3)
for _ in range(5):
    phi = e.compute_phi(th, G); gd.apply_(th, phi, e.sqnorm)
res = []
for rnd in range(3):
    steps = 10
    _lib.timing_reserve(steps)
    for _ in range(steps):
        phi = e.compute_phi(th, G, timing=True); gd.apply_(th, phi, e.sqnorm)
    torch.cuda.synchronize()
    per = _lib.timing_read(steps)
    res.append({s: sum(c[s] for c in per) / len(per) for s in _lib.T_STAGES})
print("n=%d d=%d" % (n, d), {s: round(statistics.median(r[s] for r in res), 4) for s in _lib.T_STAGES}, "checksum %.9e" % float(e.phi.double().sum()), flush=True)
