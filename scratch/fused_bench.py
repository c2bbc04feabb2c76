"""Fused single-rank steps (the path SteinSampler.update_particles takes): ms/step and the speculative-window state."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
from stein_amd.engine import SvgdEngine
from stein_amd.optimizers import AdagradGradientDescent
n, d = int(sys.argv[1]), int(sys.argv[2]); steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
torch.manual_seed(0)
theta = torch.randn(n, d, device="cuda"); G = torch.randn(n, d, device="cuda")
eng = SvgdEngine(n, d, device="cuda")
gd = AdagradGradientDescent(learning_rate=1e-3, alpha=0.9)
off = eng.layout[1][_lib.WS_SELECT] if hasattr(eng, "layout") else None
def spec():
    tot, offs, extra = _lib.workspace_layout(n, n, d, flags=eng.flags)
    o = offs[_lib.WS_SELECT] + 64
    raw = eng.ws[o:o + 64].cpu().numpy()
    u = raw.view(np.uint32)
    return dict(center=hex(int(u[1])), hw=int(u[2]), count=int(u[5]), ovf=int(u[6]), hit=int(u[7]), below=int(raw.view(np.uint64)[4]), last=hex(int(u[12])))
for i in range(6):
    phi = eng.compute_phi(theta, G); gd.apply_(theta, phi, eng.sqnorm); torch.cuda.synchronize()
    print(i, "h2 %.6f" % float(eng.h2), spec())
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(steps):
    phi = eng.compute_phi(theta, G); gd.apply_(theta, phi, eng.sqnorm)
e1.record(); torch.cuda.synchronize()
print("ms/step %.4f" % (e0.elapsed_time(e1) / steps), spec())
