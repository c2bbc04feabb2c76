"""Hit rate of the speculative median window when the score changes at random every step (minibatch-like noise)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
from stein_amd.engine import SvgdEngine
from stein_amd.optimizers import AdagradGradientDescent, AdamGradientDescent
n, d = int(sys.argv[1]), int(sys.argv[2])
for name, gd, noise in (("adagrad 1e-3, fresh random G every step", AdagradGradientDescent(learning_rate=1e-3), 1.0),
                        ("adam 1e-1, fresh random G every step", AdamGradientDescent(learning_rate=1e-1), 1.0),
                        ("adam 1e-2, G = fixed + 30% noise", AdamGradientDescent(learning_rate=1e-2), 0.3)):
    torch.manual_seed(0)
    theta = torch.randn(n, d, device="cuda"); G0 = torch.randn(n, d, device="cuda")
    eng = SvgdEngine(n, d, device="cuda")
    _, offs, _ = _lib.workspace_layout(n, n, d, flags=eng.flags)
    o = offs[_lib.WS_SELECT] + 64
    hits, hws, cnts = [], [], []
    for step in range(60):
        G = torch.randn(n, d, device="cuda") if noise == 1.0 else G0 + noise * torch.randn(n, d, device="cuda")
        phi = eng.compute_phi(theta, G); gd.apply_(theta, phi, eng.sqnorm)
        u = eng.ws[o:o + 64].cpu().numpy().view(np.uint32)
        hits.append(int(u[7])); hws.append(int(u[2])); cnts.append(int(u[5]))
    print("%-45s hits %d/60  (last 40: %d)  median halfwidth %d  median entries %d  finite %s" %
          (name, sum(hits), sum(hits[20:]), int(np.median(hws[20:])), int(np.median(cnts[20:])), bool(torch.isfinite(theta).all())))
