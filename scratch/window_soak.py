"""Soak: fused call (speculative window) against the staged radix select, bandwidth compared EVERY step, under dynamics
that stress the predictor: big Adam steps, occasional jumps / rescalings / collapses, odd and even n."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
from stein_amd.engine import SvgdEngine
from stein_amd.optimizers import AdamGradientDescent
torch.manual_seed(1)
bad = 0
SHAPES = ((2049, 24), (4096, 64), (1500, 300))
if len(sys.argv) > 1 and sys.argv[1] == "big":   # blocks that take the panel distance kernels (whole K in LDS / K in chunks)
    SHAPES = ((8192, 40), (8192, 300))
for n, d in SHAPES:
    fused, ref = SvgdEngine(n, d, device="cuda"), SvgdEngine(n, d, device="cuda")
    gd = AdamGradientDescent(learning_rate=3e-2)
    theta = torch.randn(n, d, device="cuda")
    _, offs, _ = _lib.workspace_layout(n, n, d, flags=fused.flags)
    o = offs[_lib.WS_SELECT] + 64
    hits = 0
    nsteps = 400 if n <= 4096 else 150
    for step in range(nsteps):
        G = torch.randn(n, d, device="cuda")
        if step % 57 == 56: theta.mul_(float(np.random.default_rng(step).uniform(0.3, 3.0)))      # rescale
        if step % 131 == 130: theta[: n // 2] = theta[n // 2: 2 * (n // 2)]                          # collapse half the cloud
        if step == 250: theta.mul_(1e-3)                                                            # tiny scale
        if step == 300: theta.mul_(1e5)                                                             # huge scale
        phi = fused.compute_phi(theta, G).clone()
        phi_ref = ref.compute_phi(theta, G, mark=lambda s: None)
        if n <= 4096:   # both ways run the same distance kernel: bit for bit
            same = bool(((fused.h2 == ref.h2) | (fused.h2.isnan() & ref.h2.isnan())).all())
            samephi = bool(((phi == phi_ref) | (phi.isnan() & phi_ref.isnan())).all())
        else:           # fused: panel kernel, staged: per-tile kernel (it takes the level-0 histogram) -- equal to rounding
            h2a, h2b = float(fused.h2), float(ref.h2)
            same = (h2a != h2a and h2b != h2b) or abs(h2a - h2b) <= 1e-6 * abs(h2b)
            samephi = bool(phi.isnan().any() == phi_ref.isnan().any()) and \
                (bool(phi.isnan().any()) or float((phi - phi_ref).norm() / phi_ref.norm()) <= 5e-6)
        if not (same and samephi):
            bad += 1
            print("MISMATCH n=%d step %d h2 %r vs %r phi equal %s" % (n, step, float(fused.h2), float(ref.h2), samephi))
        hits += int(fused.ws[o + 28:o + 32].view(torch.int32).item())
        gd.apply_(theta, phi, fused.sqnorm)
    print("n=%d d=%d: %d steps, window hits %d, mismatches so far %d, finite %s" % (n, d, nsteps, hits, bad, bool(torch.isfinite(theta).all())), flush=True)
print("TOTAL MISMATCHES", bad)
