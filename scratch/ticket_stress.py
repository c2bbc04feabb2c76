"""Stress of the "last workgroup out" tickets: the fused call (scales by k_colmax's last workgroup, final resolve by the
level-2 pass's last workgroup) against the staged calls (separate k_make_scales / k_resolve launches) on the same inputs,
with theta rescaled at random every step so the median window misses and the chained select really runs.
Everything must agree bit for bit."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd.engine import SvgdEngine
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
total = 0
for case in range(40):
    n = int(rng.choice([129, 200, 256, 333, 512, 777, 1024, 1500, 2048, 3000]))
    d = int(rng.choice([3, 17, 64, 128, 130, 256, 300]))
    fused = SvgdEngine(n, d, device="cuda", small=False)
    staged = SvgdEngine(n, d, device="cuda", small=False)
    for step in range(12):
        s = float(rng.uniform(0.05, 20.0)) if step % 3 != 2 else 1.0     # two jumps, then one smooth step (window may hit)
        T = torch.randn(n, d, device="cuda") * s
        G = torch.randn(n, d, device="cuda") * float(rng.uniform(0.01, 100.0))
        pf = fused.compute_phi(T, G).clone()
        ps = staged.compute_phi(T, G, mark=lambda _: None).clone()
        torch.cuda.synchronize()
        total += 1
        if float(fused.h2) != float(staged.h2) or not torch.equal(pf, ps) or float(fused.sqnorm) != float(staged.sqnorm):
            bad += 1
            print("MISMATCH n=%d d=%d step %d: h2 %r vs %r, phi max diff %.3e" %
                  (n, d, step, float(fused.h2), float(staged.h2), float((pf - ps).abs().max())), flush=True)
    del fused, staged
print("ticket stress: %d comparisons, %d mismatches" % (total, bad))
