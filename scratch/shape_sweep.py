"""One-off sweep: fused call (one-kernel path where it applies, tiled split path, tiled fp32 path) against the oracle
over awkward shapes; several steps each so the speculative window is exercised.  Prints the worst relative errors."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import svgd_oracle as orc
from stein_amd.engine import SvgdEngine
rng = np.random.default_rng(0)
shapes = [(n, d) for n in (2, 3, 31, 32, 33, 127, 128, 129, 130, 255, 256, 257, 383, 385, 511, 640, 777) for d in (1, 2, 31, 32, 33, 63, 65, 127, 128, 129, 257)]
sel = [shapes[i] for i in rng.choice(len(shapes), 48, replace=False)] + [(1023, 5), (1025, 3), (200, 513), (129, 1000)]
worst = {}
for n, d in sel:
    T64, G64 = rng.normal(size=(n, d)) * rng.uniform(0.1, 10), rng.normal(size=(n, d)) * rng.uniform(0.01, 100)
    for name, kw in (("default", {}), ("tiled-split", dict(small=False)), ("tiled-fp32", dict(small=False, x3=False))):
        eng = SvgdEngine(n, d, device="cuda", **kw)
        T = torch.tensor(T64, dtype=torch.float32, device="cuda"); G = torch.tensor(G64, dtype=torch.float32, device="cuda")
        for step in range(4):
            phi = eng.compute_phi(T, G)
            ref = orc.svgd_step(T.double().cpu().numpy(), G.double().cpu().numpy(), orc.AdagradState(), np.float32)
            e_phi = np.abs(phi.double().cpu().numpy() - ref["phi"]).max() / np.abs(ref["phi"]).max()
            e_h2 = abs(float(eng.h2) - float(ref["h2"])) / float(ref["h2"])
            if not (e_phi < 1e-5 and e_h2 < 1e-5):
                print("FAIL", name, (n, d), "step", step, "phi %.2e h2 %.2e" % (e_phi, e_h2))
            w = worst.setdefault(name, [0.0, 0.0, None])
            if e_phi > w[0]: w[0], w[2] = e_phi, (n, d)
            w[1] = max(w[1], e_h2)
            T = T + 1e-3 * phi
print({k: ("phi %.2e at %s, h2 %.2e" % (v[0], v[2], v[1])) for k, v in worst.items()})
