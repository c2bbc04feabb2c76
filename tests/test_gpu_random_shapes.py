"""GPU: seeded random shapes through the fused call against the oracle.

The hand-picked shapes of the other tests sit on round numbers; here n and d are drawn at random (n from 161 -- past the
one-kernel path -- to 1400, d from 1 to 300), so row tiles, k tiles, column blocks, split chunks and the mirrored stages
of the contraction (the k tiles left of a row tile's diagonal block, read from the upper-only distance image) all end
ragged in every combination.  Three steps each, so the speculative median window is set up and consulted too (the bandwidth is checked exactly at
every step, whichever way it was found)."""
import numpy as np
import pytest
import torch

from oracle import svgd_oracle as orc
from stein_amd.engine import SvgdEngine
from stein_amd.optimizers import AdagradGradientDescent

pytestmark = pytest.mark.gpu

_rng = np.random.default_rng(20261004)
SHAPES = sorted({(int(_rng.integers(161, 1400)), int(_rng.integers(1, 300))) for _ in range(16)})
SHAPES += [(1281, 129), (1279, 257), (385, 1), (1153, 128)]     # one past / one short of the tile edges


@pytest.mark.parametrize("n,d", SHAPES)
def test_fused_call_matches_oracle_on_random_shapes(cuda, n, d):
    rng = np.random.default_rng(n * 1009 + d)
    T0 = rng.normal(size=(n, d)) * rng.uniform(0.3, 3.0)
    G0 = rng.normal(size=(n, d)) * rng.uniform(0.1, 10.0)
    theta = torch.tensor(T0, dtype=torch.float32, device=cuda)
    score = torch.tensor(G0, dtype=torch.float32, device=cuda)
    eng = SvgdEngine(n, d, device=cuda)
    gd = AdagradGradientDescent(learning_rate=1e-2)
    gd_o = orc.AdagradState(learning_rate=1e-2, alpha=0.9)
    th = theta.cpu().numpy()
    for step in range(3):
        phi = eng.compute_phi(theta, score)
        ref = orc.svgd_step(th, score.cpu().numpy().astype(np.float64), gd_o, np.float32)
        torch.cuda.synchronize()
        D = eng.dist_matrix()
        assert torch.equal(D, D.T)                                            # built from the upper image: symmetric
        med = orc.median_all(D.cpu().numpy())
        assert float(eng.h2.item()) == float(orc.bandwidth_sq(med, n, np.float32))   # exact median of the GPU's own D
        assert abs(float(eng.h2.item()) - ref["h2"]) <= 4e-6 * ref["h2"]
        err = np.linalg.norm(phi.cpu().numpy() - ref["phi"]) / np.linalg.norm(ref["phi"])
        assert err <= 1e-5, (step, err)                                       # north-star tolerance
        assert abs(float(eng.sqnorm.item()) - ref["sqnorm"]) <= 2e-5 * ref["sqnorm"]
        gd.apply_(theta, phi, eng.sqnorm)
        th = ref["theta_new"].astype(np.float32)
        # (theta itself is not compared here: Adagrad's first step is lr * phi / (1e-6 + |phi|), which turns the absolutely
        # tiny error of an entry with |phi| ~ 1e-6 into a visible fraction of a step; the apply kernels have their own tests)
        theta.copy_(torch.tensor(th, device=cuda))                            # keep both trajectories on the same particles
    steps, hits = eng.window_stats()
    assert steps == 3 and 0 <= hits <= 2      # the first step has no window; whether the later ones hit depends on the jump
