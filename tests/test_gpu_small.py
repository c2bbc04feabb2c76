"""GPU: the one-kernel path of the fused call for n <= 160 (csrc/stein_small.hip) -- the particle counts of the
reference's own examples -- against the fp64/fp32 oracle, against the tiled kernels, and for its K / dK outputs."""
import numpy as np
import pytest
import torch

from oracle import svgd_oracle as orc
from stein_amd.engine import SvgdEngine

pytestmark = pytest.mark.gpu

SHAPES = [(2, 1), (3, 2), (7, 3), (8, 5), (20, 303), (50, 1), (100, 10), (100, 55), (127, 130), (128, 64), (64, 1000),
          (129, 3), (150, 40), (160, 64), (160, 160)]   # n^2 d > 2.2e6 (64x1000, 160x160) runs the tiled kernels


@pytest.mark.parametrize("n,d", SHAPES)
def test_small_path_matches_oracle_and_tiled_kernels(cuda, n, d):
    rng = np.random.default_rng(100 * n + d)
    T64, G64 = rng.normal(size=(n, d)), rng.normal(size=(n, d))
    T = torch.tensor(T64, dtype=torch.float32, device=cuda)
    G = torch.tensor(G64, dtype=torch.float32, device=cuda)
    small, tiled = SvgdEngine(n, d, device=cuda), SvgdEngine(n, d, device=cuda, small=False)
    K, dK = torch.empty(n, n, device=cuda), torch.empty(n, d, device=cuda)
    phi = small.compute_phi(T, G, K_out=K, dK_out=dK).clone()
    phi_t = tiled.compute_phi(T, G).clone()
    torch.cuda.synchronize()
    T32, G32 = T.double().cpu().numpy(), G.double().cpu().numpy()
    ref = orc.svgd_step(T32, G32, orc.AdagradState(), np.float32)
    Kr, dKr = orc.kernel_and_grad(T32, np.float32)[:2]
    scale = np.abs(ref["phi"]).max()
    assert abs(float(small.h2) - float(ref["h2"])) <= 2e-6 * float(ref["h2"])
    assert np.abs(phi.double().cpu().numpy() - ref["phi"]).max() <= 1e-5 * scale
    assert (phi - phi_t).abs().max().item() <= 1e-5 * scale
    assert abs(float(small.sqnorm) - float(ref["sqnorm"])) <= 1e-5 * float(ref["sqnorm"])
    assert np.abs(K.double().cpu().numpy() - Kr).max() <= 1e-5      # K <= 1: absolute = relative to the largest entry
    assert np.abs(dK.double().cpu().numpy() - dKr).max() <= 1e-5 * np.abs(dKr).max()


def test_small_path_is_taken_and_left(cuda):
    """n <= 160 takes the one-kernel path (the workspace's distance image stays untouched); larger n or small=False do not."""
    n, d = 64, 8
    T = torch.randn(n, d, device=cuda)
    G = torch.randn(n, d, device=cuda)
    for small in (True, False):
        eng = SvgdEngine(n, d, device=cuda, small=small)
        eng.dist.fill_(-7.0)
        eng.compute_phi(T, G)
        torch.cuda.synchronize()
        untouched = bool((eng.dist == -7.0).all())
        assert untouched == small


def test_even_and_odd_counts_and_ties(cuda):
    """median semantics of compute_median.py:12-15 on the LDS select: odd n*n -> middle, even -> mean of two; ties."""
    for n in (5, 6):
        d = 2
        T64 = np.zeros((n, d)); T64[:, 0] = np.arange(n) // 2          # duplicated particles -> tied distances
        G64 = np.ones((n, d))
        eng = SvgdEngine(n, d, device=cuda)
        eng.compute_phi(torch.tensor(T64, dtype=torch.float32, device=cuda), torch.tensor(G64, dtype=torch.float32, device=cuda))
        ref = orc.svgd_step(T64, G64, orc.AdagradState(), np.float32)
        assert float(eng.h2) == float(ref["h2"])
