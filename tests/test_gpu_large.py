"""GPU: BASELINE.json-sized runs checked through size-independent properties, plus a medium-size parity check
against a torch fp64 evaluation of the same formulae on the device (test-side reference for a floating-point
kernel; the product path never uses it)."""
import math

import numpy as np
import pytest
import torch

from stein_amd.engine import SvgdEngine, untile_distances

pytestmark = pytest.mark.gpu


def torch_fp64_phi(T, G):
    T, G = T.double(), G.double()
    n = T.shape[0]
    r = (T * T).sum(1)
    D = r[:, None] + r[None, :] - 2.0 * (T @ T.T)
    med = D.flatten().float().double().median() if False else None
    flat = D.flatten().sort().values
    m = flat.numel()
    med = 0.5 * (flat[m // 2 - 1] + flat[m // 2]) if m % 2 == 0 else flat[m // 2]
    h2 = med / math.log(n)
    K = torch.exp(-D / h2 / 2.0)
    dK = (K.sum(1)[:, None] * T - K @ T) / h2
    return (K @ G + dK) / n, h2


@pytest.mark.parametrize("n,d", [(4096, 128), (2048, 2001), (3000, 250)])
def test_medium_sizes_against_torch_fp64(cuda, n, d):
    gen = torch.Generator(device="cpu").manual_seed(n + d)
    T = torch.randn(n, d, generator=gen).to(cuda)
    G = torch.randn(n, d, generator=gen).to(cuda)
    eng = SvgdEngine(n, d, device=cuda)
    phi = eng.compute_phi(T, G).double()
    ref, h2 = torch_fp64_phi(T, G)
    assert abs(eng.h2.item() - h2.item()) <= 3e-6 * h2.item()
    err = ((phi - ref).norm() / ref.norm()).item()
    assert err <= 1e-5, err
    assert bool(((phi - ref).abs() <= 1e-5 * ref.abs().max() + 1e-5 * ref.abs()).all())
    assert abs(eng.sqnorm.item() - (ref ** 2).sum().item()) <= 2e-5 * (ref ** 2).sum().item()


def _exact_median_check(D, med_lo_hi, total):
    """Counting proof that lo/hi are the two middle order statistics of D (no sort of n^2 values needed)."""
    lo, hi = med_lo_hi
    k_lo = total // 2 - 1 if total % 2 == 0 else total // 2
    k_hi = total // 2
    for v, k in ((lo, k_lo), (hi, k_hi)):
        less = int((D < v).sum().item())
        leq = int((D <= v).sum().item())
        assert less <= k < leq, (v, k, less, leq)


def test_c3_full_size_properties(cuda):
    """n=16384, d=256 (roofline config): symmetry of D, exactness of the radix-select median by counting,
    dK antisymmetry (column sums vanish), linearity of phi in the score, determinism."""
    n, d = 16384, 256
    gen = torch.Generator(device="cpu").manual_seed(0)
    T = torch.randn(n, d, generator=gen).to(cuda)
    G = torch.randn(n, d, generator=gen).to(cuda)
    eng = SvgdEngine(n, d, device=cuda)
    dK = torch.empty(n, d, device=cuda)
    phi = eng.compute_phi(T, G, dK_out=dK).clone()
    D = eng.dist_matrix()
    assert torch.equal(D, D.T)
    # select state: lo / hi order statistics follow the 32 bytes of ranks/prefixes/flags: floats 8..11 = median, h2, lo, hi
    st = eng.select_state.view(torch.float32)
    med, h2, lo, hi = st[8].item(), st[9].item(), st[10].item(), st[11].item()
    _exact_median_check(D, (lo, hi), n * n)
    assert med == np.float32(0.5) * (np.float32(lo) + np.float32(hi))
    bw = np.sqrt(np.float32(med) / np.float32(math.log(n)))
    assert h2 == np.float32(bw * bw) == eng.h2.item()
    # antisymmetry: sum_i dK_i = 0 (cancellation of ~n terms of size ~|dK|)
    assert dK.double().sum(0).abs().max().item() <= 1e-3 * dK.abs().max().item() * math.sqrt(n)
    # linearity in the score: phi(T, aG1 + bG2) - dK/n = a (phi(T,G1) - dK/n) + b (phi(T,G2) - dK/n)
    G2 = torch.randn(n, d, generator=gen).to(cuda)
    phi2 = eng.compute_phi(T, G2).clone()
    mix = eng.compute_phi(T, (0.5 * G - 2.0 * G2).contiguous()).clone()
    base = dK / n
    lhs = mix - base
    rhs = 0.5 * (phi - base) - 2.0 * (phi2 - base)
    assert ((lhs - rhs).norm() / rhs.norm()).item() <= 1e-5
    # same inputs, same bits
    again = eng.compute_phi(T, G)
    assert torch.equal(again, phi)
    # |phi|^2 reduction
    assert abs(eng.sqnorm.item() - (phi.double() ** 2).sum().item()) <= 1e-9 * eng.sqnorm.item()


def test_c3_row_block_equals_full(cuda):
    """A rank owning rows [row0, row0+n_local) must produce exactly the rows of the single-rank result when it is
    given the global histogram -- emulates the 8-way sharding of config 5 on one GPU by looping over blocks."""
    from stein_amd import _lib
    n, d, parts = 4096, 128, 4
    gen = torch.Generator(device="cpu").manual_seed(5)
    T = torch.randn(n, d, generator=gen).to(cuda)
    G = torch.randn(n, d, generator=gen).to(cuda)
    full = SvgdEngine(n, d, device=cuda, x3=False)   # bitwise block == full only holds for the fp32-MFMA kernels
    phi_full = full.compute_phi(T, G).clone()

    nl = n // parts
    total, offs, extra = _lib.workspace_layout(nl, n, d)
    ld = extra[_lib.WSX_LD_DIST]
    st = full.stages
    blocks = []
    for p in range(parts):
        ws = torch.empty(total, dtype=torch.uint8, device=cuda)
        nlp = (nl + 127) // 128 * 128           # the block is stored tile-major with rows padded to 128
        D = ws[offs[_lib.WS_DIST]:offs[_lib.WS_DIST] + nlp * ld * 4].view(torch.float32).view(nlp, ld)
        r = torch.empty(n, device=cuda)
        st.rownorms(T, n, d, r)
        st.distance_block(T, r, n, d, p * nl, nl, D, ld)
        blocks.append((ws, D))
        assert torch.equal(untile_distances(D, nl, n), full.dist_matrix()[p * nl:(p + 1) * nl])
    hist = torch.zeros(_lib.HIST_LEVELS, 2, _lib.HIST_BINS, dtype=torch.int64, device=cuda)
    sel = torch.zeros(64, dtype=torch.uint8, device=cuda)
    h2 = torch.zeros(1, device=cuda)
    med = torch.zeros(1, device=cuda)
    st.median_begin(hist, sel, n * n)
    for lv in range(_lib.HIST_LEVELS):
        for ws, D in blocks:                       # "all-reduce": every block adds into the same histogram
            st.median_hist_pass(D, ld, nl, n, lv, sel, hist)
        st.median_resolve(hist, lv, n, sel, h2, med)
    assert h2.item() == full.h2.item()
    sq = 0.0
    for p, (ws, D) in enumerate(blocks):
        phi = torch.empty(nl, d, device=cuda)
        sqp = torch.zeros(1, dtype=torch.float64, device=cuda)
        st.kernel_contract(D, ld, T, G, n, d, p * nl, nl, h2, phi, sqp, None, ws)
        sq += sqp.item()
        # split factors differ between the full and the block layouts, so sums are re-associated: allow rounding
        assert ((phi - phi_full[p * nl:(p + 1) * nl]).norm() / phi_full.norm()).item() <= 1e-6
    assert abs(sq - full.sqnorm.item()) <= 1e-6 * sq
