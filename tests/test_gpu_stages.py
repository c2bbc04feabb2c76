"""Stage-by-stage parity of the fp32-MFMA HIP path (x3=False; the split-precision default is covered by test_gpu_x3.py)
through the C ABI against the CPU oracle.

Tolerances (north star: 1e-5 relative, fp32):
  * distances: |dD| <= 2e-6 * (r_i + r_j + 2|S_ij|)-scale, checked as 4e-6 * max|D|  (fp32 cancellation in
    r + r^T - 2TT^T; the oracle's BLAS sums in a different order)
  * median / h2: 2e-6 relative against the oracle's median of ITS OWN D, and bit-exact against the exact median of
    the GPU's own D (np.partition on the downloaded matrix)
  * phi: relative Frobenius error <= 1e-5 and elementwise |d| <= 1e-5 * max|phi| + 1e-5 * |phi| against the fp64 oracle
"""
import numpy as np
import pytest
import torch

from oracle import svgd_oracle as orc
from stein_amd import _lib
from stein_amd.engine import SvgdEngine

pytestmark = pytest.mark.gpu

SHAPES = [(7, 3), (8, 5), (100, 10), (257, 33), (512, 48), (1000, 130), (1536, 256)]


def _inputs(n, d, seed=0, scale=1.0):
    rng = np.random.default_rng(seed + 1000 * n + d)
    return rng.normal(size=(n, d)) * scale, rng.normal(size=(n, d))


def _staged(eng, T, G, want_K=False):
    """Run the staged calls by hand so intermediates can be inspected."""
    st, n, d = eng.stages, eng.n, eng.d
    st.rownorms(T, n, d, eng.rownorm)
    st.distance_block(T, eng.rownorm, n, d, 0, n, eng.dist, eng.ld_dist)
    st.median_begin(eng.hist, eng.select_state, n * n)
    for lv in range(_lib.HIST_LEVELS):
        st.median_hist_pass(eng.dist, eng.ld_dist, n, n, lv, eng.select_state, eng.hist)
        st.median_resolve(eng.hist, lv, n, eng.select_state, eng.h2, eng.median)
    dK = torch.empty(n, d, dtype=torch.float32, device=T.device)
    st.kernel_contract(eng.dist, eng.ld_dist, T, G, n, d, 0, n, eng.h2, eng.phi, eng.sqnorm, dK, eng.ws)
    torch.cuda.synchronize()
    return dK


@pytest.mark.parametrize("n,d", SHAPES)
def test_stages_match_oracle(cuda, n, d):
    T64, G64 = _inputs(n, d)
    T = torch.tensor(T64, dtype=torch.float32, device=cuda)
    G = torch.tensor(G64, dtype=torch.float32, device=cuda)
    eng = SvgdEngine(n, d, device=cuda, x3=False, small=False)
    dK = _staged(eng, T, G)

    T32 = T.cpu().numpy()
    # rownorms + distances
    r = eng.rownorm.cpu().numpy()
    np.testing.assert_allclose(r, (T32.astype(np.float64) ** 2).sum(1), rtol=2e-6)
    D = eng.dist_matrix().cpu().numpy()
    D64 = orc.pairwise_sq_dists(T32, np.float64)
    assert np.abs(D - D64).max() <= 4e-6 * np.abs(D64).max()
    assert np.array_equal(D, D.T), "distance block must be bitwise symmetric"

    # exact median of the GPU's own D, bit for bit
    med_gpu = eng.median.cpu().numpy()[0]
    assert med_gpu == orc.median_all(D)
    h2_gpu = eng.h2.cpu().numpy()[0]
    assert h2_gpu == orc.bandwidth_sq(med_gpu, n, np.float32)
    h2_64 = orc.bandwidth_sq(orc.median_all(D64), n, np.float64)
    assert abs(h2_gpu - h2_64) <= 2e-6 * h2_64

    # phi, dK, |phi|^2 against the fp64 oracle on the same fp32-rounded inputs
    ref = orc.svgd_step(T32.astype(np.float64), G.cpu().numpy().astype(np.float64), orc.AdagradState(), np.float64)
    phi = eng.phi.cpu().numpy().astype(np.float64)
    err = np.linalg.norm(phi - ref["phi"]) / np.linalg.norm(ref["phi"])
    assert err <= 1e-5, err
    assert np.all(np.abs(phi - ref["phi"]) <= 1e-5 * np.abs(ref["phi"]).max() + 1e-5 * np.abs(ref["phi"]))
    dk = dK.cpu().numpy().astype(np.float64)
    assert np.linalg.norm(dk - ref["dK"]) <= 1e-5 * np.linalg.norm(ref["dK"])
    assert abs(eng.sqnorm.item() - ref["sqnorm"]) <= 2e-5 * ref["sqnorm"]
    # the faithful (fp32-kernel) oracle agrees as well
    ref32 = orc.svgd_step(T32.astype(np.float64), G.cpu().numpy().astype(np.float64), orc.AdagradState(), np.float32)
    assert np.linalg.norm(phi - ref32["phi"]) <= 2e-5 * np.linalg.norm(ref32["phi"])


@pytest.mark.parametrize("n,d", [(7, 3), (100, 10), (129, 5), (512, 48), (1000, 130), (1536, 256)])
def test_fused_equals_staged(cuda, n, d):
    T64, G64 = _inputs(n, d, seed=3)
    T = torch.tensor(T64, dtype=torch.float32, device=cuda)
    G = torch.tensor(G64, dtype=torch.float32, device=cuda)
    eng = SvgdEngine(n, d, device=cuda, x3=False, small=False)
    _staged(eng, T, G)
    phi_staged, h2_staged, sq_staged = eng.phi.clone(), eng.h2.clone(), eng.sqnorm.clone()
    D_staged, hist_staged = eng.dist_matrix(), eng.hist.clone()
    eng.phi.zero_(); eng.h2.zero_(); eng.sqnorm.zero_(); eng.dist.fill_(float("nan"))
    # the fused call runs the symmetric variant (upper-triangle tiles mirrored, level-0 histogram from the
    # distance epilogue, levels 1-2 over the upper triangle with weight 2): every bit must agree
    eng.compute_phi(T, G)
    torch.cuda.synchronize()
    assert torch.equal(eng.dist_matrix(), D_staged)
    if n > 512:
        assert torch.equal(eng.hist, hist_staged)
    else:   # small blocks: the fused call selects levels 1-2 inside one workgroup (LDS histograms); level 0 is shared
        assert torch.equal(eng.hist[0], hist_staged[0])
    assert torch.equal(eng.phi, phi_staged) and torch.equal(eng.h2, h2_staged) and torch.equal(eng.sqnorm, sq_staged)
    # the marked (bench) path = staged calls with the same symmetric flags
    eng.phi.zero_()
    eng.compute_phi(T, G, mark=lambda label: None)
    assert torch.equal(eng.phi, phi_staged) and torch.equal(eng.hist, hist_staged)
    # determinism: same inputs, same bits
    again = eng.compute_phi(T, G).clone()
    assert torch.equal(again, phi_staged)


def test_kernel_matrix_output(cuda):
    n, d = 257, 33
    T64, _ = _inputs(n, d, seed=5)
    T = torch.tensor(T64, dtype=torch.float32, device=cuda)
    eng = SvgdEngine(n, d, device=cuda, x3=False, small=False)
    K = torch.empty(n, n, dtype=torch.float32, device=cuda)
    dK = torch.empty(n, d, dtype=torch.float32, device=cuda)
    eng.compute_phi(T, T, K_out=K, dK_out=dK)
    K64, dK64, h264 = orc.kernel_and_grad(T.cpu().numpy(), np.float64, return_h2=True)
    assert np.abs(K.cpu().numpy() - K64).max() <= 1e-5
    assert np.linalg.norm(dK.cpu().numpy() - dK64) <= 1e-5 * np.linalg.norm(dK64)
    Kn = K.cpu().numpy()
    assert np.array_equal(Kn, Kn.T)
    assert np.abs(np.diag(Kn) - 1).max() < 1e-4          # diag(D) is computed, not assumed zero
    assert np.abs(dK.cpu().numpy().sum(0)).max() < 1e-3  # sum_i dK_i = 0 by antisymmetry


def test_clustered_particles_with_offset(cuda):
    """Tight cluster far from the origin: r + r^T - 2TT^T cancels badly in fp32 (reference behaviour too).
    The GPU must agree with the fp32-faithful oracle's own conditioning, not with fp64."""
    n, d = 300, 20
    rng = np.random.default_rng(9)
    T64 = 0.5 + 0.01 * rng.normal(size=(n, d))
    G64 = rng.normal(size=(n, d))
    T = torch.tensor(T64, dtype=torch.float32, device=cuda)
    G = torch.tensor(G64, dtype=torch.float32, device=cuda)
    eng = SvgdEngine(n, d, device=cuda, x3=False, small=False)
    phi = eng.compute_phi(T, G).cpu().numpy()
    ref64 = orc.svgd_step(T.cpu().numpy().astype(np.float64), G.cpu().numpy().astype(np.float64), orc.AdagradState(), np.float64)
    ref32 = orc.svgd_step(T.cpu().numpy().astype(np.float64), G.cpu().numpy().astype(np.float64), orc.AdagradState(), np.float32)
    e_gpu = np.linalg.norm(phi - ref64["phi"]) / np.linalg.norm(ref64["phi"])
    e_o32 = np.linalg.norm(ref32["phi"] - ref64["phi"]) / np.linalg.norm(ref64["phi"])
    assert np.isfinite(phi).all()
    assert e_gpu <= max(5 * e_o32, 1e-5), (e_gpu, e_o32)


def test_translation_and_permutation_properties(cuda):
    n, d = 384, 40
    T64, G64 = _inputs(n, d, seed=11)
    T = torch.tensor(T64, dtype=torch.float32, device=cuda)
    G = torch.tensor(G64, dtype=torch.float32, device=cuda)
    eng = SvgdEngine(n, d, device=cuda, x3=False, small=False)
    base = eng.compute_phi(T, G).clone()
    perm = torch.randperm(n, device=cuda)
    permuted = eng.compute_phi(T[perm].contiguous(), G[perm].contiguous()).clone()
    assert (permuted - base[perm]).norm() <= 1e-5 * base.norm()
    shifted = eng.compute_phi((T + 0.25).contiguous(), G).clone()   # K, dK are translation invariant
    assert (shifted - base).norm() <= 2e-5 * base.norm()


def test_bad_arguments(cuda):
    with pytest.raises(ValueError):
        SvgdEngine(1, 4, device=cuda)                      # ln(1) = 0 in the reference -> refused here
    eng = SvgdEngine(16, 4, device=cuda)
    with pytest.raises(ValueError):
        eng.compute_phi(torch.zeros(15, 4, device=cuda), torch.zeros(16, 4, device=cuda))
    with pytest.raises(ValueError):
        eng.compute_phi(torch.zeros(16, 4, device=cuda, dtype=torch.float64), torch.zeros(16, 4, device=cuda))
    small = torch.empty(16, dtype=torch.uint8, device=cuda)
    from stein_amd._lib import SteinHipError
    with pytest.raises(SteinHipError):
        eng.stages.svgd_phi(torch.zeros(16, 4, device=cuda), torch.zeros(16, 4, device=cuda), 16, 4, eng.phi, eng.h2,
                            eng.sqnorm, None, None, small)


def test_identical_particles_are_nan_like_reference(cuda):
    """All particles equal -> median 0 -> bandwidth 0 -> exp(-0/0): the reference yields NaN; so do we (no crash)."""
    T = torch.ones(32, 8, device=cuda)
    G = torch.ones(32, 8, device=cuda)
    eng = SvgdEngine(32, 8, device=cuda, x3=False, small=False)
    phi = eng.compute_phi(T, G)
    torch.cuda.synchronize()
    assert eng.median.item() == 0.0
    assert not torch.isfinite(phi).all()


def test_unaligned_outputs_take_the_scalar_paths(cuda):
    """k_phi_finish and the Adagrad apply use 16-byte accesses only when every pointer allows it; a caller's 4-byte
    aligned phi / dK / theta / state must give the same values through the scalar loops (include/steinhip.h makes
    no alignment demand on them)."""
    import ctypes
    n, d = 300, 64
    T64, G64 = _inputs(n, d, seed=11)
    T = torch.tensor(T64, dtype=torch.float32, device=cuda)
    G = torch.tensor(G64, dtype=torch.float32, device=cuda)
    eng = SvgdEngine(n, d, device=cuda, small=False)
    _staged(eng, T, G)
    phi_ref, sq_ref = eng.phi.clone(), eng.sqnorm.clone()
    buf = torch.zeros(n * d + 1, dtype=torch.float32, device=cuda)
    dkb = torch.zeros(n * d + 1, dtype=torch.float32, device=cuda)
    phi_un, dk_un = buf[1:].view(n, d), dkb[1:].view(n, d)          # 4 bytes past a 16-byte boundary
    assert phi_un.data_ptr() % 16 == 4
    eng.stages.contract_finish(T, n, d, 0, n, eng.h2, phi_un, eng.sqnorm, dk_un, eng.ws, eng.flags)
    torch.cuda.synchronize()
    assert torch.equal(phi_un, phi_ref)
    assert abs(float(eng.sqnorm) - float(sq_ref)) <= 1e-12 * float(sq_ref)

    def apply(theta, phi, hist):
        stream = ctypes.c_void_p(torch.cuda.current_stream(cuda).cuda_stream)
        _lib.call("stein_apply_adagrad", ctypes.c_void_p(theta.data_ptr()), ctypes.c_void_p(phi.data_ptr()), _lib.F32,
                  ctypes.c_void_p(hist.data_ptr()), theta.numel(), _lib.F32, None, 1.0, 10.0, 1e-2, 0.9, 1e-6, 0, None, stream)
    th_a, hi_a = T.clone(), torch.rand(n, d, device=cuda)
    tb, hb = torch.zeros(n * d + 1, device=cuda), torch.zeros(n * d + 1, device=cuda)
    th_u, hi_u = tb[1:].view(n, d), hb[1:].view(n, d)
    th_u.copy_(th_a); hi_u.copy_(hi_a)
    apply(th_a, phi_ref, hi_a)
    apply(th_u, phi_ref, hi_u)
    torch.cuda.synchronize()
    assert torch.equal(th_a, th_u) and torch.equal(hi_a, hi_u)


@pytest.mark.gpu
def test_column_maxima_paths_agree(cuda):
    """The operand scales come from column maxima taken with 16-byte loads when d % 4 == 0 and the inputs are 16-byte
    aligned, else four bytes at a time: the same inputs 4 bytes past a 16-byte boundary must give byte-identical planes."""
    n, d = 1500, 256
    T64, G64 = _inputs(n, d, seed=12, scale=3.0)
    T = torch.tensor(T64, dtype=torch.float32, device=cuda)
    G = torch.tensor(G64 * 1e-3, dtype=torch.float32, device=cuda)
    eng = SvgdEngine(n, d, device=cuda, x3=True, small=False)
    st = eng.stages
    eng.planes.zero_()                            # (the section has padding that nothing writes)
    st.x3_prepare(T, G, n, d, eng.planes)
    torch.cuda.synchronize()
    aligned = eng.planes.clone()
    tb, gb = torch.zeros(n * d + 1, device=cuda), torch.zeros(n * d + 1, device=cuda)
    T_un, G_un = tb[1:].view(n, d), gb[1:].view(n, d)
    T_un.copy_(T); G_un.copy_(G)
    assert T_un.data_ptr() % 16 == 4 and G_un.data_ptr() % 16 == 4
    eng.planes.zero_()
    st.x3_prepare(T_un, G_un, n, d, eng.planes)
    torch.cuda.synchronize()
    assert torch.equal(eng.planes, aligned)
