"""GPU: the panel-resident distance kernel (stein_amd/csrc/stein_dpanel.hip) against the per-tile kernel and the fp64
formula D = r + r^T - 2 T T^T (stein/kernels/abstract_kernel.py:33-35), on shapes small enough to force it everywhere:
ragged k extents (d = 100, 200), bf16 inputs, a symmetric matrix (upper tiles only, diagonal tiles mirrored by value) and
a row block off the origin; the speculative median window fed by its epilogue (counting proof of the two order statistics
on the kernel's own image); and the fused call at the smallest size that takes it by itself."""
import math

import numpy as np
import pytest
import torch

from stein_amd import _lib
from stein_amd.engine import SvgdEngine, untile_distances
from test_gpu_baseline_sizes import check_exact_bandwidth

pytestmark = pytest.mark.gpu


def _inputs(n, d, device, dtype=torch.float32, seed=0):
    T = torch.tensor(np.random.default_rng(seed).normal(size=(n, d)), dtype=torch.float32)
    G = torch.tensor(np.random.default_rng(seed + 1).normal(size=(n, d)), dtype=torch.float32)
    if dtype == torch.bfloat16:
        T, G = T.bfloat16().float(), G.bfloat16().float()
    return T.to(device), G.to(device)


def _fp64_dist(T, rows=None):
    Ta = T.double()
    ra = (Ta * Ta).sum(1)
    Ti = Ta if rows is None else Ta[rows]
    ri = ra if rows is None else ra[rows]
    return ri[:, None] + ra[None, :] - 2.0 * (Ti @ Ta.T)


@pytest.mark.parametrize("n,d,dtype", [(1024, 256, torch.float32), (768, 200, torch.float32), (512, 100, torch.float32),
                                       (640, 33, torch.float32), (1024, 128, torch.bfloat16), (512, 300, torch.bfloat16),
                                       # K beyond the LDS panel (k_distance_panel_deep: K in chunks of 8 / 16 k tiles): two chunks
                                       # with a short last one, three chunks with a one-tile last one, BASELINE config 4's
                                       # d = 2001 (eight chunks), bf16 with two chunks
                                       (1024, 300, torch.float32), (640, 520, torch.float32), (512, 2001, torch.float32),
                                       (512, 600, torch.bfloat16)])
def test_panel_symmetric_matches_tile_kernel_and_fp64(cuda, n, d, dtype):
    T, G = _inputs(n, d, cuda, dtype)
    eng = SvgdEngine(n, d, device=cuda, x3=True, dtype=dtype, small=False)
    st = eng.stages
    Td = T.to(dtype)
    st.rownorms(Td, n, d, eng.rownorm)
    st.x3_prepare(Td, G.to(dtype), n, d, eng.planes)
    out = {}
    for name, kernel in (("tiles", _lib.STAGE_TILES), ("panel", _lib.STAGE_PANEL)):
        eng.dist.fill_(float("nan"))
        st.distance_block(Td, eng.rownorm, n, d, 0, n, eng.dist, eng.ld_dist, symmetric=True, planes=eng.planes, kernel=kernel)
        torch.cuda.synchronize()
        # the stored image: every tile on and above the diagonal is complete, nothing else was touched
        v = eng.dist.view(n // 128, eng.ld_dist // 32, 128, 32)
        for I in range(n // 128):
            assert bool(torch.isfinite(v[I, 4 * I:n // 32]).all()), (name, "missing entries in row tile", I)
            assert bool(torch.isnan(v[I, :4 * I]).all()), (name, "wrote below the diagonal", I)
            dg = v[I, 4 * I:4 * I + 4].permute(1, 0, 2).reshape(128, 128)
            assert torch.equal(dg, dg.T), (name, "diagonal tile not symmetric", I)
        out[name] = untile_distances(eng.dist, n, n, upper=True)
    ref = _fp64_dist(T)
    scale = ref.abs().max().item()
    tol = 4e-6 if dtype == torch.float32 else 2e-5
    for name, M in out.items():
        assert torch.equal(M, M.T)
        assert (M.double() - ref).abs().max().item() <= tol * scale, name
    # the two kernels sum the same products in a different order: equal to rounding of the accumulation
    assert (out["panel"] - out["tiles"]).abs().max().item() <= 2e-6 * scale


@pytest.mark.parametrize("n,d,row0,nl", [(1024, 256, 256, 512), (1536, 96, 1280, 256), (512, 256, 0, 512),
                                         (1024, 400, 256, 512), (1280, 1030, 1152, 128)])   # (the last two: K in chunks)
def test_panel_row_block_matches_fp64(cuda, n, d, row0, nl):
    """non-symmetric row block [row0, row0 + nl) of the n columns, every tile stored"""
    T, G = _inputs(n, d, cuda, seed=3)
    st = SvgdEngine(8, 2, device=cuda).stages
    total, offs, extra = st.workspace_layout(nl, n, d, _lib.FLAG_X3)
    ws = torch.zeros(total, dtype=torch.uint8, device=cuda)
    ld = extra[_lib.WSX_LD_DIST]
    r = ws[offs[_lib.WS_ROWNORM]:offs[_lib.WS_ROWNORM] + 4 * n].view(torch.float32)
    D = ws[offs[_lib.WS_DIST]:offs[_lib.WS_DIST] + nl * ld * 4].view(torch.float32).view(nl, ld)
    planes = ws[offs[_lib.WS_PLANES]:total]
    st.rownorms(T, n, d, r)
    st.x3_prepare(T, G, n, d, planes)
    ref = _fp64_dist(T, torch.arange(row0, row0 + nl, device=cuda))
    mats = {}
    for name, kernel in (("tiles", _lib.STAGE_TILES), ("panel", _lib.STAGE_PANEL)):
        D.fill_(float("nan"))
        st.distance_block(T, r, n, d, row0, nl, D, ld, planes=planes, kernel=kernel)
        torch.cuda.synchronize()
        mats[name] = untile_distances(D, nl, n)
        assert (mats[name].double() - ref).abs().max().item() <= 4e-6 * ref.abs().max().item(), name
    assert (mats["panel"] - mats["tiles"]).abs().max().item() <= 2e-6 * ref.abs().max().item()


@pytest.mark.parametrize("sym,d", [(True, 64), (False, 64), (True, 320), (False, 700)])   # (d > 256: k_distance_panel_deep)
def test_panel_feeds_the_median_window(cuda, sym, d):
    """Window form of the staged calls on one rank: the panel kernel's epilogue counts the entries below the window and
    collects those inside it; tally + pick then deliver the exact order statistics of the kernel's own image."""
    n = 1024
    T, G = _inputs(n, d, cuda, seed=5)
    eng = SvgdEngine(n, d, device=cuda, x3=True, small=False)
    for _ in range(4):      # fused steps give the predictor its history (the particles drift a little per step)
        eng.compute_phi(T, G)
        T = T + 1e-3 * eng.phi
    st, hist, sel, spec = eng.stages, eng.hist, eng.select_state, eng.spec_section
    state = sel.clone()
    h2, med = torch.zeros(1, device=cuda), torch.zeros(1, device=cuda)
    st.rownorms(T, n, d, eng.rownorm)
    st.x3_prepare(T, G, n, d, eng.planes)
    got = {}
    for name, kernel in (("tiles", _lib.STAGE_TILES), ("panel", _lib.STAGE_PANEL)):
        sel.copy_(state)
        st.spec_begin(hist, sel, spec, n * n)
        st.distance_block_spec(T, eng.rownorm, n, d, 0, n, eng.dist, eng.ld_dist, hist[0], sel, spec, planes=eng.planes,
                               kernel=kernel, symmetric=sym)
        st.spec_tally(sel, spec)
        st.spec_pick(sel, spec, n, h2, med)
        torch.cuda.synchronize()
        flags = sel[_lib.SPEC_HIT_OFFSET:_lib.SPEC_SKIP_L0_OFFSET + 4].view(torch.int32).cpu()
        assert int(flags[0]) == 1, (name, "the window missed")
        got[name] = check_exact_bandwidth([eng.dist], sel.view(torch.float32), n, h2.item(), upper=sym)
    assert abs(got["panel"] - got["tiles"]) <= 4e-7 * got["tiles"]


def test_fused_call_takes_the_panel_kernel_and_stays_exact(cuda):
    """n = 8192, d = 40: the smallest block the fused call gives to the panel kernel by itself.  Every step's bandwidth
    is the exact median of that step's own distances (first steps: radix select with the level-0 pass the panel kernel
    asks for; later: the window), equal to the per-tile kernel's to rounding, and so is phi."""
    n, d = 8192, 40
    T, G = _inputs(n, d, cuda, seed=7)
    eng = SvgdEngine(n, d, device=cuda, x3=True)
    ref = SvgdEngine(n, d, device=cuda, x3=True, tile_distance=True)
    hits = 0
    for step in range(5):
        phi = eng.compute_phi(T, G).clone()
        phi_ref = ref.compute_phi(T, G)
        torch.cuda.synchronize()
        h2 = check_exact_bandwidth([eng.dist], eng.select_state.view(torch.float32), n, eng.h2.item(), upper=True)
        assert abs(h2 - ref.h2.item()) <= 4e-7 * h2
        assert ((phi - phi_ref).norm() / phi_ref.norm()).item() <= 2e-6
        M = eng.dist_matrix()
        assert torch.equal(M, M.T)
        hits = eng.window_stats()[1]
        T = T + 1e-3 * phi
    assert hits >= 2
    assert not math.isnan(eng.h2.item())


@pytest.mark.parametrize("n,d,dtype,sym", [(2048, 256, torch.float32, True), (2048, 256, torch.float32, False),
                                           (1024, 2001, torch.float32, True), (2048, 256, torch.bfloat16, True),
                                           (1536, 600, torch.bfloat16, False)])
def test_panel_kernels_are_deterministic_under_memory_pressure(cuda, n, d, dtype, sym):
    """The panel kernels stream their operand through inline-asm loads with hand-counted s_waitcnt vmcnt: a wait that is
    one operation too loose reads a register before its load has landed -- only sometimes, depending on timing.  So the
    same launch is repeated while a second stream hammers HBM and the L2 with copies (latencies move around), and every
    repetition must reproduce the first image bit for bit, which itself matches fp64."""
    T, G = _inputs(n, d, cuda, dtype, seed=11)
    eng = SvgdEngine(n, d, device=cuda, x3=True, dtype=dtype, small=False)
    st = eng.stages
    Td = T.to(dtype)
    st.rownorms(Td, n, d, eng.rownorm)
    st.x3_prepare(Td, G.to(dtype), n, d, eng.planes)

    def run():
        eng.dist.fill_(float("nan"))
        st.distance_block(Td, eng.rownorm, n, d, 0, n, eng.dist, eng.ld_dist, symmetric=sym, planes=eng.planes,
                          kernel=_lib.STAGE_PANEL)

    run()
    torch.cuda.synchronize()
    first = eng.dist.clone()
    M = untile_distances(first, n, n, upper=sym)
    ref = _fp64_dist(T)
    assert (M.double() - ref).abs().max().item() <= (4e-6 if dtype == torch.float32 else 2e-5) * ref.abs().max().item()
    side = torch.cuda.Stream(device=cuda)
    a = torch.empty(64 << 20, dtype=torch.float32, device=cuda)      # 256 MB each: beyond the caches
    b = torch.empty_like(a)
    for rep in range(12):
        with torch.cuda.stream(side):
            for _ in range(1 + rep % 4):
                b.copy_(a)
        run()
        torch.cuda.synchronize()
        same = (eng.dist == first) | (eng.dist.isnan() & first.isnan())
        assert bool(same.all()), "repetition %d differs in %d entries" % (rep, int((~same).sum()))


def _level0_counts(M, sym):
    """level-0 digits (top 11 bits of the order-preserving key, stein_common.h: f32_key) of a row-major distance matrix"""
    bits = M.contiguous().view(torch.int32).to(torch.int64) & 0xffffffff
    key = torch.where(bits >= (1 << 31), (~bits) & 0xffffffff, bits | (1 << 31))
    return torch.bincount((key >> 21).flatten(), minlength=_lib.HIST_BINS)


@pytest.mark.parametrize("n,d,sym,row0,nl,spread", [(1024, 64, True, 0, 1024, False), (1024, 64, False, 256, 512, False),
                                                    (768, 320, True, 0, 768, False), (640, 700, False, 128, 384, False),
                                                    (1024, 48, True, 0, 1024, True), (1024, 16, False, 0, 512, True)])
def test_panel_epilogue_takes_the_level0_histogram(cuda, n, d, sym, row0, nl, spread):
    """A step without a window: the panel kernels (d <= 256: k_distance_panel, beyond: k_distance_panel_deep) count the
    level-0 radix-select digits in their epilogue (per-lane 8-bit slots around the wave's first digit, LDS histogram per
    workgroup).  The histogram must equal a direct count over the kernel's own distance image -- also when the distances
    span many digits (`spread`: particle scales from 1e-3 to 1e3, so most entries miss the slots and take the direct path) and
    when a few of them are negative (identical particles: r_i + r_j - 2 <t_i, t_j> cancels to +-tiny)."""
    T, G = _inputs(n, d, cuda, seed=13)
    if spread:
        T = T * torch.logspace(-3, 3, n, device=cuda)[:, None]
        T[5] = T[4]; T[n - 2] = T[n - 1]                                # coincident pairs: distances around +-0
    st = SvgdEngine(8, 2, device=cuda).stages
    total, offs, extra = st.workspace_layout(nl, n, d, _lib.FLAG_X3)
    ws = torch.zeros(total, dtype=torch.uint8, device=cuda)
    ld = extra[_lib.WSX_LD_DIST]
    r = ws[offs[_lib.WS_ROWNORM]:offs[_lib.WS_ROWNORM] + 4 * n].view(torch.float32)
    D = ws[offs[_lib.WS_DIST]:offs[_lib.WS_DIST] + nl * ld * 4].view(torch.float32).view(nl, ld)
    hist = ws[offs[_lib.WS_HIST]:offs[_lib.WS_HIST] + _lib.HIST_LEVELS * 2 * _lib.HIST_BINS * 8].view(torch.int64)
    hist = hist.view(_lib.HIST_LEVELS, 2, _lib.HIST_BINS)
    sel = ws[offs[_lib.WS_SELECT]:offs[_lib.WS_SELECT] + 128]
    planes = ws[offs[_lib.WS_PLANES]:total]
    st.rownorms(T, n, d, r)
    st.x3_prepare(T, G, n, d, planes)
    for name, kernel in (("tiles", _lib.STAGE_TILES), ("panel", _lib.STAGE_PANEL)):
        st.median_begin(hist, sel, n * n)
        st.distance_block(T, r, n, d, row0, nl, D, ld, hist0=hist[0], symmetric=sym, planes=planes, kernel=kernel)
        torch.cuda.synchronize()
        M = untile_distances(D, nl, n, upper=sym)
        want = _level0_counts(M, sym)
        got = hist[0, 0]
        assert int(got.sum()) == nl * n, (name, int(got.sum()))
        assert torch.equal(got, want), (name, (got - want).nonzero().flatten().tolist()[:8])
        assert int(hist[0, 1].abs().sum()) == 0 and int(hist[1:].abs().sum()) == 0
