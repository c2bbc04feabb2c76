"""GPU: the speculative median window of the fused single-rank call (stein_common.h: SpecState).

The fused call predicts this step's median from the two previous steps, counts the distances below a narrow window around
the prediction and selects the median inside the window; when the prediction fails it falls back to the radix-select
passes.  Either way the bandwidth must be bit-identical to the staged calls (which always take the radix-select
passes) on the same particles, at every step, and phi must agree to the last bit as well."""
import numpy as np
import pytest
import torch

from stein_amd import _lib
from stein_amd.engine import SvgdEngine

pytestmark = pytest.mark.gpu


def _spec_state(eng):
    _, offs, _ = _lib.workspace_layout(eng.n_local, eng.n, eng.d, flags=eng.flags)
    o = offs[_lib.WS_SELECT] + 64
    u = eng.ws[o:o + 64].cpu().numpy().view(np.uint32)
    return dict(magic=int(u[0]), center=int(u[1]), halfwidth=int(u[2]), lo_key=int(u[3]), width=int(u[4]),
                count=int(u[5]), overflow=int(u[6]), hit=int(u[7]))


def _same(a, b):
    """elementwise equal, NaN == NaN (identical particles give h2 = 0 and NaN phi on every path, as in the reference)"""
    return bool(((a == b) | (a.isnan() & b.isnan())).all())


def _staged(eng, T, G):
    return eng.compute_phi(T, G, mark=lambda label: None)


@pytest.mark.parametrize("n,d,x3,dtype", [(2048, 32, True, torch.float32), (1001, 17, True, torch.float32),
                                          (1536, 64, False, torch.float32), (1280, 48, True, torch.bfloat16)])
def test_window_hits_and_matches_radix_select(cuda, n, d, x3, dtype):
    g = torch.Generator(device="cpu").manual_seed(n + d)
    T = torch.randn(n, d, generator=g).to(cuda)
    G = torch.randn(n, d, generator=g).to(cuda).to(dtype)
    V = (1e-3 if dtype == torch.float32 else 2e-2) * torch.randn(n, d, generator=g).to(cuda)   # a steady drift
    fused, ref = SvgdEngine(n, d, device=cuda, x3=x3, dtype=dtype), SvgdEngine(n, d, device=cuda, x3=x3, dtype=dtype)
    hits = []
    for step in range(10):
        Tq = T.to(dtype)
        phi = fused.compute_phi(Tq, G).clone()
        h2 = fused.h2.clone()
        phi_ref = _staged(ref, Tq, G)
        torch.cuda.synchronize()
        assert torch.equal(h2, ref.h2), "step %d: h2 %r vs %r (%r)" % (step, float(h2), float(ref.h2), _spec_state(fused))
        assert torch.equal(phi, phi_ref)
        assert torch.equal(fused.sqnorm, ref.sqnorm)
        st = _spec_state(fused)
        hits.append(st["hit"])
        T = T + V * (1.0 + 0.05 * step)                          # slightly accelerating
    assert hits[0] == 0                                           # no history yet: the radix passes ran
    assert sum(hits[2:]) >= 6, hits                               # from the third step on the window carries the median


def test_jump_and_degenerate_particles_fall_back(cuda):
    n, d = 1024, 16
    g = torch.Generator(device="cpu").manual_seed(5)
    T = torch.randn(n, d, generator=g).to(cuda)
    G = torch.randn(n, d, generator=g).to(cuda)
    fused, ref = SvgdEngine(n, d, device=cuda), SvgdEngine(n, d, device=cuda)

    def check(T, expect_hit=None):
        phi = fused.compute_phi(T, G).clone()
        phi_ref = _staged(ref, T, G)
        torch.cuda.synchronize()
        assert _same(fused.h2, ref.h2) and _same(phi, phi_ref)
        st = _spec_state(fused)
        if expect_hit is not None:
            assert st["hit"] == expect_hit, st
        return st

    for k in range(4):
        check(T * (1.0 + 1e-4 * k))
    check(T * 3.0, expect_hit=0)                                  # the median jumps by 9x: outside any window
    check(T * 3.0003)
    st = check(T * 3.0006)
    # all particles identical: every distance is 0, the window around the prediction cannot separate them
    same = T[:1].expand(n, d).contiguous()
    check(same, expect_hit=0)
    check(same)
    check(same)
    # two clusters: the two middle order statistics straddle two different values (n*n even)
    two = torch.cat([T[:1].expand(n // 2, d), (T[:1] + 1.0).expand(n // 2, d)]).contiguous()
    for _ in range(4):
        check(two)
    # and back to a spread cloud
    for k in range(4):
        st = check(T * (1.0 + 1e-4 * k))
    assert st["hit"] == 1


def test_timing_flag_reports_every_stage(cuda):
    n, d = 512, 32
    T = torch.randn(n, d, device=cuda)
    G = torch.randn(n, d, device=cuda)
    eng = SvgdEngine(n, d, device=cuda)
    _lib.timing_reserve(3)
    for _ in range(4):                                            # the fourth call finds no free slot and is not timed
        eng.compute_phi(T, G, timing=True)
    calls = _lib.timing_read(8)
    assert len(calls) == 3
    for c in calls:
        assert set(c) == set(_lib.T_STAGES) and all(v > 0.0 for v in c.values())


@pytest.mark.parametrize("seed", [0, 1])
def test_last_workgroup_tickets_match_separate_launches(cuda, seed):
    """The fused call lets the last workgroup of k_colmax write the scales and the last workgroup of the level-2 pass do
    the final resolve (stein_common.h: last_workgroup_out); the staged calls launch k_make_scales / k_resolve instead.
    Random rescaling makes the window miss, so the chained select really runs.  Both must agree bit for bit."""
    rng = np.random.default_rng(seed)
    for n, d in [(129, 17), (333, 130), (777, 64), (1500, 3), (2048, 256), (3000, 128)]:
        fused = SvgdEngine(n, d, device=cuda, small=False)
        staged = SvgdEngine(n, d, device=cuda, small=False)
        for step in range(6):
            s = float(rng.uniform(0.05, 20.0)) if step % 3 != 2 else 1.0
            T = torch.randn(n, d, device=cuda) * s
            G = torch.randn(n, d, device=cuda) * float(rng.uniform(0.01, 100.0))
            pf = fused.compute_phi(T, G).clone()
            ps = _staged(staged, T, G).clone()
            assert float(fused.h2) == float(staged.h2), (n, d, step)
            assert torch.equal(pf, ps), (n, d, step)
            assert float(fused.sqnorm) == float(staged.sqnorm), (n, d, step)


@pytest.mark.parametrize("n,d,dtype", [(513, 5, torch.float32), (1000, 33, torch.float32), (2049, 64, torch.bfloat16),
                                        (4096, 128, torch.bfloat16), (4224, 9, torch.float32)])   # (the last: 2048 virtual workgroups)
def test_one_launch_select_without_the_window(cuda, n, d, dtype):
    """window=False, n > 512: every step takes the one-launch chained radix select (k_hist_all: all levels in one
    launch, workgroups meeting at in-launch barriers, the last arrival resolving a level for all of them; the level-0 counts
    come from the distance kernel or, with skip_l0 clear, from the launch itself).  Bandwidth, phi and |phi|^2 equal the
    staged calls' (separate k_hist / k_resolve launches) bit for bit at every step, for odd and even n^2."""
    g = torch.Generator(device="cpu").manual_seed(7 * n + d)
    fused = SvgdEngine(n, d, device=cuda, dtype=dtype, window=False, small=False)
    staged = SvgdEngine(n, d, device=cuda, dtype=dtype, small=False)
    T = torch.randn(n, d, generator=g).to(cuda)
    for step in range(5):
        G = torch.randn(n, d, generator=g).to(cuda).to(dtype)
        Tq = (T * (1.0 + 0.7 * step)).to(dtype)
        pf = fused.compute_phi(Tq, G).clone()
        ps = _staged(staged, Tq, G).clone()
        torch.cuda.synchronize()
        assert float(fused.h2) == float(staged.h2) and float(fused.h2) > 0.0, (step, float(fused.h2), float(staged.h2))
        assert torch.equal(pf, ps) and float(fused.sqnorm) == float(staged.sqnorm), step
        assert _spec_state(fused)["hit"] == 0


@pytest.mark.parametrize("grid,n", [(1, 2304), (3, 2304), (40, 2304), (4096, 2304), (7, 4608), (3000, 4608)])
def test_one_launch_select_needs_no_co_residency(cuda, grid, n):
    """k_hist_all's level barriers never wait for a workgroup that has not started: the work of a level is cut into 512
    virtual workgroups that the running workgroups DRAW (FuseState::draw / done / gen).  Forced here: the launch gets 1, 3
    or 40 workgroups (far fewer than the 512 virtual ones: the few present take them all over) or 4096 (more than the chip
    holds at once: the late ones find every counter exhausted and fall through).  The median -- hence bandwidth, phi and
    |phi|^2 -- equals the staged calls' bit for bit every time, and no error is raised."""
    d = 24                                                           # n = 2304: 512 virtual workgroups; 4608: 2048
    g = torch.Generator(device="cpu").manual_seed(grid)
    fused = SvgdEngine(n, d, device=cuda, window=False, small=False)
    staged = SvgdEngine(n, d, device=cuda, small=False)
    _lib.call("stein_debug_hist_all_grid", grid)
    try:
        for step in range(3):
            T = (torch.randn(n, d, generator=g) * (1.0 + 2.0 * step)).to(cuda)
            G = torch.randn(n, d, generator=g).to(cuda)
            pf = fused.compute_phi(T, G).clone()
            ps = _staged(staged, T, G).clone()
            torch.cuda.synchronize()
            assert float(fused.h2) == float(staged.h2) and float(fused.h2) > 0.0, (grid, step)
            assert torch.equal(pf, ps) and float(fused.sqnorm) == float(staged.sqnorm), (grid, step)
    finally:
        _lib.call("stein_debug_hist_all_grid", 0)
    _lib.call("stein_take_device_error")                             # nothing gave up


def test_device_error_word_is_reported_by_the_next_call(cuda):
    """A kernel that gives up raises the device's error word (page-locked host memory); the next fused call or optimizer
    apply on that device returns STEIN_E_HIP once instead of queueing more work on poisoned state, then things go on."""
    from stein_amd.optimizers import AdagradGradientDescent
    n, d = 640, 8
    T, G = torch.randn(n, d, device=cuda), torch.randn(n, d, device=cuda)
    eng = SvgdEngine(n, d, device=cuda)
    eng.compute_phi(T, G)
    _lib.call("stein_debug_raise_device_error")
    with pytest.raises(_lib.SteinHipError, match="k_hist_all"):
        eng.compute_phi(T, G)
    phi = eng.compute_phi(T, G)                                      # reported once; the word is lowered again
    _lib.call("stein_debug_raise_device_error")
    gd = AdagradGradientDescent(learning_rate=1e-3)
    with pytest.raises(_lib.SteinHipError, match="k_hist_all"):
        gd.apply_(T, phi, eng.sqnorm)
    gd.apply_(T, phi, eng.sqnorm)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(T).all())


def test_window_half_width_follows_the_documented_rule(cuda):
    """The predictor on the device (spec_update_dev, steinhip.hip): the next half-width is max(4 * |this step's prediction
    error| + 48, 3/4 of the last half-width that was itself earned from an error) -- the 4096-key window of a predictor
    without a velocity does not count -- capped at 32767.  Replayed here from the state words of consecutive steps."""
    n, d = 2048, 32
    g = torch.Generator(device="cpu").manual_seed(5)
    T = torch.randn(n, d, generator=g).to(cuda)
    G = torch.randn(n, d, generator=g).to(cuda)
    eng = SvgdEngine(n, d, device=cuda)
    _, offs, _ = _lib.workspace_layout(eng.n_local, eng.n, eng.d, flags=eng.flags)
    o = offs[_lib.WS_SELECT] + 64

    def words():
        torch.cuda.synchronize()
        return eng.ws[o:o + 64].cpu().numpy().view(np.uint32).astype(np.int64)

    prev, checked, shrank_by_floor = None, 0, 0
    for step in range(40):
        eng.compute_phi(T, G)
        u = words()   # [1] centre of the NEXT window, [2] its half-width, [4] this step's width, [8] earned half-width, [12] this step's key
        if prev is not None and prev[0] == 0x5EED0002 and u[4] != 0:      # the window of this step was a real prediction
            err = abs(int(u[12]) - int(prev[1]))
            want = 32767 if err > 32767 // 4 else 4 * err + 48
            want = max(want, int(prev[8]) - int(prev[8]) // 4)
            if not (u[7] and u[5] > ((1 << 21) - 2048) // 2):              # (the "keep the buffer small" halving: not at this size)
                assert int(u[2]) == min(want, 32767), (step, err, int(prev[8]), int(u[2]))
                assert int(u[8]) == min(want, 32767)
                checked += 1
                shrank_by_floor += int(want > 4 * err + 48)
        prev = u
        T = T + 1e-2 * torch.randn(n, d, generator=g).to(cuda) + 2e-3     # a noisy drift: the errors jump around
    assert checked >= 30 and shrank_by_floor >= 3, (checked, shrank_by_floor)
