"""GPU: the stein.{kernels,samplers,optimizers,utilities} API surface of stein_amd against the golden vectors
produced by the reference's own NumPy code, and against the oracle.  Every call goes through the C ABI."""
import numpy as np
import pytest
import torch

from oracle import svgd_oracle as orc

pytestmark = pytest.mark.gpu


class Var:
    """Stand-in for a TF variable: .name and .get_shape().as_list(), all the converters use."""

    def __init__(self, name, shape):
        self.name, self._s = name, list(shape)

    def get_shape(self):
        s = self._s

        class _S:
            def as_list(self_inner):
                return list(s)
        return _S()


# ---- optimizers (a9, a10) --------------------------------------------------------------------------------
def test_g1_optimizers_numpy_api(cuda, golden):
    from stein_amd.optimizers import AdagradGradientDescent, AdamGradientDescent
    g = golden("g1_optimizers.npz")
    ada, adam = AdagradGradientDescent(0.1), AdamGradientDescent(0.1, decay=0.999)
    for t, p in enumerate(g["phis"]):
        p32 = p.astype(np.float32).astype(np.float64)       # the device consumes phi as fp32
        s_ada, s_adam = ada.update(p), adam.update(p)
        assert s_ada.dtype == np.float64 and s_ada.shape == p.shape
        np.testing.assert_allclose(s_ada, g["adagrad_steps"][t], rtol=1e-5, atol=1e-12)
        np.testing.assert_allclose(s_adam, g["adam_steps"][t], rtol=1e-5, atol=1e-12)
        np.testing.assert_allclose(ada.hist.cpu().numpy(), g["adagrad_hist"][t], rtol=1e-6)
        np.testing.assert_allclose(adam.mu.cpu().numpy(), g["adam_mu"][t], rtol=1e-6, atol=2e-7)  # phi is consumed as fp32
        np.testing.assert_allclose(adam.nu.cpu().numpy(), g["adam_nu"][t], rtol=1e-6)
        assert adam.learning_rate == pytest.approx(g["adam_lr"][t], rel=1e-15)
        del p32
    assert ada.learning_rate == 0.1 and ada.decay == 1.0 and ada.n_iters == 5 and adam.n_iters == 5


def test_optimizer_exact_in_fp64_given_fp32_representable_phi(cuda):
    """With phi exactly representable in fp32 the fp64 state path reproduces the reference arithmetic to rounding."""
    from stein_amd.optimizers import AdagradGradientDescent, AdamGradientDescent
    rng = np.random.default_rng(3)
    ada, adam = AdagradGradientDescent(0.05, alpha=0.8), AdamGradientDescent(0.05, decay=0.9, beta_1=0.8, beta_2=0.99)
    oa, om = orc.AdagradState(0.05, alpha=0.8), orc.AdamState(0.05, decay=0.9, beta_1=0.8, beta_2=0.99)
    for _ in range(4):
        p = rng.normal(size=(33, 7)).astype(np.float32).astype(np.float64)
        np.testing.assert_allclose(ada.update(p), oa.update(p), rtol=1e-13)
        np.testing.assert_allclose(adam.update(p), om.update(p), rtol=1e-13)


def test_optimizer_device_tensor_api_and_fused_clip(cuda):
    from stein_amd.optimizers import AdagradGradientDescent
    rng = np.random.default_rng(5)
    phi = rng.normal(size=(64, 16)) * 3.0
    theta0 = rng.normal(size=(64, 16))
    sq = float((phi.astype(np.float32).astype(np.float64) ** 2).sum())
    assert np.sqrt(sq) > 10
    gd = AdagradGradientDescent(0.01)
    theta = torch.tensor(theta0, dtype=torch.float64, device=cuda)
    gd.apply_(theta, torch.tensor(phi, dtype=torch.float32, device=cuda), torch.tensor([sq], dtype=torch.float64, device=cuda))
    ref = orc.AdagradState(0.01)
    p32 = phi.astype(np.float32).astype(np.float64)
    expect = theta0 + ref.update(p32 * orc.clip_scale(sq))
    np.testing.assert_allclose(theta.cpu().numpy(), expect, rtol=1e-12)
    # float32 tensors in -> float32 tensors out
    gd32 = AdagradGradientDescent(0.01)
    step = gd32.update(torch.tensor(phi, dtype=torch.float32, device=cuda))
    assert step.dtype == torch.float32 and step.is_cuda
    np.testing.assert_allclose(step.cpu().numpy(), orc.AdagradState(0.01).update(p32), rtol=2e-6)


# ---- kernel (a1-a6) -------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,d", [(7, 3), (8, 5), (100, 10), (257, 33)])
def test_g5_kernel_and_grad(cuda, golden, n, d):
    from stein_amd.kernels import SquaredExponentialKernel
    g = golden("g5_kernel_restated.npz")
    T = g[f"T_{n}x{d}"]
    kern = SquaredExponentialKernel(n, None)
    K, dK = kern.kernel_and_grad(T)
    assert K.shape == (n, n) and dK.shape == (n, d) and K.dtype == np.float32 and dK.dtype == np.float32
    ref = g[f"dK64_{n}x{d}"]
    assert np.linalg.norm(dK - ref) <= 1e-5 * np.linalg.norm(ref)
    assert abs(kern.bandwidth ** 2 - float(g[f"h264_{n}x{d}"])) <= 3e-6 * float(g[f"h264_{n}x{d}"])
    if n <= 100:
        assert np.abs(K - g[f"K64_{n}x{d}"]).max() <= 1e-5
    with pytest.raises(ValueError):
        kern.kernel_and_grad(T[:-1])
    D = kern.squared_distances(T)
    assert np.abs(D - orc.pairwise_sq_dists(T, np.float64)).max() <= 4e-6 * np.abs(D).max()


def test_compute_median_utility(cuda):
    from stein_amd.utilities import compute_median
    rng = np.random.default_rng(8)
    for shape in [(7, 7), (8, 8), (33, 21), (1, 9), (300, 1100)]:
        D = rng.normal(size=shape).astype(np.float32)
        assert compute_median(D) == orc.median_all(D)
    D = np.round(rng.normal(size=(64, 64)) * 2).astype(np.float32)      # ties, signed zeros
    assert compute_median(D) == orc.median_all(D)
    Dt = torch.tensor(D, device=cuda)
    assert compute_median(Dt).item() == orc.median_all(D)


# ---- converters (a12) --------------------------------------------------------------------------------------
def test_g4_converters_numpy_and_device(cuda, golden):
    from stein_amd.utilities import convert_array_to_dictionary, convert_dictionary_to_array
    g = golden("g4_converters.npz")
    vz, va, vm = Var("model/zeta:0", [3, 1]), Var("model/alpha:0", []), Var("model/mid:0", [2, 2])
    d = {vz: g["zeta"], va: g["alpha"], vm: g["mid"]}
    arr, access = convert_dictionary_to_array(d)
    assert arr.dtype == np.float64
    np.testing.assert_array_equal(arr, g["array"])
    assert access[va] == (0, 1) and access[vm] == (1, 5) and access[vz] == (5, 8)
    back = convert_array_to_dictionary(arr, access)
    for v in d:
        np.testing.assert_array_equal(back[v], d[v])
    # device tensors: packing keeps the order, unpacking returns views of the packed matrix
    dt = {v: torch.tensor(x, device=cuda) for v, x in d.items()}
    arr_t, access_t = convert_dictionary_to_array(dt)
    np.testing.assert_array_equal(arr_t.cpu().numpy(), g["array"])
    views = convert_array_to_dictionary(arr_t, access_t)
    views[va].add_(1.0)
    assert torch.equal(arr_t[:, 0], torch.tensor(g["alpha"], device=cuda) + 1.0)


# ---- sampler (a7, a8, a11) -----------------------------------------------------------------------------------
@pytest.mark.parametrize("n,d", [(7, 3), (8, 5), (100, 10)])
def test_g2_compute_phi(cuda, golden, n, d):
    from stein_amd.samplers import SteinSampler
    from stein_amd.optimizers import AdamGradientDescent
    g = golden("g2_compute_phi.npz")
    T, G = g[f"T_{n}x{d}"], g[f"G_{n}x{d}"]
    s = SteinSampler(n, None, AdamGradientDescent(), theta={Var("model/w:0", [d]): T.copy()})
    phi = s.compute_phi(T, G)
    assert phi.dtype == np.float64 and phi.shape == (n, d)
    ref = g[f"phi_{n}x{d}"]
    assert np.linalg.norm(phi - ref) <= 1e-5 * np.linalg.norm(ref)
    assert np.all(np.abs(phi - ref) <= 1e-5 * np.abs(ref).max() + 1e-5 * np.abs(ref))


@pytest.mark.parametrize("tag", ["noclip", "clip"])
@pytest.mark.parametrize("oname", ["adagrad", "adam"])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_g3_update_particles(cuda, golden, tag, oname, dtype):
    from stein_amd.samplers import SteinSampler
    from stein_amd.optimizers import AdagradGradientDescent, AdamGradientDescent
    g = golden("g3_update_particles.npz")
    key = f"{tag}_{oname}"
    gd = AdagradGradientDescent(0.05) if oname == "adagrad" else AdamGradientDescent(0.05, decay=0.99)
    v = Var("model/w:0", [10])
    s = SteinSampler(100, None, gd, theta={v: g[key + "_T0"].copy()}, dtype=dtype)
    for t, G in enumerate(g[key + "_G"]):
        s.update_particles(G)
        ref = g[key + "_theta"][t]
        got = s.samples
        # a step is at most ~lr per coordinate; 1e-5 of the particle scale is the north-star tolerance
        assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max(), (t, np.abs(got - ref).max())
        assert torch.equal(s.theta[v], s.theta_matrix.reshape(100, 10))
    assert gd.n_iters == 3


def test_sampler_train_on_batch_autograd_score_and_kat(cuda, golden):
    """Linear-regression KAT (examples/linear_regression/main.py:20-48): SVGD particles converge to the
    closed-form posterior N(0.383949, 0.031917^2).  The score comes from autograd of a batched log_p."""
    from stein_amd.samplers import SteinSampler
    from stein_amd.optimizers import AdamGradientDescent
    g = golden("g6_linear_regression.npz")
    X = torch.tensor(g["X"], dtype=torch.float32, device=cuda)
    y = torch.tensor(g["y"], dtype=torch.float32, device=cuda)

    def log_p(theta, feed):
        w = theta["model/w:0"]                                   # [n, 1, 1]
        resid = feed["X"] @ w[:, :, 0].T - feed["y"][:, None]    # [samples, n]
        return -0.5 * (resid ** 2).sum(0) - 0.5 * (w ** 2).sum((1, 2))

    torch.manual_seed(0)
    s = SteinSampler(50, log_p, AdamGradientDescent(learning_rate=1e-1), model_vars={"model/w:0": [1, 1]}, seed=0)
    assert abs(s.samples.std() - 0.01) < 0.005                   # N(0, 0.01^2) init (abstract_stein_sampler.py:72)
    score = s.score_matrix({"X": X, "y": y}).cpu().numpy()
    w0 = s.samples
    expect = (g["X"].T @ (g["y"][None, :] - w0 @ g["X"].T).T).T - w0
    np.testing.assert_allclose(score, expect, rtol=2e-4, atol=1e-2)
    for _ in range(300):
        s.train_on_batch({"X": X, "y": y})
    est = s.samples
    assert abs(est.mean() - float(g["post_mean"][0])) < 0.01
    assert 0.4 * float(g["post_std"][0]) < est.std() < 1.6 * float(g["post_std"][0])
    post = s.function_posterior(lambda th, feed: th["model/w:0"][:, 0, :] * 2.0, None)
    assert post.shape == (50, 1)
    assert s.function_posterior(lambda th, feed: th["model/w:0"][:, 0, :], None, axis=0).shape == (1,)


def test_sampler_state_roundtrip(cuda):
    from stein_amd.samplers import SteinSampler
    from stein_amd.optimizers import AdamGradientDescent
    rng = np.random.default_rng(1)
    T0, G = rng.normal(size=(64, 12)), rng.normal(size=(64, 12))
    a = SteinSampler(64, None, AdamGradientDescent(0.05, decay=0.99), theta=T0.copy())
    a.update_particles(G)
    state = a.state_dict()
    b = SteinSampler(64, None, AdamGradientDescent(), theta=np.zeros_like(T0))
    b.load_state_dict(state)
    a.update_particles(G)
    b.update_particles(G)
    assert torch.equal(a.theta_matrix, b.theta_matrix)
    assert b.gd.n_iters == 2 and b.gd.learning_rate == pytest.approx(0.05 * 0.99 ** 2)


# ---- the reference's duck-typed seams (abstract_stein_sampler.py:103, :126) ---------------------------------------------
class _UpdateOnlyOptimizer:
    """What the reference's sampler needs of `gd`: update(phi) -> step, nothing else.  Wraps the oracle's restatement
    of adagrad_gradient_descent.py:37-44 / adam_gradient_descent.py:45-58 (pinned bit for bit by G1)."""

    def __init__(self, state):
        self._s = state

    def update(self, phi):
        return self._s.update(phi)


class _RestatedKernel:
    """What the reference's sampler needs of `kernel`: kernel_and_grad(theta) -> (K, dK); the stand-in the golden
    generator used (tests/golden/make_golden.py: RestatedKernel)."""
    calls = 0

    def kernel_and_grad(self, theta):
        from oracle import svgd_oracle as orc
        _RestatedKernel.calls += 1
        return orc.kernel_and_grad(theta, np.float32)


@pytest.mark.parametrize("tag", ["noclip", "clip"])
@pytest.mark.parametrize("oname", ["adagrad", "adam"])
@pytest.mark.parametrize("standin", ["both", "gd", "kernel"])
def test_g3_through_the_reference_seams(cuda, golden, tag, oname, standin):
    """SteinSampler driven with the golden generator's stand-ins: an optimizer object exposing only .update and / or a
    kernel object exposing only .kernel_and_grad.  With both, the reference's own NumPy lines run on the reference's own
    stand-ins: G3 bit for bit.  With one, the HIP engine supplies phi (or the HIP optimizer the step): 1e-5."""
    from oracle import svgd_oracle as orc
    from stein_amd.samplers import SteinSampler
    from stein_amd.optimizers import AdagradGradientDescent, AdamGradientDescent
    g = golden("g3_update_particles.npz")
    key = f"{tag}_{oname}"
    if standin in ("both", "gd"):
        gd = _UpdateOnlyOptimizer(orc.AdagradState(learning_rate=0.05) if oname == "adagrad"
                                  else orc.AdamState(learning_rate=0.05, decay=0.99))
        assert not hasattr(gd, "apply_")
    else:
        gd = AdagradGradientDescent(0.05) if oname == "adagrad" else AdamGradientDescent(0.05, decay=0.99)
    v = Var("model/w:0", [10])
    s = SteinSampler(100, None, gd, theta={v: g[key + "_T0"].copy()}, dtype=torch.float64)
    calls0 = _RestatedKernel.calls
    if standin in ("both", "kernel"):
        s.kernel = _RestatedKernel()
    for t, G in enumerate(g[key + "_G"]):
        s.update_particles(G)
        ref = g[key + "_theta"][t]
        got = s.samples
        if standin == "both":
            assert np.array_equal(got, ref), (t, np.abs(got - ref).max())
        else:
            assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max(), (t, np.abs(got - ref).max())
        assert torch.equal(s.theta[v], s.theta_matrix.reshape(100, 10))
    assert _RestatedKernel.calls - calls0 == (3 if standin in ("both", "kernel") else 0)
    # compute_phi goes through the same seam (abstract_stein_sampler.py:103-105)
    if standin in ("both", "kernel"):
        T, G = g[key + "_T0"], g[key + "_G"][0]
        K, dK = orc.kernel_and_grad(T, np.float32)
        assert np.array_equal(s.compute_phi(T, G), (K.dot(G) + dK) / 100)


def test_function_posterior_values_and_axis(cuda):
    """function_posterior against the reference's per-particle loop (abstract_stein_sampler.py:157-168: np.ravel of
    func under particle i, stacked to [n, out]; dist.mean(axis=axis) when axis is given), on the logistic example's
    readout: `logits = X w` (examples/logistic_regression/main.py:39) as evaluate() uses it (:52-61)."""
    from stein_amd.samplers import SteinSampler
    from stein_amd.optimizers import AdamGradientDescent
    rng = np.random.default_rng(12)
    n, n_feats, n_test = 37, 6, 11
    W = rng.normal(size=(n, n_feats, 1))
    log_alpha = rng.normal(size=(n,))
    X = rng.normal(size=(n_test, n_feats))
    y = (rng.uniform(size=(n_test, 1)) > 0.5).astype(np.float64)
    vw, va = Var("model/Variable:0", [n_feats, 1]), Var("model/Variable_1:0", [])
    s = SteinSampler(n, None, AdamGradientDescent(), theta={vw: W.copy(), va: log_alpha.copy()}, dtype=torch.float64)
    Xd = torch.tensor(X, dtype=torch.float64, device=cuda)

    def logits(theta, feed):                       # batched over particles: [n, n_test, 1]
        return feed["X"] @ theta[vw]

    # the reference's loop, restated: one evaluation per particle, raveled, stacked
    dist = np.array([np.ravel(X @ W[i]) for i in range(n)])
    got = s.function_posterior(logits, {"X": Xd})
    assert got.shape == (n, n_test) and got.dtype == np.float64
    np.testing.assert_allclose(got, dist, rtol=1e-12, atol=1e-12)
    for axis in (0, 1):
        np.testing.assert_allclose(s.function_posterior(logits, {"X": Xd}, axis=axis), dist.mean(axis=axis), rtol=1e-12, atol=1e-12)
    # evaluate() of the example: average the logits over the particles, threshold, compare with the labels
    acc = np.mean((s.function_posterior(logits, {"X": Xd}).mean(axis=0) > 0.) == y.ravel())
    assert acc == np.mean((dist.mean(axis=0) > 0.) == y.ravel())
    # a func with a matrix-valued output per particle is raveled per particle (np.ravel), a scalar one gives [n, 1]
    both = s.function_posterior(lambda th, feed: torch.stack([feed["X"] @ th[vw], -(feed["X"] @ th[vw])], dim=1), {"X": Xd})
    np.testing.assert_allclose(both, np.array([np.ravel(np.stack([X @ W[i], -(X @ W[i])])) for i in range(n)]), rtol=1e-12)
    sc = s.function_posterior(lambda th, feed: th[va].exp(), None)
    assert sc.shape == (n, 1)
    np.testing.assert_allclose(sc[:, 0], np.exp(log_alpha), rtol=1e-12)
    assert s.function_posterior(lambda th, feed: th[va].exp(), None, axis=0).shape == (1,)
