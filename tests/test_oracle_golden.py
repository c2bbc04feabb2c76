"""CPU: the oracle against the golden vectors produced by the reference's own NumPy code
(tests/golden/make_golden.py), plus the independent checks that stand in for the TF-side pinning."""
import numpy as np
import pytest

from oracle import svgd_oracle as orc
from oracle import staged_model as sm


class Var:
    def __init__(self, name, shape):
        self.name, self._s = name, list(shape)

    def get_shape(self):
        s = self._s

        class _S:
            def as_list(self_inner):
                return list(s)
        return _S()


def test_g1_optimizer_trajectories(golden):
    g = golden("g1_optimizers.npz")
    ada, adam = orc.AdagradState(0.1), orc.AdamState(0.1, decay=0.999)
    for t, p in enumerate(g["phis"]):
        np.testing.assert_array_equal(ada.update(p.copy()), g["adagrad_steps"][t])
        np.testing.assert_array_equal(ada.hist, g["adagrad_hist"][t])
        np.testing.assert_array_equal(adam.update(p.copy()), g["adam_steps"][t])
        np.testing.assert_array_equal(adam.mu, g["adam_mu"][t])
        np.testing.assert_array_equal(adam.nu, g["adam_nu"][t])
        assert adam.learning_rate == g["adam_lr"][t]
    assert ada.learning_rate == float(g["adagrad_lr_final"]) == 0.1      # decay stored, never applied
    assert ada.n_iters == int(g["adagrad_n_iters"]) and adam.n_iters == int(g["adam_n_iters"])
    # the first Adam step is ~0.316 * sign(phi) * lr (mu = phi, nu = phi^2 at t = 0)
    first = g["adam_steps"][0]
    np.testing.assert_allclose(np.abs(first), 0.1 * 10.0 / np.sqrt(1000.0), rtol=1e-3)


@pytest.mark.parametrize("n,d", [(7, 3), (8, 5), (100, 10)])
def test_g2_compute_phi(golden, n, d):
    g = golden("g2_compute_phi.npz")
    phi = orc.compute_phi(g[f"T_{n}x{d}"], g[f"G_{n}x{d}"], np.float32)
    assert phi.dtype == np.float64
    np.testing.assert_array_equal(phi, g[f"phi_{n}x{d}"])


@pytest.mark.parametrize("tag", ["noclip", "clip"])
@pytest.mark.parametrize("oname", ["adagrad", "adam"])
def test_g3_update_particles(golden, tag, oname):
    g = golden("g3_update_particles.npz")
    key = f"{tag}_{oname}"
    gd = orc.AdagradState(0.05) if oname == "adagrad" else orc.AdamState(0.05, decay=0.99)
    theta = g[key + "_T0"].copy()
    for t, G in enumerate(g[key + "_G"]):
        nrm = np.linalg.norm(orc.compute_phi(theta, G))
        assert (nrm > 10) == (tag == "clip")
        theta, _ = orc.update_particles(theta, G, gd)
        np.testing.assert_allclose(theta, g[key + "_theta"][t], rtol=0, atol=1e-15)


def test_g4_converters(golden):
    g = golden("g4_converters.npz")
    vz, va, vm = Var("model/zeta:0", [3, 1]), Var("model/alpha:0", []), Var("model/mid:0", [2, 2])
    d = {vz: g["zeta"], va: g["alpha"], vm: g["mid"]}
    arr, access = orc.pack_dictionary(d)
    np.testing.assert_array_equal(arr, g["array"])
    assert access[va] == tuple(g["access_alpha"]) == (0, 1)         # sorted by name: alpha, mid, zeta
    assert access[vm] == tuple(g["access_mid"]) == (1, 5)
    assert access[vz] == tuple(g["access_zeta"]) == (5, 8)
    back = orc.unpack_array(arr, access)
    for v in d:
        np.testing.assert_array_equal(back[v], d[v])


@pytest.mark.parametrize("n,d", [(7, 3), (8, 5), (100, 10), (257, 33)])
def test_g5_kernel_restated(golden, n, d):
    g = golden("g5_kernel_restated.npz")
    T = g[f"T_{n}x{d}"]
    K, dK, h2 = orc.kernel_and_grad(T, np.float32, return_h2=True)
    np.testing.assert_array_equal(dK, g[f"dK_{n}x{d}"])
    assert h2 == g[f"h2_{n}x{d}"]
    if n <= 100:
        np.testing.assert_array_equal(K, g[f"K_{n}x{d}"])
    # fp32 flow vs the fp64 twin
    np.testing.assert_allclose(dK, g[f"dK64_{n}x{d}"], rtol=0, atol=2e-5 * np.abs(g[f"dK64_{n}x{d}"]).max())


@pytest.mark.parametrize("n", [7, 8])   # odd and even n*n
def test_dk_is_minus_half_gradient_of_sum_k(n):
    """dK = -0.5 d(sum K)/d(theta) with the bandwidth held constant (squared_exponential_kernel.py:23,32):
    central finite differences of sum K in fp64."""
    d = 4
    T = np.random.default_rng(n).normal(size=(n, d))
    K, dK, h2 = orc.kernel_and_grad(T, np.float64, return_h2=True)

    def sum_k(X):
        return np.exp(-orc.pairwise_sq_dists(X, np.float64) / h2 / 2.0).sum()
    eps = 1e-6
    fd = np.zeros_like(T)
    for i in range(n):
        for k in range(d):
            P, M = T.copy(), T.copy()
            P[i, k] += eps
            M[i, k] -= eps
            fd[i, k] = (sum_k(P) - sum_k(M)) / (2 * eps)
    np.testing.assert_allclose(dK, -0.5 * fd, atol=5e-8)
    np.testing.assert_allclose(dK.sum(axis=0), 0, atol=1e-12)      # antisymmetry
    np.testing.assert_allclose(dK, (K.sum(1)[:, None] * T - K @ T) / h2, atol=1e-12)


@pytest.mark.parametrize("shape", [(7, 7), (8, 8), (1, 5), (3, 4)])
def test_median_semantics(shape):
    D = np.random.default_rng(sum(shape)).normal(size=shape).astype(np.float32)
    m = orc.median_all(D)
    assert m.dtype == np.float32
    flat = np.sort(D.reshape(-1))[::-1]                       # top_k order of compute_median.py:7-15
    k = flat.size // 2 + 1
    expect = flat[:k][k - 2:].mean(dtype=np.float32) if flat.size % 2 == 0 else flat[:k][k - 1]
    assert m == expect == np.float32(np.median(D))


def test_g6_linear_regression_posterior(golden):
    g = golden("g6_linear_regression.npz")
    assert g["X"].shape == (1000, 1)
    np.testing.assert_allclose(g["post_mean"], 0.383949, atol=1e-6)
    np.testing.assert_allclose(g["post_std"], 0.031917, atol=1e-6)
    np.testing.assert_allclose(g["post_precision"], 981.628, atol=1e-3)


def test_oracle_svgd_on_linear_regression_converges(golden):
    """End-to-end KAT on CPU with the oracle: particles converge to the closed-form posterior."""
    g = golden("g6_linear_regression.npz")
    X, y = g["X"], g["y"]
    n = 50
    theta = np.random.default_rng(0).normal(size=(n, 1)) * 0.01
    gd = orc.AdamState(learning_rate=0.1)
    for _ in range(300):
        score = (X.T @ (y[None, :] - theta @ X.T).T).T - theta      # X^T (y - X w) - w per particle
        theta, _ = orc.update_particles(theta, score, gd)
    assert abs(theta.mean() - float(g["post_mean"][0])) < 0.01
    assert 0.4 * float(g["post_std"][0]) < theta.std() < 1.6 * float(g["post_std"][0])


# ---- the radix-select model -------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(6))
def test_radix_select_model_matches_partition(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(2, 60))
    kind = seed % 3
    if kind == 0:
        D = orc.pairwise_sq_dists(rng.normal(size=(n, 5)), np.float32)
    elif kind == 1:
        D = rng.normal(size=(n, n)).astype(np.float32) * 1e-3         # negatives, tiny magnitudes
    else:
        D = np.round(rng.normal(size=(n, n)) * 2).astype(np.float32)  # heavy ties, +-0
    assert sm.radix_median(D) == orc.median_all(D)


def test_radix_keys_are_monotone():
    x = np.array([-np.inf, -3.5, -1e-30, -0.0, 0.0, 1e-38, 2.0, 7e37, np.inf], dtype=np.float32)
    k = sm.f32_keys(x).astype(np.int64)
    assert np.all(np.diff(k) >= 0) and k[3] < k[4]
    for v in x:
        assert sm.key_to_f32(sm.f32_keys(np.array([v], np.float32))[0]) == v or (v == 0)
