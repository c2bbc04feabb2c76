"""CPU: the closed-form scores of oracle/score_oracle.py against central finite differences of the restated log
posteriors (the reference differentiates its TF graphs with tf.gradients; parity is unpinned, see the oracle header)."""
import numpy as np

from oracle import score_oracle as so


def _fd(f, x, h=1e-6):
    g = np.zeros_like(x)
    for i in range(x.size):
        e = np.zeros_like(x); e[i] = h
        g[i] = (f(x + e) - f(x - e)) / (2 * h)
    return g


def test_linear_score_matches_finite_differences():
    rng = np.random.default_rng(0)
    X, y, w = rng.normal(size=(40, 7)), rng.normal(size=40), rng.normal(size=7)
    g = so.linear_score(w, X, y)
    assert np.allclose(g, _fd(lambda v: so.linear_log_p(v, X, y), w), rtol=1e-6, atol=1e-6)


def test_logistic_score_matches_finite_differences():
    rng = np.random.default_rng(1)
    X, w = rng.normal(size=(50, 9)), rng.normal(size=9) * 0.5
    y = (rng.uniform(size=50) < 0.5).astype(np.float64)
    for la in (-1.0, 0.0, 0.7):
        gw, ga = so.logistic_score(w, la, X, y, scale=16000 / 50)
        fw = _fd(lambda v: so.logistic_log_p(v, la, X, y, 16000 / 50), w)
        fa = _fd(lambda v: so.logistic_log_p(w, v[0], X, y, 16000 / 50), np.array([la]))[0]
        assert np.allclose(gw, fw, rtol=1e-6, atol=1e-5) and abs(ga - fa) < 1e-5 * max(1.0, abs(fa))


def test_matrix_packing():
    rng = np.random.default_rng(2)
    X, y = rng.normal(size=(10, 3)), (rng.uniform(size=10) < 0.5).astype(np.float64)
    th = rng.normal(size=(4, 6))                       # columns: [pad, log_alpha, w0, w1, w2, pad]
    S = so.glm_score_matrix(th, "logistic", 2, 3, 1, X, y, scale=2.0)
    assert (S[:, 0] == 0).all() and (S[:, 5] == 0).all()
    gw, ga = so.logistic_score(th[3, 2:5], th[3, 1], X, y, 2.0)
    assert np.allclose(S[3, 2:5], gw) and np.isclose(S[3, 1], ga)


def test_bnn_score_matches_finite_differences():
    rng = np.random.default_rng(3)
    n_in, H, B = 2, 7, 12
    X, y = rng.uniform(size=(B, n_in)), rng.normal(size=B)
    cols = (2, 2 + n_in * H, 2 + n_in * H + H, 2 + n_in * H + 2 * H, 0, 1)      # log_lambda, log_gamma first, as the reference sorts them
    row = rng.normal(size=2 + n_in * H + 2 * H + 1)
    f = lambda v: so.bnn_log_p(so.bnn_unpack(v, n_in, H, cols), X, y, n_train=40.0)
    g = so.bnn_score_matrix(row[None, :], n_in, H, cols, X, y, n_train=40.0)[0]
    assert np.allclose(g, _fd(f, row), rtol=1e-5, atol=1e-7)
