"""CPU restatement of the score (gradient of the log posterior) of the reference's example models.

TEST INFRASTRUCTURE ONLY: imported by tests/, never by stein_amd/.

The reference builds these models as TensorFlow 1 graphs and differentiates them with tf.gradients inside
sess.run (stein/samplers/abstract_stein_sampler.py:53-55, stein/samplers/stein_sampler.py:59-68); TensorFlow is
absent here and the reference holds no vectors for them: PARITY UNPINNED.  The log posteriors below restate the
graphs line by line; the closed-form gradients are checked against central finite differences of those log
posteriors (tests/test_score_oracle.py).
"""
import numpy as np


def linear_log_p(w, X, y, prior_precision=1.0):
    """examples/linear_regression/main.py:18-31: log_l = -1/2 sum (X w - y)^2; w ~ Normal(0, 1) (log density incl. its constant)"""
    resid = X @ w - y
    prior = -0.5 * prior_precision * (w ** 2).sum() + 0.5 * len(w) * (np.log(prior_precision) - np.log(2.0 * np.pi))
    return -0.5 * (resid ** 2).sum() + prior


def linear_score(w, X, y, prior_precision=1.0, scale=1.0):
    return scale * X.T @ (y - X @ w) - prior_precision * w


def logistic_log_p(w, log_alpha, X, y, scale, gamma_rate=0.01):
    """examples/logistic_regression/main.py:23-49.
    log_l = -sum sigmoid_cross_entropy_with_logits(labels=y, logits=X w)            (:41-44)
    w_prior = Normal(0, 1 / sqrt(alpha)), alpha_prior = Gamma(1, 0.01), alpha = exp(log_alpha)   (:30, :33-38)
    log_p = log_l * (n_train / n_batch) + sum w_prior.log_prob(w) + alpha_prior.log_prob(alpha)  (:46-50)"""
    alpha = np.exp(log_alpha)
    z = X @ w
    log_l = (y * z - np.logaddexp(0.0, z)).sum()
    prior_w = (-0.5 * np.log(2.0 * np.pi) + 0.5 * log_alpha - 0.5 * alpha * w ** 2).sum()
    prior_alpha = np.log(gamma_rate) - gamma_rate * alpha          # Gamma(concentration 1, rate): rate * exp(-rate a)
    return scale * log_l + prior_w + prior_alpha


def logistic_score(w, log_alpha, X, y, scale, gamma_rate=0.01):
    """-> (d/dw [F], d/dlog_alpha)"""
    alpha = np.exp(log_alpha)
    z = X @ w
    gw = scale * X.T @ (y - 1.0 / (1.0 + np.exp(-z))) - alpha * w
    ga = 0.5 * len(w) - alpha * (0.5 * (w ** 2).sum() + gamma_rate)
    return gw, ga


def glm_score_matrix(theta, kind, w_col, n_feats, alpha_col, X, y, scale=1.0, prior_precision=1.0, gamma_rate=0.01):
    """The [n, d] score matrix with the packing of stein_score_glm (include/steinhip.h); fp64."""
    theta = np.asarray(theta, dtype=np.float64)
    out = np.zeros_like(theta)
    for i in range(theta.shape[0]):
        w = theta[i, w_col:w_col + n_feats]
        if kind == "linear":
            out[i, w_col:w_col + n_feats] = linear_score(w, X, y, prior_precision, scale)
        elif alpha_col >= 0:
            gw, ga = logistic_score(w, theta[i, alpha_col], X, y, scale, gamma_rate)
            out[i, w_col:w_col + n_feats] = gw
            out[i, alpha_col] = ga
        else:
            z = X @ w
            out[i, w_col:w_col + n_feats] = scale * X.T @ (y - 1.0 / (1.0 + np.exp(-z))) - prior_precision * w
    return out


def bnn_predict(p, X):
    """examples/regression_neural_network/main.py:45-49: relu(X w1 + b1) w2 + b2"""
    return np.maximum(X @ p["w1"] + p["b1"], 0.0) @ p["w2"] + p["b2"]


def bnn_log_p(p, X, y, n_train, ga=1.0, gb=0.01):
    """examples/regression_neural_network/main.py:29-85; p = dict(w1 [F,H], b1 [H], w2 [H], b2, log_lambda, log_gamma).
    log_l = sum log N(y; pred, 1/sqrt(gamma))  (:51-53); priors Gamma(alpha, beta) on lambda, gamma (:56-57) and
    N(0, 1/sqrt(lambda)) on every weight (:58-73); log_p = (log_l n_train / n_batch + priors) / n_train (:75-85)."""
    from math import lgamma, log, pi
    lam, gam = np.exp(p["log_lambda"]), np.exp(p["log_gamma"])
    e = y - bnn_predict(p, X)
    log_l = (0.5 * p["log_gamma"] - 0.5 * gam * e ** 2 - 0.5 * log(2 * pi)).sum()
    gamma_pdf = lambda x, lx: ga * log(gb) - lgamma(ga) + (ga - 1.0) * lx - gb * x
    prior = gamma_pdf(lam, p["log_lambda"]) + gamma_pdf(gam, p["log_gamma"])
    for w in (p["w1"], p["b1"], p["w2"], np.atleast_1d(p["b2"])):
        prior += (0.5 * p["log_lambda"] - 0.5 * lam * np.asarray(w) ** 2 - 0.5 * log(2 * pi)).sum()
    return (log_l * n_train / len(y) + prior) / n_train


def bnn_score(p, X, y, n_train, ga=1.0, gb=0.01):
    """closed-form gradient of bnn_log_p, same dict layout"""
    lam, gam = np.exp(p["log_lambda"]), np.exp(p["log_gamma"])
    z = X @ p["w1"] + p["b1"]
    a = np.maximum(z, 0.0)
    e = y - (a @ p["w2"] + p["b2"])
    s = n_train / len(y)
    t = (e[:, None] * p["w2"][None, :]) * (z > 0)                 # [B, H]
    nw = p["w1"].size + p["b1"].size + p["w2"].size + 1
    sw2 = (p["w1"] ** 2).sum() + (p["b1"] ** 2).sum() + (p["w2"] ** 2).sum() + p["b2"] ** 2
    return dict(w1=(s * gam * X.T @ t - lam * p["w1"]) / n_train, b1=(s * gam * t.sum(0) - lam * p["b1"]) / n_train,
                w2=(s * gam * a.T @ e - lam * p["w2"]) / n_train, b2=(s * gam * e.sum() - lam * p["b2"]) / n_train,
                log_gamma=(s * (0.5 * len(y) - 0.5 * gam * (e ** 2).sum()) + (ga - 1.0) - gb * gam) / n_train,
                log_lambda=(0.5 * nw - 0.5 * lam * sw2 + (ga - 1.0) - gb * lam) / n_train)


BNN_ORDER = ("w1", "b1", "w2", "b2", "log_lambda", "log_gamma")


def bnn_unpack(row, n_in, n_hidden, cols):
    """one packed particle -> the dict of bnn_log_p; cols = first column of (w1, b1, w2, b2, log_lambda, log_gamma)"""
    c = dict(zip(BNN_ORDER, cols))
    return dict(w1=row[c["w1"]:c["w1"] + n_in * n_hidden].reshape(n_in, n_hidden), b1=row[c["b1"]:c["b1"] + n_hidden],
                w2=row[c["w2"]:c["w2"] + n_hidden], b2=row[c["b2"]], log_lambda=row[c["log_lambda"]], log_gamma=row[c["log_gamma"]])


def bnn_score_matrix(theta, n_in, n_hidden, cols, X, y, n_train, ga=1.0, gb=0.01):
    theta = np.asarray(theta, dtype=np.float64)
    out = np.zeros_like(theta)
    c = dict(zip(BNN_ORDER, cols))
    for i in range(theta.shape[0]):
        g = bnn_score(bnn_unpack(theta[i], n_in, n_hidden, cols), X, y, n_train, ga, gb)
        out[i, c["w1"]:c["w1"] + n_in * n_hidden] = g["w1"].reshape(-1)
        out[i, c["b1"]:c["b1"] + n_hidden] = g["b1"]
        out[i, c["w2"]:c["w2"] + n_hidden] = g["w2"]
        out[i, c["b2"]], out[i, c["log_lambda"]], out[i, c["log_gamma"]] = g["b2"], g["log_lambda"], g["log_gamma"]
    return out
