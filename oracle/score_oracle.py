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
