"""Bayesian neural-network regression with SVGD -- the stein_amd version of the reference's
examples/regression_neural_network/main.py: one hidden layer of 100 ReLU units on 20 synthetic points,
Gamma(1, 0.01) priors on the weight precision lambda and the noise precision gamma (both sampled in log space),
20 particles, Adam(0.1, decay 0.999).  d = 3 H + 3 = 303 parameters per particle.

    python examples/regression_neural_network/main.py [--particles 20] [--iters 3000] [--autograd]
"""
import argparse
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from stein_amd.optimizers import AdamGradientDescent  # noqa: E402
from stein_amd.samplers import SteinSampler  # noqa: E402
from stein_amd.scores import BnnScore  # noqa: E402

H = 100


def predict(theta, X):
    """[n_particles, n_points] network outputs for every particle."""
    w1, b1 = theta["model/w_1:0"], theta["model/b_1:0"]         # [n, 1, H], [n, H]
    w2, b2 = theta["model/w_2:0"], theta["model/b_2:0"]         # [n, H, 1], [n]
    hidden = torch.relu(torch.einsum("pf,nfh->nph", X, w1) + b1[:, None, :])
    return torch.einsum("nph,nh->np", hidden, w2[:, :, 0]) + b2[:, None]


def make_log_posterior(n_train, n_batch, a=1.0, b=0.01):
    def normal_logpdf(x, log_prec):
        return 0.5 * log_prec - 0.5 * log_prec.exp() * x ** 2 - 0.5 * math.log(2 * math.pi)

    def gamma_logpdf_of_exp(log_x):   # Gamma(a, b) density evaluated at x = exp(log_x)
        return a * math.log(b) - math.lgamma(a) + (a - 1) * log_x - b * log_x.exp()

    def log_posterior(theta, feed):
        log_lam, log_gam = theta["model/log_lambda:0"], theta["model/log_gamma:0"]   # [n]
        pred = predict(theta, feed["X"])
        log_l = normal_logpdf(pred - feed["y"][None, :], log_gam[:, None]).sum(1)
        prior = gamma_logpdf_of_exp(log_lam) + gamma_logpdf_of_exp(log_gam)
        for name in ("model/w_1:0", "model/b_1:0", "model/w_2:0"):
            v = theta[name].reshape(theta[name].shape[0], -1)
            prior = prior + normal_logpdf(v, log_lam[:, None]).sum(1)
        prior = prior + normal_logpdf(theta["model/b_2:0"], log_lam)
        return (log_l * n_train / n_batch + prior) / n_train
    return log_posterior


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=int, default=20)
    ap.add_argument("--iters", type=int, default=1500)
    ap.add_argument("--autograd", action="store_true", help="differentiate log_posterior with torch instead of the HIP score producer")
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    X = rng.uniform(size=(20, 1))
    y = rng.normal(np.cos(10 * X) * (5 * X), 0.1)[:, 0]
    dev = "cuda"
    feed = {"X": torch.tensor(X, dtype=torch.float32, device=dev), "y": torch.tensor(y, dtype=torch.float32, device=dev)}
    shapes = {"model/w_1:0": [1, H], "model/b_1:0": [H], "model/w_2:0": [H, 1], "model/b_2:0": [],
              "model/log_lambda:0": [], "model/log_gamma:0": []}
    # The reference draws every particle from N(0, 0.01^2) and notes itself (abstract_stein_sampler.py:60-64) that
    # "better initialization is required for ... neural networks": from that start this posterior collapses into the
    # weights-at-zero / infinite-precision mode.  Start the weights at unit scale (input layer x10: inputs are in [0,1]).
    scale = {"model/w_1:0": 10.0, "model/b_1:0": 1.0, "model/w_2:0": 1.0, "model/b_2:0": 1.0,
             "model/log_lambda:0": 0.0, "model/log_gamma:0": 0.0}
    init = {k: rng.normal(size=[args.particles] + s) * scale[k] for k, s in shapes.items()}
    sampler = SteinSampler(args.particles, make_log_posterior(len(X), len(X)),
                           AdamGradientDescent(learning_rate=5e-2, decay=0.999), theta=init, model_vars=shapes)
    assert sampler.n_params == 3 * H + 3
    if not args.autograd:   # closed-form backprop of the same log posterior in one HIP launch (stein_amd/scores.py)
        sampler.score = BnnScore(1, H, BnnScore.columns(sampler._access), n_train=len(X))
    for i in range(args.iters):
        sampler.train_on_batch(feed)
        if i % 250 == 0 or i == args.iters - 1:
            y_hat = sampler.function_posterior(lambda th, f: predict(th, f["X"]), feed)
            print("iteration %5d: mean squared error %.4f" % (i, float(np.mean((y - y_hat.mean(axis=0)) ** 2))))


if __name__ == "__main__":
    main()
