"""Bayesian linear regression with SVGD on MI355X -- the stein_amd version of the reference's
examples/linear_regression/main.py (unit-variance likelihood, N(0, 1) prior on the weights, 50 particles,
Adam(0.1), 500 iterations).  The model is a batched torch callable instead of a TF1 graph.

    python examples/linear_regression/main.py [--particles 50] [--iters 500]

The posterior of this model is Gaussian in closed form; the script prints the SVGD estimate next to it.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from stein_amd.optimizers import AdamGradientDescent  # noqa: E402
from stein_amd.samplers import SteinSampler  # noqa: E402
from stein_amd.scores import GlmScore  # noqa: E402


def make_data(n_samples=1000, n_feats=1, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n_samples, n_feats))
    w = rng.normal(size=(n_feats, 1)) * 5
    y = rng.normal(X @ w, 0.3)
    return X, y[:, 0], w[:, 0]


def log_posterior(theta, feed):
    """theta["model/w:0"]: [n_particles, n_feats, 1] -> log p(w | X, y) up to a constant, one value per particle."""
    w = theta["model/w:0"][:, :, 0]                              # [n, f]
    resid = feed["X"] @ w.T - feed["y"][:, None]                  # [samples, n]
    return -0.5 * (resid ** 2).sum(0) - 0.5 * (w ** 2).sum(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=int, default=50)
    ap.add_argument("--iters", type=int, default=500)
    ap.add_argument("--autograd", action="store_true", help="differentiate log_posterior with torch instead of the HIP score producer")
    args = ap.parse_args()
    X, y, w_true = make_data()
    feed = {"X": torch.tensor(X, dtype=torch.float32, device="cuda"),
            "y": torch.tensor(y, dtype=torch.float32, device="cuda")}
    score = None if args.autograd else GlmScore("linear", X.shape[1])
    sampler = SteinSampler(args.particles, log_posterior, AdamGradientDescent(learning_rate=1e-1), score=score,
                           model_vars={"model/w:0": [X.shape[1], 1]})
    t0 = time.time()
    for _ in range(args.iters):
        sampler.train_on_batch(feed)
    torch.cuda.synchronize()
    est = sampler.samples
    prec = X.T @ X + np.eye(X.shape[1])
    mean = np.linalg.solve(prec, X.T @ y)
    print("true coefficients      :", w_true)
    print("analytic posterior mean:", mean, " std:", np.sqrt(1.0 / np.diag(prec)))
    print("SVGD particle mean     :", est.mean(0), " std:", est.std(0))
    print("%d iterations in %.2f s" % (args.iters, time.time() - t0))


if __name__ == "__main__":
    main()
