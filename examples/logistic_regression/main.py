"""Bayesian logistic regression with SVGD -- the stein_amd version of the reference's
examples/logistic_regression/main.py: hierarchical prior w ~ N(0, 1/alpha), alpha ~ Gamma(1, 0.01) sampled as
log alpha, minibatches of 50 rescaled to the training-set size, 100 particles, Adam(0.1).  The Covertype file the
reference reads is not shipped with it, so the data here are synthetic with the same shape (54 features).

    python examples/logistic_regression/main.py [--particles 100] [--iters 2000] [--autograd]

The score matrix comes from the HIP score producer (stein_amd.scores.GlmScore, the closed-form gradient of the model
below); --autograd differentiates log_posterior with torch instead.
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from stein_amd.optimizers import AdamGradientDescent  # noqa: E402
from stein_amd.samplers import SteinSampler  # noqa: E402
from stein_amd.scores import GlmScore  # noqa: E402


def make_data(n=20000, n_feats=54, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, n_feats))
    w = rng.normal(size=n_feats)
    y = (rng.uniform(size=n) < 1.0 / (1.0 + np.exp(-(X @ w)))).astype(np.float64)
    cut = int(0.8 * n)
    return (X[:cut], y[:cut]), (X[cut:], y[cut:])


def make_log_posterior(n_train, n_batch):
    def log_posterior(theta, feed):
        w = theta["model/w:0"][:, :, 0]                       # [n, f]
        log_alpha = theta["model/log_alpha:0"]                # [n]
        alpha = log_alpha.exp()
        logits = feed["X"] @ w.T                              # [batch, n]
        log_l = -F.binary_cross_entropy_with_logits(logits, feed["y"][:, None].expand_as(logits),
                                                    reduction="none").sum(0)
        n_feats = w.shape[1]
        log_prior_w = 0.5 * n_feats * log_alpha - 0.5 * alpha * (w ** 2).sum(1)     # N(0, 1/alpha), constants dropped
        log_prior_alpha = -0.01 * alpha                       # Gamma(1, 0.01) density evaluated at alpha
        return log_l * (n_train / n_batch) + log_prior_w + log_prior_alpha
    return log_posterior


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=int, default=100)
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=50)
    ap.add_argument("--autograd", action="store_true", help="differentiate log_posterior with torch instead of the HIP score producer")
    args = ap.parse_args()
    (Xtr, ytr), (Xte, yte) = make_data()
    dev = "cuda"
    Xtr_t, ytr_t = torch.tensor(Xtr, dtype=torch.float32, device=dev), torch.tensor(ytr, dtype=torch.float32, device=dev)
    Xte_t = torch.tensor(Xte, dtype=torch.float32, device=dev)
    # packed columns follow the sorted variable names: log_alpha first, then the weights
    score = None if args.autograd else GlmScore("logistic", Xtr.shape[1], w_col=1, alpha_col=0, n_train=len(Xtr))
    sampler = SteinSampler(args.particles, make_log_posterior(len(Xtr), args.batch),
                           AdamGradientDescent(learning_rate=1e-1), score=score,
                           model_vars={"model/w:0": [Xtr.shape[1], 1], "model/log_alpha:0": []})
    gen = torch.Generator(device=dev).manual_seed(0)

    def accuracy():
        logits = sampler.function_posterior(lambda th, feed: (feed["X"] @ th["model/w:0"][:, :, 0].T).T, {"X": Xte_t})
        return float(((logits.mean(axis=0) > 0) == (yte > 0.5)).mean())

    for i in range(args.iters):
        if i % 200 == 0:
            print("iteration %5d / %d: held-out accuracy %.4f" % (i, args.iters, accuracy()))
        idx = torch.randint(0, len(Xtr), (args.batch,), device=dev, generator=gen)
        sampler.train_on_batch({"X": Xtr_t[idx], "y": ytr_t[idx]})
    print("final held-out accuracy %.4f" % accuracy())


if __name__ == "__main__":
    main()
