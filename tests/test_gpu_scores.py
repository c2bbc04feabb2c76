"""GPU: the HIP score producer (stein_score_glm, csrc/stein_score.hip) against the NumPy oracle and against torch
autograd of the example models' log posteriors, and end to end through SteinSampler(score=...)."""
import numpy as np
import pytest
import torch

from oracle import score_oracle as so
from stein_amd.optimizers import AdamGradientDescent
from stein_amd.samplers import SteinSampler
from stein_amd.scores import GlmScore

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,F,batch,alpha", [(1, 1, 1, False), (37, 5, 50, True), (300, 54, 50, True), (129, 255, 50, True),
                                             (64, 300, 17, False), (50, 1, 1000, False), (33, 70, 400, True)])
def test_logistic_score_matches_oracle(cuda, n, F, batch, alpha):
    rng = np.random.default_rng(n + F + batch)
    d = F + (1 if alpha else 0) + 1                       # one spare column that must come back as zero
    w_col, a_col = (2, 0) if alpha else (1, None)
    th = rng.normal(size=(n, d)) * 0.5
    X = rng.normal(size=(batch, F))
    y = (rng.uniform(size=batch) < 0.5).astype(np.float64)
    ref = so.glm_score_matrix(th, "logistic", w_col, F, -1 if a_col is None else a_col, X, y, scale=16000.0 / batch,
                              prior_precision=0.7)
    prod = GlmScore("logistic", F, w_col=w_col, alpha_col=a_col, n_train=16000, prior_precision=0.7)
    out = prod(torch.tensor(th, dtype=torch.float32, device=cuda),
               {"X": torch.tensor(X, dtype=torch.float32, device=cuda), "y": torch.tensor(y, dtype=torch.float32, device=cuda)})
    got = out.double().cpu().numpy()
    th32, X32 = th.astype(np.float32).astype(np.float64), X.astype(np.float32).astype(np.float64)
    ref = so.glm_score_matrix(th32, "logistic", w_col, F, -1 if a_col is None else a_col, X32, y, scale=16000.0 / batch,
                              prior_precision=0.7)
    # fp32 sums of `batch` terms scaled by n_train / batch: error relative to the column's magnitude
    tol = 2e-5 * np.abs(ref).max() + 1e-6
    assert np.abs(got - ref).max() <= tol, np.abs(got - ref).max() / np.abs(ref).max()


def test_linear_score_matches_oracle_and_autograd(cuda):
    rng = np.random.default_rng(3)
    n, F, batch = 50, 3, 1000
    th, X, y = rng.normal(size=(n, F)), rng.normal(size=(batch, F)), rng.normal(size=batch)
    t = torch.tensor(th, dtype=torch.float32, device=cuda)
    feed = {"X": torch.tensor(X, dtype=torch.float32, device=cuda), "y": torch.tensor(y, dtype=torch.float32, device=cuda)}
    got = GlmScore("linear", F)(t, feed).double().cpu().numpy()
    ref = so.glm_score_matrix(t.double().cpu().numpy(), "linear", 0, F, -1, feed["X"].double().cpu().numpy(), y)
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()
    tt = t.clone().requires_grad_(True)
    lp = -0.5 * ((feed["X"] @ tt.T - feed["y"][:, None]) ** 2).sum(0) - 0.5 * (tt ** 2).sum(1)
    (ga,) = torch.autograd.grad(lp.sum(), tt)
    assert np.abs(got - ga.double().cpu().numpy()).max() <= 2e-5 * np.abs(ref).max()


def test_sampler_with_device_score_reaches_the_analytic_posterior(cuda):
    """Linear-regression KAT: the SVGD particle mean with the HIP score lands on the closed-form posterior mean."""
    rng = np.random.default_rng(0)
    X = rng.normal(size=(1000, 1)); w = np.array([2.5]); y = rng.normal(X @ w, 0.3)
    feed = {"X": torch.tensor(X, dtype=torch.float32, device=cuda), "y": torch.tensor(y, dtype=torch.float32, device=cuda)}
    sampler = SteinSampler(50, None, AdamGradientDescent(learning_rate=1e-1), score=GlmScore("linear", 1),
                           model_vars={"model/w:0": [1, 1]}, device=cuda, seed=0)
    for _ in range(300):
        sampler.train_on_batch(feed)
    mean = np.linalg.solve(X.T @ X + np.eye(1), X.T @ y)
    assert abs(sampler.samples.mean() - mean[0]) < 5e-3


def test_bad_arguments_are_refused(cuda):
    t = torch.zeros(4, 6, device=cuda)
    feed = {"X": torch.zeros(5, 3, device=cuda), "y": torch.zeros(5, device=cuda)}
    with pytest.raises(ValueError):
        GlmScore("logistic", 3, w_col=4)(t, feed)                  # weights run past d
    with pytest.raises(ValueError):
        GlmScore("logistic", 3, w_col=0, alpha_col=1)(t, feed)     # log-alpha inside the weights
    with pytest.raises(ValueError):
        GlmScore("logistic", 4)(t, feed)                           # X has 3 features


@pytest.mark.parametrize("n,n_in,H,batch", [(5, 1, 3, 4), (40, 1, 100, 20), (33, 2, 70, 50), (17, 4, 300, 9), (9, 1, 666, 20)])
def test_bnn_score_matches_oracle(cuda, n, n_in, H, batch):
    from stein_amd.scores import BnnScore
    rng = np.random.default_rng(n + H)
    # packed order of the reference's variables (sorted TF names): log_lambda, log_gamma, w_1, b_1, w_2, b_2, + a spare column
    cols = (2, 2 + n_in * H, 2 + n_in * H + H, 2 + n_in * H + 2 * H, 0, 1)
    d = 2 + n_in * H + 2 * H + 1 + 1
    th = rng.normal(size=(n, d)) * 0.7
    X, y = rng.uniform(size=(batch, n_in)), rng.normal(size=batch)
    th32, X32, y32 = (a.astype(np.float32).astype(np.float64) for a in (th, X, y))
    ref = so.bnn_score_matrix(th32, n_in, H, cols, X32, y32, n_train=float(5 * batch))
    got = BnnScore(n_in, H, cols, n_train=5 * batch)(
        torch.tensor(th, dtype=torch.float32, device=cuda),
        {"X": torch.tensor(X, dtype=torch.float32, device=cuda), "y": torch.tensor(y, dtype=torch.float32, device=cuda)})
    got = got.double().cpu().numpy()
    assert (got[:, -1] == 0).all()
    assert np.abs(got - ref).max() <= 3e-5 * np.abs(ref).max() + 1e-7, np.abs(got - ref).max() / np.abs(ref).max()


def test_bnn_score_matches_autograd_of_the_example(cuda):
    import importlib.util
    import os
    from stein_amd.scores import BnnScore
    spec = importlib.util.spec_from_file_location(
        "bnn_example", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples",
                                    "regression_neural_network", "main.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    n, B = 64, 20
    shapes = {"model/w_1:0": [1, ex.H], "model/b_1:0": [ex.H], "model/w_2:0": [ex.H, 1], "model/b_2:0": [],
              "model/log_lambda:0": [], "model/log_gamma:0": []}
    g = torch.Generator(device="cpu").manual_seed(0)
    feed = {"X": torch.rand(B, 1, generator=g).to(cuda), "y": torch.randn(B, generator=g).to(cuda)}
    init = {k: torch.randn([n] + s, generator=g).numpy() for k, s in shapes.items()}
    auto = SteinSampler(n, ex.make_log_posterior(B, B), AdamGradientDescent(1e-2), theta=init, model_vars=shapes, device=cuda)
    hip = SteinSampler(n, None, AdamGradientDescent(1e-2), theta=init, model_vars=shapes, device=cuda,
                       score=BnnScore(1, ex.H, BnnScore.columns(auto._access), n_train=B))
    a, b = auto.score_matrix(feed), hip.score_matrix(feed)
    assert (a - b).abs().max() <= 2e-5 * a.abs().max()
