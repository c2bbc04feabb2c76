"""Wall time per train_on_batch iteration at the reference's own example sizes (HIP score producers)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd.samplers import SteinSampler
from stein_amd.optimizers import AdamGradientDescent
from stein_amd.scores import GlmScore, BnnScore
dev = "cuda"
def run(name, s, feed, iters=2000):
    for _ in range(50): s.train_on_batch(feed)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): s.train_on_batch(feed)
    torch.cuda.synchronize()
    print("%-46s %7.1f us per iteration" % (name, (time.perf_counter() - t0) / iters * 1e6))
g = torch.Generator(device="cpu").manual_seed(0)
# linear regression: 50 particles, 1 feature, 1000 points
feed = {"X": torch.randn(1000, 1, generator=g).to(dev), "y": torch.randn(1000, generator=g).to(dev)}
run("linear regression n=50 d=1 batch=1000", SteinSampler(50, None, AdamGradientDescent(1e-1), score=GlmScore("linear", 1), model_vars={"model/w:0": [1, 1]}, seed=0), feed)
# logistic regression: 100 particles, 54 features + log alpha, minibatch 50
feed = {"X": torch.randn(50, 54, generator=g).to(dev), "y": (torch.rand(50, generator=g) < 0.5).float().to(dev)}
run("logistic regression n=100 d=55 batch=50", SteinSampler(100, None, AdamGradientDescent(1e-1), score=GlmScore("logistic", 54, w_col=1, alpha_col=0, n_train=16000),
    model_vars={"model/w:0": [54, 1], "model/log_alpha:0": []}, seed=0), feed)
# BNN: 20 particles, H=100 -> d=303, 20 points
H = 100
shapes = {"model/w_1:0": [1, H], "model/b_1:0": [H], "model/w_2:0": [H, 1], "model/b_2:0": [], "model/log_lambda:0": [], "model/log_gamma:0": []}
feed = {"X": torch.rand(20, 1, generator=g).to(dev), "y": torch.randn(20, generator=g).to(dev)}
s = SteinSampler(20, None, AdamGradientDescent(5e-2, decay=0.999), model_vars=shapes, seed=0)
s.score = BnnScore(1, H, BnnScore.columns(s._access), n_train=20)
run("BNN regression n=20 d=303 batch=20", s, feed)
