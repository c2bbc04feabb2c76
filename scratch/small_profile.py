import os, sys, time, cProfile, pstats, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd.samplers import SteinSampler
from stein_amd.optimizers import AdamGradientDescent
from stein_amd.scores import GlmScore
rng = np.random.default_rng(0)
X = rng.normal(size=(1000, 1)); y = rng.normal(X @ np.array([2.5]), 0.3)
feed = {"X": torch.tensor(X, dtype=torch.float32, device="cuda"), "y": torch.tensor(y, dtype=torch.float32, device="cuda")}
s = SteinSampler(50, None, AdamGradientDescent(learning_rate=1e-1), score=GlmScore("linear", 1), model_vars={"model/w:0": [1, 1]}, seed=0)
for _ in range(50): s.train_on_batch(feed)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2000): s.train_on_batch(feed)
torch.cuda.synchronize(); t1 = time.perf_counter()
print("us per iteration (wall): %.1f" % ((t1 - t0) / 2000 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): s.train_on_batch(feed)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
