import os, sys, math, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/stein_amd") else os.getcwd())
sys.path.insert(0, os.path.join(sys.path[0], "examples", "regression_neural_network"))
import importlib.util
spec = importlib.util.spec_from_file_location("bnn", os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "examples/regression_neural_network/main.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
n, H, B = 8192, 666, 20
m.H = H
dev = "cuda"
shapes = {"model/w_1:0": [1, H], "model/b_1:0": [H], "model/w_2:0": [H, 1], "model/b_2:0": [], "model/log_lambda:0": [], "model/log_gamma:0": []}
X = torch.rand(B, 1, device=dev); y = torch.randn(B, device=dev)
feed = {"X": X, "y": y}
from stein_amd.samplers import SteinSampler
from stein_amd.optimizers import AdamGradientDescent
s = SteinSampler(n, m.make_log_posterior(B, B), AdamGradientDescent(1e-2), model_vars=shapes, seed=0)
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
print("BNN n=%d d=%d: autograd score %.3f ms, full train_on_batch %.3f ms" % (n, s.n_params, t(lambda: s.score_matrix(feed)), t(lambda: s.train_on_batch(feed))))
from stein_amd.scores import BnnScore
s.score = BnnScore(1, H, BnnScore.columns(s._access), n_train=B)
print("BNN n=%d d=%d: HIP score      %.3f ms, full train_on_batch %.3f ms" % (n, s.n_params, t(lambda: s.score_matrix(feed)), t(lambda: s.train_on_batch(feed))))
