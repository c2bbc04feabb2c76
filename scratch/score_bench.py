"""Score production at the C3 shape (n=16384 particles, 255 weights + log alpha, minibatch 50): the HIP producer vs
torch autograd of the same log posterior, and a full train_on_batch iteration with each."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd.optimizers import AdagradGradientDescent
from stein_amd.samplers import SteinSampler
from stein_amd.scores import GlmScore
n, nf, batch, ntrain = 16384, 255, 50, 16000
dev = "cuda"
X = torch.randn(batch, nf, device=dev); y = (torch.rand(batch, device=dev) < 0.5).float()
feed = {"X": X, "y": y}
def log_p(theta, feed):
    w, la = theta[:, 1:], theta[:, 0]
    logits = feed["X"] @ w.T
    ll = -F.binary_cross_entropy_with_logits(logits, feed["y"][:, None].expand_as(logits), reduction="none").sum(0)
    return ll * (ntrain / batch) + 0.5 * nf * la - 0.5 * la.exp() * (w ** 2).sum(1) - 0.01 * la.exp()
def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
theta0 = 0.1 * torch.randn(n, nf + 1, device=dev)
prod = GlmScore("logistic", nf, w_col=1, alpha_col=0, n_train=ntrain)
out = torch.empty_like(theta0)
def autograd_score():
    t = theta0.detach().clone().requires_grad_(True)
    return torch.autograd.grad(log_p(t, feed).sum(), t)[0]
a, b = prod(theta0, feed), autograd_score()
print("max |hip - autograd| / max|autograd| = %.2e" % ((a - b).abs().max() / b.abs().max()).item())
print("score only      : HIP %.3f ms   torch autograd %.3f ms" % (timeit(lambda: prod(theta0, feed, out)), timeit(autograd_score)))
for name, kw in (("HIP score", dict(log_p=None, score=prod)), ("autograd", dict(log_p=log_p))):
    s = SteinSampler(n, kw.get("log_p"), AdagradGradientDescent(learning_rate=1e-3), theta=theta0.clone(), score=kw.get("score"), device=dev)
    print("train_on_batch  : %-10s %.3f ms / iteration" % (name, timeit(lambda: s.train_on_batch(feed))))
