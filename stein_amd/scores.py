"""Score producers on the device: d log p / d theta for every particle in one launch.

The reference obtains the score matrix with n sequential ``sess.run(grad_log_p)`` calls on a TF1 graph of the
model (stein/samplers/stein_sampler.py:59-68).  For the generalised linear models of its examples the gradient has
a closed form; ``GlmScore`` evaluates it with one hand-written HIP kernel (csrc/stein_score.hip) and plugs into
``SteinSampler(score=...)``.  Other models keep the autograd route of ``SteinSampler.score_matrix``.
"""
import torch

from . import _lib


class GlmScore:
    """score(theta_matrix, feed) -> [n, d] float32 device tensor, feed = {"X": [batch, n_feats], "y": [batch]}.

    kind        : "linear" (examples/linear_regression/main.py:18-31) or "logistic"
                  (examples/logistic_regression/main.py:23-49)
    w_col       : first column of the weights in the packed particle matrix (columns are ordered by sorted variable
                  name, stein/utilities/converters.py:40)
    alpha_col   : column of log(alpha) for the hierarchical prior w ~ N(0, 1/alpha), alpha ~ Gamma(1, gamma_rate);
                  None -> fixed prior precision
    n_train     : the log-likelihood of a batch is scaled by n_train / batch (logistic_regression/main.py:47);
                  None -> no scaling
    """
    wants_matrix = True   # SteinSampler hands this callable the packed [n, d] matrix, not the dict of views

    def __init__(self, kind, n_feats, w_col=0, alpha_col=None, n_train=None, prior_precision=1.0, gamma_rate=0.01):
        if kind not in ("linear", "logistic"):
            raise ValueError("kind must be 'linear' or 'logistic'")
        self.kind = _lib.GLM_LINEAR if kind == "linear" else _lib.GLM_LOGISTIC
        self.n_feats, self.w_col = int(n_feats), int(w_col)
        self.alpha_col = -1 if alpha_col is None else int(alpha_col)
        self.n_train = n_train
        self.prior_precision, self.gamma_rate = float(prior_precision), float(gamma_rate)

    def __call__(self, theta, feed, out=None):
        X, y = feed["X"], feed["y"]
        for name, t in (("theta", theta), ("X", X), ("y", y)):
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise ValueError("%s must be a contiguous float32 device tensor" % name)
        n, d = theta.shape
        batch = X.shape[0]
        if X.shape != (batch, self.n_feats) or y.numel() != batch:
            raise ValueError("X must be [batch, %d] and y [batch]" % self.n_feats)
        if out is None:
            out = torch.empty_like(theta)
        scale = 1.0 if self.n_train is None else float(self.n_train) / batch
        _lib.call_on(theta.device, "stein_score_glm", theta.data_ptr(), n, d, self.kind, self.w_col, self.n_feats, self.alpha_col,
                  X.data_ptr(), y.data_ptr(), batch, scale, self.prior_precision, self.gamma_rate, out.data_ptr(),
                  torch.cuda.current_stream(theta.device).cuda_stream)
        return out


class BnnScore:
    """Score of the one-hidden-layer Bayesian neural-network regression model of
    examples/regression_neural_network/main.py:29-85 (ReLU, Gamma(a, b) priors on the weight precision lambda and the
    noise precision gamma, both sampled in log space), for every particle in one launch.

    cols: first column of (w1, b1, w2, b2, log_lambda, log_gamma) in the packed particle matrix -- use
    ``BnnScore.columns(sampler)`` for a SteinSampler built with the variable names of the example.
    """
    wants_matrix = True
    NAMES = ("model/w_1:0", "model/b_1:0", "model/w_2:0", "model/b_2:0", "model/log_lambda:0", "model/log_gamma:0")

    def __init__(self, n_in, n_hidden, cols, n_train, gamma_a=1.0, gamma_b=0.01):
        import ctypes
        self.n_in, self.n_hidden, self.n_train = int(n_in), int(n_hidden), float(n_train)
        self.gamma_a, self.gamma_b = float(gamma_a), float(gamma_b)
        self._cols = (ctypes.c_int64 * 6)(*[int(c) for c in cols])

    @staticmethod
    def columns(access, names=NAMES):
        """first packed column of each parameter block from a sampler's access map {name: (start, end)}"""
        return tuple(access[k][0] for k in names)

    def __call__(self, theta, feed, out=None):
        X, y = feed["X"], feed["y"]
        for name, t in (("theta", theta), ("X", X), ("y", y)):
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise ValueError("%s must be a contiguous float32 device tensor" % name)
        n, d = theta.shape
        batch = X.shape[0]
        if X.shape != (batch, self.n_in) or y.numel() != batch:
            raise ValueError("X must be [batch, %d] and y [batch]" % self.n_in)
        if out is None:
            out = torch.empty_like(theta)
        _lib.call_on(theta.device, "stein_score_bnn", theta.data_ptr(), n, d, self.n_in, self.n_hidden, self._cols, X.data_ptr(),
                  y.data_ptr(), batch, self.n_train, self.gamma_a, self.gamma_b, out.data_ptr(),
                  torch.cuda.current_stream(theta.device).cuda_stream)
        return out
