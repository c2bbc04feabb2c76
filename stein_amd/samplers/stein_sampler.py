"""SteinSampler: the user-facing SVGD sampler (stein/samplers/stein_sampler.py:8-78).

``SteinSampler(n_particles, log_p, gd, theta=None)`` keeps the reference's argument order.
Because there is no TF graph here, ``log_p`` is a callable:

    log_p(theta, batch_feed) -> tensor [n_local]     log-posterior of every particle,

written with torch ops on ``theta`` (the dict of per-variable views, or the packed matrix
when no model_vars were given), batched over the particle axis.  The score matrix the
reference obtains with n sequential ``sess.run(grad_log_p)`` calls (:59-68) comes from one
reverse pass: particles are independent, so d(sum_i log_p_i)/d(theta) is the per-particle
score.  A closed-form ``score(theta, batch_feed) -> [n_local, d]`` may be supplied instead
(keyword ``score=``), or the caller can skip both and drive ``update_particles`` directly.
"""
import torch

from ..utilities.converters import convert_array_to_dictionary
from .abstract_stein_sampler import AbstractSteinSampler


class SteinSampler(AbstractSteinSampler):
    def __init__(self, n_particles, log_p, gd, theta=None, *, score=None, **kwargs):
        super().__init__(n_particles, log_p, theta, **kwargs)
        self.gd = gd
        self.score = score
        # stein_sampler.py:48.  While this is the package's own RBF kernel the sampler goes through the fused engine (K is
        # never materialised); assign any other object with kernel_and_grad(theta) -> (K, dK) and update_particles /
        # compute_phi call it exactly as the reference does (abstract_stein_sampler.py:103)
        from ..kernels import SquaredExponentialKernel
        self.kernel = SquaredExponentialKernel(self.n_particles, None, device=self.device) \
            if self._group is None else None

    def score_matrix(self, batch_feed=None):
        """[n_local, d] float32 device tensor of d log_p / d theta for every particle."""
        if self.score is not None:
            if getattr(self.score, "wants_matrix", False):   # device score producers (stein_amd.scores) take the packed matrix
                if self.dtype != torch.float32:
                    raise ValueError("device score producers need float32 particles")
                return self.score(self.theta_matrix, batch_feed)
            arg = self.theta if self._access is not None else self.theta_matrix
            g = self.score(arg, batch_feed)
            if isinstance(g, dict):
                from ..utilities.converters import convert_dictionary_to_array
                g, _ = convert_dictionary_to_array(g)
            return torch.as_tensor(g).to(device=self.device, dtype=torch.float32).reshape(
                self.n_local, self.n_params).contiguous()
        if self.log_p is None:
            raise ValueError("no log_p / score callable: call update_particles(score_matrix) directly")
        t = self.theta_matrix.detach().clone().requires_grad_(True)
        arg = convert_array_to_dictionary(t, self._access, self._shapes) if self._access is not None else t
        lp = self.log_p(arg, batch_feed)
        if lp.shape != (self.n_local,):
            raise ValueError("log_p must return one value per particle, shape (%d,), got %s" %
                             (self.n_local, tuple(lp.shape)))
        (g,) = torch.autograd.grad(lp.sum(), t)
        return g.to(torch.float32).contiguous()

    def train_on_batch(self, batch_feed=None):
        """One SVGD iteration on a batch (stein_sampler.py:50-71)."""
        self.update_particles(self.score_matrix(batch_feed))

    @property
    def samples(self):
        """Packed [n_local, d] particle matrix as NumPy (stein_sampler.py:73-78)."""
        return self.theta_matrix.detach().cpu().numpy()
