"""Adam particle optimizer on device.

Follows stein/optimizers/adam_gradient_descent.py:15-58 including its quirks: the
first call sets ``mu = phi, nu = phi**2`` (no (1-beta) factor), both moments are
bias-corrected with the incremented iteration count, eps = 1e-8 is added to
sqrt(nu_hat), and ``learning_rate *= decay`` happens after the step is formed.
"""
import ctypes

import torch

from .. import _lib
from .abstract_gradient_descent import AbstractGradientDescent, CLIP_THRESHOLD, _code, _vp

EPS = 1e-8  # adam_gradient_descent.py:55


class AdamGradientDescent(AbstractGradientDescent):
    def __init__(self, learning_rate=1e-3, decay=1., beta_1=0.9, beta_2=0.999):
        super().__init__(learning_rate, decay)
        self.beta_1 = beta_1
        self.beta_2 = beta_2
        self._mu = self._nu = None

    @property
    def mu(self):
        return self._mu

    @property
    def nu(self):
        return self._nu

    def _state_tensors(self):
        return {"mu": self._mu, "nu": self._nu}

    def _launch(self, theta, phi, state_dtype, sqnorm_dev, clip_scale, step_out):
        first = self.n_iters == 0
        mu = self._state_for("_mu", phi.shape, state_dtype, phi.device, first)
        nu = self._state_for("_nu", phi.shape, state_dtype, phi.device, first)
        t = self.n_iters + 1
        stream = ctypes.c_void_p(torch.cuda.current_stream(phi.device).cuda_stream)
        _lib.call_on(phi.device, "stein_apply_adam", _vp(theta), _vp(phi), _code(phi.dtype), _vp(mu), _vp(nu),
                     phi.numel(), _code(state_dtype), _vp(sqnorm_dev), float(clip_scale), CLIP_THRESHOLD,
                     float(self.learning_rate), float(self.beta_1), float(self.beta_2), EPS, t, _vp(step_out), stream)
        self.n_iters = t
        self.learning_rate *= self.decay
