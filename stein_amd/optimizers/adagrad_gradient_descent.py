"""Adagrad-named (RMSprop-form) particle optimizer on device.

Follows stein/optimizers/adagrad_gradient_descent.py:13-44: the first call sets
``hist = phi**2``, later calls keep an EMA with weight ``alpha``; the step is
``phi / (1e-6 + sqrt(hist)) * learning_rate``; ``decay`` is accepted, stored and
never applied (as in the reference).
"""
import ctypes

import torch

from .. import _lib
from .abstract_gradient_descent import AbstractGradientDescent, CLIP_THRESHOLD, _code, _vp

EPS = 1e-6  # adagrad_gradient_descent.py:44


class AdagradGradientDescent(AbstractGradientDescent):
    def __init__(self, learning_rate=1e-3, decay=1., alpha=0.9):
        super().__init__(learning_rate, decay)
        self.alpha = alpha
        self._hist = None

    @property
    def hist(self):
        """Running average of squared directions (device tensor; None before the first update)."""
        return self._hist

    def _state_tensors(self):
        return {"hist": self._hist}

    def _launch(self, theta, phi, state_dtype, sqnorm_dev, clip_scale, step_out):
        first = self.n_iters == 0
        hist = self._state_for("_hist", phi.shape, state_dtype, phi.device, first)
        stream = ctypes.c_void_p(torch.cuda.current_stream(phi.device).cuda_stream)
        _lib.call_on(phi.device, "stein_apply_adagrad", _vp(theta), _vp(phi), _code(phi.dtype), _vp(hist), phi.numel(),
                     _code(state_dtype), _vp(sqnorm_dev), float(clip_scale), CLIP_THRESHOLD, float(self.learning_rate),
                     float(self.alpha), EPS, 1 if first else 0, _vp(step_out), stream)
        self.n_iters += 1
