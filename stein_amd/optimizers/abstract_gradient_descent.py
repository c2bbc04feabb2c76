"""Base class of the particle optimizers (device-resident state).

API of stein/optimizers/abstract_gradient_descent.py:13-30: ``learning_rate``,
``decay``, ``n_iters`` attributes and ``update(phi) -> step``.  On top of that
each optimizer has ``apply_(theta, phi, sqnorm_dev)``, the fused device form the
sampler uses: norm clip + optimizer map + ``theta += step`` in one HIP pass with
no host round trip (stein/samplers/abstract_stein_sampler.py:125-126).
"""
import ctypes

import numpy as np
import torch

from .. import _lib

CLIP_THRESHOLD = 10.0  # stein/samplers/abstract_stein_sampler.py:125


def _vp(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _code(dtype):
    if dtype == torch.float32:
        return _lib.F32
    if dtype == torch.float64:
        return _lib.F64
    raise ValueError("optimizer state must be float32 or float64, got %s" % dtype)


class AbstractGradientDescent:
    def __init__(self, learning_rate, decay):
        self.learning_rate = learning_rate
        self.decay = decay
        self.n_iters = 0
        self._device = None

    # -- helpers shared by the subclasses ---------------------------------------------------
    def _new_state(self, like_shape, dtype, device):
        return torch.zeros(like_shape, dtype=dtype, device=device)

    @staticmethod
    def _as_device_phi(phi, device=None):
        """-> (contiguous device tensor, state dtype, was_numpy).

        float64 input (every NumPy array, float64 tensors) stays float64 on the device and gets float64 state:
        the reference's ``gd.update(phi)`` is pure fp64 NumPy (adagrad_gradient_descent.py:37-44,
        adam_gradient_descent.py:44-58), so phi must not be rounded to fp32 on the way in.  Anything else is
        float32 with float32 state."""
        if isinstance(phi, torch.Tensor):
            dt = torch.float64 if phi.dtype == torch.float64 else torch.float32
            if not phi.is_cuda:
                phi = phi.to(device or "cuda")
            return phi.to(dt).contiguous(), dt, False
        arr = np.ascontiguousarray(np.asarray(phi), dtype=np.float64)
        return torch.from_numpy(arr).to(device or "cuda"), torch.float64, True

    def _state_for(self, name, shape, dtype, device, first):
        """The state tensor `name`, created on the first update.  A later call in the other precision (``update`` on a
        NumPy array after ``apply_`` on float32 particles, or the reverse) converts the state instead of failing; a
        change of shape is an error."""
        cur = getattr(self, name)
        if cur is None or cur.shape != shape:
            if not first:
                raise ValueError("phi shape changed between updates")
            cur = self._new_state(shape, dtype, device)
        elif cur.dtype != dtype:
            cur = cur.to(dtype)
        setattr(self, name, cur)
        return cur

    def _launch(self, theta, phi, state_dtype, sqnorm_dev, clip_scale, step_out):
        raise NotImplementedError()

    def _state_tensors(self):
        raise NotImplementedError()

    # -- public ---------------------------------------------------------------------------------
    def update(self, phi):
        """step for `phi` (same container type as the input), advancing the optimizer state.

        Equivalent of ``gd.update(phi)`` in the reference: no clipping here, no theta.
        """
        phid, state_dtype, was_numpy = self._as_device_phi(phi, self._device)
        step = torch.empty(phid.shape, dtype=state_dtype, device=phid.device)
        self._launch(None, phid, state_dtype, None, 1.0, step)
        if was_numpy:
            return step.cpu().numpy()
        return step if phi.dtype == state_dtype else step.to(phi.dtype)

    def apply_(self, theta, phi, sqnorm_dev=None, clip_scale=1.0):
        """In place: theta += update(phi * clip), with clip = 10/max(10, sqrt(*sqnorm_dev)) read on device
        (or `clip_scale` when sqnorm_dev is None).  theta: float32/float64 device tensor, phi: float32."""
        if phi.dtype != torch.float32 or not phi.is_contiguous() or not theta.is_contiguous():
            raise ValueError("apply_ needs contiguous tensors and a float32 phi")
        if theta.shape != phi.shape:
            raise ValueError("theta %s and phi %s differ in shape" % (tuple(theta.shape), tuple(phi.shape)))
        self._launch(theta, phi, theta.dtype, sqnorm_dev, clip_scale, None)

    def state_dict(self):
        """Host copy of everything needed to resume (the reference keeps these as plain attributes)."""
        out = dict(learning_rate=self.learning_rate, decay=self.decay, n_iters=self.n_iters)
        for k, v in self._state_tensors().items():
            out[k] = None if v is None else v.detach().cpu().numpy()
        return out

    def load_state_dict(self, state, device="cuda"):
        self.learning_rate, self.decay, self.n_iters = state["learning_rate"], state["decay"], state["n_iters"]
        for k in self._state_tensors():
            v = state.get(k)
            setattr(self, "_" + k, None if v is None else torch.as_tensor(v).to(device))
