"""Kernel base class: pairwise squared distances and the median-heuristic bandwidth on device.

Counterpart of stein/kernels/abstract_kernel.py:17-63.  The reference builds a TF1
graph with n placeholders; here the constructor only records ``n_particles`` and
the numbers are produced per call by libsteinhip (rownorms -> fp32-MFMA distance
pass -> exact radix-select median -> h^2).  ``sess`` is accepted for signature
compatibility and ignored.  ``bandwidth`` holds the last computed value (a float),
standing in for the reference's TF tensor of the same name.
"""
from abc import abstractmethod

import numpy as np
import torch

from ..engine import SvgdEngine


def as_device_matrix(x, device, name="theta"):
    """-> (float32 contiguous [n, d] device tensor, came_from_numpy)."""
    if isinstance(x, torch.Tensor):
        was_numpy = False
        t = x
    else:
        was_numpy = True
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float64)))
    if t.dim() != 2:
        raise ValueError("%s must be a [n_particles, n_params] matrix, got shape %s" % (name, tuple(t.shape)))
    return t.to(device=device, dtype=torch.float32).contiguous(), was_numpy


class AbstractKernel:
    def __init__(self, n_particles, sess=None, device="cuda"):
        self.n_particles = int(n_particles)
        self.sess = sess
        self.device = torch.device(device)
        self.bandwidth = None
        self._engine = None

    def _engine_for(self, d):
        if self._engine is None or self._engine.d != d:
            self._engine = SvgdEngine(self.n_particles, d, device=self.device)
        return self._engine

    def squared_distances(self, theta):
        """D = r + r^T - 2 theta theta^T in fp32 (abstract_kernel.py:33-35); returns [n, n]."""
        T, was_numpy = as_device_matrix(theta, self.device)
        self._check_n(T)
        eng = self._engine_for(T.shape[1])
        st = eng.stages
        st.rownorms(T, eng.n, eng.d, eng.rownorm)
        st.distance_block(T, eng.rownorm, eng.n, eng.d, 0, eng.n, eng.dist, eng.ld_dist)
        eng.dist_upper, eng._have_dist = False, True      # a full (fp32-MFMA, non-symmetric) image
        D = eng.dist_matrix()
        return D.cpu().numpy() if was_numpy else D

    def _check_n(self, T):
        if T.shape[0] != self.n_particles:
            raise ValueError("theta has %d rows but the kernel was built for %d particles" %
                             (T.shape[0], self.n_particles))

    @abstractmethod
    def kernel_and_grad(self, theta):
        raise NotImplementedError()
