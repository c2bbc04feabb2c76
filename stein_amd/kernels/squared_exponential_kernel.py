"""Isotropic RBF kernel with median-heuristic bandwidth, and its repulsion term, on MI355X.

Drop-in for stein/kernels/squared_exponential_kernel.py:18-35:
``SquaredExponentialKernel(n_particles, sess).kernel_and_grad(theta) -> (K, dK)`` with
K = exp(-D / bw^2 / 2) of shape [n, n] and dK = -0.5 * d(sum K)/d(theta) of shape [n, d],
both float32.  NumPy in -> NumPy out; torch device tensors in -> device tensors out.
The sampler does not call this (it never materialises K); it exists for callers that use
the kernel on its own.
"""
import torch

from .abstract_kernel import AbstractKernel, as_device_matrix


class SquaredExponentialKernel(AbstractKernel):
    def __init__(self, n_particles, sess=None, device="cuda"):
        super().__init__(n_particles, sess, device)

    def kernel_and_grad(self, theta):
        T, was_numpy = as_device_matrix(theta, self.device)
        self._check_n(T)
        n, d = T.shape
        eng = self._engine_for(d)
        K = torch.empty(n, n, dtype=torch.float32, device=self.device)
        dK = torch.empty(n, d, dtype=torch.float32, device=self.device)
        # dK does not depend on the score; feed theta itself as the score operand
        eng.compute_phi(T, T, K_out=K, dK_out=dK)
        self.bandwidth = float(eng.h2.sqrt().item())
        if was_numpy:
            return K.cpu().numpy(), dK.cpu().numpy()
        return K, dK
