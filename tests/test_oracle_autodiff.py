"""CPU: an independent restatement of the reference's TF-1.12 kernel graph, op for op, in torch -- with the
repulsion term obtained by AUTODIFF, as the reference obtains it -- compared with the closed form the oracle
(and the HIP kernels) use.

The TF graph itself cannot run anywhere in this pipeline (parity for a1-a6 stays "unpinned").
What this test adds: the closed-form dK of oracle/svgd_oracle.py::kernel_and_grad is no longer checked only against
itself and finite differences, but against reverse-mode differentiation of the very graph the reference builds:

    theta_i placeholders -> tf.stack                      stein/kernels/abstract_kernel.py:30-33
    r = reduce_sum(T*T, 1) reshaped [-1, 1]               :34
    D = r + transpose(r) - 2 matmul(T, transpose(T))      :35
    V = reshape(D, [-1]); top_k(V, dim//2 + 1)            stein/utilities/compute_median.py:7-10
    even: mean of the last two of the top-k; odd: last    :12-15
    bandwidth = stop_gradient(sqrt(m / log n))            abstract_kernel.py:38-40
    K = exp(-D / bandwidth**2 / 2)                        stein/kernels/squared_exponential_kernel.py:22
    dK = tf.gradients(K, [theta_1 .. theta_n])            :23   (gradient of sum(K) w.r.t. each row)
    return K, -0.5 * vstack(dK)                           :28-35
"""
import numpy as np
import pytest
import torch

from oracle import svgd_oracle as orc


def tf_graph_in_torch(theta, dtype):
    n = theta.shape[0]
    rows = [torch.tensor(theta[i], dtype=dtype, requires_grad=True) for i in range(n)]   # n placeholders [d]
    T = torch.stack(rows)                                           # tf.stack(self.theta)
    r = torch.reshape(torch.sum(T * T, 1), [-1, 1])
    D = r + torch.transpose(r, 0, 1) - 2 * torch.matmul(T, torch.transpose(T, 0, 1))
    V = torch.reshape(D, [-1])
    dim = V.shape[0]
    m = dim // 2 + 1
    top = torch.topk(V, m).values                                   # tf.nn.top_k: the m LARGEST, descending
    med = torch.mean(top[-2:]) if dim % 2 == 0 else top[-1]
    bandwidth = torch.sqrt(med / np.log(n)).detach()                # tf.stop_gradient
    K = torch.exp(-D / bandwidth ** 2 / 2)
    grads = torch.autograd.grad(K.sum(), rows)                      # tf.gradients(K, theta): d(sum K)/d(theta_i)
    return K.detach().numpy(), -0.5 * np.vstack([g.numpy() for g in grads]), float(bandwidth) ** 2, D.detach().numpy()


@pytest.mark.parametrize("n,d", [(7, 3), (8, 5), (33, 4), (100, 10)])     # n*n odd: 7, 33; even: 8, 100
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_closed_form_equals_autodiff_of_the_reference_graph(n, d, dtype):
    rng = np.random.default_rng(100 * n + d)
    theta = rng.normal(size=(n, d))
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    K_t, dK_t, h2_t, D_t = tf_graph_in_torch(theta, tdt)
    K_o, dK_o, h2_o = orc.kernel_and_grad(theta, dtype, return_h2=True)
    D_o = orc.pairwise_sq_dists(theta, dtype)
    eps = np.finfo(dtype).eps
    # D: same formula; torch's matmul may sum in another order
    assert np.abs(D_t - D_o).max() <= 8 * eps * np.abs(D_o).max()
    # the median is an order statistic: top_k semantics == the oracle's partition semantics on the SAME values
    V = D_t.reshape(-1)
    assert orc.median_all(D_t) == (np.sort(V)[[V.size // 2 - 1, V.size // 2]].mean(dtype=V.dtype) if V.size % 2 == 0
                                   else np.sort(V)[V.size // 2])
    assert abs(h2_t - float(h2_o)) <= 64 * eps * float(h2_o)
    assert np.abs(K_t - K_o).max() <= 256 * eps
    scale = np.abs(dK_o).max()
    assert np.abs(dK_t - dK_o).max() <= (2e-12 if dtype == np.float64 else 2e-5) * scale
    # and the gradient really is that of sum(K) with the bandwidth frozen: rows of dK sum to zero (antisymmetry)
    assert np.abs(dK_t.sum(0)).max() <= (1e-11 if dtype == np.float64 else 1e-4) * scale * n


def test_median_semantics_match_top_k_for_ties_and_negative_zero():
    """top_k on a tensor with repeated values and a computed (not assumed) diagonal: same answer as the oracle."""
    for vals in ([0.0, 0.0, 1.0, 1.0], [3.0, -0.0, 0.0, 2.0, 2.0], [5.0] * 9, [1.0, 2.0]):
        V = torch.tensor(vals, dtype=torch.float32)
        dim = V.numel()
        top = torch.topk(V, dim // 2 + 1).values
        med = torch.mean(top[-2:]) if dim % 2 == 0 else top[-1]
        assert float(med) == float(orc.median_all(np.asarray(vals, dtype=np.float32)))


def test_phi_from_autodiff_kernel_equals_oracle_phi():
    """compute_phi (stein/samplers/abstract_stein_sampler.py:100-105) on top of the autodiff kernel == the oracle's."""
    n, d = 24, 6
    rng = np.random.default_rng(3)
    theta, grads = rng.normal(size=(n, d)), rng.normal(size=(n, d))
    K, dK, _, _ = tf_graph_in_torch(theta, torch.float64)
    phi = (K.dot(grads) + dK) / n
    assert np.abs(phi - orc.compute_phi(theta, grads, np.float64)).max() <= 1e-13 * np.abs(phi).max() + 1e-15
