"""CPU oracle for the SVGD particle-update path  --  TEST INFRASTRUCTURE ONLY.

This module is a NumPy restatement of the arithmetic on the reference's hot
path.  It exists to *check* the HIP implementation; nothing under
``stein_amd/`` may import it.  Only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` use it.

Pinning status
--------------
* ``adagrad_update`` / ``adam_update`` / ``compute_phi`` / ``update_particles``
  / the converters are pinned against golden vectors produced by running the
  reference's own NumPy methods in the build container
  (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
* ``pairwise_sq_dists`` / ``median_all`` / ``bandwidth_sq`` /
  ``kernel_and_grad`` restate arithmetic that the reference executes inside
  TensorFlow 1.12 (``tensorflow-gpu==1.12.0``, stein/requirements.txt:25), which
  is not present anywhere in this pipeline and for which the reference holds
  no test vectors:  **parity unpinned** for those four functions.  They are
  cross-checked instead against central finite differences of sum(K) and
  against ``np.median`` (tests/test_oracle.py).

Each function cites the reference file:line it follows (paths relative to
/root/reference).  ``dtype`` selects the "faithful" flow (fp32 kernel, fp64
contraction, exactly the reference's promotion chain) or an all-fp64 twin.
"""
import numpy as np

CLIP_NORM = 10.0          # stein/samplers/abstract_stein_sampler.py:125
ADAGRAD_EPS = 1e-6        # stein/optimizers/adagrad_gradient_descent.py:44
ADAM_EPS = 1e-8           # stein/optimizers/adam_gradient_descent.py:55


# --------------------------------------------------------------------------
# kernel side (TF graph in the reference)
# --------------------------------------------------------------------------
def pairwise_sq_dists(theta, dtype=np.float32):
    """D_ij = r_i + r_j - 2 <theta_i, theta_j>.

    stein/kernels/abstract_kernel.py:33-35 (stack -> r -> r + r^T - 2 T T^T),
    evaluated in fp32 there because the placeholders are tf.float32 (:31).
    """
    T = np.ascontiguousarray(theta, dtype=dtype)
    r = np.sum(T * T, axis=1).reshape(-1, 1)
    return r + r.T - dtype(2) * (T @ T.T)


def median_all(D):
    """Exact median over every entry of D (diagonal and both triangles).

    stein/utilities/compute_median.py:7-15: flatten, take the top m = dim//2+1
    values; even dim -> mean of the two smallest of those (the two middle order
    statistics), odd dim -> the smallest of those (the middle one).  The mean
    of two fp32 values is taken in fp32 (tf.reduce_mean on an fp32 tensor).
    """
    V = np.asarray(D).reshape(-1)
    dim = V.shape[0]
    if dim % 2 == 0:
        lo, hi = dim // 2 - 1, dim // 2
        part = np.partition(V, (lo, hi))
        two = part[[lo, hi]]
        return two.dtype.type(two.mean(dtype=two.dtype))
    mid = dim // 2
    return np.partition(V, mid)[mid]


def bandwidth_sq(med, n, dtype=np.float32):
    """h^2 as the reference's graph produces it: square(sqrt(med / ln n)).

    stein/kernels/abstract_kernel.py:40 (bandwidth = sqrt(m / np.log(n))) then
    squared again at stein/kernels/squared_exponential_kernel.py:22.
    """
    bw = np.sqrt(dtype(med) / dtype(np.log(n)))
    return dtype(bw * bw)


def kernel_and_grad(theta, dtype=np.float32, return_h2=False):
    """(K, dK) as SquaredExponentialKernel.kernel_and_grad returns them.

    stein/kernels/squared_exponential_kernel.py:22-35.
      K  = exp(-D / bw^2 / 2)                                            (:22)
      dK = -0.5 * d(sum K)/d(theta)   with the bandwidth held constant   (:23,:32)
         = (rowsum(K) * theta - K @ theta) / bw^2
    The closed form is what tf.gradients evaluates for this graph: with
    A = dsumK/dD = -K / (2 bw^2) (symmetric), the chain rule through
    D = r + r^T - 2 T T^T gives 2 T (rowsum A + colsum A) - 2 (A + A^T) T.
    """
    T = np.ascontiguousarray(theta, dtype=dtype)
    n = T.shape[0]
    D = pairwise_sq_dists(T, dtype)
    h2 = bandwidth_sq(median_all(D), n, dtype)
    K = np.exp(-D / h2 / dtype(2))
    A = -K / (dtype(2) * h2)
    grad = dtype(2) * T * (A.sum(axis=1) + A.sum(axis=0))[:, None] \
        - dtype(2) * ((A + A.T) @ T)
    dK = dtype(-0.5) * grad
    if return_h2:
        return K, dK, h2
    return K, dK


# --------------------------------------------------------------------------
# sampler side (NumPy in the reference)
# --------------------------------------------------------------------------
def compute_phi(theta_array, grads_array, dtype=np.float32):
    """phi = (K . grads + dK) / n.

    stein/samplers/abstract_stein_sampler.py:100-105.  K, dK arrive as `dtype`
    (fp32 in the reference) and the product with the fp64 score matrix
    promotes to fp64.
    """
    n = grads_array.shape[0]
    K, dK = kernel_and_grad(theta_array, dtype)
    return (K.dot(np.asarray(grads_array, dtype=np.float64)) + dK) / n


def clip_scale(sqnorm):
    """10 / max(10, ||phi||_F)  (stein/samplers/abstract_stein_sampler.py:125)."""
    return CLIP_NORM / max(CLIP_NORM, float(np.sqrt(sqnorm)))


class AdagradState:
    """State of stein/optimizers/adagrad_gradient_descent.py (:13-44)."""

    def __init__(self, learning_rate=1e-3, decay=1.0, alpha=0.9):
        self.learning_rate, self.decay, self.alpha = learning_rate, decay, alpha
        self.n_iters = 0
        self.hist = None

    def update(self, phi):
        # :37-44 -- first call hist = phi^2, later an EMA; `decay` is stored
        # (abstract_gradient_descent.py:28-30) and never applied.
        if self.n_iters == 0:
            self.hist = phi ** 2
        else:
            self.hist = self.alpha * self.hist + (1.0 - self.alpha) * phi ** 2
        self.n_iters += 1
        return phi / (ADAGRAD_EPS + np.sqrt(self.hist)) * self.learning_rate


class AdamState:
    """State of stein/optimizers/adam_gradient_descent.py (:15-58)."""

    def __init__(self, learning_rate=1e-3, decay=1.0, beta_1=0.9, beta_2=0.999):
        self.learning_rate, self.decay = learning_rate, decay
        self.beta_1, self.beta_2 = beta_1, beta_2
        self.n_iters = 0
        self.mu = self.nu = None

    def update(self, phi):
        # :45-58 -- first call mu = phi, nu = phi^2 (no (1-beta) factor), bias
        # correction, step, then learning_rate *= decay.
        if self.n_iters == 0:
            self.mu, self.nu = phi, phi ** 2
        else:
            self.mu = self.beta_1 * self.mu + (1.0 - self.beta_1) * phi
            self.nu = self.beta_2 * self.nu + (1.0 - self.beta_2) * phi ** 2
        self.n_iters += 1
        mup = self.mu / (1.0 - self.beta_1 ** self.n_iters)
        nup = self.nu / (1.0 - self.beta_2 ** self.n_iters)
        step = mup / (ADAM_EPS + np.sqrt(nup)) * self.learning_rate
        self.learning_rate *= self.decay
        return step


def update_particles(theta_array, grads_array, gd, dtype=np.float32):
    """One SVGD step on the packed [n, d] matrix; returns (theta_new, phi_clipped).

    stein/samplers/abstract_stein_sampler.py:121-127 minus the dict packing.
    """
    phi = compute_phi(theta_array, grads_array, dtype)
    phi = phi * clip_scale(np.sum(phi * phi))
    return np.asarray(theta_array, dtype=np.float64) + gd.update(phi), phi


# --------------------------------------------------------------------------
# dict <-> matrix packing (stein/utilities/converters.py)
# --------------------------------------------------------------------------
def pack_dictionary(dictionary, name_of=lambda v: v.name):
    """converters.py:30-55: columns ordered by sorted variable name (:40)."""
    keys = sorted(dictionary.keys(), key=name_of)
    n = next(iter(dictionary.values())).shape[0]
    cols, access, at = [], {}, 0
    for v in keys:
        block = np.reshape(dictionary[v], (n, -1))
        access[v] = (at, at + block.shape[1])
        at += block.shape[1]
        cols.append(block)
    return np.concatenate(cols, axis=1).astype(np.float64), access


def unpack_array(array, access, shape_of=lambda v: v.get_shape().as_list()):
    """converters.py:76-89."""
    n = array.shape[0]
    return {v: np.reshape(array[:, a:b], [n] + list(shape_of(v)))
            for v, (a, b) in access.items()}


# --------------------------------------------------------------------------
# whole step, used by bench.py's cpu_baseline leg and by the parity tests
# --------------------------------------------------------------------------
def svgd_step(theta, grads, gd, dtype=np.float32):
    """distance -> median -> K -> phi -> clip -> optimizer apply.  Returns a
    dict with every intermediate so the GPU stages can be compared one by one."""
    T = np.ascontiguousarray(theta, dtype=dtype)
    n = T.shape[0]
    D = pairwise_sq_dists(T, dtype)
    med = median_all(D)
    h2 = bandwidth_sq(med, n, dtype)
    K = np.exp(-D / h2 / dtype(2))
    dK = (K.sum(axis=1)[:, None] * T - K @ T) / h2
    phi = (K.dot(np.asarray(grads, dtype=np.float64)) + dK) / n
    sq = float(np.sum(phi * phi))
    phi_c = phi * clip_scale(sq)
    theta_new = np.asarray(theta, dtype=np.float64) + gd.update(phi_c)
    return dict(D=D, median=med, h2=h2, K=K, dK=dK, phi=phi, sqnorm=sq,
                phi_clipped=phi_c, theta_new=theta_new)


def svgd_step_rows(theta, grads, row0, m, gd, dtype=np.float32):
    """The same step restricted to the row block [row0, row0+m): distances of those rows to all n particles,
    the median of that block (a bounded stand-in for the n^2 select: same algorithm, m*n values), K rows,
    phi rows, clip with the block's norm, optimizer apply on those rows.  Used by bench.py to time a BOUNDED
    sample of a large workload on the host; every stage of the step is row-separable except the median,
    whose cost is proportional to the number of values selected over."""
    T = np.ascontiguousarray(theta, dtype=dtype)
    n = T.shape[0]
    Tb = T[row0:row0 + m]
    r = np.sum(T * T, axis=1)
    D = r[row0:row0 + m, None] + r[None, :] - dtype(2) * (Tb @ T.T)
    h2 = bandwidth_sq(median_all(D), n, dtype)
    K = np.exp(-D / h2 / dtype(2))
    dK = (K.sum(axis=1)[:, None] * Tb - K @ T) / h2
    phi = (K.dot(np.asarray(grads, dtype=np.float64)) + dK) / n
    phi = phi * clip_scale(np.sum(phi * phi))
    return np.asarray(theta[row0:row0 + m], dtype=np.float64) + gd.update(phi)
