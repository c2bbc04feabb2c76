#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference itself.

Run in the build container only (it reads /root/reference, which does not
exist on the GPU box):   python tests/golden/make_golden.py

What is driven, and how
-----------------------
* ``stein.optimizers`` (NumPy only) is imported as-is and stepped ->  G1.
* ``stein/utilities/converters.py`` (NumPy only) is loaded by file path -> G4.
* ``AbstractSteinSampler.compute_phi`` / ``update_particles`` / ``samples`` are
  pure NumPy methods, but their module does ``import tensorflow`` at the top
  and TensorFlow is not installed.  An EMPTY module object is registered under
  that name so the import statement succeeds; it provides no attribute and no
  TensorFlow behaviour is emulated.  Instances are made with
  ``object.__new__`` (the constructor opens a tf.Session) and given the
  attributes the NumPy methods read.  The `.kernel` they call is the oracle's
  restated kernel, so G2/G3 pin `/n`, the K.G contraction, dtype promotion,
  the norm clip and the optimizer hand-off -- not the TF kernel graph (G5 is
  labelled "restated, TF-unpinned").
* G6 copies the reference's linear-regression CSV *data* and stores the
  closed-form posterior of that model as the end-to-end known answer.

Outputs are small .npz files; inputs, outputs and seeds are all stored so the
tests never need the reference.
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from oracle import svgd_oracle as orc  # noqa: E402


class StandInVariable:
    """Exposes only what converters.py:40,86 touch: .name and .get_shape().as_list()."""

    def __init__(self, name, shape):
        self.name, self._shape = name, list(shape)

    def get_shape(self):
        return types.SimpleNamespace(as_list=lambda: list(self._shape))


class RestatedKernel:
    def __init__(self, dtype=np.float32):
        self.dtype = dtype

    def kernel_and_grad(self, theta):
        return orc.kernel_and_grad(theta, self.dtype)


def load_reference():
    from stein.optimizers import AdagradGradientDescent, AdamGradientDescent
    spec = importlib.util.spec_from_file_location(
        "ref_converters", os.path.join(REF, "stein/utilities/converters.py"))
    conv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(conv)
    if "tensorflow" not in sys.modules:
        sys.modules["tensorflow"] = types.ModuleType("tensorflow")  # empty, see docstring
    from stein.samplers.stein_sampler import SteinSampler
    return AdagradGradientDescent, AdamGradientDescent, conv, SteinSampler


def make_sampler(SteinSampler, n, gd, theta_dict):
    s = object.__new__(SteinSampler)
    s.n_particles, s.gd, s.theta = n, gd, theta_dict
    s.kernel = RestatedKernel()
    return s


def g1_optimizers(Adagrad, Adam, out):
    rng = np.random.default_rng(11)
    phis = rng.normal(size=(5, 8, 5))
    ada, adam = Adagrad(0.1), Adam(0.1, decay=0.999)
    ada_steps, ada_hist, adam_steps, adam_mu, adam_nu, adam_lr = [], [], [], [], [], []
    for p in phis:
        ada_steps.append(ada.update(p.copy()))
        ada_hist.append(np.array(ada.hist))
        adam_steps.append(adam.update(p.copy()))
        adam_mu.append(np.array(adam.mu)); adam_nu.append(np.array(adam.nu))
        adam_lr.append(adam.learning_rate)
    np.savez(os.path.join(out, "g1_optimizers.npz"), phis=phis,
             adagrad_steps=np.array(ada_steps), adagrad_hist=np.array(ada_hist),
             adagrad_lr_final=ada.learning_rate, adagrad_n_iters=ada.n_iters,
             adam_steps=np.array(adam_steps), adam_mu=np.array(adam_mu),
             adam_nu=np.array(adam_nu), adam_lr=np.array(adam_lr),
             adam_n_iters=adam.n_iters)


def g2_phi(SteinSampler, Adam, out):
    data = {}
    for n, d in [(7, 3), (8, 5), (100, 10)]:
        rng = np.random.default_rng(100 * n + d)
        T, G = rng.normal(size=(n, d)), rng.normal(size=(n, d))
        v = StandInVariable("model/w:0", [d])
        s = make_sampler(SteinSampler, n, Adam(), {v: T.copy()})
        phi = s.compute_phi(T, G)
        assert phi.dtype == np.float64
        data[f"T_{n}x{d}"], data[f"G_{n}x{d}"], data[f"phi_{n}x{d}"] = T, G, phi
    np.savez(os.path.join(out, "g2_compute_phi.npz"), **data)


def g3_update(SteinSampler, Adagrad, Adam, out):
    data = {}
    n, d = 100, 10
    for tag, gscale in [("noclip", 1.0), ("clip", 400.0)]:
        for oname, mk in [("adagrad", lambda: Adagrad(0.05)),
                          ("adam", lambda: Adam(0.05, decay=0.99))]:
            rng = np.random.default_rng({"noclip": 1, "clip": 2}[tag] * 10
                                        + {"adagrad": 1, "adam": 2}[oname])
            T0 = rng.normal(size=(n, d))
            Gs = rng.normal(size=(3, n, d)) * gscale
            v = StandInVariable("model/w:0", [d])
            s = make_sampler(SteinSampler, n, mk(), {v: T0.copy()})
            traj, norms = [], []
            for G in Gs:
                theta_before = s.samples.copy()
                norms.append(np.linalg.norm(s.compute_phi(theta_before, G)))
                s.update_particles(G)
                traj.append(s.samples.copy())
            key = f"{tag}_{oname}"
            data[key + "_T0"], data[key + "_G"] = T0, Gs
            data[key + "_theta"] = np.array(traj)
            data[key + "_phi_norm"] = np.array(norms)
    assert data["clip_adam_phi_norm"].min() > 10 and data["noclip_adam_phi_norm"].max() < 10
    np.savez(os.path.join(out, "g3_update_particles.npz"), **data)


def g4_converters(conv, out):
    rng = np.random.default_rng(4)
    n = 6
    vb = StandInVariable("model/zeta:0", [3, 1])
    va = StandInVariable("model/alpha:0", [])
    vm = StandInVariable("model/mid:0", [2, 2])
    d = {vb: rng.normal(size=(n, 3, 1)), va: rng.normal(size=(n,)),
         vm: rng.normal(size=(n, 2, 2))}
    arr, access = conv.convert_dictionary_to_array(d)
    back = conv.convert_array_to_dictionary(arr, access)
    for v in d:
        assert np.array_equal(back[v], d[v])
    np.savez(os.path.join(out, "g4_converters.npz"),
             zeta=d[vb], alpha=d[va], mid=d[vm], array=arr,
             access_zeta=np.array(access[vb]), access_alpha=np.array(access[va]),
             access_mid=np.array(access[vm]))


def g5_kernel(out):
    """restated, TF-unpinned: outputs of the oracle's own kernel restatement."""
    data = {}
    for n, d in [(7, 3), (8, 5), (100, 10), (257, 33)]:
        rng = np.random.default_rng(5000 + n)
        T = rng.normal(size=(n, d))
        K, dK, h2 = orc.kernel_and_grad(T, np.float32, return_h2=True)
        K64, dK64, h264 = orc.kernel_and_grad(T, np.float64, return_h2=True)
        data[f"T_{n}x{d}"] = T
        data[f"dK_{n}x{d}"], data[f"h2_{n}x{d}"] = dK, h2
        data[f"dK64_{n}x{d}"], data[f"h264_{n}x{d}"] = dK64, h264
        if n <= 100:   # keep the fixture small: full K only for the small cases
            data[f"K_{n}x{d}"], data[f"K64_{n}x{d}"] = K, K64
    np.savez_compressed(os.path.join(out, "g5_kernel_restated.npz"), **data)


def g6_linear_regression(out):
    X = np.loadtxt(os.path.join(REF, "examples/linear_regression/data/data_X.csv"), delimiter=",")
    y = np.loadtxt(os.path.join(REF, "examples/linear_regression/data/data_y.csv"), delimiter=",")
    w = np.loadtxt(os.path.join(REF, "examples/linear_regression/data/data_w.csv"), delimiter=",")
    X = np.atleast_2d(X).T if X.ndim == 1 else X
    prec = X.T @ X + np.eye(X.shape[1])          # unit-variance likelihood, N(0,1) prior
    mean = np.linalg.solve(prec, X.T @ y)        # examples/linear_regression/main.py:25-31
    np.savez(os.path.join(out, "g6_linear_regression.npz"), X=X, y=y, w_true=w,
             post_precision=prec, post_mean=mean, post_std=np.sqrt(1.0 / np.diag(prec)))


def main():
    Adagrad, Adam, conv, SteinSampler = load_reference()
    g1_optimizers(Adagrad, Adam, HERE)
    g2_phi(SteinSampler, Adam, HERE)
    g3_update(SteinSampler, Adagrad, Adam, HERE)
    g4_converters(conv, HERE)
    g5_kernel(HERE)
    g6_linear_regression(HERE)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
