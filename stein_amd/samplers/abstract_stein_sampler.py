"""SVGD driver with device-resident particles.

Counterpart of stein/samplers/abstract_stein_sampler.py:7-168.  What stays the same:
the constructor arguments, ``compute_phi(theta_array, grads_array)``,
``update_particles(grads_array)``, ``function_posterior(func, feed_dict, axis)``,
``train_on_batch(batch_feed)``, the ``theta`` dictionary keyed by model variable and
the N(0, 0.01^2) initialisation (:69-74).

What differs, deliberately: there is no TensorFlow session.  Particles live in ONE
contiguous [n, d] device tensor (``theta_matrix``); ``theta`` is a dict of zero-copy
views into it (columns ordered by sorted variable name, stein/utilities/converters.py:40).
``log_p`` is a callable instead of a TF tensor (see SteinSampler).  K is never
materialised and nothing on the update path synchronises with the host.

With a ``group`` (torch.distributed) each rank holds n/P particles; the engine gathers
rows once per step and reduces histograms and |phi|^2 (see engine.py).
"""
from abc import abstractmethod

import numpy as np
import torch

from ..engine import SvgdEngine
from ..utilities.converters import convert_array_to_dictionary, convert_dictionary_to_array


class AbstractSteinSampler:
    INIT_SCALE = 0.01  # abstract_stein_sampler.py:72

    def __init__(self, n_particles, log_p, theta=None, *, model_vars=None, device="cuda", dtype=torch.float32,
                 group=None, seed=None, kernel_dtype=torch.float32, x3=None):
        """
        n_particles : total number of particles n (across all ranks).
        log_p       : see SteinSampler.
        theta       : initial particles -- a dict {variable: [n_local, *shape]} (reference form) or a packed
                      [n_local, d] matrix; NumPy or torch.  None -> N(0, 0.01^2) draws, which needs model_vars.
        model_vars  : {name: shape} (or a list of objects with .name / .get_shape()) describing the parameters
                      of one particle; required when theta is None or a bare matrix that should be exposed as a dict.
        dtype       : storage type of particles and optimizer state: float32 (device-native) or float64 (the
                      reference's host dtype; the kernel path still runs in fp32 exactly as the reference's
                      fp32 placeholders do, stein/kernels/abstract_kernel.py:31).
        kernel_dtype: torch.float32 (default) or torch.bfloat16 -- what theta and the score are rounded to when they
                      are fed to the kernel / contraction (the reference rounds fp64 -> fp32 at that point).
        x3          : None (default: split-fp16 GEMMs on the 16-bit matrix cores) / False (fp32-input MFMA GEMMs); see engine.SvgdEngine.
        """
        self.n_particles = int(n_particles)
        self.log_p = log_p
        self.device = torch.device(device)
        self.dtype = dtype
        if dtype not in (torch.float32, torch.float64):
            raise ValueError("dtype must be torch.float32 or torch.float64")
        world = 1
        if group is not None:
            import torch.distributed as dist
            world = dist.get_world_size(group)
        if self.n_particles % world:
            raise ValueError("n_particles must be divisible by the number of ranks")
        self.n_local = self.n_particles // world
        self._group = group

        self._shapes = None
        if model_vars is not None:
            if isinstance(model_vars, dict):
                self._shapes = {k: list(v) for k, v in model_vars.items()}
            else:
                self._shapes = {v: list(v.get_shape().as_list()) for v in model_vars}
        self.model_vars = list(self._shapes.keys()) if self._shapes else None

        if theta is None:
            if not self._shapes:
                raise ValueError("theta=None needs model_vars={name: shape} to size the particles")
            if seed is not None:
                # every rank draws the FULL [n, ...] arrays from the same seeded stream and keeps its own rows: the
                # gathered particles equal the single-rank draw, and no two ranks start from the same particles (with
                # one stream per rank and the same seed, all ranks would hold identical copies and, SVGD being
                # deterministic, keep them forever)
                gen = np.random.default_rng(seed)
                rank = 0
                if group is not None:
                    rank = dist.get_rank(group)
                lo = rank * self.n_local
                theta = {v: (gen.normal(size=[self.n_particles] + s) * self.INIT_SCALE)[lo:lo + self.n_local]
                         for v, s in self._shapes.items()}
            else:   # the reference's unseeded global stream (abstract_stein_sampler.py:69-74); processes differ
                theta = {v: np.random.normal(size=[self.n_local] + s) * self.INIT_SCALE for v, s in self._shapes.items()}
        if isinstance(theta, dict):
            packed, self._access = convert_dictionary_to_array(theta)
            if self._shapes is None:
                self._shapes = {v: list(theta[v].shape[1:]) for v in theta}
                self.model_vars = list(self._shapes.keys())
        else:
            packed = theta
            if self._shapes:
                self._access, at = {}, 0
                for v in sorted(self._shapes, key=lambda k: k if isinstance(k, str) else k.name):
                    w = int(np.prod(self._shapes[v])) if self._shapes[v] else 1
                    self._access[v] = (at, at + w)
                    at += w
            else:
                self._access = None
        packed = torch.as_tensor(np.asarray(packed) if not isinstance(packed, torch.Tensor) else packed)
        if packed.dim() != 2 or packed.shape[0] != self.n_local:
            raise ValueError("theta must pack to [%d, d], got %s" % (self.n_local, tuple(packed.shape)))
        self.theta_matrix = packed.to(device=self.device, dtype=dtype).contiguous()
        self.n_params = self.theta_matrix.shape[1]
        self.kernel_dtype = kernel_dtype
        self.engine = SvgdEngine(self.n_particles, self.n_params, device=self.device, group=group, x3=x3,
                                 dtype=kernel_dtype)
        self._theta32 = (self.theta_matrix if dtype == kernel_dtype else
                         torch.empty(self.n_local, self.n_params, dtype=kernel_dtype, device=self.device))

    # -- particle access ---------------------------------------------------------------------------
    @property
    def theta(self):
        """{variable: view [n_local, *shape]} over the packed device matrix (no copies)."""
        if self._access is None:
            raise AttributeError("the sampler was given a bare matrix and no model_vars; use theta_matrix / samples")
        return convert_array_to_dictionary(self.theta_matrix, self._access, self._shapes)

    def _theta_f32(self):
        if self.dtype != self.kernel_dtype:
            self._theta32.copy_(self.theta_matrix)  # the reference's fp64 -> fp32 feed (squared_exponential_kernel.py:27)
        return self._theta32

    def _score_to_device(self, grads_array):
        g = grads_array
        if isinstance(g, dict):
            g, _ = convert_dictionary_to_array(g)
        if not isinstance(g, torch.Tensor):
            g = np.asarray(g)
            if g.dtype not in (np.float32, np.float64):     # the reference hands over float64; float32 goes up as it is
                g = g.astype(np.float64)
            g = torch.from_numpy(np.ascontiguousarray(g))
        if tuple(g.shape) != (self.n_local, self.n_params):
            raise ValueError("score must be [%d, %d], got %s" % (self.n_local, self.n_params, tuple(g.shape)))
        return g.to(device=self.device, dtype=self.kernel_dtype).contiguous()

    # -- the hot path ---------------------------------------------------------------------------------
    def _foreign_kernel(self):
        """The reference's kernel seam (abstract_stein_sampler.py:103): any object with ``kernel_and_grad(theta) ->
        (K, dK)`` drops in as ``sampler.kernel``.  Returns it when it is NOT this package's fused RBF kernel (whose K the
        sampler never materialises), else None."""
        k = getattr(self, "kernel", None)
        if k is None:
            return None
        from ..kernels import SquaredExponentialKernel
        return None if isinstance(k, SquaredExponentialKernel) else k

    def compute_phi(self, theta_array, grads_array):
        """phi = (K . grads + dK) / n for the given particles (abstract_stein_sampler.py:100-105).

        NumPy in -> float64 NumPy out (values carry fp32 precision); device tensors in -> float32 tensor out.
        With a user-supplied ``self.kernel`` the reference's own lines run instead: ``K, dK =
        self.kernel.kernel_and_grad(theta_array)``; ``(K.dot(grads_array) + dK) / n_particles`` in NumPy.
        """
        was_numpy = not isinstance(theta_array, torch.Tensor)
        foreign = self._foreign_kernel()
        if foreign is not None:
            if self._group is not None:
                raise ValueError("a user-supplied kernel needs all particles on one rank")
            th = theta_array if was_numpy else theta_array.detach().double().cpu().numpy()
            g = grads_array
            if isinstance(g, dict):
                g, _ = convert_dictionary_to_array(g)
            g = g.detach().double().cpu().numpy() if isinstance(g, torch.Tensor) else np.asarray(g)
            n_particles, n_params = g.shape                                  # abstract_stein_sampler.py:100
            K, dK = foreign.kernel_and_grad(th)                              # :103
            K, dK = (x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x) for x in (K, dK))
            phi = (K.dot(g) + dK) / n_particles                              # :105
            return phi if was_numpy else torch.as_tensor(phi).to(device=self.device, dtype=torch.float32)
        T = theta_array if not was_numpy else torch.from_numpy(np.ascontiguousarray(np.asarray(theta_array, dtype=np.float64)))
        T = T.to(device=self.device, dtype=self.kernel_dtype).contiguous()
        G = self._score_to_device(grads_array)
        phi = self.engine.compute_phi(T, G)
        return phi.double().cpu().numpy() if was_numpy else phi.clone()

    def update_particles(self, grads_array):
        """One SVGD step from the score matrix: phi, norm clip, optimizer apply
        (abstract_stein_sampler.py:121-127), all on device."""
        if self._foreign_kernel() is not None or not hasattr(self.gd, "apply_"):
            return self._update_particles_by_the_seams(grads_array)
        G = self._score_to_device(grads_array)
        phi = self.engine.compute_phi(self._theta_f32(), G)
        self.gd.apply_(self.theta_matrix, phi, self.engine.sqnorm)

    def _update_particles_by_the_seams(self, grads_array):
        """The reference's duck-typed seams, line for line (abstract_stein_sampler.py:121-127): taken when ``self.gd`` is a
        reference-style optimizer (only ``update(phi) -> step``, no fused ``apply_``) or ``self.kernel`` is a user's
        object (only ``kernel_and_grad``).  phi still comes from the HIP engine unless the kernel is foreign; the clip and
        the optimizer run on the host in float64 as the reference's NumPy does."""
        g = grads_array
        if isinstance(g, dict):
            g, _ = convert_dictionary_to_array(g)
        g = g.detach().double().cpu().numpy() if isinstance(g, torch.Tensor) else np.asarray(g, dtype=np.float64)
        theta_array = self.theta_matrix.detach().double().cpu().numpy()      # :121 (a float64 copy, as the reference packs)
        phi = self.compute_phi(theta_array, g)                               # :123
        # :125 -- on one rank NumPy's norm of phi, as the reference; sharded, phi holds this rank's rows and the norm is the
        # engine's all-reduced one
        norm = np.linalg.norm(phi) if self._group is None else float(self.engine.sqnorm.sqrt().item())
        phi *= 10. / max(10., norm)
        theta_array += self.gd.update(phi)                                   # :126
        self.theta_matrix.copy_(torch.as_tensor(theta_array).to(self.theta_matrix))   # :127

    def function_posterior(self, func, feed_dict=None, axis=None, gather=True):
        """Evaluate `func` under every particle (abstract_stein_sampler.py:157-168).

        func(theta_dict_or_matrix, feed_dict) must be batched over particles and return [n_local, ...]; the
        result is flattened per particle to [n_local, out] like the reference's np.ravel.  Sharded sampler: every rank
        evaluates its own particles and the rows are all-gathered in rank order (rank p holds particles
        [p n/P, (p + 1) n/P), so the gathered array is indexed like the reference's, all n particles: :160-162), unless
        gather=False.  The mean over `axis` is taken of that [n, out] array, as the reference does (:165-168).  Returns
        NumPy, the same array on every rank.  Collective when sharded: every rank must call it.
        """
        arg = self.theta if self._access is not None else self.theta_matrix
        with torch.no_grad():
            out = func(arg, feed_dict)
        out = torch.as_tensor(out).reshape(self.n_local, -1).detach()
        if self._group is not None and gather:
            import torch.distributed as dist
            world = dist.get_world_size(self._group)
            # (the per-particle output width is the same on every rank: the same func on the same shapes)
            mine = out.to(self.device).contiguous()
            full = torch.empty(world * self.n_local, mine.shape[1], dtype=mine.dtype, device=mine.device)
            dist.all_gather_into_tensor(full, mine, group=self._group)
            out = full
        out = out.cpu().numpy()
        return out.mean(axis=axis) if axis is not None else out

    def samples_all(self):
        """All n particles as one float64 [n, d] NumPy array on every rank (the reference's `samples`, stein_sampler.py:73-78,
        for a sharded sampler; on one rank it equals `samples`).  Collective when sharded."""
        if self._group is None:
            return self.theta_matrix.detach().double().cpu().numpy()
        import torch.distributed as dist
        world = dist.get_world_size(self._group)
        full = torch.empty(world * self.n_local, self.n_params, dtype=self.theta_matrix.dtype, device=self.device)
        dist.all_gather_into_tensor(full, self.theta_matrix.contiguous(), group=self._group)
        return full.double().cpu().numpy()

    @abstractmethod
    def train_on_batch(self, batch_feed):
        raise NotImplementedError()

    # -- save / restore (the reference leaves this to pickling its attributes) -----------------------
    def state_dict(self):
        return {"theta": self.theta_matrix.detach().cpu().numpy(), "gd": self.gd.state_dict()}

    def load_state_dict(self, state):
        self.theta_matrix.copy_(torch.as_tensor(state["theta"]).to(self.theta_matrix))
        self.gd.load_state_dict(state["gd"], device=self.device)
