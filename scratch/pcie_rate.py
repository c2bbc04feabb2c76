"""PCIe-inclusive step: the reference-style call update_particles(grads_array) with a HOST fp64 NumPy score matrix
(uploaded and cast every step) against the device-resident path, C3 shape."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd.samplers import SteinSampler
from stein_amd.optimizers import AdagradGradientDescent
n, d = 16384, 256
rng = np.random.default_rng(0)
T0 = rng.normal(size=(n, d)); G = rng.normal(size=(n, d))
for label, grads in (("host fp64 ndarray (pageable)", G), ("host fp32 ndarray (pageable)", G.astype(np.float32)),
                     ("host fp32 pinned tensor", torch.tensor(G, dtype=torch.float32).pin_memory()),
                     ("device fp32 tensor", torch.tensor(G, dtype=torch.float32, device="cuda"))):
    s = SteinSampler(n, None, AdagradGradientDescent(learning_rate=1e-3), theta=T0.copy(), device="cuda:0")
    for _ in range(3): s.update_particles(grads)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): s.update_particles(grads)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("%-32s %.3f ms/step  %.3g particle-updates/s" % (label, dt * 1e3, n / dt), flush=True)
