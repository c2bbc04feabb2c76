"""GPU: the C ABI driven by a plain C11 host program -- no Python, no PyTorch on the product side
(examples/c_host/svgd_steps.c: hipMalloc'ed buffers, caller-owned workspace and stream, stein_workspace_bytes /
stein_workspace_layout / stein_svgd_phi / stein_apply_adagrad).  The program's output is compared with the oracle's
update_particles sequence (stein/samplers/abstract_stein_sampler.py:107-127 with AdagradGradientDescent) on the same
inputs."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def build(tmp_path):
    import __graft_entry__ as ge
    return ge.build_c_host(str(tmp_path / "svgd_steps"))


@pytest.mark.parametrize("n,d,steps", [(100, 10, 4), (300, 20, 4), (1000, 130, 3)])
def test_c_host_program_matches_oracle(cuda, tmp_path, n, d, steps):
    from oracle import svgd_oracle as orc
    exe = build(tmp_path)
    rng = np.random.default_rng(n + d)
    T = rng.normal(size=(n, d)).astype(np.float32)
    G = rng.normal(size=(n, d)).astype(np.float32)
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    np.concatenate([T.ravel(), G.ravel()]).tofile(inp)
    res = subprocess.run([exe, inp, outp, str(n), str(d), str(steps)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    raw = np.fromfile(outp, dtype=np.uint8)
    cnt = n * d
    phi = raw[:4 * cnt].view(np.float32).reshape(n, d)
    theta = raw[4 * cnt:8 * cnt].view(np.float32).reshape(n, d)
    h2 = raw[8 * cnt:8 * cnt + 4 * steps].view(np.float32)
    sq = raw[8 * cnt + 4 * steps:].view(np.float64)
    assert sq.shape == (steps,)
    gd = orc.AdagradState(learning_rate=1e-3, alpha=0.9)
    th32 = T.copy()                                                # the program keeps theta in fp32 on the device
    for s in range(steps):
        ref = orc.svgd_step(th32, G.astype(np.float64), gd, np.float32)
        assert abs(h2[s] - ref["h2"]) <= 2e-6 * ref["h2"], (s, h2[s], ref["h2"])
        assert abs(sq[s] - ref["sqnorm"]) <= 1e-5 * ref["sqnorm"]
        last_phi = ref["phi"]
        th32 = ref["theta_new"].astype(np.float32)                 # theta + step, rounded to the state's type
    err = np.linalg.norm(phi - last_phi) / np.linalg.norm(last_phi)
    assert err <= 1e-5, err                                        # north-star tolerance (BASELINE.json)
    assert np.abs(theta - th32).max() <= 1e-5 * np.abs(th32).max()
