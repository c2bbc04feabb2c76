"""GPU: bench.py prints ONE JSON line with the keys the driver's contract names (a short run of the C2 workload, which
takes a second; the default run is the same code on C3)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_line_has_the_contract_keys(cuda):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c2", "--steps", "4", "--warmup", "3",
           "--secondary", "none", "--no-other-configs", "--no-variants"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout                                   # exactly one line on stdout
    b = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in b, key
    assert b["n_gpus"] == 1 and b["steps"] == 4 and b["warmup"] == 3 and b["higher_is_better"] is True
    assert b["vs_baseline"] is None and b["data"] == "synthetic" and "workload" in b["config"]
    assert abs(b["value"] - b["config"]["n"] * 1e3 / b["ms_per_step"]) <= 1e-6 * b["value"]     # particle-updates/s
    roof = b["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] == "mfma" and 0.0 < roof["frac"] <= 1.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    cpu = b["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] == "port" and cpu["value"] > 0
    # round 3: the fraction is named three ways, the host baseline says how many threads / cores it ran on and is timed on
    # one thread as well, the distance pass reports its HBM write rate
    for key in ("frac_executed", "frac_algorithmic", "mfma_busy_frac"):
        assert key in roof, key
    assert roof["frac_executed"] == roof["frac"] and 0.0 < roof["frac_algorithmic"] <= roof["frac_executed"]
    for key in ("blas_threads", "logical_cpus", "physical_cores", "one_thread"):
        assert key in cpu, key
    assert cpu["one_thread"]["cores"] == 1 and cpu["one_thread"]["value"] > 0 and cpu["cores"] == cpu["blas_threads"]
    dp = b["distance_pass"]
    assert dp["bound"] == "hbm" and dp["achieved_write_GBps"] > 0 and 0.0 < dp["frac_of_hbm_peak"] < 1.0
    assert b["finite"] is True and b["parity_sample_relerr"] < 4e-3                            # bf16 inputs: K rounded to bf16


def test_train_on_batch_entry(cuda):
    """the C3-shaped SteinSampler.train_on_batch entry of the bench line: score recomputed every step, window statistics"""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    ent = bench.train_on_batch_entry(torch, cuda, steps=6, warmup=3)
    assert ent["finite"] and ent["ms_per_step"] > 0 and ent["window"]["timed_steps"] == 6
    assert 0 <= ent["window"]["hits"] <= 6
