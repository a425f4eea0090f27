"""bench.py prints one JSON line with the fields its contract names (a short run on a small batch)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields(gpu):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
                          "--spp-per-step", "16", "--cpu-spp", "1"], capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "rmse_vs_cpu"):
        assert k in d, k
    assert d["metric"] == "path samples/s" and d["unit"] == "samples/s" and d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 800 * 800 * 16 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    # the final scene's tables live in LDS: the closest-hit kernel is bound by f64 vector issue, not by HBM, and the
    # line says so (there is no MFMA work in a path tracer); the byte formula is reported beside it
    assert r["bound"] == "valu_f64" and r["served_from"] == "lds" and r["unit"] == "TFLOP/s" and r["peak"] == 78.6
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    assert r["bytes"]["unit"] == "GB/s" and r["bytes"]["peak"] == 8000.0 and r["bytes"]["achieved"] > 0
    assert r["traffic"] is None  # the committed counter passes are for the default batch size only
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] == 8 and c["value"] > 0
    # BASELINE.json's metric ends "image RMSE vs CPU": the GPU's picture of the CPU leg's samples against the oracle's under
    # the reference's own box rule, far below the noise between two seeds
    m = d["rmse_vs_cpu"]
    assert m["spp"] == 1 and m["cpu_box_rule"] == "reference AABB test" and m["noise_floor"] > 0
    assert m["value"] <= 1e-3 * m["noise_floor"] + 1e-6 or m["pixels_off"] > 0, m
    assert m["value"] <= 1.5 * m["noise_floor"] and m["pixels_off"] <= 0.002, m
