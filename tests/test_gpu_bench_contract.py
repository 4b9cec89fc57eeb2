"""bench.py prints ONE JSON line with the fields the driver and the judge read."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "3", "--cpu-seconds", "2"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["metric"] == "logp+grad evals/sec" and d["unit"] == "evals/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 3 and d["scaling"] == "weak"
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["config"]["workload"].startswith("synthetic 10000 ind x 200 gaps") and "model" not in d["config"]
    assert d["value"] > 1e4 and abs(d["value"] - 40 * 4 / (d["ms_per_step"] * 40 / 1e3)) < 1e-3 * d["value"]
    ro = d["roofline"]
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and ro["peak"] == 8000.0
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3
    assert abs(ro["achieved"] - ro["algorithmic_bytes_per_launch"] / (ro["kernel_us"] * 1e-6) / 1e9) < 0.01 * ro["achieved"]
    assert ro["traffic"] is None or ro["traffic"] >= 0.9 * ro["algorithmic_bytes_per_launch"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "evals/s" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert d["value"] > 20 * cb["value"]
