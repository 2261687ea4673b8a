"""The benchmark's output contract: one JSON line on stdout, with the measurement records the harness reads."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_exactly_one_json_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-extra", "--no-cpu"],
                         capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[:500]
    rec = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in rec, key
    assert rec["n_gpus"] == 1 and rec["steps"] == 3 and rec["scaling"] == "weak" and rec["vs_baseline"] is None
    assert "workload" in rec["config"] and "model" not in rec["config"]
    r = rec["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0.3 < r["frac"] < 1.0 and rec["value"] > 1e7
