"""bench.py keeps the driver's contract: ONE JSON line from rank 0 with the agreed keys (small config, GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "sfno_debug", "--steps", "2", "--warmup", "1",
                          "--no-sht-metric"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "samples/s" and d["value"] > 0 and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-6          # B = 1: samples/s = 1 / step time
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3


@pytest.mark.gpu
def test_bench_multistep_rollout_runs():
    """--multistep-count 3 --multistep-checkpoint (makani's multistep flags, BASELINE configs[4] shape of work)"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "sfno_debug", "--steps", "2", "--warmup", "1",
                          "--no-sht-metric", "--no-cpu-baseline", "--multistep-count", "3", "--multistep-checkpoint"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["multistep_count"] == 3 and d["config"]["multistep_checkpoint"] is True
    assert d["value"] > 0 and d["final_loss"] == d["final_loss"]
