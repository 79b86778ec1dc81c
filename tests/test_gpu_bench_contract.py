"""GPU: `python bench.py` prints ONE JSON line with the keys of the measurement contract (driver + judge read them)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    env = dict(os.environ, BENCH_CPU_WORKERS="2", BENCH_CPU_SECONDS="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                          "--envs", "512"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 20 and j["warmup"] == 5 and j["higher_is_better"] is True
    assert j["scaling"] == "weak" and j["vs_baseline"] is None and j["data"] == "synthetic" and j["dtype"] == "f64"
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 2 and c["value"] > 0 and "sample" in c
    assert j["value"] > 1e5 and abs(j["ms_per_step"] * j["value"] / 1e3 / (512 * j["config"]["env_step_fraction"]) - 1) < 0.02
    assert set(j["other_modes"]) == {"sparse_raster_update", "bit_packed_rasters_only"}
