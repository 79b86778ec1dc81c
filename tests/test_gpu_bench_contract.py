"""GPU: `python bench.py` prints ONE JSON line with the keys of the measurement contract (driver + judge read them)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    env = dict(os.environ, BENCH_CPU_WORKERS="2", BENCH_CPU_SECONDS="1", BENCH_NUMPY_STEPS="20", BENCH_TRAIN_MODES="0")
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
    assert c["one_core"]["value"] > 0 and c["one_core"]["cores"] == 1 and c["numpy_highs"]["value"] > 0
    assert j["ranks_seen"] == 1 and "traffic_source" in r and j["config"]["debug"] == 0
    assert j["config"]["seeds"] == [0, 1, 2] and sorted(p["seed"] for p in j["config"]["per_seed"]) == [0, 1, 2]
    vals = sorted(p["value"] for p in j["config"]["per_seed"])
    assert j["value"] == pytest.approx(vals[1]) and j["config"]["seed_of_value"] in (0, 1, 2)      # the median run
    om = j["other_modes"]
    assert set(om) == {"sparse_raster_update", "bit_packed_rasters_only", "candidate_stability", "candidate_stability_3groups",
                       "config5_hexagon_bridge"}
    for k, v in om.items():
        assert "error" not in v, (k, v)
        assert v["value"] > 1e4
    for leg, groups in (("candidate_stability", 1), ("candidate_stability_3groups", 3)):
        cs = om[leg]["candidate_stability"]
        assert cs["decisions_per_s"] > 1e5 and cs["decisions_per_s_wall"] > 1e5 and cs["last_lockstep"]["errors"] == 0 and cs["groups"] == groups
    assert "hexagon" in om["config5_hexagon_bridge"]["workload"]


def test_bench_launches_its_own_ranks_for_gpus_2():
    """`python bench.py --gpus 2` with no torchrun environment starts two ranks itself (gloo here: both share the one
    card of the test box; on a node the same path runs RCCL) and rank 0 prints the one line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(BENCH_DIST_BACKEND="gloo", BENCH_TRAIN_LOCKSTEPS="4", BENCH_TRAIN_WARMUP="4")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3",
                          "--envs", "256"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["scaling"] == "weak"
    assert "cpu_baseline" not in j                                     # rank 0 at N = 1 only
    # N > 1 carries BASELINE.json configs[3]'s data path: per-rank env shards + one all-gather of records per lock-step
    leg = j["other_modes"]["train_config4"]
    assert "error" not in leg, leg
    assert leg["ranks_seen"] == 2 and leg["dist_backend"] == "gloo" and leg["allgather_ms_per_lockstep"] > 0
    assert leg["allgather_rows_received"] == 2 * 256            # every rank received every rank's rows
    assert leg["ring_hash_equal"] is True and leg["policy_hash_equal"] is True and leg["last_losses_finite"] is True
    assert leg["value"] > 1e3 and leg["ring_records"] > 256
    assert j["config"]["seeds"] == [0, 1, 2] and len(j["config"]["per_seed"]) == 3
    # both ranks' env-steps are in `value`: 2 x 256 envs per lock-step
    assert abs(j["ms_per_step"] * j["value"] / 1e3 / (2 * 256 * j["config"]["env_step_fraction"]) - 1) < 0.05
