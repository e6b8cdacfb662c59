"""bench.py keeps its contract: one JSON line with the fields the driver reads (small batch, 2 steps)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_bench_json_contract(native_lib):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "8",
                        "--no-cpu-baseline", "--no-kernel-rates"], capture_output=True, text=True, timeout=540, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "images/sec" and d["scaling"] == "weak"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and rf["peak"] > 0 and 0 < rf["frac"] < 1
    assert d["value"] > 0 and abs(d["value"] - 8 * 1000.0 / d["ms_per_step"]) < 0.02 * d["value"]


@pytest.mark.timeout(900)
def test_bench_starts_its_own_workers(native_lib):
    """`bench.py --gpus 2` started directly (no torchrun, no WORLD_SIZE): the parent spawns both ranks before touching the GPU and relays
    rank 0's line.  Rehearsal on ONE GPU: both ranks on device 0, gloo transport (RCCL refuses two ranks per device)."""
    env = dict(os.environ, BENCH_ONE_DEVICE="1", BENCH_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8",
                        "--no-cpu-baseline", "--no-kernel-rates", "--no-extras"], capture_output=True, text=True, timeout=800, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp2"
    assert abs(d["value"] - 16 * 1000.0 / d["ms_per_step"]) < 0.02 * d["value"]
    # the N > 1 record proves what ran: group size, transport, one entry per rank, the bucket plan, the exposed collective time
    rc = d["rccl"]
    assert rc["world_size"] == 2 and rc["backend"] == "gloo" and rc["one_device_rehearsal"] is True and rc["nccl_version"] is None
    assert [e["rank"] for e in rc["devices"]] == [0, 1] and all(e["device"] for e in rc["devices"])
    assert rc["bucket_count"] >= 1 and rc["gradient_bytes_per_step"] > 4 * 21_000_000 and rc["exposed_allreduce_ms"] >= 0.0
    # a group of another size than --gpus is refused
    env3 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29917")
    r3 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "8", "--no-cpu-baseline",
                         "--no-kernel-rates", "--no-extras"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env3)
    assert r3.returncode != 0 and "WORLD_SIZE" in (r3.stderr + r3.stdout)
    # a failing worker makes the launcher fail (the driver must not read a half result as success)
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "8",
                          "--backend", "no_such_backend", "--no-cpu-baseline", "--no-kernel-rates", "--no-extras"], capture_output=True, text=True,
                         timeout=600, cwd=ROOT, env=env)
    assert bad.returncode != 0
