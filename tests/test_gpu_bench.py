"""-m gpu: the bench.py contract the driver depends on (one JSON line with the keys of the round prompt), on a small
configuration so that it runs in seconds: single process, and two ranks over gloo sharing the one GPU of the test box
(the N > 1 code path: barrier, MAX over ranks, whole-job value)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--backbone", "t5-small", "--batch", "6", "--passages", "3", "--passage-len", "32", "--beams", "5", "--dataset", "Toys",
         "--steps", "2", "--warmup", "1"]
CONTRACT = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config"}


def _run(cmd, env=None):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env or {}))
    p = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def _check(r, n_gpus, steps):
    assert CONTRACT <= set(r)
    assert r["unit"] == "users/s" and r["n_gpus"] == n_gpus and r["steps"] == steps and r["higher_is_better"] is True
    assert r["scaling"] == "weak" and r["vs_baseline"] is None and r["dtype"] == "f16x3" and r["data"] == "synthetic"
    assert "workload" in r["config"] and "model" not in r["config"]
    assert abs(r["value"] - n_gpus * 6 * steps / (r["ms_per_step"] * steps / 1e3)) < 1e-6 * r["value"]
    assert r["output_check"]["all_in_trie"] is True


def test_bench_contract_single_process():
    r = _run([sys.executable, "bench.py", *SMALL, "--cpu-users", "1", "--ragged", "--item-pool", "9"])
    _check(r, 1, 2)
    roof = r["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(roof) and roof["bound"] in ("hbm", "mfma")
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and roof["achieved"] > 0
    cpu = r["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cpu) and cpu["kind"] == "port" and cpu["value"] > 0
    ic = r["config"]["item_cache"]
    assert ic["pool"] == 9 and ic["passages_from_cache_per_step"] > 0 and ic["passages_encoded_per_step"] >= 6


def test_bench_spawns_its_own_ranks_and_exchanges_hit_ranks():
    """`python bench.py --gpus 2` with no launcher: the parent starts both ranks before touching the GPU; the timed region ends
    with the hit-rank all-gather + all-reduce cross-check (gloo here: the two ranks share the one test GPU)."""
    r = _run([sys.executable, "bench.py", "--gpus", "2", *SMALL, "--cpu-users", "0", "--backend", "gloo", "--share-device"])
    _check(r, 2, 2)
    ex = r["exchange"]
    assert ex["users_gathered"] == 12 and "all_gather_into_tensor" in ex["collective"]
    assert set(ex["metrics_vs_synthetic_gold"]) == {"hit@5", "hit@10", "ndcg@5", "ndcg@10"}


def test_bench_contract_two_ranks_gloo_one_device():
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", "29531", "bench.py", "--gpus", "2", *SMALL, "--cpu-users", "0", "--backend", "gloo", "--share-device"])
    _check(r, 2, 2)
    assert "cpu_baseline" not in r  # rank 0 at N = 1 only
