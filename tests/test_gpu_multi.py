"""-m gpu, self-skipping below two devices: the multi-GPU path over RCCL itself (`backend="nccl"` is RCCL on ROCm), one rank per GPU.
The builder's GPU box has ONE device, so these tests are skipped there and in the driver's single-GPU tier; on any box with two or
more MI355X they run without anyone writing new code:

  * the distributed runner at world size 2 on a user count the world size does not divide (sums = the single runner's), the hit-rank
    all-gather and the reference's all_reduce cross-check on device tensors (distributed_runner_gram.py:351,832-836;
    main_generative_gram.py:31-49: one process per GPU);
  * `python bench.py --gpus 2` (self-spawned ranks, RCCL): both ranks on distinct devices, `exchange.users_gathered == 2 * B`.

`torch.cuda.device_count()` does not initialise the GPU in the collecting process; every rank is a spawned child."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.test_gpu_runner import _cfg, _checkpoint
from tests.test_runner_loaders import PieceTokenizer, fixture_args

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (one rank per device over RCCL)")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank(rank, world, rdzv, ckpt, q):
    import gram_amd
    from gram_amd.runner import get_runner
    dev = torch.device(f"cuda:{rank}")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"file://{rdzv}", rank=rank, world_size=world, device_id=dev)
    try:
        args = fixture_args(eval_batch_size=2, rank=rank, eval_check_allreduce=1)
        model = gram_amd.create_model("gram", _cfg()).to(dev)
        runner = get_runner("distributed", model, None, PieceTokenizer(), None, None, None, dev, args, rank)
        runner.test(ckpt)
        r = runner.last_results
        q.put((rank, len(r["local_hit_ranks"]), r["total"], r["sums"].tolist(), sorted(r["hit_ranks"].tolist()), dist.get_backend(),
               torch.cuda.current_device()))
    finally:
        dist.destroy_process_group()


def test_distributed_runner_world_2_over_rccl(tmp_path):
    import gram_amd
    from gram_amd.runner import get_runner
    ckpt = str(tmp_path / "model_rec_best.pt")
    _checkpoint(ckpt)
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, world, str(tmp_path / "rdzv"), ckpt, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=900) for _ in procs)
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    # the single runner, in a child as well (this process never touches a GPU)
    single = ctx.Process(target=_single, args=(ckpt, q))
    single.start()
    total, sums, ranks = q.get(timeout=900)
    single.join(120)
    assert [r[1] for r in res] == [6, 6] and all(r[2] == total == 12 for r in res)
    assert all(np.allclose(r[3], sums) for r in res) and all(r[4] == ranks for r in res)
    assert all(r[5] == "nccl" for r in res) and sorted(r[6] for r in res) == [0, 1]


def _single(ckpt, q):
    import gram_amd
    from gram_amd.runner import get_runner
    model = gram_amd.create_model("gram", _cfg()).to("cuda:0")
    runner = get_runner("single", model, None, PieceTokenizer(), None, None, None, "cuda:0", fixture_args(eval_batch_size=5))
    runner.test(ckpt)
    r = runner.last_results
    q.put((r["total"], r["sums"].tolist(), sorted(r["hit_ranks"].tolist())))


def test_bench_two_gpus_over_rccl():
    small = ["--backbone", "t5-small", "--batch", "6", "--passages", "3", "--passage-len", "32", "--beams", "5", "--dataset", "Toys",
             "--steps", "2", "--warmup", "1", "--cpu-users", "0", "--check-allreduce"]
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", *small], cwd=ROOT, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak"
    ex = r["exchange"]
    assert ex["users_gathered"] == 2 * 6 and "backend nccl" in ex["collective"] and ex["devices"] == [0, 1]
