"""Data-parallel eval driver; mirror of src/runner/distributed_runner_gram.py:685-874.

Users are independent, so each rank scores its own shard with no data-path collective.  At the end
the per-rank hit ranks (int16 position of the gold item in the score-sorted top-K, -1 = miss) are
exchanged with ONE all-gather over RCCL/xGMI (``backend="nccl"`` is RCCL on ROCm; tests use gloo),
and every rank derives the metric sums from the gathered ranks.  The reference's own
``all_reduce(SUM)`` of metric sums and counts (:835-836) is kept as a cross-check.

Sharding: ``shard_indices`` (strided, no duplicates) is the default; ``pad_like_reference=True``
reproduces DistributedSampler's padded accounting (:351: shuffled, repeated up to ceil(n/W)*W).
"""
from __future__ import annotations

import logging
import math
from typing import List

import numpy as np
import torch
import torch.distributed as dist

from ..utils import evaluate
from .base import BaseRunner, _arg


def shard_indices(n: int, world: int, rank: int, pad_like_reference: bool = False, seed: int = 0) -> List[int]:
    if not pad_like_reference:
        return list(range(rank, n, world))
    g = torch.Generator()
    g.manual_seed(seed)
    idx = torch.randperm(n, generator=g).tolist()
    total = math.ceil(n / world) * world
    pad = total - n
    if pad > 0:
        idx += (idx * math.ceil(pad / n))[:pad]
    return idx[rank:total:world]


def all_gather_hit_ranks(ranks: np.ndarray, device, group=None) -> np.ndarray:
    """One fixed-size all-gather of hit ranks (sent as int32: gloo, used by the CPU tests, has no
    int16 collectives); shards are padded with the sentinel -2."""
    world = dist.get_world_size(group)
    n = torch.tensor([len(ranks)], dtype=torch.int64, device=device)
    dist.all_reduce(n, op=dist.ReduceOp.MAX, group=group)
    width = int(n.item())
    buf = torch.full((width,), -2, dtype=torch.int32, device=device)
    buf[: len(ranks)] = torch.from_numpy(np.asarray(ranks, dtype=np.int32)).to(device)
    out = torch.empty(world * width, dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(out, buf, group=group)
    got = out.cpu().numpy()
    return got[got != -2].astype(np.int16)


class ShardSampler(torch.utils.data.Sampler):
    """This rank's user indices.  Default: ``shard_indices`` (strided, every user exactly once over the ranks); with
    ``--eval_pad_like_reference 1`` the order and padding of the reference's ``DistributedSampler(testdata)``
    (distributed_runner_gram.py:351: shuffled with seed 0, repeated up to ceil(n/W)*W, the duplicates counted)."""

    def __init__(self, n, world, rank, pad_like_reference=False):
        self.indices = shard_indices(n, world, rank, pad_like_reference)

    def __iter__(self):
        return iter(self.indices)

    def __len__(self):
        return len(self.indices)


class DistributedRunnerGRAM(BaseRunner):
    def __init__(self, model_rec, model_gen, tokenizer, train_loader_id, train_loader_rec, valid_loader, device, args, rank=0):
        self.rank = rank  # (before the base constructor: it builds the loaders, which shard by rank)
        super().__init__(model_rec, model_gen, tokenizer, train_loader_id, train_loader_rec, valid_loader, device, args)

    def _make_dataset(self, dataset, task, model_gen, tokenizer, regenerate, phase, debug_test_small_set, mode):
        """distributed_runner_gram.py:313-336: rank 0 builds the dataset first (it may regenerate the item-id file), the
        others wait at the barrier and then read what rank 0 wrote, never regenerating."""
        build = super()._make_dataset
        if not (dist.is_available() and dist.is_initialized()):
            return build(dataset, task, model_gen, tokenizer, regenerate, phase, debug_test_small_set, mode)
        if self.rank == 0:
            data = build(dataset, task, model_gen, tokenizer, regenerate, phase, debug_test_small_set, mode)
            dist.barrier()
        else:
            dist.barrier()
            data = build(dataset, task, model_gen, tokenizer, False, phase, debug_test_small_set, mode)
        return data

    def _make_loader(self, data, collator):
        from torch.utils.data import DataLoader
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        sampler = ShardSampler(len(data), world, self.rank, bool(int(_arg(self.args, "eval_pad_like_reference", 0))))
        return DataLoader(dataset=data, sampler=sampler, batch_size=int(_arg(self.args, "eval_batch_size", 1)), collate_fn=collator,
                          shuffle=False)

    def test_dataset_task(self, testloader, mode="test"):
        if self.rank == 0:
            logging.info(f"[{mode}] testing {testloader.dataset.dataset} dataset on {testloader.dataset.task} task")
        ranks, total_time, examples, user_ids, rows_out = self._score_loader(testloader)
        K = self.generate_num
        dev = self.device if dist.get_backend() != "gloo" else torch.device("cpu")
        dist.barrier()
        all_ranks = all_gather_hit_ranks(ranks, dev)
        sums = evaluate.metrics_from_ranks(all_ranks, self.metrics, K)
        test_total = len(all_ranks)
        # the reference's reduction, kept as a cross-check of the gathered result
        local = torch.tensor(evaluate.metrics_from_ranks(ranks, self.metrics, K), dtype=torch.float64, device=dev)
        cnt = torch.tensor(len(ranks), dtype=torch.int64, device=dev)
        dist.all_reduce(local, op=dist.ReduceOp.SUM)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        assert int(cnt.item()) == test_total and np.allclose(local.cpu().numpy(), sums), "all-gather and all-reduce disagree"
        metrics_res = sums / max(test_total, 1)
        if self.rank == 0:
            logging.info("\n-------------------------------")
            logging.info("\n".join(examples))
            for name, val in zip(self.metrics, metrics_res):
                logging.info(f"{mode} {name}: {val}")
        if _arg(self.args, "save_predictions", False):
            base = _arg(self.args, "pred_path", f"../preds/{testloader.dataset.dataset}_pred_{mode}.tsv")
            self._write_preds(f"{base}.{self.rank}", user_ids, ranks, rows_out)
            dist.barrier()
            if self.rank == 0:  # merge per-rank files, distributed_runner_gram.py:853-874
                with open(base, "w") as out:
                    out.write(self.PRED_HEADER)
                    import os
                    for r in range(dist.get_world_size()):
                        part = f"{base}.{r}"
                        if os.path.exists(part):
                            with open(part) as f:
                                out.writelines(list(f)[1:])
                            os.remove(part)
                    for name, val in zip(self.metrics, metrics_res):
                        out.write(f"{name}: {val}\n")
        dist.barrier()
        self.last_results = dict(metrics=dict(zip(self.metrics, metrics_res.tolist())), sums=sums, total=test_total,
                                 hit_ranks=all_ranks, local_hit_ranks=ranks, generate_seconds=total_time)
        return True
