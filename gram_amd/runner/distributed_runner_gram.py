"""Data-parallel eval driver; mirror of src/runner/distributed_runner_gram.py:685-874.

Users are independent, so each rank scores its own shard with no data-path collective.  At the end
ONE fixed-size ``all_gather_into_tensor`` over RCCL/xGMI (``backend="nccl"`` is RCCL on ROCm; tests use
gloo) exchanges, per user, the packed 6-byte record ``{user_idx: int32, hit_rank: int16}`` (SURVEY.md §8e:
dataset index; position of the gold item in the score-sorted top-K, -1 = miss); every rank derives the
metric sums from the gathered records.  The shard width ``ceil(n / W)`` is known without a collective.
The reference's own ``all_reduce(SUM)`` of metric sums and counts (:835-836) runs only as a cross-check,
behind ``--eval_check_allreduce 1``.

Sharding: ``shard_indices`` (strided, no duplicates) is the default; ``pad_like_reference=True``
reproduces DistributedSampler's padded accounting (:351: shuffled, repeated up to ceil(n/W)*W).
"""
from __future__ import annotations

import logging
import math
import os
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


HIT_RECORD = np.dtype([("user_idx", "<i4"), ("hit_rank", "<i2")])  # packed: 6 bytes per user, as SURVEY.md §8e specifies


def all_gather_hits(user_idx: np.ndarray, ranks: np.ndarray, n_total: int, device, group=None) -> np.ndarray:
    """THE collective of an evaluation: one ``all_gather_into_tensor`` of ``ceil(n_total / W)`` packed
    ``{user_idx:int32, hit_rank:int16}`` records per rank (sent as bytes: neither RCCL nor gloo has an int16 type), short
    shards padded with user_idx = -1.  Returns the valid records of all ranks, in rank order (structured array)."""
    world = dist.get_world_size(group)
    width = math.ceil(n_total / world) if n_total > 0 else 0
    if len(ranks) > width or len(user_idx) != len(ranks):
        raise ValueError(f"shard of {len(ranks)} users does not fit the fixed width ceil({n_total}/{world}) = {width}")
    rec = np.zeros(width, dtype=HIT_RECORD)
    rec["user_idx"] = -1
    rec["user_idx"][: len(ranks)] = np.asarray(user_idx, dtype=np.int32)
    rec["hit_rank"][: len(ranks)] = np.asarray(ranks, dtype=np.int16)
    buf = torch.from_numpy(rec.view(np.uint8).copy()).to(device)
    out = torch.empty(world * buf.numel(), dtype=torch.uint8, device=device)
    if width:
        dist.all_gather_into_tensor(out, buf, group=group)
    got = out.cpu().numpy().view(HIT_RECORD)
    return got[got["user_idx"] >= 0]


def all_gather_hit_ranks(ranks: np.ndarray, device, group=None, n_total=None, user_idx=None) -> np.ndarray:
    """Hit ranks of all ranks' users (rank order) through ``all_gather_hits``.  ``n_total`` = users over all ranks; without it
    every shard must have the same length (then the width is that length and still no collective is needed to agree on it)."""
    world = dist.get_world_size(group)
    if n_total is None:
        n_total = len(ranks) * world
    if user_idx is None:
        user_idx = np.arange(len(ranks), dtype=np.int32)
    return all_gather_hits(user_idx, ranks, n_total, device, group)["hit_rank"].astype(np.int16)


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

    def _timestamp(self, dev):
        """distributed_runner_gram.py:695-716: rank 0's wall-clock stamp, broadcast so that every rank names the same files."""
        import datetime
        if self.rank == 0:
            t = torch.tensor([int(x) for x in datetime.datetime.now().strftime("%Y%m%d_%H%M%S").split("_")], dtype=torch.int64, device=dev)
        else:
            t = torch.zeros(2, dtype=torch.int64, device=dev)
        dist.broadcast(t, src=0)
        stamp = "_".join(str(x) for x in t.tolist())
        dist.barrier()
        return stamp

    def test_dataset_task(self, testloader, mode="test"):
        if self.rank == 0:
            logging.info(f"[{mode}] testing {testloader.dataset.dataset} dataset on {testloader.dataset.task} task")
        K = self.generate_num
        dev = self.device if dist.get_backend() != "gloo" else torch.device("cpu")
        save = bool(_arg(self.args, "save_predictions", False))
        stamp = self._timestamp(dev) if save and not _arg(self.args, "pred_path", None) else None
        world = dist.get_world_size()
        # dataset indices of this rank's users, in scoring order; the sampler fixes the shard width ceil(n / W) for every rank
        sampler = getattr(testloader, "sampler", None)
        # The all-gather's width is ceil(n / W), known without a collective; a caller-supplied loader whose split gives one rank
        # more than that (contiguous chunks with a bigger last one, say) would fail only AFTER its whole shard was scored.
        width = math.ceil(len(testloader.dataset) / world)
        if sampler is not None and hasattr(sampler, "__len__") and len(sampler) > width:
            raise ValueError(f"rank {self.rank}: the loader's sampler yields {len(sampler)} users, more than the shard width "
                             f"ceil({len(testloader.dataset)}/{world}) = {width} of the hit-record all-gather; use ShardSampler "
                             f"(gram_amd.runner) or any split with at most that many users per rank")
        ranks, total_time, examples, user_ids, rows_out = self._score_loader(testloader)
        idx = np.asarray(getattr(sampler, "indices", range(len(ranks))), dtype=np.int32)[: len(ranks)]
        n_users = len(testloader.dataset)
        n_total = math.ceil(n_users / world) * world if bool(int(_arg(self.args, "eval_pad_like_reference", 0))) else n_users
        dist.barrier()
        hits = all_gather_hits(idx, ranks, n_total, dev)
        all_ranks = hits["hit_rank"].astype(np.int16)
        sums = evaluate.metrics_from_ranks(all_ranks, self.metrics, K)
        test_total = len(all_ranks)
        if int(_arg(self.args, "eval_check_allreduce", 0)):
            # the reference's reduction (distributed_runner_gram.py:835-836), as a cross-check of the gathered result
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
        self.last_pred_file = None
        if save:
            # distributed_runner_gram.py:717-720,853-874: one file per rank, merged by rank 0 into ..._all.tsv with the metric footer.
            # (--pred_path, an extension, names the merged file directly.)
            explicit = _arg(self.args, "pred_path", None)
            stem = (explicit[:-4] if explicit and explicit.endswith(".tsv") else explicit) if explicit else os.path.join(
                _arg(self.args, "pred_dir", "../preds"), f"{stamp}_{testloader.dataset.dataset}_{testloader.dataset.task}_pred_{mode}")
            self._write_preds(f"{stem}_{self.rank}.tsv", user_ids, ranks, rows_out)
            dist.barrier()
            merged = explicit if explicit else f"{stem}_all.tsv"
            if self.rank == 0:
                with open(merged, "w") as out:
                    out.write(self.PRED_HEADER)
                    for r in range(world):
                        part = f"{stem}_{r}.tsv"
                        if os.path.exists(part):
                            with open(part) as f:
                                out.writelines(list(f)[1:])
                            os.remove(part)
                    for name, val in zip(self.metrics, metrics_res):
                        out.write(f"{name}: {val}\n")
                logging.info(f">> preds saved to {merged}")
            self.last_pred_file = merged
        dist.barrier()
        self.last_results = dict(metrics=dict(zip(self.metrics, metrics_res.tolist())), sums=sums, total=test_total,
                                 hit_ranks=all_ranks, hit_user_idx=hits["user_idx"].astype(np.int32), local_hit_ranks=ranks,
                                 generate_seconds=total_time)
        return True
