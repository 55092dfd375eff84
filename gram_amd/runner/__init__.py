"""``get_runner`` with the reference's dispatch and error behaviour (src/runner/__init__.py:12-58)."""
from .base import BaseRunner
from .distributed_runner_gram import DistributedRunnerGRAM, HIT_RECORD, ShardSampler, all_gather_hit_ranks, all_gather_hits, shard_indices
from .single_runner_gram import SingleRunnerGRAM


def get_runner(runner_type, model_rec, model_gen, tokenizer, train_loader_id, train_loader_rec, valid_loader, device, args,
               rank=0):
    if runner_type == "single":
        return SingleRunnerGRAM(model_rec, model_gen, tokenizer, train_loader_id, train_loader_rec, valid_loader, device, args)
    elif runner_type == "distributed":
        return DistributedRunnerGRAM(model_rec, model_gen, tokenizer, train_loader_id, train_loader_rec, valid_loader, device,
                                     args, rank)
    raise ValueError(f"Unknown runner type: {runner_type}")


__all__ = ["get_runner", "BaseRunner", "SingleRunnerGRAM", "DistributedRunnerGRAM", "shard_indices", "all_gather_hit_ranks", "all_gather_hits", "HIT_RECORD", "ShardSampler"]
