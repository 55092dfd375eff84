"""CPU, world_size 2 (gloo): the data-parallel eval path -- users sharded across ranks, per-rank
{user_idx:int32, hit_rank:int16} records combined by ONE fixed-width all-gather (no collective to agree on the
width), metric sums identical to the single-process run and to the reference's all_reduce(SUM) accounting
(distributed_runner_gram.py:832-838; run as a cross-check under --eval_check_allreduce 1).  The model is a stub
whose generate() returns canned beams (the HIP model itself is covered by the -m gpu tests); what is
under test is sharding, the collective, and the metric reconstruction."""
import os
import tempfile
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gram_amd.runner import DistributedRunnerGRAM, SingleRunnerGRAM, shard_indices
from gram_amd.utils import evaluate as ev

K = 5
N_USERS = 23  # not a multiple of the world size
ITEMS = [[0, a, b, 1] for a in range(2, 8) for b in range(2, 8)]


class StubModel(torch.nn.Module):
    """generate(): user u (read from input_ids[b,0,0]) gets K distinct items in a seeded order."""

    def forward(self, *a, **k):
        raise NotImplementedError

    def generate(self, input_ids, attention_mask, max_length, num_beams, **kw):
        seqs, scores = [], []
        for u in input_ids[:, 0, 0].tolist():
            g = torch.Generator().manual_seed(1000 + u)
            pick = torch.randperm(len(ITEMS), generator=g)[:num_beams].tolist()
            seqs += [ITEMS[i] for i in pick]
            scores += sorted((-torch.rand(num_beams, generator=g)).tolist(), reverse=True)
        return {"sequences": torch.tensor(seqs), "sequences_scores": torch.tensor(scores)}


def _gold(u):
    g = torch.Generator().manual_seed(1000 + u)
    pick = torch.randperm(len(ITEMS), generator=g)[:K].tolist()
    # gold sits at rank u % (K+2); ranks >= K mean "not retrieved"
    r = u % (K + 2)
    return ITEMS[pick[r]] if r < K else ITEMS[(pick[0] + 17) % len(ITEMS)] if ITEMS[(pick[0] + 17) % len(ITEMS)] not in [ITEMS[i] for i in pick] else [0, 9, 9, 1]


class _Dataset(SimpleNamespace):
    def __len__(self):
        return self.n


class Loader(list):
    def __init__(self, users, bs=4, n_total=None):
        batches = []
        for i in range(0, len(users), bs):
            us = users[i:i + bs]
            ids = torch.zeros(len(us), 2, 8, dtype=torch.long)
            ids[:, 0, 0] = torch.tensor(us)
            tgt = torch.full((len(us), 6), -100)
            for j, u in enumerate(us):
                gseq = _gold(u)[1:]
                tgt[j, : len(gseq)] = torch.tensor(gseq)
            batches.append({"item_text_ids": ids, "item_text_masks": torch.ones_like(ids, dtype=torch.bool), "target_ids": tgt,
                            "user_ids": [f"u{u}" for u in us]})
        super().__init__(batches)
        self.dataset = _Dataset(all_items=ITEMS, dataset="Synthetic", task="sequential", n=len(users) if n_total is None else n_total)
        self.sampler = SimpleNamespace(indices=list(users))  # what ShardSampler exposes: the dataset indices of this rank's users


ARGS = SimpleNamespace(metrics="hit@1,hit@5,ndcg@3,ndcg@5", beam_size=K, length_penalty=1.0, item_id_type="split", save_predictions=False)


def _single():
    r = SingleRunnerGRAM(StubModel(), None, None, None, None, None, "cpu", ARGS)
    r.test_dataset_task(Loader(list(range(N_USERS))))
    return r.last_results


def _worker(rank, world, path, pad, q):
    dist.init_process_group("gloo", init_method=f"file://{path}", rank=rank, world_size=world)
    try:
        args = SimpleNamespace(**{**vars(ARGS), "eval_pad_like_reference": int(pad), "eval_check_allreduce": 1})
        r = DistributedRunnerGRAM(StubModel(), None, None, None, None, None, "cpu", args, rank)
        r.test_dataset_task(Loader(shard_indices(N_USERS, world, rank, pad_like_reference=pad), n_total=N_USERS))
        by_user = sorted(zip(r.last_results["hit_user_idx"].tolist(), r.last_results["hit_ranks"].tolist()))
        q.put((rank, r.last_results["sums"].tolist(), r.last_results["total"], by_user))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pad", [False, True])
def test_world2_allgather_matches_single_process(pad):
    single = _single()
    assert single["total"] == N_USERS
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.TemporaryDirectory() as d:
        procs = [ctx.Process(target=_worker, args=(r, 2, os.path.join(d, "rdzv"), pad, q)) for r in range(2)]
        [p.start() for p in procs]
        res = [q.get(timeout=120) for _ in procs]
        [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (r0, s0, t0, h0), (r1, s1, t1, h1) = sorted(res)
    assert s0 == s1 and t0 == t1 and h0 == h1  # every rank reconstructs the same global result
    if not pad:
        assert t0 == N_USERS
        assert np.allclose(s0, single["sums"])
        # the gathered records carry the dataset index: every user exactly once, with the single run's rank for THAT user
        assert h0 == [(u, int(single["hit_ranks"][u])) for u in range(N_USERS)]
    else:
        # DistributedSampler accounting: ceil(n/W)*W samples, the padded duplicate is counted twice
        assert t0 == 24
        dup = shard_indices(N_USERS, 2, 0, True) + shard_indices(N_USERS, 2, 1, True)
        ranks = np.array([single["hit_ranks"][u] for u in dup])
        assert np.allclose(s0, ev.metrics_from_ranks(ranks, ARGS.metrics.split(","), K))
        assert h0 == sorted((u, int(single["hit_ranks"][u])) for u in dup)


def _preds_worker(rank, world, path, pred_path):
    dist.init_process_group("gloo", init_method=f"file://{path}", rank=rank, world_size=world)
    try:
        args = SimpleNamespace(**{**vars(ARGS), "save_predictions": True, "pred_path": pred_path})
        r = DistributedRunnerGRAM(StubModel(), None, None, None, None, None, "cpu", args, rank)
        r.test_dataset_task(Loader(shard_indices(N_USERS, world, rank), n_total=N_USERS))
    finally:
        dist.destroy_process_group()


def test_preds_tsv_single_and_world2_merge(tmp_path):
    """Preds file (single_runner_gram.py:580-588,675-694,709-710; merge distributed_runner_gram.py:853-874): literal
    header, one row per user, closing `metric: value` lines; the world-2 merge holds the same rows as the single run."""
    p1 = str(tmp_path / "single.tsv")
    args = SimpleNamespace(**{**vars(ARGS), "save_predictions": True, "pred_path": p1})
    r = SingleRunnerGRAM(StubModel(), None, None, None, None, None, "cpu", args)
    r.test_dataset_task(Loader(list(range(N_USERS))))
    lines = open(p1).read().splitlines()
    metrics = ARGS.metrics.split(",")
    assert lines[0] == "idx\tH@5\tH@10\tNDCG@5\tNDCG@10\tgold\tpred\tscores"
    rows, foot = lines[1:1 + N_USERS], lines[1 + N_USERS:]
    assert [f.split(": ")[0] for f in foot] == metrics
    assert np.allclose([float(f.split(": ")[1]) for f in foot], [r.last_results["metrics"][m] for m in metrics])
    for row in rows:
        cols = row.split("\t")
        assert len(cols) == 1 + len(metrics) + 3
        assert len(cols[-2].split("||")) == K and len(cols[-1].split("||")) == K
        sc = [float(x) for x in cols[-1].split("||")]
        assert sc == sorted(sc, reverse=True)
    p2 = str(tmp_path / "merged.tsv")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_preds_worker, args=(rk, 2, str(tmp_path / "rdzv"), p2)) for rk in range(2)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    merged = open(p2).read().splitlines()
    assert merged[0] == lines[0] and sorted(merged[1:1 + N_USERS]) == sorted(rows)
    mfoot = merged[1 + N_USERS:]
    assert [f.split(": ")[0] for f in mfoot] == metrics
    assert np.allclose([float(f.split(": ")[1]) for f in mfoot], [float(f.split(": ")[1]) for f in foot], rtol=1e-12)
    assert not [f for f in os.listdir(tmp_path) if f.startswith("merged_")]  # the per-rank parts merged_{rank}.tsv are removed


def _named_worker(rank, world, path, pred_dir, q):
    dist.init_process_group("gloo", init_method=f"file://{path}", rank=rank, world_size=world)
    try:
        args = SimpleNamespace(**{**vars(ARGS), "save_predictions": True, "pred_dir": pred_dir})
        r = DistributedRunnerGRAM(StubModel(), None, None, None, None, None, "cpu", args, rank)
        r.test_dataset_task(Loader(shard_indices(N_USERS, world, rank), n_total=N_USERS), mode="validation")
        q.put((rank, r.last_pred_file))
    finally:
        dist.destroy_process_group()


def test_preds_file_names_mirror_the_reference(tmp_path):
    """single_runner_gram.py:580-588: ../preds/{timestamp}_{dataset}_{task}_pred_{mode}.tsv; distributed_runner_gram.py:695-720,
    853-874: rank 0's timestamp broadcast to all ranks, per-rank files ..._{mode}_{rank}.tsv merged into ..._{mode}_all.tsv and removed."""
    import re
    d1 = str(tmp_path / "p1")
    args = SimpleNamespace(**{**vars(ARGS), "save_predictions": True, "pred_dir": d1})
    r = SingleRunnerGRAM(StubModel(), None, None, None, None, None, "cpu", args)
    r.test_dataset_task(Loader(list(range(N_USERS))))
    (f1,) = os.listdir(d1)
    assert re.fullmatch(r"\d{8}_\d{6}_Synthetic_sequential_pred_test\.tsv", f1) and r.last_pred_file == os.path.join(d1, f1)
    d2 = str(tmp_path / "p2")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_named_worker, args=(rk, 2, str(tmp_path / "rdzv2"), d2, q)) for rk in range(2)]
    [p.start() for p in procs]
    got = dict(q.get(timeout=120) for _ in procs)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (f2,) = os.listdir(d2)  # the per-rank parts are gone
    # (the reference joins the broadcast ints back with "_": a time such as 09:05:07 loses its leading zero there too)
    assert re.fullmatch(r"\d{8}_\d{1,6}_Synthetic_sequential_pred_validation_all\.tsv", f2)
    assert got[0] == got[1] == os.path.join(d2, f2)  # both ranks derived the same name from rank 0's stamp
    assert len(open(os.path.join(d2, f2)).read().splitlines()) == 1 + N_USERS + len(ARGS.metrics.split(","))
