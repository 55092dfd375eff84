"""CPU: the runners build their own evaluation loaders from the reference's flags (single_runner_gram.py:296-358,
distributed_runner_gram.py:300-359) -- runner.test(path) / validate(path), the methods main_generative_gram.py:127,209
call, must score the dataset the args name, and an empty loader list is an error, not a silent no-op.  The model is a
stub (the HIP model is covered by tests/test_gpu_runner.py); the dataset is tests/golden/dataset_fixture."""
import json
import os
import tempfile
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gram_amd.runner import ShardSampler, get_runner, shard_indices
from tests.stub_tokenizer import StubTokenizer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
K = 5


class PieceTokenizer(StubTokenizer):
    """StubTokenizer that also breaks lexical ids ('|▁soap|▁rene') at their '|' separators, like SentencePiece does
    (the separator pieces 1820 / 9175 are what the runner and the collator strip)."""

    def convert_tokens_to_ids(self, tokens):
        out = []
        for t in tokens:
            for piece in t.replace("|", " | ").split():
                out.append(self._id(piece))
        return out


def fixture_args(**kw):
    case = json.load(open(os.path.join(GOLDEN, "dataset_cases.json")))[0]["args"]
    a = dict(case)
    a.update(data_path=os.path.join(GOLDEN, "dataset_fixture"), prompt_file=os.path.join(GOLDEN, "dataset_fixture", "prompt.txt"),
             datasets="Beauty", tasks="sequential", eval_batch_size=5, metrics="hit@1,hit@5,ndcg@5", beam_size=K, length_penalty=1.0,
             item_id_type="split", item_prompt_max_len=64, target_max_len=16, save_predictions=False, debug_test_small_set=0,
             passage_cache=0)
    a.update(kw)
    return SimpleNamespace(**a)


class StubModel(torch.nn.Module):
    """generate(): K distinct Trie members per user, seeded by the user's first tokens; counts the users it saw."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))
        self.users = 0

    def generate(self, input_ids, attention_mask, max_length, prefix_allowed_tokens_fn, num_beams, **kw):
        trie = prefix_allowed_tokens_fn.__closure__[0].cell_contents
        seqs, scores = [], []
        for b in range(input_ids.shape[0]):
            self.users += 1
            g = torch.Generator().manual_seed(int(input_ids[b, 0, :4].sum()))
            got = set()
            while len(got) < num_beams:
                seq = [0]
                while True:
                    allowed = trie.get(seq)
                    if not allowed:
                        break
                    seq.append(allowed[int(torch.randint(0, len(allowed), (1,), generator=g))])
                got.add(tuple(seq))
            for s in sorted(got):
                seqs.append(list(s) + [0] * (max_length - len(s)))
            scores += sorted((-torch.rand(num_beams, generator=g)).tolist(), reverse=True)
        return {"sequences": torch.tensor(seqs), "sequences_scores": torch.tensor(scores)}


def test_single_runner_builds_and_scores_its_loaders(tmp_path):
    args = fixture_args()
    model = StubModel()
    ckpt = str(tmp_path / "model_rec_best.pt")
    torch.save(model.state_dict(), ckpt)
    runner = get_runner("single", model, None, PieceTokenizer(), None, None, None, "cpu", args)
    # built at construction, like the reference (single_runner_gram.py:46,51)
    assert len(runner.testloaders) == 1 and len(runner.validloaders) == 1
    n = len(runner.testloaders[0].dataset)
    assert n == 12 and runner.testloaders[0].batch_size == 5
    runner.test(ckpt)
    assert model.users == n and runner.last_results["total"] == n
    assert len(runner.last_results["hit_ranks"]) == n
    model.users = 0
    runner.validate(str(tmp_path))  # a directory: the first file in it (single_runner_gram.py:377-379)
    assert model.users == len(runner.validloaders[0].dataset) == 12
    assert runner.validloaders[0].dataset.mode == "validation"
    # two datasets -> two loaders, in order
    r2 = get_runner("single", model, None, PieceTokenizer(), None, None, None, "cpu", fixture_args(datasets="Beauty,Yelp"))
    assert [l.dataset.dataset for l in r2.testloaders] == ["Beauty", "Yelp"]


def test_empty_loader_list_is_an_error():
    args = SimpleNamespace(metrics="hit@5", beam_size=K, length_penalty=1.0)
    runner = get_runner("single", StubModel(), None, None, None, None, None, "cpu", args)
    with pytest.raises(RuntimeError):
        runner.test(None)
    with pytest.raises(RuntimeError):
        runner.validate(None)
    with pytest.raises(ValueError):
        runner.get_testloader()


def test_shard_sampler_orders():
    assert list(ShardSampler(10, 4, 1)) == [1, 5, 9]
    padded = [list(ShardSampler(10, 4, r, pad_like_reference=True)) for r in range(4)]
    assert all(len(p) == 3 for p in padded)
    flat = sorted(i for p in padded for i in p)
    assert set(flat) == set(range(10)) and len(flat) == 12  # ceil(10/4)*4 samples: two users counted twice
    ref = torch.utils.data.DistributedSampler(list(range(10)), num_replicas=4, rank=2)  # the reference's sampler (:351)
    assert list(ref) == padded[2] == shard_indices(10, 4, 2, True)


def _worker(rank, world, path, pad, q):
    dist.init_process_group("gloo", init_method=f"file://{path}", rank=rank, world_size=world)
    try:
        args = fixture_args(eval_batch_size=2, eval_pad_like_reference=int(pad), rank=rank)
        model = StubModel()
        runner = get_runner("distributed", model, None, PieceTokenizer(), None, None, None, "cpu", args, rank)
        runner.test(None)
        q.put((rank, model.users, runner.last_results["total"], runner.last_results["sums"].tolist(),
               sorted(runner.last_results["hit_ranks"].tolist())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pad", [False, True])
def test_distributed_runner_world5_matches_single(pad):
    """12 users over 5 ranks (shards of 3, 3, 2, 2, 2): the rank-0-first dataset build with its barriers, the sampler, the
    all-gather and the all-reduce cross-check.  Default sharding reproduces the single-process result; with
    --eval_pad_like_reference the DistributedSampler accounting of the reference (15 samples, three users counted twice)."""
    from gram_amd.utils import evaluate as ev
    args = fixture_args()
    single = get_runner("single", StubModel(), None, PieceTokenizer(), None, None, None, "cpu", args)
    single.test(None)
    want = single.last_results
    world = 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.TemporaryDirectory() as d:
        procs = [ctx.Process(target=_worker, args=(r, world, os.path.join(d, "rdzv"), pad, q)) for r in range(world)]
        [p.start() for p in procs]
        res = sorted(q.get(timeout=240) for _ in procs)
        [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert all(r[3] == res[0][3] and r[4] == res[0][4] for r in res)  # every rank reconstructs the same global result
    if not pad:
        assert [r[1] for r in res] == [3, 3, 2, 2, 2] and all(r[2] == 12 for r in res)
        assert np.allclose(res[0][3], want["sums"]) and res[0][4] == sorted(want["hit_ranks"].tolist())
    else:
        assert [r[1] for r in res] == [3] * 5 and all(r[2] == 15 for r in res)
        dup = [i for r in range(world) for i in shard_indices(12, world, r, True)]
        ranks = np.array([want["hit_ranks"][u] for u in dup])
        assert np.allclose(res[0][3], ev.metrics_from_ranks(ranks, args.metrics.split(","), K))
