"""CPU: the host half of the drop-in runner at the headline rate (SURVEY.md §8d second metric, §8f N2):
re-batching whatever --eval_batch_size is, the per-item token-row table of the collator, the vectorised separator strip and
metrics, item indices instead of decoded rows.  Every fast form is compared with the slow form it replaces."""

import numpy as np
import pytest
import torch

from gram_amd.processor import collator as col
from gram_amd.runner import get_runner
from gram_amd.utils import evaluate
from gram_amd.utils import generation_trie as gt
from tests.test_runner_loaders import PieceTokenizer, StubModel, fixture_args


def _strip_rows_loop(ids, mask, limit):
    """The per-row form (Collator.py:281-340 restated row by row): what `_strip_rows` vectorises."""
    keep = (ids != col.SPLIT_IDS[0]) & (ids != col.SPLIT_IDS[1])
    out_ids = torch.zeros(ids.size(0), limit, dtype=torch.long)
    out_mask = torch.zeros(ids.size(0), limit, dtype=torch.long)
    for r in range(ids.size(0)):
        row, m = ids[r][keep[r]][:limit].clone(), mask[r][keep[r]][:limit]
        if row.numel() == 0:
            raise ValueError("a passage consists of separator tokens only")
        if not bool((row == 1).any()):
            row[-1] = 1
        out_ids[r, : row.numel()] = row
        out_mask[r, : m.numel()] = m
    return out_ids, out_mask


@pytest.mark.parametrize("W,limit", [(40, 16), (12, 16), (99, 32), (16, 16)])
def test_strip_rows_vectorised_equals_the_row_loop(W, limit):
    g = torch.Generator().manual_seed(W * 100 + limit)
    R = 200
    ids = torch.randint(2, 50, (R, W), generator=g)
    ids[torch.rand(R, W, generator=g) < 0.3] = 1820
    ids[torch.rand(R, W, generator=g) < 0.1] = 9175
    lens = torch.randint(1, W + 1, (R,), generator=g)
    ids[torch.arange(R), lens - 1] = 1  # EOS at the end of the content ...
    mask = (torch.arange(W)[None, :] < lens[:, None]).long()
    ids = ids * mask                    # ... zero padding behind it
    ids[0, :] = 7                       # a row without any EOS that is longer than the limit: EOS forced into the last slot
    mask[0, :] = 1
    a, b = col._strip_rows(ids, mask, limit), _strip_rows_loop(ids, mask, limit)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    ids[3, :] = 1820
    with pytest.raises(ValueError):
        col._strip_rows(ids, mask, limit)


def _dataset_and_collator(**kw):
    args = fixture_args(**kw)
    runner = get_runner("single", StubModel(), None, PieceTokenizer(), None, None, None, "cpu", args)
    loader = runner.testloaders[0]
    return runner, loader, loader.dataset, loader.collate_fn


def test_collator_row_table_is_result_neutral():
    """A batch collated by a collator that has seen other users before (rows cached), by a fresh one, and user by user
    (then padded like the Collator pads) are the same tensors."""
    _, _, data, warm = _dataset_and_collator()
    samples = [data[i] for i in range(len(data))]
    for s in samples:  # fill the table in another order, one user at a time
        warm([s])
    fresh = col.CollatorGRAM(PieceTokenizer(), args=fixture_args(), mode="test")
    a, b = warm(samples), fresh(samples)
    for k in ("item_text_ids", "item_text_masks", "target_ids", "target_masks"):
        assert torch.equal(a[k], b[k]), k
    assert a["item_text_masks"].dtype == torch.bool and a["user_ids"] == b["user_ids"]
    # one user at a time, merged by the runner = the big batch
    from gram_amd.runner.base import BaseRunner
    merged = BaseRunner._merge_batches([warm([s]) for s in samples])
    assert torch.equal(merged["item_text_ids"], a["item_text_ids"]) and torch.equal(merged["item_text_masks"], a["item_text_masks"])
    assert torch.equal(merged["target_ids"], a["target_ids"]) and merged["user_ids"] == a["user_ids"]
    # the per-user prompts are not kept (they would never be looked up again); item prompts are
    n_items_seen = len({p for s in samples for p in s["input"][1:]})
    assert len(warm._passage_rows.index) == n_items_seen
    ids, mask = warm.encode_passages(sorted(set(data.item2input.values())))
    assert ids.shape[1] == warm.item_prompt_max_len and mask.dtype == torch.bool


@pytest.mark.parametrize("per_call", [1, 5, 7, 100])
def test_gpu_batches_rebatch_the_loader_in_order(per_call):
    runner, loader, data, _ = _dataset_and_collator(eval_batch_size=1)
    got = list(runner._gpu_batches(loader, per_call))
    users = [u for b in got for u in b["user_ids"]]
    assert users == [data[i]["user_id"] for i in range(len(data))]
    sizes = [b["item_text_ids"].shape[0] for b in got]
    assert sizes[0] == min(max(1, per_call // 4), len(data)) and all(z == per_call for z in sizes[1:-1]) and sizes[-1] <= per_call

    class Plain:  # a loader that is not a DataLoader: iterated as it is, its batches merged
        def __init__(self, inner):
            self.inner, self.dataset = inner, inner.dataset

        def __iter__(self):
            return iter(self.inner)

    got2 = list(runner._gpu_batches(Plain(loader), per_call))
    assert [u for b in got2 for u in b["user_ids"]] == users
    whole, whole2 = runner._merge_batches(got), runner._merge_batches(got2)
    assert torch.equal(whole["item_text_ids"], whole2["item_text_ids"]) and torch.equal(whole["target_ids"], whole2["target_ids"])


class FillerStub(StubModel):
    """StubModel that returns a non-candidate filler row for every third user, like HF does when fewer than K hypotheses finish."""

    def generate(self, input_ids, attention_mask, max_length, prefix_allowed_tokens_fn, num_beams, **kw):
        out = StubModel.generate(self, input_ids, attention_mask, max_length, prefix_allowed_tokens_fn, num_beams, **kw)
        seqs = out["sequences"]
        for b in range(0, input_ids.shape[0], 3):
            seqs[b * num_beams + num_beams - 1] = 0  # start token + padding only
            out["sequences_scores"][b * num_beams + num_beams - 1] = -1e9
        return out


class ItemStub(FillerStub):
    """... and answers `sequence_items` (host walk of the flat Trie), like gram_amd.GRAM does on the device."""

    def sequence_items(self, sequences, fn, candidates):
        trie = fn.__closure__[0].cell_contents
        flat = gt.FlatTrie(trie)
        node_item = flat.node_items(candidates)
        out = []
        for row in sequences.tolist():
            while row and row[-1] == 0:
                row.pop()
            leaf = flat.leaf_of(row) if row else -1
            is_leaf = leaf > 0 and flat.child_off[leaf + 1] == flat.child_off[leaf]
            out.append(int(node_item[leaf]) if is_leaf else -1)
        return torch.tensor(out, dtype=torch.int32)


@pytest.mark.parametrize("eval_batch_size,gpu_batch", [(1, 0), (5, 4), (3, 1)])
def test_item_index_path_equals_decoding_every_row(tmp_path, eval_batch_size, gpu_batch):
    """The same evaluation through (a) a model that returns item indices (strings from one decode of the candidate list) and (b) one
    that does not (every generated row decoded, as the reference does): same hit ranks, sums and preds TSV, filler rows included."""
    outs = []
    for cls in (ItemStub, FillerStub):
        pred = str(tmp_path / f"{cls.__name__}.tsv")
        args = fixture_args(eval_batch_size=eval_batch_size, eval_gpu_batch=gpu_batch, save_predictions=True, pred_path=pred)
        model = cls()
        runner = get_runner("single", model, None, PieceTokenizer(), None, None, None, "cpu", args)
        runner.test_dataset_task(runner.testloaders[0])
        outs.append((runner.last_results, open(pred).read()))
    (a, ta), (b, tb) = outs
    assert a["total"] == b["total"] == 12
    assert a["hit_ranks"].tolist() == b["hit_ranks"].tolist() and np.array_equal(a["sums"], b["sums"])
    assert ta == tb


def test_node_items_first_duplicate_wins_and_rejects_foreign_candidates():
    cands = [[0, 2, 3, 1], [0, 2, 4, 1], [0, 5, 1], [0, 2, 3, 1]]
    flat = gt.FlatTrie(gt.Trie(cands))
    ni = flat.node_items(cands)
    assert ni[flat.leaf_of(cands[0])] == 0 and ni[flat.leaf_of(cands[1])] == 1 and ni[flat.leaf_of(cands[2])] == 2
    assert (ni >= 0).sum() == 3 and ni[0] == -1
    with pytest.raises(ValueError):
        flat.node_items([[0, 9, 1]])


def test_vectorised_metrics_equal_the_loops():
    rng = np.random.default_rng(3)
    m = "hit@1,hit@5,hit@10,ndcg@5,ndcg@10".split(",")
    ranks = rng.integers(-1, 20, size=3000)
    assert np.array_equal(evaluate.metrics_from_ranks(ranks, m, 20), evaluate.get_metrics_results(evaluate.rel_rows_from_ranks(ranks, 20), m))
    assert evaluate.metrics_from_ranks([], m, 20).tolist() == [0.0] * len(m)
    B, Kk = 500, 20
    scores = np.round(rng.standard_normal((B, Kk)), 1).astype(np.float32)  # many ties: the sort must be stable like the reference's
    pred, gold = rng.integers(0, 30, size=(B, Kk)), rng.integers(0, 35, size=B)
    rel = evaluate.rel_results([int(x) for x in pred.reshape(-1)], [int(x) for x in gold], scores.reshape(-1), Kk)
    assert np.array_equal(evaluate.hit_ranks(rel), evaluate.hit_ranks_from_ids(pred, scores, gold))


def test_flat_trie_from_sequences_equals_the_nested_dict_walk():
    """FlatTrie built straight from the added sequences (the lazy Trie never materialises its nested dict on the generate() path) is
    the CSR the breadth-first walk of the reference-shaped dict gives: random trees with shared prefixes, sequences that are
    prefixes of others, duplicates; and the lazily built dict is the dict the eager insert gives."""
    import random
    rng = random.Random(1)
    for _ in range(150):
        seqs = [[0] + [rng.randint(2, 6) for _ in range(rng.randint(0, 4))] + ([1] if rng.random() < 0.8 else [])
                for _ in range(rng.randint(1, 40))]
        fast = gt.FlatTrie(gt.Trie(seqs))
        t = gt.Trie(seqs)
        nested = t.trie_dict           # handing the dict out makes the sequence list non-authoritative: the walk is used
        assert t.sequences() is None
        slow = gt.FlatTrie(t)
        for name in ("child_off", "child_tok", "child_node"):
            assert np.array_equal(getattr(fast, name), getattr(slow, name)), (name, seqs)
        assert (fast.n_nodes, fast.n_edges, fast.max_fanout, fast.min_seq_len) == (slow.n_nodes, slow.n_edges, slow.max_fanout, slow.min_seq_len)
        leafy = [q for q in seqs if fast.child_off[fast.leaf_of(q) + 1] == fast.child_off[fast.leaf_of(q)]]
        if leafy:
            assert np.array_equal(fast.node_items(seqs), slow.node_items(seqs))
        eager = {}
        for q in seqs:
            node = eager
            for tok in q:
                node = node.setdefault(tok, {})
        assert nested == eager and sorted(map(tuple, t)) == sorted({tuple(q) for q in seqs if not any(
            len(o) > len(q) and o[: len(q)] == q for o in seqs)})
    t = gt.Trie([[0, 2, 1]])
    t.add([0, 3, 1])                  # add() after construction is seen by both forms
    assert gt.FlatTrie(t).n_nodes == 6 and t.get([0]) == [2, 3] and len(t) == 2


class TermStub(StubModel):
    """Users whose passage tokens sum to a multiple of 3 have one hypothesis short: decoded to the Trie's depth they get a -inf filler row
    (start token + padding); decoded with the reference's max_length = 50 the filler is a 50-token row, as HF's would be."""

    def __init__(self):
        super().__init__()
        self.calls = []

    @staticmethod
    def short(input_ids):
        return [b for b in range(input_ids.shape[0]) if int(input_ids[b].sum()) % 3 == 0]

    def generate(self, input_ids, attention_mask, max_length, prefix_allowed_tokens_fn, num_beams, **kw):
        out = StubModel.generate(self, input_ids, attention_mask, max_length, prefix_allowed_tokens_fn, num_beams, **kw)
        self.users -= input_ids.shape[0] if max_length == 50 else 0
        self.calls.append((input_ids.shape[0], max_length))
        for b in self.short(input_ids):
            row = b * num_beams + num_beams - 1
            out["sequences"][row] = 0
            if max_length == 50:
                out["sequences"][row, 1:] = 7
            out["sequences_scores"][row] = float("-inf")
        return out


@pytest.mark.parametrize("gpu_batch", [0, 5])
def test_term_ids_decode_to_the_trie_depth_and_rescore_unfinished_users(tmp_path, gpu_batch):
    """single_runner_gram.py:629-637: every id type but "t5_token" / "split" is decoded with max_length = 50.  The runner decodes to
    the Trie's depth and scores again, with 50, exactly the users that come back with a -inf row (fewer than K finished hypotheses:
    the only users for whom HF's result depends on max_length); their rows are replaced, everyone else's kept."""
    pred = str(tmp_path / "term.tsv")
    args = fixture_args(item_id_type="term", eval_batch_size=2, eval_gpu_batch=gpu_batch, save_predictions=True, pred_path=pred)
    model = TermStub()
    runner = get_runner("single", model, None, PieceTokenizer(), None, None, None, "cpu", args)
    loader = runner.testloaders[0]
    n_short = sum(len(TermStub.short(b["item_text_ids"])) for b in loader)
    assert 0 < n_short < len(loader.dataset)
    runner.test_dataset_task(loader)
    depth = max(len(c) for c in runner.encode_candidates(loader.dataset.all_items))
    assert depth < 50
    first = [c for c in model.calls if c[1] == depth]
    again = [c for c in model.calls if c[1] == 50]
    assert len(first) + len(again) == len(model.calls) and sum(n for n, _ in first) == len(loader.dataset)
    assert sum(n for n, _ in again) == n_short
    rows = open(pred).read().splitlines()[1:1 + len(loader.dataset)]
    garbage = " ".join(["7"] * 49)
    assert sum(garbage in r.split("\t")[-2].split("||") for r in rows) == n_short  # the 50-token fillers, decoded like any other row
    assert sum("-inf" in r.split("\t")[-1] for r in rows) == n_short
    assert runner.last_results["total"] == len(loader.dataset)
