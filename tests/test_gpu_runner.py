"""-m gpu: the drop-in entry points with the HIP model and nothing else -- ``get_runner(kind, ...).test(ckpt_path)`` as
main_generative_gram.py:107-127,191-209 calls them, with a reference-shaped args namespace, on the dataset directory
tests/golden/dataset_fixture: the runner builds its own dataset / collator / loaders, loads the checkpoint, scores every
user through gram_generate and writes the preds TSV.  The distributed runner is run as 5 ranks sharing the one test GPU
(gloo for the collectives; RCCL needs a GPU per rank) on a user count the world size does not divide, and must
reproduce the single runner's result."""

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.test_runner_loaders import K, PieceTokenizer, fixture_args

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfg():
    import gram_amd
    return gram_amd.T5Config(vocab_size=32128, d_model=256, d_ff=512, num_layers=2, num_decoder_layers=2, num_heads=4, max_item_num=20)


def _checkpoint(path):
    import gram_amd
    torch.manual_seed(77)
    m = gram_amd.create_model("gram", _cfg())
    torch.save(m.state_dict(), path)


def test_single_runner_test_and_validate_from_checkpoint(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import gram_amd
    from gram_amd.runner import get_runner
    ckpt = str(tmp_path / "model_rec_best.pt")
    _checkpoint(ckpt)
    args = fixture_args(save_predictions=True, pred_path=str(tmp_path / "preds.tsv"), eval_batch_size=5, passage_cache=1)
    torch.manual_seed(1)  # different init: the scores must come from the checkpoint
    model = gram_amd.create_model("gram", _cfg()).to(DEV)
    runner = get_runner("single", model, None, PieceTokenizer(), None, None, None, DEV, args)
    runner.test(ckpt)
    res = runner.last_results
    n = len(runner.testloaders[0].dataset)
    assert res["total"] == n == 12
    lines = open(args.pred_path).read().splitlines()
    rows = [ln.split("\t") for ln in lines[1:1 + n]]
    items = {" ".join(str(t) for t in c if t > 1) for c in runner.encode_candidates(runner.testloaders[0].dataset.all_items)}
    for r in rows:
        preds, scores = r[-2].split("||"), [float(x) for x in r[-1].split("||")]
        assert len(preds) == K and len(set(preds)) == K and all(p in items for p in preds)
        assert scores == sorted(scores, reverse=True)
    # the checkpoint decides the result: the freshly initialised weights give other scores
    torch.manual_seed(1)
    other = gram_amd.create_model("gram", _cfg()).to(DEV)
    r2 = get_runner("single", other, None, PieceTokenizer(), None, None, None, DEV, fixture_args(eval_batch_size=5))
    r2.test_dataset_task(r2.testloaders[0])
    assert r2.last_results["total"] == n
    runner.validate(ckpt)
    assert runner.last_results["total"] == len(runner.validloaders[0].dataset) == 12


def test_rebatched_runner_equals_one_user_per_call(tmp_path):
    """The drop-in at the reference's default flag (--eval_batch_size 1): the runner scores many users per generate() call and
    takes item indices from the device; the result -- hit ranks, sums, the whole preds TSV -- is that of one user per call with
    every generated row decoded on the host (a user's scores do not depend on the batch it is scored in, bit for bit)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import gram_amd
    from gram_amd.runner import get_runner
    ckpt = str(tmp_path / "model_rec_best.pt")
    _checkpoint(ckpt)
    res = []
    for tag, kw in (("rebatched", dict(eval_batch_size=1)), ("one_by_one", dict(eval_batch_size=1, eval_gpu_batch=1)),
                    ("sevens", dict(eval_batch_size=4, eval_gpu_batch=7))):
        pred = str(tmp_path / f"{tag}.tsv")
        model = gram_amd.create_model("gram", _cfg()).to(DEV)
        runner = get_runner("single", model, None, PieceTokenizer(), None, None, None, DEV,
                            fixture_args(save_predictions=True, pred_path=pred, passage_cache=int(tag != "one_by_one"), **kw))
        if tag == "one_by_one":
            runner._generate_model = lambda m=model: _NoItems(m)
        runner.test(ckpt)
        res.append((runner.last_results, open(pred).read(), runner.last_host["users_per_call"]))
    (a, ta, ca), (b, tb, cb), (c, tc, cc) = res
    assert ca >= 12 and cb == 1 and cc == 7
    assert a["total"] == b["total"] == c["total"] == 12
    assert a["hit_ranks"].tolist() == b["hit_ranks"].tolist() == c["hit_ranks"].tolist()
    assert np.array_equal(a["sums"], b["sums"]) and np.array_equal(a["sums"], c["sums"])
    assert ta == tb == tc


class _At50:
    """A model view whose generate() always runs with the reference's max_length = 50 for the "term" id type (single_runner_gram.py:637),
    whatever the runner passes."""

    def __init__(self, m):
        self._m = m
        self.lengths = []

    def __getattr__(self, name):
        return getattr(self._m, name)

    def generate(self, *a, **kw):
        self.lengths.append(kw["max_length"])
        kw["max_length"] = 50
        return self._m.generate(*a, **kw)


def test_term_id_type_decoded_to_the_trie_depth_equals_max_length_50(tmp_path):
    """`--item_id_type term`: the reference decodes with max_length = 50; the runner decodes to the candidate Trie's depth.  Same hit ranks,
    sums and preds TSV, byte for byte, as the same runner over a model that decodes to 50 tokens."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import gram_amd
    from gram_amd.runner import get_runner
    ckpt = str(tmp_path / "model_rec_best.pt")
    _checkpoint(ckpt)
    res = []
    for tag in ("depth", "fifty"):
        pred = str(tmp_path / f"{tag}.tsv")
        model = gram_amd.create_model("gram", _cfg()).to(DEV)
        runner = get_runner("single", model, None, PieceTokenizer(), None, None, None, DEV,
                            fixture_args(item_id_type="term", save_predictions=True, pred_path=pred, eval_batch_size=3))
        view = _At50(model)
        if tag == "fifty":
            runner._generate_model = lambda v=view: v
        runner.test(ckpt)
        res.append((runner.last_results, open(pred).read(), view.lengths))
    (a, ta, _), (b, tb, lens) = res
    depth = max(len(c) for c in runner.encode_candidates(runner.testloaders[0].dataset.all_items))
    assert lens and all(x == depth for x in lens) and depth < 50  # (what the runner asked for; the view decoded to 50 instead)
    assert a["total"] == b["total"] == 12
    assert a["hit_ranks"].tolist() == b["hit_ranks"].tolist() and np.array_equal(a["sums"], b["sums"])
    assert ta == tb and "-inf" not in ta


class _NoItems:
    """A model view without `sequence_items`: the runner then decodes every generated row, like the reference."""

    def __init__(self, m):
        self._m = m

    def generate(self, *a, **kw):
        return self._m.generate(*a, **kw)

    def cache_passages(self, *a, **kw):
        return self._m.cache_passages(*a, **kw)


def _worker(rank, world, rdzv, ckpt, pred, q):
    import gram_amd
    from gram_amd.runner import get_runner
    dist.init_process_group("gloo", init_method=f"file://{rdzv}", rank=rank, world_size=world)
    try:
        args = fixture_args(eval_batch_size=2, rank=rank, save_predictions=True, pred_path=pred)
        model = gram_amd.create_model("gram", _cfg()).to(DEV)
        runner = get_runner("distributed", model, None, PieceTokenizer(), None, None, None, DEV, args, rank)
        runner.test(ckpt)
        q.put((rank, len(runner.last_results["local_hit_ranks"]), runner.last_results["total"], runner.last_results["sums"].tolist(),
               sorted(runner.last_results["hit_ranks"].tolist())))
    finally:
        dist.destroy_process_group()


def test_distributed_runner_five_ranks_one_gpu_matches_single(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import gram_amd
    from gram_amd.runner import get_runner
    ckpt = str(tmp_path / "model_rec_best.pt")
    _checkpoint(ckpt)
    p_single, p_dist = str(tmp_path / "single.tsv"), str(tmp_path / "dist.tsv")
    model = gram_amd.create_model("gram", _cfg()).to(DEV)
    single = get_runner("single", model, None, PieceTokenizer(), None, None, None, DEV,
                        fixture_args(eval_batch_size=2, save_predictions=True, pred_path=p_single))
    single.test(ckpt)
    want = single.last_results
    world = 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, str(tmp_path / "rdzv"), ckpt, p_dist, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=600) for _ in procs)
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert [r[1] for r in res] == [3, 3, 2, 2, 2] and all(r[2] == 12 for r in res)
    assert all(np.allclose(r[3], want["sums"]) for r in res)
    assert all(r[4] == sorted(want["hit_ranks"].tolist()) for r in res)
    # same rows in the merged TSV (batch invariance of the kernels: a user's scores do not depend on its batch)
    a, b = open(p_single).read().splitlines(), open(p_dist).read().splitlines()
    assert a[0] == b[0] and sorted(a[1:13]) == sorted(b[1:13])


def _rccl_worker(rdzv, ckpt, q):
    import gram_amd
    from gram_amd.runner import all_gather_hit_ranks, get_runner
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"file://{rdzv}", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        got = all_gather_hit_ranks(np.array([3, -1, 0, 7], dtype=np.int16), torch.device(DEV))
        args = fixture_args(eval_batch_size=4, rank=0, eval_check_allreduce=1)  # (the reference's all_reduce cross-check over RCCL too)
        model = gram_amd.create_model("gram", _cfg()).to(DEV)
        runner = get_runner("distributed", model, None, PieceTokenizer(), None, None, None, DEV, args, 0)
        runner.test(ckpt)
        q.put((got.tolist(), runner.last_results["total"], runner.last_results["sums"].tolist(), dist.get_backend()))
    finally:
        dist.destroy_process_group()


def test_distributed_runner_over_rccl_single_rank(tmp_path):
    """backend="nccl" (= RCCL on ROCm) on the one GPU of the test box: the runner's all-gather / all-reduce run on device tensors
    through RCCL itself (world size 1 -- more ranks need more GPUs; the 5-rank test above covers the sharding over gloo)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import gram_amd
    from gram_amd.runner import get_runner
    ckpt = str(tmp_path / "model_rec_best.pt")
    _checkpoint(ckpt)
    model = gram_amd.create_model("gram", _cfg()).to(DEV)
    single = get_runner("single", model, None, PieceTokenizer(), None, None, None, DEV, fixture_args(eval_batch_size=4))
    single.test(ckpt)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(str(tmp_path / "rdzv"), ckpt, q))
    p.start()
    got, total, sums, backend = q.get(timeout=600)
    p.join(120)
    assert p.exitcode == 0 and backend == "nccl"
    assert got == [3, -1, 0, 7] and total == 12 and np.allclose(sums, single.last_results["sums"])
