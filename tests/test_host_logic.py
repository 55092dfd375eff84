"""CPU: host-side mirrors of the reference interface (Trie / FlatTrie, metrics, model wrapper,
runner factory, sharding) against the oracle and the golden vectors."""
import os
import random

import numpy as np
import pytest
import torch

import gram_amd
from gram_amd.model.gram import relative_position_bucket
from gram_amd.runner import get_runner, shard_indices
from gram_amd.utils import evaluate as ev
from gram_amd.utils import generation_trie as gt
from oracle import gram_oracle as O


def _rand_seqs(rng, n, lo, hi, vocab):
    return [[0] + [rng.randrange(2, vocab) for _ in range(rng.randrange(lo, hi + 1))] + [1] for _ in range(n)]


def test_trie_mirror_matches_oracle_and_flat_form():
    rng = random.Random(3)
    seqs = _rand_seqs(rng, 300, 1, 5, 12)
    a, b = gt.Trie(seqs), O.Trie(seqs)
    flat = gt.FlatTrie(a)
    assert len(a) == len(b) == 300
    probes = [[]] + [s[:k] for s in seqs[:80] for k in range(1, len(s) + 1)] + [[0, 99], [5], [0, 2, 2, 2, 2, 2, 2, 2]]
    for p in probes:
        assert sorted(a.get(p)) == sorted(b.get(p)) == flat.get(p), p
    fn = gt.prefix_allowed_tokens_fn(a)
    assert sorted(fn(0, torch.tensor([0]))) == sorted(b.get([0]))
    assert sorted(map(tuple, a)) == sorted({tuple(s) for s in seqs})
    assert gt.Trie.load_from_dict(a.trie_dict).len == len({tuple(s) for s in seqs})
    assert a[[0]] == a.get([0])
    assert gt.exact_match(["a", "b", "c", "d"], ["b", "x"], 2) == 1
    # CSR invariants
    assert flat.child_off[0] == 0 and flat.child_off[-1] == flat.n_edges == len(flat.child_tok)
    for n in range(flat.n_nodes):
        toks = flat.child_tok[flat.child_off[n]:flat.child_off[n + 1]]
        assert np.all(np.diff(toks) > 0)


def test_trie_golden_and_real_shapes(golden_dir):
    z = np.load(os.path.join(golden_dir, "ref_tiny.npz"))
    unpad = lambda rows: [[int(x) for x in r if x >= 0] for r in rows]
    t = gt.Trie(unpad(z["trie_cands"]))
    flat = gt.FlatTrie(t)
    for probe, ans in zip(unpad(z["trie_probes"]), unpad(z["trie_answers"])):
        assert sorted(t.get(probe)) == ans == flat.get(probe)
    tries = np.load(os.path.join(golden_dir, "tries.npz"))
    # node counts / fan-outs recorded in SURVEY.md §8 A11 (incl. EOS leaves and the start token)
    expect = {"Beauty": (12101, 255, 108), "Toys": (11924, 186, 30), "Sports": (18357, 175, 28), "Yelp": (20033, 263, 21)}
    for ds, (n_items, fan, root_fan) in expect.items():
        cands = unpad(tries[f"{ds}_cands"])
        f = gt.FlatTrie(gt.Trie(cands))
        assert f.n_sequences == n_items and f.max_fanout == fan and len(f.get([0])) == root_fan, ds


def test_metrics_mirror(golden_dir):
    z = np.load(os.path.join(golden_dir, "ref_tiny.npz"))
    rel = ev.rel_results(["a", "b", "c", "d"], ["c"], [-1.0, -3.0, -2.0, -4.0], 4)
    assert rel == z["rel_known"].tolist()
    assert np.array_equal(ev.get_metrics_results(rel, ["hit@1", "hit@2", "ndcg@2", "ndcg@4"]), z["met_known"])
    k = int(z["rand_k"])
    rel = ev.rel_results(z["rand_preds"].tolist(), z["rand_golds"].tolist(), z["rand_scores"].tolist(), k)
    assert rel == z["rel_rand"].tolist()
    mets = z["rand_metrics"].tolist()
    assert np.array_equal(ev.get_metrics_results(rel, mets), z["met_rand"])
    # compact hit-rank form used by the DP runner reproduces the sums when predictions are unique
    uniq = ev.rel_results(["a", "b", "c", "x", "y", "z"], ["c", "q"], [3, 2, 1, 3, 2, 1], 3)
    ranks = ev.hit_ranks(uniq)
    assert ranks.tolist() == [2, -1]
    assert np.array_equal(ev.metrics_from_ranks(ranks, ["hit@1", "hit@3", "ndcg@3"], 3), ev.get_metrics_results(uniq, ["hit@1", "hit@3", "ndcg@3"]))


def test_model_wrapper_contract():
    oc = O.OracleConfig(vocab_size=256, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_decoder_layers=2, num_heads=2, max_item_num=5)
    cfg = gram_amd.T5Config(vocab_size=256, d_model=128, d_ff=256, num_layers=2, num_decoder_layers=2, num_heads=2, max_item_num=5)
    m = gram_amd.create_model("gram", config=cfg)
    sd = O.init_state_dict(oc, 1)
    assert set(m.state_dict()) == set(sd)  # the reference checkpoint key layout
    m.load_state_dict(sd)
    assert torch.equal(m.state_dict()["lm_head.weight"], sd["shared.weight"])
    assert m.state_dict()["encoder.position_embedding.weight"].data_ptr() == m.state_dict()["position_embedding.weight"].data_ptr()
    # load_t5: plain T5 key layout, non-strict, position embedding untouched
    t5 = {k.replace("encoder.encoder.block.", "encoder.block.").replace(".module.", ".").replace("encoder.encoder.", "encoder."): v + 1
          for k, v in sd.items() if "position_embedding" not in k}
    pos_before = m.state_dict()["position_embedding.weight"].clone()
    m.load_t5(t5)
    assert torch.equal(m.state_dict()["encoder.encoder.block.1.module.layer.1.DenseReluDense.wi.weight"],
                       sd["encoder.encoder.block.1.module.layer.1.DenseReluDense.wi.weight"] + 1)
    assert torch.equal(m.state_dict()["position_embedding.weight"], pos_before)
    with pytest.raises(ValueError):
        gram_amd.create_model("p5", config=cfg)
    with pytest.raises(ValueError):
        get_runner("multi", m, None, None, None, None, None, "cpu", None)
    # no CPU fallback: generate on a CPU model fails loudly
    fn = gt.prefix_allowed_tokens_fn(gt.Trie([[0, 2, 1]]))
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.generate(torch.zeros(1, 1, 32, dtype=torch.long), torch.ones(1, 1, 32, dtype=torch.bool), 3, prefix_allowed_tokens_fn=fn, num_beams=1)
    with pytest.raises(NotImplementedError):
        m(input_ids=torch.zeros(1, 1, 32, dtype=torch.long))


def test_relative_bucket_tables_match_oracle():
    rel = torch.arange(-127, 128)
    assert torch.equal(relative_position_bucket(rel, True, 32, 128), O.relative_position_bucket(rel, True, 32, 128))
    d = -torch.arange(0, 32)
    assert torch.equal(relative_position_bucket(d, False, 32, 128), O.relative_position_bucket(d, False, 32, 128))


def test_shard_indices_vs_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler
    n, W = 23, 4
    data = list(range(n))
    seen = []
    for r in range(W):
        ref = list(DistributedSampler(data, num_replicas=W, rank=r))  # shuffle=True, seed=0 (distributed_runner_gram.py:351)
        assert shard_indices(n, W, r, pad_like_reference=True) == ref == O.distributed_sampler_indices(n, W, r)
        seen += shard_indices(n, W, r)
    assert sorted(seen) == data  # default sharding: every user exactly once


def test_collator_matches_reference_fixtures(golden_dir):
    """gram_amd.processor.CollatorGRAM against outputs of the reference's own CollatorGRAM (Collator.py:152-450) on
    the same batches and the same stub tokenizer (oracle/make_collator_fixtures.py): split / t5_token / plain id
    types, passage padding, separator stripping with forced EOS, target -100 padding, L trimming -- bit for bit."""
    import json
    from types import SimpleNamespace

    from gram_amd.processor import CollatorGRAM
    from tests.stub_tokenizer import StubTokenizer
    cases = json.load(open(os.path.join(golden_dir, "collator_cases.json")))
    assert {c["args"]["item_id_type"] for c in cases} == {"split", "t5_token", "other"}
    for c in cases:
        out = CollatorGRAM(StubTokenizer(), SimpleNamespace(**c["args"]), mode="test")(c["batch"])
        assert out["item_text_ids"].dtype == torch.int64 and out["item_text_masks"].dtype == torch.bool
        assert out["item_text_ids"].tolist() == c["item_text_ids"]
        assert out["item_text_masks"].long().tolist() == c["item_text_masks"]
        assert out["target_ids"].tolist() == c["target_ids"]
        assert out["target_masks"].long().tolist() == c["target_masks"]
        assert out["user_ids"] == c["user_ids"] and out["neg_item_ids"] is None
        # stand-alone tokenisation of the same passages (what the passage cache is filled with) = the stacked rows
        col = CollatorGRAM(StubTokenizer(), SimpleNamespace(**c["args"]), mode="test")
        W = out["item_text_ids"].shape[-1]
        for b, x in enumerate(c["batch"]):
            p_ids, p_mask = col.encode_passages(x["input"])
            n = len(x["input"])
            assert torch.equal((p_ids * p_mask)[:, :W], out["item_text_ids"][b, :n] * out["item_text_masks"][b, :n])
            assert torch.equal(p_mask[:, :W], out["item_text_masks"][b, :n]) and not p_mask[:, W:].any()
    # the reference cannot stack a user with more passages than slots (it raises inside torch.cat); the mirror says why
    args = SimpleNamespace(item_prompt_max_len=16, target_max_len=8, max_his=1, item_id_type="split", hierarchical_id_type="none")
    with pytest.raises(ValueError):
        CollatorGRAM(StubTokenizer(), args)([{"input": ["a b"], "output": "x", "user_id": "u"},
                                              {"input": ["a", "b", "c"], "output": "y", "user_id": "v"}])


def test_dataset_and_indexing_match_reference_fixtures(golden_dir, monkeypatch):
    """gram_amd.data.TestDatasetGRAM / gram_amd.utils.indexing.gram_indexing against the outputs of the reference's own
    classes (test_dataset_gram.py:19-231, indexing.py:132-322) on the synthetic dataset directory
    tests/golden/dataset_fixture (oracle/make_dataset_fixtures.py): every item_prompt kind, id_linking, max_his, history
    order, validation/test hold-out, an alternative item id file, Amazon- and Yelp-style attribute lines."""
    import json
    from types import SimpleNamespace

    from gram_amd.data import TestDatasetGRAM
    from gram_amd.utils import indexing
    cases = json.load(open(os.path.join(golden_dir, "dataset_cases.json")))
    assert {c["args"]["item_prompt"] for c in cases} == {"all_text", "lexical_id", "nothing", "only_title", "only_brand",
                                                         "only_category", "only_tbc"}
    monkeypatch.chdir(golden_dir)  # the fixtures hold paths relative to tests/golden
    for c in cases:
        ds = TestDatasetGRAM(SimpleNamespace(**c["args"]), c["dataset"], "sequential", None, None, mode=c["mode"])
        assert len(ds) == c["len"] and ds.all_items == c["all_items"]
        assert ds.item2input == c["item2input"] and list(ds.item2input) == list(c["item2input"])
        assert ds.item2lexid == c["item2lexid"] and ds.user_seq_dict == c["user_seq_dict"]
        assert [ds[i] for i in range(len(ds))] == c["samples"]
        assert [s["history"] for s in ds.data_samples] == c["history"] and sorted(ds.info) == c["info"]
    # failure modes of the reference that callers rely on
    a = dict(cases[0]["args"])
    with pytest.raises(ValueError):
        TestDatasetGRAM(SimpleNamespace(**a), "Beauty", "sequential", None, None, mode="train")
    with pytest.raises(FileNotFoundError):  # indexing.py:169-172
        indexing.gram_indexing("dataset_fixture", "Beauty", None, None, args=SimpleNamespace(**dict(a, hierarchical_id_type="nope")))
    with pytest.raises(ValueError):  # text prompts need top_k_similar_item > 0 (indexing.py:183-207)
        indexing.gram_indexing("dataset_fixture", "Beauty", None, None, args=SimpleNamespace(**dict(a, top_k_similar_item=0)))
    with pytest.raises(AssertionError):
        TestDatasetGRAM(SimpleNamespace(**a), "Beauty", "rating", None, None)


def test_flat_trie_cache_tracks_the_trie_object():
    """Two same-sized Tries built one after the other (CPython recycles the address of the first) must not share a
    cached CSR: the cache entry pins its source Trie and is keyed by object identity."""
    cfg = gram_amd.T5Config(vocab_size=256, d_model=128, d_ff=256, num_layers=1, num_decoder_layers=1, num_heads=2, max_item_num=3)
    m = gram_amd.create_model("gram", cfg)
    seen = []
    for base in (10, 20, 30, 40):
        trie = gt.Trie([[0, base + i, base + 5 + i, 1] for i in range(4)])
        flat = m._flat_trie(gt.prefix_allowed_tokens_fn(trie))
        seen.append(sorted(int(t) for t in flat.child_tok))
        del trie, flat
    assert all(seen[i] != seen[j] for i in range(4) for j in range(i))
    trie = gt.Trie([[0, 2, 3, 1]])
    fn = gt.prefix_allowed_tokens_fn(trie)
    a = m._flat_trie(fn)
    assert m._flat_trie(fn) is a          # same object: cached
    trie.add([0, 2, 4, 1])
    assert m._flat_trie(fn) is not a      # grown in place: rebuilt


def test_dropin_launcher_resolves_bare_names_to_gram_amd(tmp_path):
    """`python -m gram_amd.dropin script.py`: a script living next to its OWN `model` / `runner` packages (like
    main_generative_gram.py in src/) gets gram_amd's instead, while its other sibling imports still resolve locally."""
    import subprocess
    import sys
    src = tmp_path / "src"
    for pkg in ("model", "runner"):
        (src / pkg).mkdir(parents=True)
        (src / pkg / "__init__.py").write_text("raise ImportError('the reference package was imported')\n")
    (src / "arguments.py").write_text("FLAG = 'reference arguments module'\n")
    (src / "main.py").write_text(
        "import sys\nfrom runner import get_runner\nfrom model import create_model\nfrom arguments import FLAG\n"
        "print(get_runner.__module__, create_model.__module__, FLAG, sys.argv[1:])\n"
        "try:\n    get_runner('nope', *[None] * 8)\nexcept ValueError as e:\n    print('ValueError', e)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "gram_amd.dropin", str(src / "main.py"), "--datasets", "Beauty"], cwd=root,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "gram_amd.runner gram_amd.model reference arguments module ['--datasets', 'Beauty']" in out.stdout
    assert "ValueError Unknown runner type: nope" in out.stdout


def test_interleaved_two_piece_layout():
    """gram_hip.h: a two-piece GEMM operand is [rows][cols / 32][2][32] -- element (n, piece p) of a row sits at
    (n / 32) * 64 + p * 32 + n % 32 -- and ``_lib.deinterleave`` inverts ``_lib.interleave``."""
    from gram_amd import _lib
    rows, cols = 5, 96
    p0 = torch.arange(rows * cols, dtype=torch.float32).view(rows, cols)
    pieces = torch.stack([p0, -p0 - 1])
    x = _lib.interleave(pieces)
    assert x.shape == (rows, 2 * cols)
    for n in (0, 1, 31, 32, 63, 64, 95):
        for p in (0, 1):
            assert torch.equal(x[:, (n // 32) * 64 + p * 32 + n % 32], pieces[p][:, n]), (n, p)
    assert torch.equal(_lib.deinterleave(x), pieces)


def test_error_codes_of_the_c_abi_map_to_messages():
    """include/gram_hip.h's GRAM_E_* codes and the text GramHipError carries (the header is the contract: the numbers are read from it)."""
    import re
    from gram_amd import _lib
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "gram_hip.h")).read()
    codes = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define (GRAM_E_\w+) \((-\d+)\)", hdr)}
    assert codes == {"GRAM_E_ARG": _lib.E_ARG, "GRAM_E_WORKSPACE": _lib.E_WORKSPACE, "GRAM_E_BEAM": _lib.E_BEAM, "GRAM_E_NONFINITE": _lib.E_NONFINITE}
    _lib.check(0, "ok")
    for name, code in codes.items():
        with pytest.raises(_lib.GramHipError, match=name):
            _lib.check(code, "call")
    with pytest.raises(_lib.GramHipError, match="bfloat16"):  # the way out of an activation overflow is named
        _lib.check(_lib.E_NONFINITE, "gram_generate")
    with pytest.raises(_lib.GramHipError, match="hipError 719"):
        _lib.check(719, "call")
