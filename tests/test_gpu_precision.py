"""-m gpu: the north-star precision bound, asserted in the arithmetic bench.py times, on every population.

Recall@5 / NDCG@5 of the HIP path must be within 1e-4 of the reference's, which computes in fp32 end to end.  The
reference arithmetic at population scale is the oracle itself (oracle/gram_oracle.py, plain torch fp32 ops) with its
tensors on the GPU -- see tests/precision_population.py.  Populations: T5-base users, 3 x 128-token passages, the Beauty
Trie, beam 20; the gold item of user u sits at the reference's rank u mod 10, so every rank flip inside the top 10 moves a
metric.  16 384 users per population: ONE user across the rank-5 boundary is 6.1e-5 of Recall@5, so the instrument
resolves the 1e-4 bound it asserts (two flipped users would already fail it).

  plain       random-init weights, all-valid masks
  sharpened   every attention q projection x 4: peaky attention, the encoder states actually steer the scores as in a
              trained model (at T5's random init every query averages ~140 keys and all users get similar beams)
  ragged      Collator-shaped masks: valid lengths U[32, 128], fully padded passages (4 096 users: zero flips required)

  trained     weights shaped by gradient descent: Adam steps on a synthetic retrieval task through the oracle's own functions (plain torch
              fp32 autograd on the GPU), then users of that task.  In the suite on T5-small (200 steps, 4 096 users); on T5-base, 1 200
              steps: profiles/r04v_precision_trained_t5base.json (1 rank flip at the rank-10 line, Recall@5 / NDCG@5 unchanged)

Parity on a trained GRAM checkpoint is unpinned: none exists offline.

One 16-bit piece per value misses the bound by ~10-50x (profiles/r03a_precision_*); that is measured again here on a small
sample and reported, not asserted, since bench.py does not time that mode."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BOUND = 1e-4
USERS = 16384


def _run(**kw):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import bench
    from tests import precision_population as pp
    mode = bench.DEFAULT_PRECISION
    kw = dict(kw)
    res = pp.run(chunk=256, backbone=kw.pop("backbone", "t5-base"), dataset="Beauty", modes=(mode,), log=lambda *_: None, **kw)
    m = res["modes"][mode]
    print(f"\n[precision] {mode} {kw}: rank flips {m['rank_flips']}, membership changes {m['membership_changes']}, |delta| {m['abs_delta']}, "
          f"max |score dev| {m['max_abs_score_dev']:.2e}, swaps by gap {[(g['gap'], g['pairs'], g['swapped']) for g in m['adjacent_pair_swaps_by_reference_gap']]}")
    assert m["users"] == kw["users"]
    m["population"] = res["population"]
    return m


def test_headline_mode_meets_the_recall_ndcg_bound():
    m = _run(users=USERS)
    assert m["abs_delta"]["hit@5"] <= BOUND and m["abs_delta"]["ndcg@5"] <= BOUND, m["abs_delta"]
    # and the scores themselves are fp32-class (observed 4e-6; one bf16 piece is at ~7e-3 here)
    assert m["max_abs_score_dev"] < 4e-5


def test_attention_sharpened_population():
    m = _run(users=USERS, sharpen=4.0)
    assert m["abs_delta"]["hit@5"] <= BOUND and m["abs_delta"]["ndcg@5"] <= BOUND, m["abs_delta"]
    assert m["max_abs_score_dev"] < 1e-4  # observed 1.1e-5


def test_ragged_mask_population():
    m = _run(users=4096, ragged=True)
    assert m["abs_delta"]["hit@5"] <= BOUND and m["abs_delta"]["ndcg@5"] <= BOUND, m["abs_delta"]
    assert m["max_abs_score_dev"] < 1e-4


def test_briefly_trained_population():
    """4 096 users after 200 steps (the fp32 reference's host-side search is most of this test's time): ONE user across the rank-5 line is
    2.4e-4 here, so the assertion allows exactly that one; the 16 384-user measurements of this population (T5-small, 300 steps: 2 rank
    flips, none across the line; T5-base, 1 200 steps: profiles/r04v_precision_trained_t5base.json) are inside the 1e-4 bound."""
    m = _run(users=4096, backbone="t5-small", train_steps=200)
    tr = m["population"]["trained"]
    assert tr["loss_last10"] < 0.5 * tr["loss_first10"], tr  # the optimiser did shape the weights
    assert m["abs_delta"]["hit@5"] <= 2.5e-4 and m["abs_delta"]["ndcg@5"] <= BOUND, m["abs_delta"]
    assert m["max_abs_score_dev"] < 1e-3


def test_one_piece_is_reported_not_asserted():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import bench
    from tests import precision_population as pp
    mode = bench.DEFAULT_PRECISION[:-2]  # "f16x3" -> "f16"
    res = pp.run(users=512, chunk=256, backbone="t5-base", dataset="Beauty", modes=(mode,), log=lambda *_: None)
    m = res["modes"][mode]
    print(f"\n[precision] {mode}, 512 users: rank flips {m['rank_flips']}, |delta| {m['abs_delta']}, max |score dev| {m['max_abs_score_dev']:.2e}")
    assert m["max_abs_score_dev"] < 0.05  # sanity only
