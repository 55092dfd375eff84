"""-m gpu: the north-star precision bound, asserted in the arithmetic bench.py times.

Recall@5 / NDCG@5 of the HIP path must be within 1e-4 of the reference's, which computes in fp32 end to end.  The
reference arithmetic at population scale is the oracle itself (oracle/gram_oracle.py, plain torch fp32 ops) with its
tensors on the GPU -- see tests/precision_population.py.  Population: 4 096 T5-base users, 3 x 128-token passages, the
Beauty Trie, beam 20; the gold item of user u sits at the reference's rank u mod 10, so every rank flip inside the top
10 moves a metric (one flipped user of 4 096 is 2.4e-4 of Recall@5: the bound allows none across the rank-5 boundary).

Plain bf16 operands miss the bound by ~50x (profiles/r02a_precision_bf16_*.json: |dRecall@5| 5.9e-3 on 2 048 users);
that is measured again here on a small sample and reported, not asserted, since bench.py does not time that mode."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BOUND = 1e-4


def test_headline_mode_meets_the_recall_ndcg_bound():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import bench
    from tests import precision_population as pp
    mode = bench.DEFAULT_PRECISION
    res = pp.run(users=4096, chunk=256, backbone="t5-base", dataset="Beauty", modes=(mode,), log=lambda *_: None)
    m = res["modes"][mode]
    print(f"\n[precision] {mode}, 4096 users: rank flips {m['rank_flips']}, |delta| {m['abs_delta']}, max |score dev| "
          f"{m['max_abs_score_dev']:.2e}, swaps by gap {[(g['gap'], g['pairs'], g['swapped']) for g in m['adjacent_pair_swaps_by_reference_gap']]}")
    assert m["users"] == 4096
    assert m["abs_delta"]["hit@5"] <= BOUND and m["abs_delta"]["ndcg@5"] <= BOUND, m["abs_delta"]
    # and the scores themselves are fp32-class: plain bf16 operands are at ~7e-3 here
    assert m["max_abs_score_dev"] < 2e-4


def test_attention_sharpened_population():
    """The same bound on weights whose attention is peaky (every q projection x4: the encoder states actually steer the
    scores, as in a trained model -- at T5's random init every query averages ~140 keys and all users get similar beams)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import bench
    from tests import precision_population as pp
    mode = bench.DEFAULT_PRECISION
    res = pp.run(users=2048, chunk=256, backbone="t5-base", dataset="Beauty", modes=(mode,), sharpen=4.0, log=lambda *_: None)
    m = res["modes"][mode]
    print(f"\n[precision, q x4] {mode}, 2048 users: rank flips {m['rank_flips']}, |delta| {m['abs_delta']}, max |score dev| {m['max_abs_score_dev']:.2e}")
    assert m["abs_delta"]["hit@5"] <= 5e-4 and m["abs_delta"]["ndcg@5"] <= 5e-4, m["abs_delta"]  # one user of 2 048 = 4.9e-4


def test_plain_bf16_is_reported_not_asserted():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from tests import precision_population as pp
    res = pp.run(users=512, chunk=256, backbone="t5-base", dataset="Beauty", modes=("bf16",), log=lambda *_: None)
    m = res["modes"]["bf16"]
    print(f"\n[precision] bf16, 512 users: rank flips {m['rank_flips']}, |delta| {m['abs_delta']}, max |score dev| {m['max_abs_score_dev']:.2e}")
    assert m["max_abs_score_dev"] < 0.05  # sanity only
