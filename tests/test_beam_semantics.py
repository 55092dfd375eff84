"""CPU: the restated HF-4.26 beam search (oracle) on hand-worked variable-length cases.

The reference leaves beam search to transformers==4.26.0, whose source is not in the container, and
holds no tests for it: these cases are worked by hand against the published 4.26
rules -- top-2K candidates in rank order; an EOS candidate is banked only if its rank < K, scored
sum_logprobs / len(prefix)^length_penalty; the worst of K+1 banked hypotheses is evicted; a user is
done when K are banked and the worst banked score >= best candidate / cur_len^lp; finalize returns the
banked hypotheses best-first, EOS appended, 0-padded."""
import math

import pytest
import torch

from oracle import gram_oracle as O


def _rows(*dists, V=8):
    """logits whose log_softmax is exactly log(p): rows given as {token: prob}, remainder on token 7."""
    out = []
    for d in dists:
        p = torch.full((V,), 1e-30)
        p[7] = 1.0 - sum(d.values())
        for t, v in d.items():
            p[t] = v
        out.append(torch.log(p))
    return torch.stack(out)


def _run(step_rows, cands, K, lp=1.0, early_exit=True):
    it = iter(step_rows)
    trie = O.Trie(cands)
    return O.beam_search(lambda tok: next(it), lambda idx: None, 1, K, max(len(c) for c in cands),
                         O.prefix_allowed_tokens_fn(trie), lp, early_exit=early_exit)


def test_hand_worked_short_item_wins_on_length_normalised_score():
    cands = [[0, 2, 1], [0, 3, 4, 1], [0, 3, 5, 1]]
    steps = [
        _rows({2: 0.6, 3: 0.3}, {2: 0.6, 3: 0.3}),  # cur_len 1: both beams at [0] (beam 1 carries -1e9)
        _rows({1: 0.9}, {4: 0.5, 5: 0.4}),           # cur_len 2: [0,2] -> EOS ; [0,3] -> 4 | 5
        _rows({1: 0.8}, {1: 0.95}),                  # cur_len 3: [0,3,4] -> EOS ; [0,3,5] -> EOS
    ]
    seqs, scores = _run(steps, cands, K=2)
    a = (math.log(0.6) + math.log(0.9)) / 2          # [0,2] + EOS, normalised by len([0,2]) = 2
    b = (math.log(0.3) + math.log(0.5) + math.log(0.8)) / 3
    c = (math.log(0.3) + math.log(0.4) + math.log(0.95)) / 3
    assert c < b < a                                  # C is banked third and evicted at once
    assert seqs.tolist() == [[0, 2, 1, 0], [0, 3, 4, 1]]
    assert scores.tolist() == pytest.approx([a, b], abs=1e-6)


def test_eos_candidate_beyond_rank_k_is_dropped():
    # K=1: ranks 0..1 are inspected; the EOS continuation sits at rank 1 (>= K) and must be skipped
    cands = [[0, 2, 1], [0, 2, 3, 1]]
    steps = [_rows({2: 0.9}), _rows({3: 0.6, 1: 0.3}), _rows({1: 0.5})]
    seqs, scores = _run(steps, cands, K=1)
    assert seqs.tolist() == [[0, 2, 3, 1]]
    assert scores.tolist() == pytest.approx([(math.log(0.9) + math.log(0.6) + math.log(0.5)) / 3], abs=1e-6)


def test_length_penalty_changes_the_winner():
    cands = [[0, 2, 1], [0, 3, 4, 5, 1]]
    # beams are kept in score order: after step 0 beam 0 = [0,3] (0.6), beam 1 = [0,2] (0.3)
    steps = [_rows({2: 0.3, 3: 0.6}, {2: 0.3, 3: 0.6}), _rows({4: 0.9}, {1: 0.9}), _rows({5: 0.9}, {7: 1.0 - 1e-6}),
             _rows({1: 0.9}, {7: 1.0 - 1e-6})]
    short = math.log(0.3) + math.log(0.9)
    long = math.log(0.6) + 3 * math.log(0.9)
    for lp, first in [(0.0, [0, 3, 4, 5, 1]), (-1.0, [0, 2, 1, 0, 0])]:  # scores are negative: lp < 0 favours short
        seqs, scores = _run([s.clone() for s in steps], cands, K=2, lp=lp)
        assert seqs[0].tolist() == first, lp
        want = sorted([short / 2 ** lp, long / 4 ** lp], reverse=True)
        assert scores.tolist() == pytest.approx(want, abs=1e-6)


@pytest.mark.parametrize("seed", range(6))
def test_early_exit_is_result_neutral(seed):
    """Stepping on to max_length after every user is done (what the device loop does, to avoid a
    per-step host sync) gives exactly the results of HF's early exit."""
    g = torch.Generator().manual_seed(seed)
    cands = set()
    while len(cands) < 30:
        n = int(torch.randint(1, 5, (1,), generator=g))
        cands.add(tuple([0] + [int(x) for x in torch.randint(2, 9, (n,), generator=g)] + [1]))
    cands = [list(c) for c in sorted(cands)]
    B, K, V = 3, 4, 16
    T = max(len(c) for c in cands)
    logits = [torch.randn(B * K, V, generator=g) * 3 for _ in range(T)]
    outs = []
    for early in (True, False):
        it = iter([l.clone() for l in logits])
        outs.append(O.beam_search(lambda tok: next(it), lambda idx: None, B, K, T, O.prefix_allowed_tokens_fn(O.Trie(cands)),
                                  1.0, early_exit=early))
    assert outs[0][0].tolist() == outs[1][0].tolist()
    assert torch.equal(outs[0][1], outs[1][1])


def test_greedy_search_hand_worked():
    """HF greedy_search (num_beams = 1): argmax over the allowed tokens of the RAW logits, finished rows
    emit pad, the loop stops when every row has produced EOS; tokens outside the Trie never win."""
    cands = [[0, 2, 1], [0, 3, 4, 1]]

    def rows(*dists, V=8):
        out = []
        for d in dists:
            r = torch.full((V,), -5.0)
            for k, v in d.items():
                r[k] = v
            out.append(r)
        return torch.stack(out)

    steps = iter([rows({2: 1.0, 3: 2.0, 5: 9.0}, {2: 3.0, 3: 2.0}),   # token 5 has the largest logit but is not allowed
                  rows({4: 0.5, 1: 0.1}, {1: 0.3}),                  # row 0: only 4 is allowed after [0,3]; row 1: EOS
                  rows({1: 1.0}, {6: 5.0})])                         # row 1 is finished: emits pad whatever its logits
    seq = O.greedy_search(lambda tok: next(steps), 2, 4, O.prefix_allowed_tokens_fn(O.Trie(cands)))
    assert seq.tolist() == [[0, 3, 4, 1], [0, 2, 1, 0]]
    # early stop: both rows finish at length 3 although max_length is 4
    steps = iter([rows({2: 1.0}, {2: 1.0}), rows({1: 0.0}, {1: 0.0}), rows({1: 0.0}, {1: 0.0})])
    seq = O.greedy_search(lambda tok: next(steps), 2, 4, O.prefix_allowed_tokens_fn(O.Trie([[0, 2, 1], [0, 3, 4, 1]])))
    assert seq.tolist() == [[0, 2, 1], [0, 2, 1]]
