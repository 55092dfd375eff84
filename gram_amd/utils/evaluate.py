"""Ranking metrics of the eval path; mirror of the reference's src/utils/evaluate.py (same function
names, arguments and return conventions: *sums* over users, the caller divides).

``hit_ranks``/``metrics_from_ranks`` are the compact form used by the data-parallel runner: the
position of the gold item in a user's score-sorted top-K (or -1) is all that the metrics depend on,
so ranks -- not strings -- are what ranks exchange (one RCCL all-gather).
"""
from __future__ import annotations

import math
from typing import List, Sequence

import numpy as np


def rel_results(predictions: Sequence, targets: Sequence, scores: Sequence[float], k: int) -> List[List[int]]:
    """evaluate.py:5-22.  Per user: order the k (prediction, score) pairs by score, best first
    (stable), and flag the entries equal to the gold."""
    results = []
    for b in range(len(targets)):
        pairs = list(zip(predictions[b * k:(b + 1) * k], scores[b * k:(b + 1) * k]))
        pairs.sort(key=lambda ps: ps[1], reverse=True)
        gold = targets[b]
        results.append([1 if pred == gold else 0 for pred, _ in pairs])
    return results


def get_metrics_results(rel_results: Sequence[Sequence[int]], metrics: Sequence[str]) -> np.ndarray:
    """evaluate.py:25-35."""
    res = []
    for m in metrics:
        name = m.lower()
        if name.startswith("hit"):
            res.append(hit_at_k(rel_results, int(m.split("@")[1])))
        elif name.startswith("ndcg"):
            res.append(ndcg_at_k(rel_results, int(m.split("@")[1])))
    return np.array(res)


def ndcg_at_k(relevance: Sequence[Sequence[int]], k: int) -> float:
    """evaluate.py:38-49.  Leave-one-out: one gold item per user, so IDCG = 1."""
    ndcg = 0.0
    for row in relevance:
        one = 0.0
        for i, r in enumerate(row[:k]):
            one += r / math.log(i + 2, 2)
        ndcg += one
    return ndcg


def hit_at_k(relevance: Sequence[Sequence[int]], k: int) -> float:
    """evaluate.py:52-58."""
    return float(sum(1 for row in relevance if sum(row[:k]) > 0))


def hit_ranks(rel_rows: Sequence[Sequence[int]]) -> np.ndarray:
    """First position of a 1 in each rel row, -1 if none (int16)."""
    out = np.full(len(rel_rows), -1, dtype=np.int16)
    for i, row in enumerate(rel_rows):
        for j, r in enumerate(row):
            if r:
                out[i] = j
                break
    return out


def rel_rows_from_ranks(ranks: Sequence[int], k: int) -> List[List[int]]:
    """Inverse of :func:`hit_ranks` for candidate sets without duplicates (a Trie emits each item
    at most once, so a gold item matches at most one of the k predictions)."""
    rows = []
    for r in ranks:
        row = [0] * k
        if r >= 0:
            row[int(r)] = 1
        rows.append(row)
    return rows


def metric_table(metrics: Sequence[str], k: int) -> np.ndarray:
    """float64 [k + 1][len(metrics)]: row r + 1 holds a user's metric values when the gold item sits at position r of its
    score-sorted top-k (row 0: absent) -- the same floats evaluate.py:38-58 produces for that rel row."""
    tab = np.zeros((k + 1, len(metrics)), dtype=np.float64)
    for r in range(-1, k):
        tab[r + 1] = get_metrics_results(rel_rows_from_ranks([r], k), metrics)
    return tab


def metrics_from_ranks(ranks: Sequence[int], metrics: Sequence[str], k: int) -> np.ndarray:
    """Metric SUMS over users from their hit ranks.  Same floats as ``get_metrics_results(rel_rows_from_ranks(ranks, k), metrics)``:
    per user the table row, added up in user order (np.cumsum is a sequential sum, like the reference's ``+=`` loop)."""
    ranks = np.asarray(ranks, dtype=np.int64)
    if ranks.size == 0:
        return np.zeros(len(metrics), dtype=np.float64)
    return np.cumsum(metric_table(metrics, k)[ranks + 1], axis=0)[-1]


def hit_ranks_from_ids(pred_ids: np.ndarray, scores: np.ndarray, gold_ids: np.ndarray) -> np.ndarray:
    """Vectorised ``hit_ranks(rel_results(...))`` on integer identities: pred_ids / scores (B, k), gold_ids (B,) where two
    predictions (or a prediction and the gold) carry the same id exactly when the reference's decoded strings are equal, and a
    gold that equals no candidate carries an id no prediction has.  Stable descending order by score, as evaluate.py:14-15 sorts."""
    order = np.argsort(-scores.astype(np.float64), axis=1, kind="stable")
    rel = np.take_along_axis(pred_ids, order, axis=1) == gold_ids[:, None]
    first = rel.argmax(axis=1)
    return np.where(rel.any(axis=1), first, -1).astype(np.int16)
