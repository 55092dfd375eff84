#!/usr/bin/env python3
"""Population-scale precision measurement of the HIP path against the fp32 reference arithmetic.

TEST INFRASTRUCTURE (imports oracle/).  The north star asks for Recall@5 / NDCG@5 within 1e-4 of the reference,
which computes in fp32 end to end (SURVEY.md "Facts").  The CPU oracle cannot score thousands of T5-base users
(0.1 users/s), so this script runs THE SAME oracle code (oracle/gram_oracle.py, plain torch fp32 ops) with its
tensors on the GPU -- rocBLAS fp32 GEMMs, fp32 softmax, the pinned HF-4.26 beam search on the host -- as the
on-GPU fp32 reference, and compares ``gram_amd.GRAM.generate`` in each precision mode with it on identical weights
and inputs:

  * hit@5 / ndcg@5 / hit@10 / ndcg@10 of both sides with the gold item of user u placed at the reference's rank
    u mod 10 (so every rank flip inside the top 10 moves a metric), and their absolute differences;
  * rank flips: users whose gold item sits at a different rank; membership changes of the top-K;
  * adjacent-pair order swaps as a function of the reference's score gap between the two items;
  * max |score difference| over sequences both sides returned.

    python tests/precision_population.py --users 4096 --chunk 256 --modes f16x3,f16 --out gpurun_out/precision.json
    python tests/precision_population.py --users 2048 --sweep f16x3      # every stage at one piece, the rest at two

``--outliers F`` gives the model the residual stream of a TRAINED T5 (none exists offline): a few feature dimensions carry values F times
the ordinary ones from the embedding on and are fed by every sublayer's output projection, while the layer norms' gains shrink them
back (gain / F on those dimensions, and all gains x the factor by which the outliers inflate a row's rms) -- so the stream leaves the
IEEE-half range for large F, its rms is dominated by the outliers, and the information sits in dimensions far below the rms: the case
the per-row power-of-two factors of the 16-bit copy exist for.

``--train-steps S`` runs S Adam steps on the weights first (plain torch fp32 autograd through the oracle's own functions, on the GPU): a
synthetic retrieval task -- the gold item of a user is one of 256 items, announced by that item's marker token scattered over the user's
passages; the decoder is taught (teacher forcing, cross-entropy) to spell the item's id pieces -- so that the comparison runs on weights
that gradient descent has shaped (attention that looks for particular keys, logits that commit to a piece) instead of a random
initialisation or a rescaled one.  No trained GRAM checkpoint exists offline; this is the closest population that can be made here.  The
evaluation users are drawn from the task's distribution (markers included), over the full candidate Trie.

``--sharpen F`` multiplies every attention q projection by F: at T5's random init the attention logits are ~N(0,1)
and every query averages ~140 keys, which washes the encoder out of the scores (all users get nearly the same
beams); F = 4 gives peaky attention (a few keys per query), i.e. scores that depend on the passages as a trained
model's do.  Both populations are reported.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GAP_EDGES = [0.0, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, float("inf")]
METRICS = ["hit@5", "ndcg@5", "hit@10", "ndcg@10"]


def _strip(row):
    return tuple(int(t) for t in row if int(t) not in (0, 1))


OUTLIER_DIMS = (7, 200, 450, 701)


def add_outlier_features(sd, oc, F):
    """In place: see --outliers in the module docstring."""
    dims = [d for d in OUTLIER_DIMS if d < oc.d_model]
    infl = float(np.sqrt(1.0 + len(dims) * F * F / oc.d_model))  # how much the outliers inflate a row's rms
    done = set()
    for k in list(sd):
        v = sd[k]
        if id(v) in done:
            continue
        if k.endswith("shared.weight") or k.endswith("embed_tokens.weight"):
            v[:, dims] *= F
        elif k.endswith(".o.weight") or k.endswith(".wo.weight"):      # [d_model][inner | d_ff]: the rows that write the outlier features
            v[dims, :] *= F
        elif k.endswith("layer_norm.weight"):
            v *= infl
            v[dims] /= F
        elif k == "lm_head.weight":
            v *= oc.d_model ** -0.5  # untied (see build): the scale the tied head gets from gram_t5.py:249-252, so the logits stay O(1)
        else:
            continue
        done.add(id(v))


MARKER_COPIES = 12


def marker_range(vmax, n_cands):
    """(first marker token, number of marked items): the top of the vocabulary below `vmax`, 256 items for a real vocabulary."""
    n = max(1, min(256, n_cands, (vmax - 2) // 4))
    return vmax - n, n


def plant_markers(ids, mask, gold, g, base):
    """In place: MARKER_COPIES copies of token base + gold[b] at random valid positions of user b's passages (not the last position)."""
    B, N, L = ids.shape
    for b in range(B):
        valid = torch.nonzero(mask[b, :, : L - 1].reshape(-1)).flatten()
        if valid.numel() == 0:
            continue
        pick = valid[torch.randint(0, valid.numel(), (MARKER_COPIES,), generator=g)]
        ids[b].view(-1)[(pick // (L - 1)) * L + pick % (L - 1)] = base + int(gold[b])


def train_briefly(sd, oc, cands, steps, N, L, vmax, seed, dev, log, batch=32, lr=5e-4):
    """See --train-steps in the module docstring.  sd: the state dict ON THE DEVICE (aliases aliased); trained in place.
    Returns (mean loss of the first 10 steps, of the last 10)."""
    from oracle import gram_oracle as O
    uniq = {}
    for k, v in sd.items():
        uniq.setdefault(id(v), v)
    params = [v.requires_grad_(True) for v in uniq.values()]
    opt = torch.optim.Adam(params, lr=lr)
    g = torch.Generator().manual_seed(seed + 7)
    base, n_items = marker_range(vmax, len(cands))
    items = cands[:n_items]
    T = max(len(c) for c in items)
    losses = []
    for step in range(steps):
        gold = torch.randint(0, len(items), (batch,), generator=g)
        ids = torch.randint(2, base, (batch, N, L), generator=g)
        ids[:, :, -1] = 1
        mask = torch.ones(batch, N, L, dtype=torch.bool)
        plant_markers(ids, mask, gold, g, base)
        tgt = torch.full((batch, T), -100, dtype=torch.long)
        dec_in = torch.zeros(batch, T - 1, dtype=torch.long)
        for b, gi in enumerate(gold.tolist()):
            seq = items[gi]
            tgt[b, : len(seq) - 1] = torch.tensor(seq[1:])
            dec_in[b, : len(seq) - 1] = torch.tensor(seq[:-1])
        ids_d, mask_d, tgt, dec_in = ids.to(dev), mask.to(dev), tgt.to(dev), dec_in.to(dev)
        enc = O.encode_fused(sd, oc, ids_d, mask_d)
        ext = (1.0 - mask_d.reshape(batch, 1, 1, N * L).to(torch.float32)) * O.FMIN
        st = O.DecodeState(cross=O.cross_kv(sd, oc, enc), enc_mask_ext=ext, rows_per_bank=1)
        loss = 0.0
        n_tok = int((tgt[:, : T - 1] >= 0).sum())
        for t in range(T - 1):
            logits = O.decoder_step(sd, oc, dec_in[:, t], st)
            loss = loss + torch.nn.functional.cross_entropy(logits, tgt[:, t], ignore_index=-100, reduction="sum")
        loss = loss / n_tok
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        lr_t = lr * min(1.0, (step + 1) / 20)
        for grp in opt.param_groups:
            grp["lr"] = lr_t
        opt.step()
        losses.append(float(loss.detach()))
        if step % 50 == 0 or step == steps - 1:
            log(f"[precision] training step {step + 1}/{steps}: loss {losses[-1]:.3f}")
    for v in params:
        v.requires_grad_(False)
    k = min(10, len(losses))
    return sum(losses[:k]) / k, sum(losses[-k:]) / k


def build(backbone, seed, sharpen, dev, outliers=0.0, train=None):
    import gram_amd
    from oracle import gram_oracle as O

    if backbone == "tiny":
        oc = O.OracleConfig(vocab_size=256, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_decoder_layers=2, num_heads=2,
                            max_item_num=5)
    else:
        oc = O.OracleConfig.named(backbone)
    if outliers:  # an untied lm_head: the logits keep their ordinary scale (a tied one would read the outlier columns of the table)
        import dataclasses
        oc = dataclasses.replace(oc, tie_word_embeddings=False)
    gc = gram_amd.T5Config(vocab_size=oc.vocab_size, d_model=oc.d_model, d_ff=oc.d_ff, num_layers=oc.num_layers,
                           num_decoder_layers=oc.num_decoder_layers, num_heads=oc.num_heads, max_item_num=oc.max_item_num,
                           tie_word_embeddings=oc.tie_word_embeddings)
    sd = O.init_state_dict(oc, seed)
    if sharpen != 1.0:
        for k in sd:
            if k.endswith(".q.weight"):
                sd[k] = sd[k] * sharpen
    if outliers:
        sd = {k: v for k, v in sd.items()}
        uniq = {}
        for k, v in sd.items():  # clone once per distinct tensor: the tied tables stay tied
            uniq.setdefault(id(v), v.clone())
            sd[k] = uniq[id(v)]
        add_outlier_features(sd, oc, outliers)
    sd_dev = {}
    seen = {}
    for k, v in sd.items():  # keep the aliases aliased on the device
        if id(v) not in seen:
            seen[id(v)] = v.to(dev)
        sd_dev[k] = seen[id(v)]
    trained = None
    if train:  # (steps, cands, N, L, vmax, log): a few hundred optimiser steps on the device copy, then the model gets THOSE weights
        steps, cands, N, L, vmax, log = train
        trained = train_briefly(sd_dev, oc, cands, steps, N, L, vmax, seed, dev, log)
        back = {}
        sd = {k: back.setdefault(id(v), v.detach().cpu()) for k, v in sd_dev.items()}
    model = gram_amd.create_model("gram", gc)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    return oc, sd_dev, model, trained


def compare(ref_seqs, ref_scores, out_seqs, out_scores, B, K, user0, acc):
    """Accumulate the statistics of one chunk into ``acc`` (see the module docstring)."""
    from gram_amd.utils import evaluate as ev

    opred = [_strip(s) for s in ref_seqs.tolist()]
    dpred = [_strip(s) for s in out_seqs.tolist()]
    osc, dsc = ref_scores.tolist(), out_scores.tolist()
    gold = [opred[b * K + (user0 + b) % 10 % K] for b in range(B)]
    orel = ev.rel_results(opred, gold, osc, K)
    drel = ev.rel_results(dpred, gold, dsc, K)
    acc["flips"] += int((ev.hit_ranks(orel) != ev.hit_ranks(drel)).sum())
    acc["o_sum"] += ev.get_metrics_results(orel, METRICS)
    acc["d_sum"] += ev.get_metrics_results(drel, METRICS)
    acc["users"] += B
    for b in range(B):
        o_items, d_items = opred[b * K:(b + 1) * K], dpred[b * K:(b + 1) * K]
        o_s, d_s = osc[b * K:(b + 1) * K], dsc[b * K:(b + 1) * K]
        dpos = {it: i for i, it in enumerate(d_items)}
        acc["membership_changes"] += sum(1 for it in o_items if it not in dpos)
        acc["top1_same"] += int(o_items[0] == d_items[0])
        acc["order_same"] += int(o_items == d_items)
        for i, it in enumerate(o_items):
            if it in dpos:
                acc["max_score_dev"] = max(acc["max_score_dev"], abs(o_s[i] - d_s[dpos[it]]))
                acc["sum_score_dev"] += abs(o_s[i] - d_s[dpos[it]])
                acc["n_score_dev"] += 1
        for r in range(min(K - 1, 10)):  # adjacent pairs of the reference's top 11
            a, c = o_items[r], o_items[r + 1]
            gap = o_s[r] - o_s[r + 1]
            bucket = next(i for i in range(len(GAP_EDGES) - 1) if GAP_EDGES[i] <= gap < GAP_EDGES[i + 1])
            acc["pairs"][bucket] += 1
            if a in dpos and c in dpos:
                acc["swaps"][bucket] += int(dpos[a] > dpos[c])
            else:
                acc["swaps"][bucket] += 1


def new_acc():
    return dict(flips=0, o_sum=np.zeros(len(METRICS)), d_sum=np.zeros(len(METRICS)), users=0, membership_changes=0, top1_same=0,
                order_same=0, max_score_dev=0.0, sum_score_dev=0.0, n_score_dev=0, pairs=[0] * (len(GAP_EDGES) - 1),
                swaps=[0] * (len(GAP_EDGES) - 1))


def summarise(acc):
    n = max(acc["users"], 1)
    delta = np.abs(acc["o_sum"] - acc["d_sum"]) / n
    return {
        "users": acc["users"],
        "rank_flips": acc["flips"],
        "metrics_reference": dict(zip(METRICS, (acc["o_sum"] / n).round(6).tolist())),
        "metrics_device": dict(zip(METRICS, (acc["d_sum"] / n).round(6).tolist())),
        "abs_delta": dict(zip(METRICS, [float(f"{x:.3e}") for x in delta])),
        "top1_same": acc["top1_same"], "topK_order_same": acc["order_same"], "membership_changes": acc["membership_changes"],
        "max_abs_score_dev": acc["max_score_dev"], "mean_abs_score_dev": acc["sum_score_dev"] / max(acc["n_score_dev"], 1),
        "adjacent_pair_swaps_by_reference_gap": [
            {"gap": f"[{GAP_EDGES[i]:g}, {GAP_EDGES[i + 1]:g})", "pairs": acc["pairs"][i], "swapped": acc["swaps"][i]}
            for i in range(len(GAP_EDGES) - 1)],
    }


def parse_mode(spec):
    """'bf16x6' or 'bf16x6/enc_attn=1+bank_v=2' (stage caps of a sensitivity sweep, GRAM.set_stage_pieces)."""
    mode, _, caps = spec.partition("/")
    return mode, {k: int(v) for k, v in (kv.split("=") for kv in caps.split("+") if kv)}


def sweep_modes(base, pieces):
    """Every stage at 1 .. pieces-1 pieces with the rest at `base` (plus `base` itself)."""
    from gram_amd import _lib
    return [base] + [f"{base}/{st}={n}" for st in _lib.STAGES for n in range(1, pieces)]


def run(users=4096, chunk=256, backbone="t5-base", dataset="Beauty", modes=("f16x3",), seed=2023, sharpen=1.0, N=3, L=128, K=20,
        dev="cuda:0", log=print, n_items=0, ragged=False, outliers=0.0, train_steps=0):
    from gram_amd.utils import generation_trie as gt
    from oracle import gram_oracle as O

    if backbone == "tiny":
        g0 = torch.Generator().manual_seed(5)
        cands = sorted({tuple([0] + torch.randint(2, 60, (3,), generator=g0).tolist() + [1]) for _ in range(n_items or 400)})
        cands = [list(c) for c in cands]
        vmax = 256
    else:
        z = np.load(os.path.join(ROOT, "tests", "golden", "tries.npz"))
        cands = [[int(x) for x in row if x >= 0] for row in z[f"{dataset}_cands"]]
        vmax = 32100
    oc, sd, model, trained = build(backbone, seed, sharpen, dev, outliers, (train_steps, cands, N, L, vmax, log) if train_steps else None)
    max_length = max(len(c) for c in cands)
    dfn = gt.prefix_allowed_tokens_fn(gt.Trie(cands))
    ofn = O.prefix_allowed_tokens_fn(O.Trie(cands))
    accs = {m: new_acc() for m in modes}
    t_ref = t_dev = 0.0
    g = torch.Generator().manual_seed(seed + 1)
    # pass 1: the fp32 reference of every chunk (kept on the host: K sequences + scores per user); pass 2: one mode at a time over all
    # the chunks, so that the weights are re-packed once per mode (a sweep runs dozens of modes)
    chunks = []
    for u0 in range(0, users, chunk):
        B = min(chunk, users - u0)
        ids = torch.randint(2, vmax, (B, N, L), generator=g)
        ids[:, :, -1] = 1
        mask = torch.ones(B, N, L, dtype=torch.bool)
        if ragged:  # Collator-shaped: valid lengths U[L/4, L] with EOS at the valid end, one passage in eight fully padded (never passage 0)
            lens = torch.randint(L // 4, L + 1, (B, N), generator=g)
            lens[:, 1:][torch.rand(B, N - 1, generator=g) < 0.125] = 0
            mask = torch.arange(L)[None, None, :] < lens[:, :, None]
            ids[torch.arange(B)[:, None], torch.arange(N)[None, :], (lens - 1).clamp(min=0)] = 1
            ids[~mask] = 0
        if train_steps:  # users of the task the weights were trained on
            base, n_marked = marker_range(vmax, len(cands))
            ids[(ids >= base) & mask] = 2
            plant_markers(ids, mask, torch.randint(0, n_marked, (B,), generator=g), g, base)
        t0 = time.perf_counter()
        ref = O.generate(sd, oc, ids.to(dev), mask.to(dev), max_length, ofn, K, K, 1.0)
        torch.cuda.synchronize()
        t_ref += time.perf_counter() - t0
        chunks.append((u0, B, ids, mask, ref["sequences"].cpu(), ref["sequences_scores"].cpu()))
        log(f"[precision] reference users {u0 + B}/{users}  {t_ref:.0f}s")
    del sd
    torch.cuda.empty_cache()
    for m in modes:
        mode, caps = parse_mode(m)
        model.set_precision(mode)
        model.set_stage_pieces(caps)
        t0 = time.perf_counter()
        for u0, B, ids, mask, rseq, rsc in chunks:
            out = model.generate(input_ids=ids.to(dev), attention_mask=mask.to(dev), max_length=max_length, prefix_allowed_tokens_fn=dfn,
                                 num_beams=K, num_return_sequences=K, length_penalty=1.0)
            compare(rseq, rsc, out["sequences"].cpu(), out["sequences_scores"].cpu(), B, K, u0, accs[m])
        torch.cuda.synchronize()
        t_dev += time.perf_counter() - t0
        a = accs[m]
        log(f"[precision] {m}: flips {a['flips']} membership {a['membership_changes']} max|ds| {a['max_score_dev']:.2e} "
            f"|dhit@5| {abs(a['o_sum'][0] - a['d_sum'][0]) / max(a['users'], 1):.2e} |dndcg@5| {abs(a['o_sum'][1] - a['d_sum'][1]) / max(a['users'], 1):.2e}"
            f"  (dev {t_dev:.0f}s)")
    model.set_stage_pieces(None)
    return {
        "population": {"backbone": backbone, "dataset": dataset, "items": len(cands), "users": users, "N": N, "L": L, "K": K,
                       "seed": seed, "q_sharpen": sharpen, "ragged_masks": bool(ragged), "outlier_features_x": outliers,
                       "trained": ({"adam_steps": train_steps, "loss_first10": round(trained[0], 4), "loss_last10": round(trained[1], 4),
                                    "task": f"{marker_range(vmax, len(cands))[1]} items announced by marker tokens, teacher-forced cross-entropy on the id pieces"}
                                   if trained else None), "gold_rank": "reference rank (user index mod 10)",
                       "reference": "oracle/gram_oracle.py run with torch fp32 tensors on the GPU (rocBLAS fp32), HF-4.26 search on the host"},
        "modes": {m: summarise(a) for m, a in accs.items()},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=4096)
    ap.add_argument("--chunk", type=int, default=256)
    ap.add_argument("--backbone", default="t5-base")
    ap.add_argument("--dataset", default="Beauty")
    ap.add_argument("--modes", default="f16x3")
    ap.add_argument("--ragged", action="store_true", help="ragged masks (valid lengths U[L/4, L], some passages fully padded)")
    ap.add_argument("--seed", type=int, default=2023)
    ap.add_argument("--sharpen", type=float, default=1.0)
    ap.add_argument("--outliers", type=float, default=0.0, help="outlier features F times the ordinary ones (a trained T5's residual stream)")
    ap.add_argument("--train-steps", type=int, default=0, help="Adam steps on a synthetic retrieval task before the comparison (a TRAINED population)")
    ap.add_argument("--beams", type=int, default=20)
    ap.add_argument("--out", default="")
    ap.add_argument("--sweep", default="", help="BASE mode: every stage at fewer pieces with the rest at BASE (per-stage sensitivity)")
    a = ap.parse_args()
    modes = tuple(a.modes.split(","))
    if a.sweep:
        import gram_amd
        modes = tuple(sweep_modes(a.sweep, gram_amd.GRAM._PIECES[a.sweep]))
    res = run(a.users, a.chunk, a.backbone, a.dataset, modes, a.seed, a.sharpen, K=a.beams, ragged=a.ragged, outliers=a.outliers, train_steps=a.train_steps)
    txt = json.dumps(res, indent=1)
    print(txt)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
