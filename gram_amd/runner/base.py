"""Shared eval logic of the GRAM runners (the scoring half of src/runner/*_runner_gram.py).

The training half of the reference runners (optimizer, schedulers, train_generator) is out of
scope of this build; the constructor keeps the reference's argument list so that
``get_runner(...)`` call sites (main_generative_gram.py:107-118,191-201) work unchanged.
"""
from __future__ import annotations

import logging
import os
from time import time
from typing import List, Optional, Sequence

import numpy as np
import torch

from ..utils import evaluate
from ..utils import generation_trie as gt

SPLIT_TOKENS = (1820, 9175)  # single_runner_gram.py:606-609 ("split token hard coded")


def _arg(args, name, default):
    return getattr(args, name, default) if args is not None else default


class _PredRows:
    """Per-user (gold string, K predicted strings in the order generate returned them, K scores) rows, materialised on demand:
    an evaluation keeps item indices and scores, the strings come from the candidate list's one decode."""

    def __init__(self, gold, items, scores, extra, cand_str, K):
        self.gold, self.items, self.scores, self.extra, self.cand_str, self.K = gold, items, scores, extra, cand_str, K

    def __len__(self):
        return len(self.gold)

    def __iter__(self):
        u = 0
        for items, scores, extra in zip(self.items, self.scores, self.extra):
            for b in range(items.shape[0]):
                gen = [self.cand_str[i] if i >= 0 else extra[b * self.K + k] for k, i in enumerate(items[b].tolist())]
                yield self.gold[u], gen, scores[b]
                u += 1

    def head(self, n):
        out = []
        for row in self:
            if len(out) >= n:
                break
            out.append(row)
        return out


class BaseRunner:
    def __init__(self, model_rec, model_gen, tokenizer, train_loader_id, train_loader_rec, valid_loader, device, args):
        self.model = model_rec
        self.model_rec = model_rec
        self.model_gen = model_gen  # unused by GRAM (single_runner_gram.py:44)
        self.tokenizer = tokenizer
        self.device = device
        self.args = args
        self.metrics = _arg(args, "metrics", "hit@5,hit@10,ndcg@5,ndcg@10").split(",")
        # single_runner_gram.py:38-39: the beam is max(largest k in the metrics, --beam_size)
        self.generate_num = max(max(int(m.split("@")[1]) for m in self.metrics), int(_arg(args, "beam_size", 1)))
        self.length_penalty = float(_arg(args, "length_penalty", 1.0))
        self.train_loader_id = train_loader_id
        self.train_loader_rec = train_loader_rec
        self.testloaders: List = []
        self.validloaders: List = [valid_loader] if valid_loader is not None else []
        self.cur_model_path = None
        self.last_results = None
        # single_runner_gram.py:46,51 / distributed_runner_gram.py:57-66: the runner builds its own test and validation
        # loaders at construction.  An args namespace without --datasets (unit tests driving test_dataset_task with
        # their own loader) skips that; test()/validate() then raise instead of silently scoring nothing.
        if _arg(args, "datasets", None) and _arg(args, "data_path", None):
            self.get_testloader(regenerate=False, phase=0)
            self.get_validloader(regenerate=False, phase=0)

    # ---------------------------------------------------------------- loaders
    def _make_dataset(self, dataset, task, model_gen, tokenizer, regenerate, phase, debug_test_small_set, mode):
        from ..data import TestDatasetGRAM
        return TestDatasetGRAM(self.args, dataset, task, model_gen, tokenizer, regenerate, phase,
                               debug_test_small_set=debug_test_small_set, mode=mode)

    def _make_loader(self, data, collator):
        """single_runner_gram.py:320-325: batch_size = --eval_batch_size, no shuffling, num_workers 0."""
        from torch.utils.data import DataLoader
        return DataLoader(dataset=data, batch_size=int(_arg(self.args, "eval_batch_size", 1)), collate_fn=collator, shuffle=False)

    def _build_loaders(self, mode, model_gen, tokenizer, regenerate, phase, debug_test_small_set):
        from ..processor import CollatorGRAM
        if not _arg(self.args, "datasets", None):
            raise ValueError("args.datasets is empty: there is nothing to build an evaluation loader from")
        collator = CollatorGRAM(self.tokenizer, args=self.args, mode="test" if mode == "test" else "valid")
        loaders = []
        for dataset in self.args.datasets.split(","):
            for task in _arg(self.args, "tasks", "sequential").split(","):
                data = self._make_dataset(dataset, task, model_gen, tokenizer, regenerate, phase, debug_test_small_set, mode)
                loaders.append(self._make_loader(data, collator))
        return loaders

    def get_testloader(self, model_gen=None, tokenizer=None, regenerate=False, phase=0, debug_test_small_set=False):
        """single_runner_gram.py:296-326 (distributed: :300-359): one loader per (dataset, task), test hold-out."""
        self.testloaders = self._build_loaders("test", model_gen, tokenizer, regenerate, phase, debug_test_small_set)

    def get_validloader(self, model_gen=None, tokenizer=None, regenerate=False, phase=0, debug_test_small_set=False):
        """single_runner_gram.py:328-358: the same with mode="validation" (second-to-last item held out)."""
        self.validloaders = self._build_loaders("validation", model_gen, tokenizer, regenerate, phase, debug_test_small_set)

    def _loaders_for(self, mode):
        """test()/validate() rebuild their loaders like the reference (single_runner_gram.py:370-375,408-413) when the
        args describe a dataset; otherwise they use what the caller put in testloaders/validloaders -- and an empty list
        is an error, never a silent no-op."""
        if _arg(self.args, "datasets", None) and _arg(self.args, "data_path", None):
            small = bool(_arg(self.args, "debug_test_small_set", False))
            (self.get_testloader if mode == "test" else self.get_validloader)(regenerate=False, phase=0, debug_test_small_set=small)
        loaders = self.testloaders if mode == "test" else self.validloaders
        if not loaders:
            raise RuntimeError(f"no {mode} loader: pass args.datasets/args.data_path (the reference's flags) or fill "
                               f"runner.{'testloaders' if mode == 'test' else 'validloaders'} before calling")
        return loaders

    # ---------------------------------------------------------------- candidates -> Trie
    def encode_candidates(self, candidates: Sequence) -> List[List[int]]:
        """single_runner_gram.py:594-616.  Candidates that are already token-id sequences (the
        offline fixtures: no tokenizer in this container) pass through."""
        if len(candidates) and not isinstance(candidates[0], str):
            return [list(map(int, c)) for c in candidates]
        id_type = _arg(self.args, "item_id_type", "split")
        out = []
        for cand in candidates:
            if id_type == "t5_token":
                out.append([0] + self.tokenizer.convert_tokens_to_ids(cand.split(" ")) + [1])
            elif id_type == "split":
                out.append([0] + [t for t in self.tokenizer.encode(cand) if t not in SPLIT_TOKENS])
            else:
                out.append([0] + self.tokenizer.encode(f"{cand}"))
        return out

    def _load(self, path: Optional[str], strict: bool = True) -> None:
        if path:
            if os.path.isdir(path):
                path = os.path.join(path, os.listdir(path)[0])
            self.model.load_state_dict(torch.load(path, map_location=self.device, weights_only=True), strict=strict)

    def _generate_model(self):
        return self.model_rec.module if hasattr(self.model_rec, "module") else self.model_rec

    def _warm_passage_cache(self, testloader, chunk: int = 2048) -> int:
        """The passage cache of an evaluation: every user's passages 1..h are `item2input[item]` texts
        (test_dataset_gram.py:115-123,203-210), the same for all users, so their encoder states are computed once per eval
        instead of once per occurrence.  `--passage_cache 1` (default): the model HARVESTS them -- an item prompt joins the cache
        when the first batch that contains it has been scored (GRAM.set_passage_harvest), so nothing runs before the first
        generate() and items no user has in its history are never encoded.  `--passage_cache 2`: all of `dataset.item2input` is
        encoded up front (needs a collate_fn with `encode_passages`, gram_amd.processor.CollatorGRAM).  `--passage_cache 0`: off.
        Results do not change (bit-identical, tests/test_gpu_configs.py)."""
        mode = int(_arg(self.args, "passage_cache", 1))
        model = self._generate_model()
        if hasattr(model, "set_passage_harvest"):
            model.set_passage_harvest(mode == 1)
        item2input = getattr(testloader.dataset, "item2input", None)
        if not mode or not item2input or not hasattr(model, "cache_passages"):
            return 0
        if hasattr(model, "reserve_passage_cache"):
            model.reserve_passage_cache(len(set(item2input.values())))
        encode = getattr(getattr(testloader, "collate_fn", None), "encode_passages", None)
        if mode != 2 or encode is None:
            return 0
        texts = sorted(set(item2input.values()))
        start, n = time(), 0
        for lo in range(0, len(texts), chunk):
            ids, mask = encode(texts[lo:lo + chunk])
            n = model.cache_passages(ids, mask)
        logging.info(f"passage cache: {n} item prompts encoded in {time() - start:.2f}s")
        return n

    # ---------------------------------------------------------------- GPU batches, whatever --eval_batch_size is
    def _gpu_batch_users(self, model, n_slots: int, L: int, K: int, max_length: int) -> int:
        """Users per ``generate`` call.  The reference scores `--eval_batch_size` users per call (default 1,
        arguments.py:84-86): 95 users/s on this path against 3 100 at 4 096 per call.  A user's result does not depend on the batch
        it is scored in (bit for bit: tests/test_gpu_configs.py), so the runner re-batches: `--eval_gpu_batch` / GRAM_EVAL_USERS
        users per call, default the largest count <= 4 096 whose workspace fits the free HBM."""
        explicit = int(_arg(self.args, "eval_gpu_batch", 0) or os.environ.get("GRAM_EVAL_USERS", 0) or 0)
        if explicit > 0:
            return explicit
        if hasattr(model, "max_users_per_call"):
            return int(model.max_users_per_call(n_slots, L, K, max_length, limit=4096))
        return 4096

    def _rescore_unfinished(self, model, input_ids, attention_mask, seqs, scores, fn, K, ref_max_length, shape):
        """Users of a batch decoded to the Trie's depth who have fewer than K finished hypotheses (a -inf score among their K rows:
        only BeamSearchScorer.finalize's filler beams carry one), scored again with the reference's own max_length
        (single_runner_gram.py:637: 50 for the "term" id type) -- HF keeps decoding such a user's -inf beams until max_length and
        finalizes THOSE.  Everyone else's rows are what HF returns at either length (see _score_loader).  Which tokens a -inf
        beam carries is torch.topk's choice among equal keys in the reference and "lowest flat index first" here; the finite rows, their
        order and their scores do not depend on it."""
        B = input_ids.shape[0]
        unfinished = (~torch.isfinite(scores.view(B, K))).any(dim=1).nonzero().flatten()
        if unfinished.numel() == 0:
            return seqs, scores
        per_call = int(model.max_users_per_call(shape[0], shape[1], K, ref_max_length, limit=4096)) if hasattr(model, "max_users_per_call") \
            else int(unfinished.numel())
        width = seqs.shape[1]
        redo = []
        for lo in range(0, int(unfinished.numel()), max(per_call, 1)):
            u = unfinished[lo:lo + per_call]
            redo.append(model.generate(input_ids=input_ids[u], attention_mask=attention_mask[u], max_length=ref_max_length,
                                       prefix_allowed_tokens_fn=fn, num_beams=K, num_return_sequences=K, output_scores=True,
                                       return_dict_in_generate=True, length_penalty=self.length_penalty))
            width = max(width, redo[-1]["sequences"].shape[1])
        out_seqs = seqs.new_zeros(B * K, width)  # (pad id 0, as BeamSearchScorer.finalize pads)
        out_seqs[:, : seqs.shape[1]] = seqs
        out_scores = scores.clone()
        rows_of = lambda u: (u[:, None] * K + torch.arange(K, device=u.device)[None, :]).flatten()
        lo = 0
        for r in redo:
            n = r["sequences"].shape[0] // K
            rows = rows_of(unfinished[lo:lo + n].to(out_seqs.device))
            out_seqs[rows] = 0
            out_seqs[rows, : r["sequences"].shape[1]] = r["sequences"].to(out_seqs.device)
            out_scores[rows.to(out_scores.device)] = r["sequences_scores"].to(out_scores.device)
            lo += n
        return out_seqs, out_scores

    @staticmethod
    def _merge_batches(parts):
        """Collated batches -> one: passages padded like the Collator pads them (all-zero ids, all-False mask; Collator.py:410-436),
        targets with -100."""
        if len(parts) == 1:
            return parts[0]
        B = sum(p["item_text_ids"].shape[0] for p in parts)
        N = max(p["item_text_ids"].shape[1] for p in parts)
        L = max(p["item_text_ids"].shape[2] for p in parts)
        T = max(p["target_ids"].shape[1] for p in parts)
        ids = torch.zeros(B, N, L, dtype=parts[0]["item_text_ids"].dtype)
        mask = torch.zeros(B, N, L, dtype=torch.bool)
        tgt = torch.full((B, T), -100, dtype=parts[0]["target_ids"].dtype)
        users, lo = [], 0
        for p in parts:
            b, n, l = p["item_text_ids"].shape
            ids[lo:lo + b, :n, :l] = p["item_text_ids"]
            mask[lo:lo + b, :n, :l] = p["item_text_masks"].bool()
            tgt[lo:lo + b, : p["target_ids"].shape[1]] = p["target_ids"]
            users += list(p.get("user_ids", [None] * b))
            lo += b
        return {"item_text_ids": ids, "item_text_masks": mask, "target_ids": tgt, "user_ids": users}

    def _gpu_batches(self, testloader, users_per_call: int):
        """The loader's users, in the loader's order, `users_per_call` at a time.  A loader built by this runner (map-style dataset,
        batch sampler, CollatorGRAM) is read through its batch sampler and collated once per GPU batch; any other loader is iterated
        as it is and its collated batches are merged."""
        bs = getattr(testloader, "batch_sampler", None)
        collate = getattr(testloader, "collate_fn", None)
        data = getattr(testloader, "dataset", None)
        if bs is not None and collate is not None and hasattr(data, "__getitem__") and getattr(testloader, "num_workers", 0) == 0 \
                and not isinstance(data, torch.utils.data.IterableDataset):
            pending = []
            want = max(1, users_per_call // 4)  # a short first batch: the GPU starts while the first full batch is being collated
            for idx in bs:
                pending.extend(idx)
                while len(pending) >= want:
                    take, pending = pending[:want], pending[want:]
                    want = users_per_call
                    yield collate([data[i] for i in take])
            if pending:
                yield collate([data[i] for i in pending])
            return
        parts, n = [], 0
        for batch in testloader:
            parts.append(batch)
            n += batch["item_text_ids"].shape[0]
            if n >= users_per_call:
                yield self._merge_batches(parts)
                parts, n = [], 0
        if parts:
            yield self._merge_batches(parts)

    def _decode_rows(self, rows) -> List:
        """batch_decode(skip_special_tokens=True) when a tokenizer exists, else id tuples with the
        specials (pad 0, eos 1, -100 label padding) dropped -- the same equality relation whenever
        decoding is injective on the candidate set."""
        if self.tokenizer is not None:
            rows = [[t for t in r if t >= 0] for r in rows]
            return self.tokenizer.batch_decode(rows, skip_special_tokens=True)
        return [tuple(t for t in r if t > 1) for r in rows]

    def _decode(self, ids: torch.Tensor) -> List:
        return self._decode_rows(ids.detach().cpu().tolist())

    def _score_loader(self, testloader):
        """Per-user hit ranks for one loader: (ranks int16 [n], total generate() seconds, examples, user ids, pred rows).

        The reference's loop body (single_runner_gram.py:622-694) per batch: collate -> H2D -> generate -> batch_decode of the B*K
        generated rows and of the targets -> rel_results -> metrics.  Here, with the same results:
          * users are scored `_gpu_batch_users` at a time whatever --eval_batch_size is (see there);
          * every generated row is a leaf of the candidate Trie, so the device returns the ITEM INDEX of each row
            (GRAM.sequence_items -> gram_trie_item_index) and the strings come from ONE batch_decode of the candidate list per
            evaluation -- the string of a row depends on its ids alone, so these are the strings the reference decodes per batch;
            rows that are not candidates (HF's -inf filler beams) are decoded individually, like the reference decodes every row;
          * string equality is evaluated on integer ids of the distinct strings (evaluate.hit_ranks_from_ids);
          * collating batch i+1 and post-processing batch i-1 overlap generate() of batch i (two worker threads; the C-ABI call
            releases the GIL)."""
        import queue
        import threading

        phases = {}
        t_phase = time()

        def lap(name):
            nonlocal t_phase
            now = time()
            phases[name] = phases.get(name, 0.0) + now - t_phase
            t_phase = now

        self._warm_passage_cache(testloader)
        lap("passage_cache")
        candidates = testloader.dataset.all_items
        encoded = self.encode_candidates(candidates)
        trie = gt.Trie(encoded)
        fn = gt.prefix_allowed_tokens_fn(trie)
        lap("candidates_and_trie")
        # single_runner_gram.py:629-637, hoisted out of the loop: max_length = the longest candidate for the "t5_token" / "split" id
        # types, 50 for every other one ("term").  The search runs to the Trie's depth D either way: past it no beam is inside the
        # Trie, every candidate of every later step is -inf, so a user whose K hypotheses have all finished by D gets from HF at 50
        # exactly what it gets at D (BeamHypotheses.add refuses a -inf score once the heap is full; is_done turns true on the first
        # all--inf step).  That is every user when each Trie leaf is an EOS: finished hypotheses + beams inside the Trie start at K
        # (HF's K start beams, all finite) and no step lowers the sum below K.  The guard of that argument: a
        # user with FEWER than K finished hypotheses at D is recognisable by a -inf score among its K rows (the filler beams
        # BeamSearchScorer.finalize adds) -- HF would have kept decoding that user's -inf beams up to 50 tokens before finalizing --
        # and is scored again with the reference's max_length (`_rescore_unfinished`).
        longest = max(len(c) for c in encoded)
        ref_max_length = longest
        if isinstance(candidates[0], str) and _arg(self.args, "item_id_type", "split") not in ("t5_token", "split"):
            ref_max_length = 50
        max_length = min(longest, ref_max_length)
        K = self.generate_num
        model = self._generate_model()
        on_device_items = hasattr(model, "sequence_items")
        # one decode of the candidate list; sid = id of a distinct decoded string (first item that decodes to it)
        cand_str = self._decode_rows(encoded)
        str2sid = {}
        for i, s in enumerate(cand_str):
            str2sid.setdefault(s, i)
        cand_sid = np.fromiter((str2sid[s] for s in cand_str), dtype=np.int64, count=len(cand_str))
        other_sid = {}  # strings that are not candidates (fillers, unknown golds): ids below zero

        def sid_of(s):
            r = str2sid.get(s)
            if r is None:
                r = other_sid.setdefault(s, -2 - len(other_sid))
            return r

        collator = getattr(testloader, "collate_fn", None)
        n_slots = int(_arg(self.args, "max_his", 20)) + 1
        L = int(getattr(collator, "item_prompt_max_len", 0) or _arg(self.args, "item_prompt_max_len", 128))
        users_per_call = self._gpu_batch_users(model, n_slots, L, K, max_length)
        lap("candidate_strings")
        logging.info(f"scoring {users_per_call} users per generate() call (--eval_batch_size {_arg(self.args, 'eval_batch_size', 1)} "
                     f"is the loader's; results do not depend on the batch)")

        batches_q: "queue.Queue" = queue.Queue(maxsize=2)
        post_q: "queue.Queue" = queue.Queue(maxsize=4)
        failure = []

        def produce():
            try:
                for batch in self._gpu_batches(testloader, users_per_call):
                    batches_q.put(batch)
            except BaseException as e:  # surfaced by the consumer
                failure.append(e)
            batches_q.put(None)

        out = dict(ranks=[], user_ids=[], gold=[], items=[], scores=[], extra=[])

        def post_one(batch, items, scores, seqs):
            B = scores.shape[0]
            gold = self._decode(batch["target_ids"])
            gold_sid = np.fromiter((sid_of(g) for g in gold), dtype=np.int64, count=B)
            extra = {}
            if items is None:  # a model without sequence_items: decode every generated row, like the reference
                gen = self._decode(seqs)
                pred_sid = np.fromiter((sid_of(s) for s in gen), dtype=np.int64, count=B * K).reshape(B, K)
                extra = dict(enumerate(gen))
                items = np.full((B, K), -1, dtype=np.int32)
            else:
                pred_sid = np.where(items >= 0, cand_sid[np.maximum(items, 0)], -1)
                miss = np.argwhere(items < 0)
                if len(miss):
                    strs = self._decode(seqs[miss[:, 0] * K + miss[:, 1]])
                    for (b, k), s in zip(miss.tolist(), strs):
                        pred_sid[b, k] = sid_of(s)
                        extra[b * K + k] = s
            out["ranks"].append(evaluate.hit_ranks_from_ids(pred_sid, scores, gold_sid))
            out["user_ids"] += list(batch.get("user_ids", [None] * B))
            out["gold"] += gold
            out["items"].append(items)
            out["scores"].append(scores)
            out["extra"].append(extra)

        def post():
            while True:
                job = post_q.get()
                if job is None:
                    return
                if failure:
                    continue  # keep draining so that the scoring loop never blocks on a full queue
                try:
                    post_one(*job)
                except BaseException as e:
                    failure.append(e)

        producer = threading.Thread(target=produce, name="gram-collate", daemon=True)
        poster = threading.Thread(target=post, name="gram-post", daemon=True)
        producer.start()
        poster.start()
        total_time = 0.0
        try:
            with torch.no_grad():
                while True:
                    lap("post_handoff")
                    batch = batches_q.get()
                    lap("wait_for_collate")
                    if batch is None or failure:
                        break
                    input_ids = batch["item_text_ids"].to(self.device)
                    attention_mask = batch["item_text_masks"].to(self.device)
                    start = time()
                    pred = model.generate(
                        input_ids=input_ids, attention_mask=attention_mask, max_length=max_length,
                        prefix_allowed_tokens_fn=fn, num_beams=K, num_return_sequences=K, output_scores=True,
                        return_dict_in_generate=True, length_penalty=self.length_penalty,
                    )
                    B = input_ids.shape[0]
                    seqs = pred["sequences"]
                    seq_scores = pred["sequences_scores"]
                    if ref_max_length > max_length:
                        seqs, seq_scores = self._rescore_unfinished(model, input_ids, attention_mask, seqs, seq_scores, fn, K,
                                                                    ref_max_length, (n_slots, L))
                    lap("h2d_and_generate")
                    total_time += time() - start  # generate() alone, as single_runner_gram.py:640-652 times it (it returns synchronised)
                    items = model.sequence_items(seqs, fn, encoded).cpu().numpy().reshape(B, K) if on_device_items else None
                    scores = seq_scores.detach().cpu().numpy().reshape(B, K)
                    need_seqs = items is None or bool((items < 0).any())
                    post_q.put((batch, items, scores, seqs.detach().cpu() if need_seqs else None))
        finally:
            post_q.put(None)
            poster.join()
            if producer.is_alive():  # a failure on this side: unblock the producer and let it finish
                while producer.is_alive():
                    try:
                        batches_q.get(timeout=0.05)
                    except queue.Empty:
                        pass
        lap("drain_post")
        if failure:
            raise failure[0]
        ranks = np.concatenate(out["ranks"]) if out["ranks"] else np.zeros(0, dtype=np.int16)
        rows_out = _PredRows(out["gold"], out["items"], out["scores"], out["extra"], cand_str, K)
        examples = [f"[GT] {g} || [top-1] {gen[0]}" for g, gen, _ in rows_out.head(10)]
        self.last_host = dict(users_per_call=users_per_call, phases=phases)
        return ranks, total_time, examples, out["user_ids"], rows_out

    # the reference writes this literal header whatever `metrics` is (single_runner_gram.py:588, distributed :720)
    PRED_HEADER = "idx\tH@5\tH@10\tNDCG@5\tNDCG@10\tgold\tpred\tscores\n"

    def _write_preds(self, fname, user_ids, ranks, rows_out, footer=None):
        """Preds TSV, single_runner_gram.py:580-588,675-694,709-710: the literal header, one row per user (id, the
        per-user metric values tab-joined, gold, '||'-joined predictions, '||'-joined scores) and, when `footer` is
        given, the reference's closing `metric: value` lines (plain floats here; the reference prints 0-d tensors)."""
        K = self.generate_num
        os.makedirs(os.path.dirname(os.path.abspath(fname)), exist_ok=True)
        with open(fname, "w") as f:
            f.write(self.PRED_HEADER)
            table = evaluate.metric_table(self.metrics, K)
            for uid, r, (gold, gen, sc) in zip(user_ids, ranks, rows_out):
                per = table[int(r) + 1]
                f.write("\t".join([str(uid), "\t".join(str(x) for x in per), str(gold), "||".join(map(str, gen)),
                                   "||".join(str(s) for s in sc)]) + "\n")
            if footer is not None:
                for name, val in zip(self.metrics, footer):
                    f.write(f"{name}: {val}\n")

    def test_from_model(self, rec_model_path=None, id_model_path=None):
        """single_runner_gram.py:359-368: the loaders built at construction, non-strict load."""
        self.model.eval()
        self._load(rec_model_path, strict=False)
        if not self.testloaders:
            raise RuntimeError("no test loader (see get_testloader)")
        for loader in self.testloaders:
            self.test_dataset_task(loader)

    def test(self, path=None, debug_test_small_set=False):
        """single_runner_gram.py:370-393 / distributed_runner_gram.py:425-443 (debug_test_on_train is a training-set
        diagnostic and stays out of this build)."""
        loaders = self._loaders_for("test")
        self.model.eval()
        self._load(path)
        for loader in loaders:
            self.test_dataset_task(loader)

    def validate_from_model(self, rec_model_path=None, id_model_path=None):
        loaders = self._loaders_for("validation")
        self.model.eval()
        self._load(rec_model_path, strict=False)
        for loader in loaders:
            self.test_dataset_task(loader, mode="validation")

    def validate(self, path=None, debug_test_small_set=False):
        loaders = self._loaders_for("validation")
        self.model.eval()
        self._load(path)
        for loader in loaders:
            self.test_dataset_task(loader, mode="validation")

    def train_generator(self):
        raise NotImplementedError("training is outside the scoring path this build accelerates (SURVEY.md §8)")

    def test_dataset_task(self, testloader, mode="test"):  # pragma: no cover - abstract
        raise NotImplementedError
