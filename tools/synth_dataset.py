"""A Beauty-sized SYNTHETIC dataset directory for end-to-end runs of the drop-in runners (bench.py `extras.e2e_runner`,
tests/test_gpu_runner.py): no tokenizer, dataset or checkpoint exists offline, so the files the reference's data path reads
(src/utils/indexing.py:132-213, src/data/test_dataset_gram.py:83-218) are generated with the SHAPE of BASELINE.json configs[1]:

  * items: the 12 101 Beauty lexical ids of tests/golden/tries.npz (the real item-ID file as token-id arrays, piece -> id by
    enumeration), written back as ``|▁w<id>|▁w<id>...`` strings, so that with `SynthTokenizer` (below) the candidate Trie the
    runner builds is exactly the real Beauty Trie (75 892 nodes, fan-out <= 255, 7 / 8 pieces);
  * item texts long enough that every item prompt fills the 128-token limit (the headline's L = 128);
  * users with 3..8 interactions; with ``--max_his 2`` every user has N = 3 passages (user prompt + 2 item prompts).

`SynthTokenizer` is tests/stub_tokenizer.py's whitespace tokenizer with the ``w<id>`` pieces mapped to ``<id>`` itself.
"""
from __future__ import annotations

import os
from types import SimpleNamespace

import numpy as np

from tests.stub_tokenizer import StubTokenizer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROMPTS = ("sequential; seen; What would user purchase after {history_lex_id} ?; {target_lex_id}\n"
           "sequential; seen; user has purchased items {history_lex_id}, predict the next one ?; {target_lex_id}\n"
           "sequential; unseen; Next item after {history_lex_id} ?; {target_lex_id}\n")


class SynthTokenizer(StubTokenizer):
    """StubTokenizer that breaks lexical ids at their '|' separators like SentencePiece does (the separator pieces 1820 / 9175 are
    what the runner and the collator strip) and maps a piece ``w<id>`` to the token id ``<id>``."""

    def __init__(self):
        self._vocab = {}  # piece -> id, filled on first sight (a real tokenizer's vocabulary lookup)

    def convert_tokens_to_ids(self, tokens):
        out, vocab = [], self._vocab
        for t in tokens:
            for piece in (t.replace("|", " | ").split() if "|" in t else (t,)):
                v = vocab.get(piece)
                if v is None:
                    v = vocab[piece] = int(piece[1:]) if piece[:1] == "w" and piece[1:].isdigit() else self._id(piece)
                out.append(v)
        return out


def make(root: str, n_users: int = 8192, dataset: str = "Beauty", seed: int = 2023, words_per_item: int = 150) -> SimpleNamespace:
    """Write the dataset directory under `root` and return the reference-shaped args namespace that reads it."""
    rng = np.random.default_rng(seed)
    cands = np.load(os.path.join(ROOT, "tests", "golden", "tries.npz"))[f"{dataset}_cands"]
    n_items = cands.shape[0]
    ddir = os.path.join(root, dataset)
    os.makedirs(ddir, exist_ok=True)
    items = [f"I{i:05d}" for i in range(n_items)]
    lex = []
    for row in cands:
        pieces = [int(t) for t in row if t > 1]  # drop start (0), EOS (1), padding (-1)
        lex.append("".join(f"|▁w{t}" for t in pieces))
    with open(os.path.join(ddir, "item_generative_indexing_synth.txt"), "w") as f:
        f.writelines(f"{it} {lx}\n" for it, lx in zip(items, lex))
    vocab = np.array([f"t{j}" for j in range(20000)])
    with open(os.path.join(ddir, "item_plain_text.txt"), "w") as f:
        for it in items:
            w = vocab[rng.integers(0, len(vocab), size=words_per_item)]
            f.write(f"{it} title: {' '.join(w[:8])}; brand: {w[8]}; categories: {' '.join(w[9:14])}; description: {' '.join(w[14:])} \n")
    with open(os.path.join(ddir, "similar_item_sasrec.txt"), "w") as f:
        f.write("anchor top1 top2 top3 top4 top5\n")
        nb = rng.integers(0, n_items, size=(n_items, 5))
        f.writelines(f"{it} {' '.join(items[j] for j in nb[i])}\n" for i, it in enumerate(items))
    with open(os.path.join(ddir, "user_sequence.txt"), "w") as f:
        lens = rng.integers(3, 9, size=n_users)
        for u in range(n_users):
            f.write(f"U{u:06d} {' '.join(items[j] for j in rng.integers(0, n_items, size=lens[u]))}\n")
    with open(os.path.join(root, "prompt.txt"), "w") as f:
        f.write(PROMPTS)
    return args_for(root, dataset)


def args_for(root: str, dataset: str = "Beauty", **kw) -> SimpleNamespace:
    """The reference's flags (src/arguments.py) for scoring the generated directory at BASELINE.json configs[1]'s shape: beam 20 /
    top-20, N = 3 passages of <= 128 tokens, and the reference's DEFAULT --eval_batch_size 1 (arguments.py:84-86)."""
    a = dict(data_path=root, prompt_file=os.path.join(root, "prompt.txt"), datasets=dataset, tasks="sequential", reverse_history=1,
             user_id_without_target_item=0, id_linking=0, max_his=2, his_sep=" ; ", item_id_path="", hierarchical_id_type="synth",
             item_prompt="all_text", cf_model="sasrec", top_k_similar_item=2, debug_test_100=0, rank=0, verbose_input_output=0,
             eval_batch_size=1, metrics="hit@5,hit@10,ndcg@5,ndcg@10", beam_size=20, length_penalty=1.0, item_id_type="split",
             item_prompt_max_len=128, target_max_len=32, save_predictions=False, debug_test_small_set=0, passage_cache=1)
    a.update(kw)
    return SimpleNamespace(**a)
