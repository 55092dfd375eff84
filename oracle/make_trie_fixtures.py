"""Derive small, shape-exact fixtures from the reference's dataset files (build container only).

    python oracle/make_trie_fixtures.py  ->  tests/golden/tries.npz

For each dataset: the candidate token sequences the runner builds at
single_runner_gram.py:594-619 ([0] + one id per '|'-separated piece + EOS=1), with the
SentencePiece ids replaced by an enumeration (piece -> 2 + first-seen index): no tokenizer is
available offline, and the Trie's shape (node count, fan-out, depth) only depends on piece
identity.  Also the histogram of N = 1 + min(len(history), 20) passages per test
user (test_dataset_gram.py:105-107, Collator.py:348-350).  Data only; no reference source.
"""
import os
import numpy as np

REF = "/root/reference/rec_datasets"
FILES = {
    "Beauty": "item_generative_indexing_hierarchy_v1_c128_l7_len32768_split.txt",
    "Toys": "item_generative_indexing_hierarchy_v1_c32_l5_len32768_split.txt",
    "Sports": "item_generative_indexing_hierarchy_v1_c32_l7_len32768_split.txt",
    "Yelp": "item_generative_indexing_hierarchy_v1_c32_l9_len128_split.txt",
}


def main():
    out = {}
    for ds, fn in FILES.items():
        vocab, seqs = {}, []
        for line in open(os.path.join(REF, ds, fn)):
            _item, rest = line.rstrip("\n").split(" ", 1)
            pieces = rest.split("|")[1:]
            seqs.append([0] + [vocab.setdefault(p, 2 + len(vocab)) for p in pieces] + [1])
        width = max(len(s) for s in seqs)
        arr = np.full((len(seqs), width), -1, dtype=np.int16 if len(vocab) < 32000 else np.int32)
        for i, s in enumerate(seqs):
            arr[i, : len(s)] = s
        out[f"{ds}_cands"] = arr
        useq = os.path.join(REF, ds, "user_sequence.txt")
        if os.path.exists(useq):
            hist = np.zeros(22, dtype=np.int64)
            for line in open(useq):
                items = line.split()[1:]
                hist[1 + min(len(items) - 1, 20)] += 1
            out[f"{ds}_npassage_hist"] = hist
        print(ds, arr.shape, "vocab", len(vocab), "N-hist" if useq else "")
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tries.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
