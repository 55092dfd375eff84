"""Golden vectors for the evaluation data path: run the REFERENCE's own ``gram_indexing``
(/root/reference/src/utils/indexing.py:132-322) and ``TestDatasetGRAM`` (src/data/test_dataset_gram.py:19-231), loaded by
path (nothing copied), on a small synthetic dataset directory written to tests/golden/dataset_fixture/, and store the
argument sets + outputs in tests/golden/dataset_cases.json.
TEST INFRASTRUCTURE: run once in the build container (the reference does not exist on the GPU box).
    PYTHONDONTWRITEBYTECODE=1 python oracle/make_dataset_fixtures.py"""
import importlib.util
import json
import os
import random
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
FIX = os.path.join(ROOT, "tests", "golden", "dataset_fixture")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    ipy = types.ModuleType("IPython")  # imported, never called
    ipy.embed = lambda *a, **k: None
    sys.modules["IPython"] = ipy
    pkg = types.ModuleType("utils")
    pkg.__path__ = []
    sys.modules["utils"] = pkg
    pkg.utils = _load("utils.utils", os.path.join(REF, "utils", "utils.py"))
    pkg.indexing = _load("utils.indexing", os.path.join(REF, "utils", "indexing.py"))
    pkg.prompt = _load("utils.prompt", os.path.join(REF, "utils", "prompt.py"))
    return _load("ref_test_dataset_gram", os.path.join(REF, "data", "test_dataset_gram.py"))


def write_dataset(rng, name, n_items, n_users, yelp=False):
    d = os.path.join(FIX, name)
    os.makedirs(d, exist_ok=True)
    words = ["rene", "furterer", "shampoo", "oil", "mango", "butter", "kernel", "organic", "salt", "soap", "serum", "nail"]
    items = [f"I{j:04d}" for j in range(n_items)]
    with open(os.path.join(d, "item_generative_indexing_tiny_split.txt"), "w") as f:
        for it in items:
            f.write(f"{it} |{'|'.join('▁' + rng.choice(words) for _ in range(rng.randint(3, 4)))}\n")
    with open(os.path.join(d, "item_ids_alt.txt"), "w") as f:
        for it in items:
            f.write(f"{it} alt {rng.choice(words)} {rng.choice(words)}\n")
    with open(os.path.join(d, "similar_item_sasrec.txt"), "w") as f:
        f.write("anchor " + " ".join(f"top{k + 1}" for k in range(5)) + "\n")
        for it in items:
            f.write(it + " " + " ".join(rng.sample([x for x in items if x != it], 5)) + "\n")
    with open(os.path.join(d, "item_plain_text.txt"), "w") as f:
        for it in items:
            t = " ".join(rng.choice(words) for _ in range(3))
            if yelp:
                f.write(f"{it} name: {t}; city: {rng.choice(words)}; categories: {rng.choice(words)}, {rng.choice(words)}\n")
            else:
                f.write(f"{it} title: {t}; brand: {rng.choice(words)}; categories: {rng.choice(words)}, {rng.choice(words)}; "
                        f"description: {' '.join(rng.choice(words) for _ in range(6))} \n")
    with open(os.path.join(d, "user_sequence.txt"), "w") as f:
        for u in range(n_users):
            f.write(f"U{u:03d} " + " ".join(rng.choice(items) for _ in range(rng.randint(3, 9))) + "\n")


def main():
    ref = load_reference()
    rng = random.Random(7)
    os.makedirs(FIX, exist_ok=True)
    write_dataset(rng, "Beauty", 25, 12)
    write_dataset(rng, "Yelp", 15, 6, yelp=True)
    with open(os.path.join(FIX, "prompt.txt"), "w") as f:
        f.write("sequential; seen; What would user purchase after {history_lex_id} ?; {target_lex_id}\n"
                "sequential; seen; user has purchased items {history_lex_id}, predict the next one ?; {target_lex_id}\n"
                "sequential; unseen; Next item after {history_lex_id} ?; {target_lex_id}\n"
                "straightforward; seen; Which item for {user_id} ?; {target_lex_id}\n")
    base = dict(data_path="dataset_fixture", prompt_file="dataset_fixture/prompt.txt", reverse_history=1,
                user_id_without_target_item=0, id_linking=0, max_his=20, his_sep=" ; ", item_id_path="",
                hierarchical_id_type="tiny_split", item_prompt="all_text", cf_model="sasrec", top_k_similar_item=2,
                debug_test_100=0, rank=0, verbose_input_output=0)
    variants = [
        ("Beauty", "test", {}),
        ("Beauty", "validation", {"max_his": 3}),
        ("Beauty", "test", {"reverse_history": 0, "max_his": 2, "id_linking": 1}),
        ("Beauty", "test", {"item_prompt": "lexical_id", "top_k_similar_item": 0, "max_his": -1}),
        ("Beauty", "test", {"item_prompt": "nothing", "top_k_similar_item": 3}),
        ("Beauty", "test", {"item_prompt": "only_title", "item_id_path": "item_ids_alt.txt"}),
        ("Beauty", "validation", {"item_prompt": "only_brand", "his_sep": ", "}),
        ("Beauty", "test", {"item_prompt": "only_category", "top_k_similar_item": 1}),
        ("Beauty", "test", {"item_prompt": "only_tbc"}),
        ("Yelp", "test", {"item_prompt": "only_title"}),
        ("Yelp", "validation", {"item_prompt": "only_tbc", "max_his": 4}),
    ]
    cases = []
    cwd = os.getcwd()
    os.chdir(os.path.join(ROOT, "tests", "golden"))  # the fixtures hold paths relative to tests/golden
    try:
        for dataset, mode, over in variants:
            a = dict(base, **over)
            ds = ref.TestDatasetGRAM(types.SimpleNamespace(**a), dataset, "sequential", None, None, mode=mode)
            cases.append({"args": a, "dataset": dataset, "mode": mode, "len": len(ds), "all_items": ds.all_items,
                          "item2input": ds.item2input, "item2lexid": ds.item2lexid, "user_seq_dict": ds.user_seq_dict,
                          "samples": [ds[i] for i in range(len(ds))],
                          "history": [s["history"] for s in ds.data_samples], "info": sorted(ds.info)})
    finally:
        os.chdir(cwd)
    path = os.path.join(ROOT, "tests", "golden", "dataset_cases.json")
    json.dump(cases, open(path, "w"), ensure_ascii=False)
    print("wrote", path, [(c["dataset"], c["mode"], c["len"]) for c in cases])


if __name__ == "__main__":
    main()
