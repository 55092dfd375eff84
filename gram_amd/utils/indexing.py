"""Item / user indexing of the GRAM data path: mirror of ``gram_indexing`` and its helpers
(src/utils/indexing.py:132-322).

Reads the dataset directory the reference reads -- ``user_sequence.txt`` (user id + item ids), the item id file
(``item_generative_indexing_<type>.txt`` or ``--item_id_path``: item id + lexical id), ``similar_item_<cf>.txt``
(collaborative neighbours) and ``item_plain_text.txt`` -- and returns the three dictionaries the datasets are built
from.  ``item2input`` (item id -> item prompt text) is what ``GRAM.cache_passages`` is filled from: the same string for
every user that has the item in the history.

Out of scope: (re)generating lexical ids with the id-generator model (``phase != 0 and regenerate``).
Parity: tests/golden/dataset_cases.json holds the reference functions' outputs on a synthetic dataset directory
(oracle/make_dataset_fixtures.py); tests/test_host_logic.py compares.
"""
from __future__ import annotations

import os
from typing import Dict, List

TEXT_PROMPTS = ("all_text", "nothing", "only_title", "only_brand", "only_category", "only_tbc")


def read_lines(path: str) -> List[str]:
    """utils/utils.py:21-28."""
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    with open(path, "r") as fd:
        return [line.rstrip("\n") for line in fd]


def get_dict_from_lines(lines) -> Dict[str, str]:
    """indexing.py:216-224: key = text before the first blank, value = the rest (a line without a blank is an
    IndexError there and here)."""
    out = {}
    for line in lines:
        parts = line.split(" ", 1)
        out[parts[0]] = parts[1]
    return out


def _attributes(text: str, key: str) -> str:
    kept = [piece.strip() for piece in text.split(";") if piece.strip().startswith(key)]
    return "; ".join(kept).strip()


def get_dict_with_similar_items(args, item2lexid, text_file, data_path, dataset) -> Dict[str, str]:
    """indexing.py:227-322: ``similar items: <lexical ids of the top-k CF neighbours>; <item description>``."""
    kind = args.item_prompt
    if kind == "only_brand":
        assert dataset != "Yelp", "Yelp dataset does not have brand information."
    neighbours = {}
    with open(os.path.join(data_path, dataset, f"similar_item_{args.cf_model}.txt"), "r") as fd:
        for line in fd:
            if line.startswith("anchor"):  # header
                continue
            item, rest = line.split(" ", 1)
            neighbours[item] = rest.split()[: args.top_k_similar_item]

    def text_rows():
        with open(os.path.join(data_path, dataset, text_file), "r") as fd:
            for line in fd:
                yield line.split(" ", 1)

    desc = {}
    if kind == "all_text":
        for item, text in text_rows():
            desc[item] = text.strip()
    elif kind == "nothing":
        desc = {item: "" for item in item2lexid}
    elif kind == "only_title":
        key = "name:" if dataset == "Yelp" else "title:"
        for item, text in text_rows():
            desc[item] = _attributes(text, key)
    elif kind == "only_brand":
        for item, text in text_rows():
            desc[item] = _attributes(text, "brand:")
    elif kind == "only_category":
        for item, text in text_rows():
            desc[item] = _attributes(text, "categories:")
    elif kind == "only_tbc":
        # the reference keeps title/brand/categories in variables that survive from line to line: an item without
        # one of the attributes inherits the previous item's (and the very first one raises); same here
        amazon = dataset in ("Beauty", "Toys", "Sports")
        found = {}
        for item, text in text_rows():
            for piece in text.split(";"):
                piece = piece.strip()
                for key in (("title:", "brand:", "categories:") if amazon else ("name:", "categories:")):
                    if piece.startswith(key):
                        found[key] = piece
            if amazon:
                desc[item] = f"{found['title:']}; {found['brand:']}; {found['categories:']}"
            else:
                desc[item] = f"{found['name:']}; {found['categories:']}"
    else:
        raise ValueError(f"item_prompt should be one of {list(TEXT_PROMPTS)}, but got {kind}")

    out = {}
    for item, text in desc.items():
        names = [item2lexid[n] for n in neighbours[item]]
        out[item] = f"similar items: {', '.join(names)}; {text}".strip()
    return out


def gram_indexing(data_path, dataset, model_gen, tokenizer, regenerate=True, phase=0, args=None,
                  user_id_without_target_item=False, id_linking=False):
    """indexing.py:132-213.  Returns (user_sequence_dict {user: [item, ...]}, item2input {item: prompt text},
    item2lexid {item: lexical id})."""
    users = get_dict_from_lines(read_lines(os.path.join(data_path, dataset, "user_sequence.txt")))
    users = {u: seq.split() for u, seq in users.items()}

    chosen = os.path.join(data_path, dataset, args.item_id_path) if len(args.item_id_path) else ""
    if chosen and os.path.exists(chosen):
        index_file = chosen
        print(f"Load item id from {index_file}")
    else:
        index_file = os.path.join(data_path, dataset, f"item_generative_indexing_{args.hierarchical_id_type}.txt")
        if not os.path.exists(index_file):
            raise FileNotFoundError(
                f"Item index file {index_file} does not exist. Please generate it first. or Check the path.")
    if phase != 0 and regenerate:
        raise NotImplementedError("regenerating lexical ids with the id-generator model is outside this build (SURVEY.md §8)")
    item2lexid = get_dict_from_lines(read_lines(index_file))

    if args.item_prompt == "lexical_id":
        item2input = dict(item2lexid)
    elif args.item_prompt in TEXT_PROMPTS and args.top_k_similar_item > 0:
        item2input = get_dict_with_similar_items(args=args, item2lexid=item2lexid, text_file="item_plain_text.txt",
                                                 data_path=data_path, dataset=dataset)
    else:
        raise ValueError(f"Invalid item_prompt: {args.item_prompt}")
    if id_linking:
        item2input = {item: f"item: {item2lexid[item]}; {text}" for item, text in item2input.items()}
    return users, item2input, item2lexid
