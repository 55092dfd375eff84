"""Evaluation dataset of the scoring path: mirror of ``TestDatasetGRAM`` (src/data/test_dataset_gram.py:19-231).

One sample per user: ``input`` = [coarse user prompt, item prompt of each history item] (the passages CollatorGRAM
stacks, most recent item first when ``reverse_history``), ``output`` = the target item's lexical id, ``user_id``.
test mode holds out the last item, validation the second to last (and drops the last from the history).
``all_items`` (every lexical id) feeds the runner's Trie; ``item2input`` feeds the passage cache.
"""
from __future__ import annotations

import logging

from torch.utils.data import Dataset

from ..utils import indexing
from ..utils.prompt import check_task_prompt, get_info_from_prompt, load_prompt_template


class TestDatasetGRAM(Dataset):
    __test__ = False  # not a pytest class

    def __init__(self, args, dataset, task, model_gen, tokenizer, regenerate=False, phase=0, debug_test_small_set=False,
                 mode="test"):
        super().__init__()
        self.args, self.dataset, self.task, self.phase, self.mode = args, dataset, task, phase, mode
        self.data_path = args.data_path
        self.model_gen, self.tokenizer = model_gen, tokenizer
        self.reverse_history = args.reverse_history
        self.user_id_without_target_item = args.user_id_without_target_item
        self.id_linking = args.id_linking

        self.prompt = load_prompt_template(args.prompt_file, [task])
        check_task_prompt(self.prompt, [task])
        self.info = get_info_from_prompt(self.prompt)
        if "history_lex_id" in self.info:
            self.max_his, self.his_sep = args.max_his, args.his_sep

        self.user_seq_dict, self.item2input, self.item2lexid = indexing.gram_indexing(
            data_path=self.data_path, dataset=dataset, model_gen=model_gen, tokenizer=tokenizer, regenerate=regenerate,
            phase=phase, args=args, user_id_without_target_item=self.user_id_without_target_item, id_linking=self.id_linking)
        self.all_items = list(self.item2lexid.values())

        if mode == "test":
            self.data_samples = self._load(holdout=1)
        elif mode == "validation":
            self.data_samples = self._load(holdout=2)
        else:
            raise ValueError(f"Invalid mode: {mode}")
        if args.debug_test_100 or debug_test_small_set:
            self.data_samples = self.data_samples[:100]
            if args.rank == 0:
                logging.info(">>>> Debug mode: only use 100 samples for test (TestDatasetGRAM)")
        self.construct_sentence()

    def _load(self, holdout: int):
        """load_test (:83-130) / load_validation (:132-177): the target is the holdout-th item from the end."""
        samples = []
        for user, items in self.user_seq_dict.items():
            target = items[-holdout]
            history = items[:-holdout]
            if self.max_his > 0:
                history = history[-self.max_his:]
            shown = history[::-1] if self.reverse_history else history
            samples.append({
                "dataset": self.dataset, "user_id": user, "target": target, "target_lex_id": self.item2lexid[target],
                "history": self.his_sep.join(shown),
                "history_input": [self.item2input[h] for h in shown],
                # the user prompt lists the lexical ids most recent first whatever reverse_history says
                "history_lex_id": self.his_sep.join(self.item2lexid[h] for h in history[::-1]),
            })
        return samples

    def load_test(self):
        return self._load(holdout=1)

    def load_validation(self):
        return self._load(holdout=2)

    def construct_sentence(self):
        """:179-216: passage 0 = the coarse user prompt over the lexical ids, then one item prompt per history item."""
        self.data = {"input": [], "output": [], "user_id": []}
        for s in self.data_samples:
            self.data["input"].append([f"What would user purchase after {s['history_lex_id']} ?"] + s["history_input"])
            self.data["output"].append(s["target_lex_id"])
            self.data["user_id"].append(s["user_id"])

    def __len__(self):
        return len(self.data_samples)

    def __getitem__(self, idx):
        return self.get_item(idx)

    def get_item(self, idx):
        return {"input": self.data["input"][idx], "output": self.data["output"][idx], "user_id": self.data["user_id"][idx]}
