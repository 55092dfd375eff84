"""Prompt-template file handling: mirror of src/utils/prompt.py:6-63.

A template line is ``task; seen|unseen; input template; output template``; templates of a (task, kind) pair are
numbered in file order."""
from __future__ import annotations

import os
import re

from .indexing import read_lines


def load_prompt_template(path, task_list):
    if not os.path.exists(path):
        raise FileNotFoundError
    templates = {}
    for line in read_lines(path):
        fields = [f.strip() for f in line.split(";")]
        if fields[0] not in task_list:
            continue
        kinds = templates.setdefault(fields[0], {})
        numbered = kinds.setdefault(fields[1], {})
        numbered[str(len(numbered))] = {"Input": fields[2], "Output": fields[3]}
    return templates


def get_info_from_prompt(prompt_templates):
    """Names of the ``{placeholders}`` used anywhere in the templates (order unspecified, as in the reference)."""
    found = set()
    for kinds in prompt_templates.values():
        for numbered in kinds.values():
            for tpl in numbered.values():
                found.update(re.findall(r"\{.*?\}", tpl["Input"]))
                found.update(re.findall(r"\{.*?\}", tpl["Output"]))
    return [name[1:-1] for name in found]


def check_task_prompt(prompt_templates, task_list):
    for task in task_list:
        assert task in prompt_templates, f"No prompt for {task} task"
