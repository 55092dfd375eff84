"""Single-process eval driver; mirror of src/runner/single_runner_gram.py:570-719."""
from __future__ import annotations

import logging

import numpy as np

from ..utils import evaluate
from .base import BaseRunner, _arg


class SingleRunnerGRAM(BaseRunner):
    def test_dataset_task(self, testloader, mode="test"):
        logging.info(f"[{mode}] testing {testloader.dataset.dataset} dataset on {testloader.dataset.task} task")
        from time import time
        t_start = time()
        ranks, total_time, examples, user_ids, rows_out = self._score_loader(testloader)
        K = self.generate_num
        sums = evaluate.metrics_from_ranks(ranks, self.metrics, K)
        test_total = len(ranks)
        metrics_res = sums / max(test_total, 1)
        logging.info("\n-------------------------------")
        logging.info("\n".join(examples))
        for name, val in zip(self.metrics, metrics_res):
            logging.info(f"{mode} {name}: {val}")  # single_runner_gram.py:708
        n_batches = max(len(testloader), 1)
        logging.info(f"Total inference time: {total_time:.2f}s for {n_batches} samples. Average: {total_time / n_batches:.4f}s")
        self.last_pred_file = None
        if _arg(self.args, "save_predictions", False):
            # single_runner_gram.py:580-588: ../preds/{timestamp}_{dataset}_{task}_pred_{mode}.tsv (--pred_dir / --pred_path: extensions)
            import datetime
            import os
            stamp = datetime.datetime.now().strftime("%Y%m%d_%H%M%S")
            fname = _arg(self.args, "pred_path", None) or os.path.join(
                _arg(self.args, "pred_dir", "../preds"), f"{stamp}_{testloader.dataset.dataset}_{testloader.dataset.task}_pred_{mode}.tsv")
            self._write_preds(fname, user_ids, ranks, rows_out, footer=metrics_res.tolist())
            self.last_pred_file = fname
        self.last_results = dict(metrics=dict(zip(self.metrics, metrics_res.tolist())), sums=sums, total=test_total,
                                 hit_ranks=ranks, generate_seconds=total_time, score_loader_seconds=time() - t_start,
                                 users_per_sec=test_total / total_time if total_time > 0 else float("nan"))
        return True
