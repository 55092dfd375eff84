"""``from runner import get_runner`` (main_generative_gram.py:13,107-118,191-201) -> the gram_amd runners."""
from gram_amd.runner import DistributedRunnerGRAM, SingleRunnerGRAM, get_runner  # noqa: F401

__all__ = ["get_runner", "SingleRunnerGRAM", "DistributedRunnerGRAM"]
