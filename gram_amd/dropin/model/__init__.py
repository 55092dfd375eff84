"""``from model import create_model`` (main_generative_gram.py:15,73,158) -> the HIP-backed GRAM."""
from gram_amd.model import GRAM, T5Config, create_model  # noqa: F401

__all__ = ["create_model", "GRAM", "T5Config"]
