"""gram_amd -- MI355X-native implementation of GRAM's multi-granular late-fusion generative
scoring path (encoder -> late fusion -> beam-shared KV bank -> Trie-constrained beam search),
behind the reference's own ``create_model`` / ``GRAM.generate`` / runner interface.

Importing the package binds libgram_hip.so; a missing library is an ImportError (no fallback)."""
from . import _lib

_lib.load()

from .model import GRAM, T5Config, create_model  # noqa: E402

__all__ = ["GRAM", "T5Config", "create_model"]
