"""Hyper-parameter container with the attribute names of the reference's T5Config
(src/model/gram_t5_config.py:85-143) plus the four attributes main_generative_gram.py:67-70 sets.
Any object exposing these attributes (e.g. a HuggingFace ``T5Config``) is accepted by GRAM."""
from __future__ import annotations

from dataclasses import dataclass


@dataclass
class T5Config:
    vocab_size: int = 32128
    d_model: int = 512
    d_kv: int = 64
    d_ff: int = 2048
    num_layers: int = 6
    num_decoder_layers: int = 6
    num_heads: int = 8
    relative_attention_num_buckets: int = 32
    relative_attention_max_distance: int = 128
    dropout_rate: float = 0.1
    layer_norm_epsilon: float = 1e-6
    initializer_factor: float = 1.0
    feed_forward_proj: str = "relu"
    is_encoder_decoder: bool = True
    use_cache: bool = True
    pad_token_id: int = 0
    eos_token_id: int = 1
    decoder_start_token_id: int = 0
    tie_word_embeddings: bool = True
    # GRAM additions (main_generative_gram.py:67-70)
    max_seq_len: int = 128
    max_item_num: int = 20
    use_position_embedding: bool = True
    sample_num: int = 1

    _PRESETS = None

    @classmethod
    def named(cls, name: str, **overrides) -> "T5Config":
        presets = {
            "t5-small": dict(d_model=512, d_ff=2048, num_layers=6, num_decoder_layers=6, num_heads=8),
            "t5-base": dict(d_model=768, d_ff=3072, num_layers=12, num_decoder_layers=12, num_heads=12),
            "t5-large": dict(d_model=1024, d_ff=4096, num_layers=24, num_decoder_layers=24, num_heads=16),
        }
        if name not in presets:
            raise ValueError(f"unknown backbone {name!r}")
        kw = dict(presets[name])
        kw.update(overrides)
        return cls(**kw)
