"""``create_model`` with the reference's dispatch and error behaviour (src/model/__init__.py:9-24)."""
from .config import T5Config
from .gram import GRAM, GenerateOutput


def create_model(model_type, config=None, **kwargs):
    if model_type == "gram":
        return GRAM(config=config, **kwargs)
    raise ValueError(f"Unknown model type: {model_type}")


__all__ = ["create_model", "GRAM", "T5Config", "GenerateOutput"]
