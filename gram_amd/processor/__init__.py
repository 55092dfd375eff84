"""Host-side input construction (mirror of the reference's src/processor package for the scoring path)."""
from .collator import CollatorGRAM  # noqa: F401
