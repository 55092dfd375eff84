"""Drop-in shims for the reference's bare-name imports.

``main_generative_gram.py`` runs as a script from ``src/`` and reaches the scoring path through two bare-name imports,
``from model import create_model`` (main:15) and ``from runner import get_runner`` (main:13).  The packages ``model`` and
``runner`` in this directory re-export the gram_amd implementations under those names; everything else the script
imports (``utils``, ``arguments``, ``data`` ...) stays the reference's own.  Python puts the SCRIPT's directory ahead of
PYTHONPATH, so the shims are activated by a launcher rather than an environment variable:

    python -m gram_amd.dropin /path/to/GRAM/src/main_generative_gram.py --datasets Beauty --train 0 ...

which runs the unmodified script with ``<this dir>`` in front of ``src/`` on sys.path (see __main__.py).
"""
