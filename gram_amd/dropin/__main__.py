"""``python -m gram_amd.dropin <script.py> [args...]``: run a reference script with ``model`` / ``runner`` resolved to gram_amd."""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit("usage: python -m gram_amd.dropin /path/to/GRAM/src/main_generative_gram.py [its arguments]")
    script = os.path.abspath(argv[0])
    here = os.path.dirname(os.path.abspath(__file__))
    # what `python script.py` would have done, with the shim directory in front of the script's own
    sys.path[:0] = [here, os.path.dirname(script)]
    for name in ("model", "runner"):  # a previously imported reference package must not win
        sys.modules.pop(name, None)
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
