"""python -m seervideoldm_amd.compat <script.py> [args...]: run a reference script with the hot-path aliases installed"""
import os
import runpy
import sys

from . import install


def main() -> None:
    if len(sys.argv) < 2:
        raise SystemExit("usage: python -m seervideoldm_amd.compat <script.py> [args...]")
    script = os.path.abspath(sys.argv[1])
    sys.argv = [script] + sys.argv[2:]
    sys.path[0] = os.path.dirname(script)          # what `python script.py` does
    install(vae=os.environ.get("SEER_COMPAT_VAE", "0") == "1")
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
