"""python scripts/check_asm.py [objdir]: run the assembly rules of seervideoldm_amd/asm_check.py over the device assembly the
build kept under seervideoldm_amd/lib/obj (build.py runs the same check and fails on a violation)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seervideoldm_amd.asm_check import check_directory  # noqa: E402

if __name__ == "__main__":
    d = Path(sys.argv[1]) if len(sys.argv) > 1 else Path(__file__).resolve().parents[1] / "seervideoldm_amd" / "lib" / "obj"
    problems = check_directory(d)
    print("\n".join(problems) if problems else f"asm check: {len(list(d.glob('*gfx950.s')))} files clean")
    sys.exit(1 if problems else 0)
