"""python scripts/check_attn40_asm.py: the register discipline of attention40.hip's split LDS reads (`lds_issue_kv` ->
`lds_wait_v`: the hardware writes the V registers after the first asm statement has ended, so nothing between the two
statements may name them).  Same rule as build.py enforces through seervideoldm_amd/asm_check.py."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seervideoldm_amd.asm_check import check_attn40_vregs  # noqa: E402

if __name__ == "__main__":
    d = Path(__file__).resolve().parents[1] / "seervideoldm_amd" / "lib" / "obj"
    files = sorted(d.glob("attention40*gfx950.s"))
    if not files:
        sys.exit(f"no attention40 assembly under {d}: run python -m seervideoldm_amd.build first")
    bad, pairs = [], 0
    for f in files:
        v, p = check_attn40_vregs(f.read_text().splitlines(keepends=True), f.name)
        bad += v
        pairs += p
    print("\n".join(bad) if bad else f"attention40: {pairs} issue/wait statement pairs, no V register named in between")
    sys.exit(1 if bad or pairs == 0 else 0)
