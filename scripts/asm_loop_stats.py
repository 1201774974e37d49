"""Instruction statistics of the main loop (the span of v_mfma instructions) of one kernel of a .s file.
usage: python scripts/asm_loop_stats.py <file.s> <kernel-name-substring>"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
funcs = re.split(r"\n(_Z[^\n:]*):[^\n]*\n", s)
for i in range(1, len(funcs), 2):
    name, body = funcs[i], funcs[i + 1]
    if sys.argv[2] not in name:
        continue
    body = body.split(".Lfunc_end")[0]
    lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
    idx = [k for k, l in enumerate(lines) if l.startswith("v_mfma")]
    print(name, "lines", len(lines), "mfma", len(idx), "span", idx[0], idx[-1])
    seg = lines[idx[0] - 80: idx[-1] + 5]
    c = collections.Counter(l.split()[0] for l in seg)
    for k, v in sorted(c.items(), key=lambda kv: -kv[1])[:45]:
        print(f"  {k:34s}{v}")
    print("  waits:", [l for l in seg if "vmcnt" in l])
    print("  labels:", [(k, l) for k, l in enumerate(lines) if l.endswith(":") and idx[0] - 300 < k < idx[-1] + 50])
    print("  branches:", [(k, l) for k, l in enumerate(lines) if l.startswith("s_cbranch") and idx[0] - 100 < k < idx[-1] + 50])
