"""Checks over the gfx950 assembly hipcc produced for libseer_hip.so (build.py keeps the device `.s` of every source under
lib/obj/ and runs these on each build; a violation fails the build).

Rule `pk_src1_hi` -- no packed (VOP3P `v_pk_*`) arithmetic instruction whose SECOND source is read with `op_sel[1] = 1`,
i.e. whose LOW result reads the HIGH half of src1 -- half-swapped or hi-broadcast alike:

    v_pk_fma_f32 v[36:37], v[76:77], v[36:37], v[84:85] op_sel:[0,1,0] op_sel_hi:[1,0,0]      (rotary epilogue, round 2)
    v_pk_add_f32 v[66:67], v[66:67], v[92:93] op_sel:[0,1] op_sel_hi:[1,0]                    (LayerNorm backward's row sums)

On MI355X, while a second process runs the denoising network on the same GPU, such an instruction computes its low result in
LANES 48..63 as if that operand were ZERO (fma: the addend alone; add: the other term alone; mul: 0) -- in three launches out of
four of `scripts/lab_pkswap.cpp`, which isolates one instruction per kernel (profiles/r03_flake_12_lab_pkswap.log): every form
with op_sel[1] = 1 fails (fma / mul / add, aliased destination or not, op_sel_hi 0 or 1, with or without neg), every other form
never does in 2e13 lane-iterations each (src0 or src2 half-swapped or hi-broadcast, lo-broadcasts, plain accumulates, v_pk_mov_b32,
scalar v_fma_f32), and nothing fails without the co-tenant or next to a co-tenant that runs the lab itself.  hipcc emits the form
freely: the SLP vectoriser builds it from scalar code that pairs the two halves of different values.  That was the "last-bit
replay difference" of round 2 and the 1 % gradient mismatches of the co-tenant training check (profiles/r03_flake_root_cause.md).
`v_pk_mov_b32` is exempt: its low result reads only src0.

Rule `attn40_vregs` -- attention40.hip issues its V'^T LDS reads in one asm statement (`lds_issue_kv`, ends with
`s_waitcnt lgkmcnt(8)`; ring form: `lds_issue_v`, and `lds_prefetch_k` for the next sub tile's K' fragments, neither with a wait)
and waits for them in a later one (`lds_wait_v` / `lds_wait_vk`, `s_waitcnt lgkmcnt(0)`): the hardware writes those registers
AFTER the issuing statement has ended, behind the register allocator's back.  Between the two statements no instruction may name
one of them (a copy or a spill would move stale data).

Rule `ff_fused_vregs` -- ff_fused.hip requests its weight fragments with `global_load_dwordx4` into registers (req4 / req10) and its
activation fragments with `ds_read_b128` (a_req) in asm statements WITHOUT a wait, and waits later with counted `s_waitcnt vmcnt(N)`
/ `lgkmcnt(0)` statements -- across the chunk loop's back edge.  The rule interprets the kernel's assembly over its control-flow
graph with the two counters modelled (in-order retirement; every vector-memory / LDS / scalar-memory instruction is an entry):
an instruction that names a V register whose load has not been retired on SOME path to it is a violation.  The compiler's own loads
pass through the same model.
"""
from __future__ import annotations

import re
from pathlib import Path
from typing import Iterable, List, Set, Tuple

_PK = re.compile(r"^\s*(v_pk_\w+)\s+v\[(\d+):(\d+)\],\s*(.*)$")
_LABEL = re.compile(r"^([A-Za-z_][\w$.]*):")


def _sel(mods: str, name: str, default: int, n: int) -> List[int]:
    m = re.search(name + r":\[([01,]+)\]", mods)
    if not m:
        return [default] * n
    v = [int(x) for x in m.group(1).split(",")]
    return v + [default] * (n - len(v))


def check_pk_src1_hi(lines: Iterable[str], fname: str = "") -> List[str]:
    out, kern = [], "?"
    for ln, line in enumerate(lines, 1):
        lab = _LABEL.match(line)
        if lab and not lab.group(1).startswith(".L"):
            kern = lab.group(1)
        m = _PK.match(line)
        if not m or m.group(1).startswith("v_pk_mov"):
            continue
        body = m.group(4).split(";")[0]
        osl = _sel(body, "op_sel", 0, 3)
        if osl[1] == 1:
            out.append(f"{fname}:{ln}: [{kern}] the low result reads the high half of src1 (op_sel[1] = 1): {line.strip()}")
    return out


_PK_F32_ARITH = re.compile(r"^\s*(v_pk_(?:fma|mul|add)_f32)\b")


def check_no_packed_fp32(lines: Iterable[str], fname: str = "") -> List[str]:
    """build.py::NO_PACKED_FP32 switches the packed fp32 arithmetic off for the device code (it serialises with MFMAs,
    profiles/r05_lab_mfma_valu.log).  The switch is a cc1 target feature the driver does not know: a toolchain that drops it would
    bring the instructions back without a word, so the emitted assembly is the check."""
    out, kern = [], "?"
    for ln, line in enumerate(lines, 1):
        lab = _LABEL.match(line)
        if lab and not lab.group(1).startswith(".L"):
            kern = lab.group(1)
        m = _PK_F32_ARITH.match(line)
        if m:
            out.append(f"{fname}:{ln}: [{kern}] packed fp32 arithmetic in device code ({m.group(1)}): NO_PACKED_FP32 did not reach "
                       f"the device compilation")
    return out


def _asm_start(line: str) -> bool:
    t = line.strip()
    return t.startswith(";;#ASMSTART") or t.startswith(";APP")


def _asm_end(line: str) -> bool:
    t = line.strip()
    return t.startswith(";;#ASMEND") or t.startswith(";NO_APP")


def _regs(text: str) -> Set[int]:
    s: Set[int] = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        s.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        s.add(int(a))
    return s


_BRANCH = re.compile(r"^\s*(s_cbranch_\w+|s_branch)\s+(\.L[\w$.]+)")
_LOCAL = re.compile(r"^(\.L[\w$.]*):")


def check_attn40_vregs(lines: Iterable[str], fname: str = "") -> Tuple[List[str], int]:
    """returns (violations, number of issue statements checked).  Control-flow aware: from every ISSUE statement (an asm block with
    LDS reads that are not all waited for inside it) every path is followed -- fall-through and branch targets, inside the kernel --
    until a WAIT statement (an asm block with `s_waitcnt lgkmcnt(0)` and no LDS read); an instruction on such a path that names
    a register the LDS is still writing is a violation.  (A scan in file order is not enough: hipcc lays the blocks of the partial-tile
    and re-run paths out between an issue and its wait.)"""
    lines = list(lines)
    out: List[str] = []
    checked = 0
    n = len(lines)
    # kernel extents
    starts = [i for i, l in enumerate(lines) if (m := _LABEL.match(l)) and not m.group(1).startswith(".L")]
    for ki, k0 in enumerate(starts):
        if "seer_attn40_kernel" not in lines[k0]:
            continue
        k1 = starts[ki + 1] if ki + 1 < len(starts) else n
        labels = {}
        for i in range(k0, k1):
            m = _LOCAL.match(lines[i])
            if m:
                labels[m.group(1)] = i
        # asm blocks: start index -> (end index, text)
        blocks = {}
        i = k0
        while i < k1:
            if _asm_start(lines[i]):
                j = i + 1
                while j < k1 and not _asm_end(lines[j]):
                    j += 1
                blocks[i] = (j, "".join(lines[i + 1:j]))
                i = j + 1
            else:
                i += 1
        for b0, (b1, text) in blocks.items():
            body = lines[b0 + 1:b1]
            dests = [r.group(1) for r in (re.match(r"\s*ds_read_\w+\s+(v\[\d+:\d+\]|v\d+)", b) for b in body) if r]
            wm = re.search(r"lgkmcnt\((\d+)\)", text)
            n_live = len(dests) if wm is None else min(int(wm.group(1)), len(dests))
            if not dests or n_live == 0:
                continue
            live: Set[int] = set()
            for dreg in dests[len(dests) - n_live:]:
                live |= _regs(dreg)
            checked += 1
            seen: Set[int] = set()
            stack = [b1 + 1]
            reached_wait = False
            while stack:
                i = stack.pop()
                while k0 <= i < k1 and i not in seen:
                    seen.add(i)
                    line = lines[i]
                    if i in blocks:
                        e1, t2 = blocks[i]
                        if "lgkmcnt(0)" in t2 and "ds_read" not in t2:
                            reached_wait = True
                            break                       # this path is closed: the data has landed
                        hit = _regs(t2) & live
                        if hit:
                            out.append(f"{fname}:{i + 1}: asm statement names V registers {sorted(hit)} before their LDS reads are waited for")
                        i = e1 + 1
                        continue
                    code = line.split(";")[0].strip()
                    if code and not code.startswith(".") and not _LABEL.match(line) and not _LOCAL.match(line):
                        if code.startswith("s_endpgm"):
                            break
                        hit = _regs(code) & live
                        if hit:
                            out.append(f"{fname}:{i + 1}: `{code}` names V registers {sorted(hit)} that an LDS issue statement "
                                       f"(line {b0 + 1}) is still writing")
                        br = _BRANCH.match(line)
                        if br:
                            tgt = labels.get(br.group(2))
                            if tgt is not None:
                                stack.append(tgt)
                            if br.group(1) == "s_branch":
                                break
                    i += 1
            if not reached_wait:
                out.append(f"{fname}:{b0 + 1}: LDS issue statement without a wait statement on any path behind it")
    return out, checked


_VM_LOAD = re.compile(r"^(global_load_|buffer_load_|flat_load_|scratch_load_)(\w+)\s+([va]\[\d+:\d+\]|[va]\d+),")
_VM_OTHER = re.compile(r"^(global_|buffer_|flat_|scratch_)(load_lds_|store_|atomic_)")
_DS_READ = re.compile(r"^ds_(read|load)\w*\s+([va]\[\d+:\d+\]|[va]\d+),")
_CNT = re.compile(r"(vmcnt|lgkmcnt|expcnt)\((\d+)\)")


def _regs_va(text: str) -> Set[int]:
    """V registers as their numbers, accumulation registers as 1000 + number"""
    s: Set[int] = set()
    for cls, a, b in re.findall(r"\b([va])\[(\d+):(\d+)\]", text):
        s.update(range(int(a) + (1000 if cls == "a" else 0), int(b) + 1 + (1000 if cls == "a" else 0)))
    for cls, a in re.findall(r"\b([va])(\d+)\b", text):
        s.add(int(a) + (1000 if cls == "a" else 0))
    return s


def _strip(q: tuple) -> tuple:
    """entries older than the oldest one that still owns registers retire first and own nothing: irrelevant"""
    i = 0
    while i < len(q) and not q[i]:
        i += 1
    return q[i:]


def check_async_vregs(lines: Iterable[str], kernel_substr: str, fname: str = "") -> Tuple[List[str], int]:
    """returns (violations, number of register-destined vector-memory loads seen inside asm statements of the kernel)"""
    lines = list(lines)
    out: List[str] = []
    n = len(lines)
    starts = [i for i, l in enumerate(lines) if (m := _LABEL.match(l)) and not m.group(1).startswith(".L")]
    n_async = 0
    for ki, k0 in enumerate(starts):
        if kernel_substr not in lines[k0]:
            continue
        k1 = starts[ki + 1] if ki + 1 < len(starts) else n
        labels = {m.group(1): i for i in range(k0, k1) if (m := _LOCAL.match(lines[i]))}
        in_asm = False
        asm_line = [False] * n
        for i in range(k0, k1):
            if _asm_start(lines[i]):
                in_asm = True
            elif _asm_end(lines[i]):
                in_asm = False
            asm_line[i] = in_asm
        seen = set()
        reported = set()
        work = [(k0 + 1, (), ())]
        while work:
            i, vm, lg = work.pop()
            while k0 <= i < k1:
                key = (i, vm, lg)
                if key in seen:
                    break
                seen.add(key)
                line = lines[i]
                code = line.split(";")[0].strip()
                if not code or code.startswith(".") or _LABEL.match(line) or _LOCAL.match(line) or code.startswith("//"):
                    i += 1
                    continue
                if code.startswith("s_endpgm"):
                    break
                named = _regs_va(code)
                pending = set().union(*vm, *lg) if (vm or lg) else set()
                hit = named & pending
                if hit and (i, tuple(sorted(hit))) not in reported:
                    reported.add((i, tuple(sorted(hit))))
                    out.append(f"{fname}:{i + 1}: `{code}` names registers {sorted(hit)} (accumulation registers: 1000 +) whose load is still in flight on some path")
                if code.startswith("s_waitcnt"):
                    if not _CNT.search(code):                       # raw immediate: assume it waits for everything
                        vm, lg = (), ()
                    for name, cnt in _CNT.findall(code):
                        c = int(cnt)
                        if name == "vmcnt":
                            vm = _strip(vm[len(vm) - c:] if c < len(vm) else vm) if c else ()
                        elif name == "lgkmcnt":
                            lg = _strip(lg[len(lg) - c:] if c < len(lg) else lg) if c else ()
                    i += 1
                    continue
                m = _VM_LOAD.match(code)
                if m and "_lds_" not in code:
                    vm = _strip(vm + (frozenset(_regs_va(m.group(3))),))
                    if asm_line[i]:
                        n_async += 1
                elif _VM_OTHER.match(code) or m:
                    vm = _strip(vm + (frozenset(),))
                else:
                    d = _DS_READ.match(code)
                    if d:
                        lg = _strip(lg + (frozenset(_regs_va(d.group(2))),))
                    elif code.startswith("ds_") or code.startswith("s_load_") or code.startswith("s_buffer_load_"):
                        lg = _strip(lg + (frozenset(),))
                br = _BRANCH.match(line)
                if br:
                    tgt = labels.get(br.group(2))
                    if tgt is not None:
                        work.append((tgt, vm, lg))
                    if br.group(1) == "s_branch":
                        break
                i += 1
    return out, n_async


def check_directory(objdir: Path, sources: Iterable[str] = ()) -> List[str]:
    """run every rule over the device assembly under `objdir`; returns the violations.  `sources`: the .hip files of the build --
    only their assembly is read (a stale .s of a removed source must neither fail nor mask a build); empty = every *gfx950.s.
    Scope of the packed-instruction rule: packed fp32 arithmetic (v_pk_{fma,mul,add}_f32: destinations are register PAIRS, which
    is what _PK matches) -- the forms measured in profiles/r03_flake_root_cause.md.  16-bit packed forms (one destination
    VGPR) were not measured and are not judged."""
    problems: List[str] = []
    files = sorted(objdir.glob("*gfx950.s"))
    stems = {Path(x).stem for x in sources}
    if stems:
        files = [f for f in files if f.name.split("-hip-")[0] in stems]
    if not files:
        return [f"no device assembly (*gfx950.s) under {objdir}: build with seervideoldm_amd.build"]
    pairs = 0
    ff_seen = None
    for f in files:
        lines = f.read_text().splitlines(keepends=True)
        problems += check_pk_src1_hi(lines, f.name)
        problems += check_no_packed_fp32(lines, f.name)[:5]
        if f.name.startswith("attention40"):
            v, p = check_attn40_vregs(lines, f.name)
            problems += v
            pairs += p
        if f.name.startswith("ff_fused"):
            v, ff_seen = check_async_vregs(lines, "seer_ff_fused_c320_kernel", f.name)
            problems += v
        if f.name.startswith("rowchain"):       # the same discipline: weight fragments requested by asm, waited for by count
            v, rc_seen = check_async_vregs(lines, "seer_rowchain_c320_kernel", f.name)
            problems += v
            if rc_seen < 30:
                problems.append(f"rowchain: only {rc_seen} asynchronous fragment requests found in the kernel's assembly (the check no longer sees them)")
    if pairs == 0:
        problems.append("attention40: no lds_issue_kv / lds_wait_v statement pair found in the assembly (the check no longer sees the kernel)")
    if ff_seen is not None and ff_seen < 30:
        problems.append(f"ff_fused: only {ff_seen} asynchronous fragment requests found in the kernel's assembly (the check no longer sees them)")
    return problems
