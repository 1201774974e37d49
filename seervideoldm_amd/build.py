"""Build libseer_hip.so (gfx950) in-tree with hipcc.

`python -m seervideoldm_amd.build` or `build_library()`; hipcc cross-compiles without a GPU.  The .so lands in
seervideoldm_amd/lib/ (git-ignored, but it travels with the gpurun snapshot).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIBDIR = PKG / "lib"
LIB = LIBDIR / "libseer_hip.so"
SOURCES = ["gemm.hip", "gemm_ws.hip", "gemm_t320.hip", "ff_fused.hip", "rowchain.hip", "attention.hip", "attention40.hip", "attention_bwd.hip", "gemm_tn.hip", "norm.hip", "elementwise.hip", "train.hip"]
ARCH = "gfx950"
# per-source extra flags.  attention: keep the MFMA accumulators in VGPRs (gfx950's unified file allows it) -- the online
# softmax touches S and O every key tile, and the default AGPR form costs ~220 v_accvgpr moves per tile per wave.
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "attention40.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "attention_bwd.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "gemm_tn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               # GEMM: the epilogue reads every accumulator once; VGPR form saves those moves (+0.7 % on the step, A/B in one run)
               "gemm.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "gemm_ws.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "gemm_t320.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               # rowchain: 120 accumulators + 80 weight-fragment registers + 48 of activation fragments fit the V file; its
               # epilogues read every accumulator once
               "rowchain.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}
# ff_fused.hip: default form -- its accumulators and its W1 fragment ring live in the accumulation registers (120 + 48 + 80), the
# asynchronously written fragment registers must never meet a register-allocator copy (asm_check.py::check_async_vregs)


# No packed fp32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) anywhere: on gfx950 those instructions do not run in the
# shadow of MFMAs -- a loop of MFMAs with v_pk_fma_f32 between them takes the SUM of the two loops' times (plain v_fma_f32: 0.7 of
# the shorter one hidden), also across the waves of a SIMD, so a wave in its epilogue stalls its neighbour's MFMAs; and with two
# waves per SIMD a packed instruction costs 3.9 ns against 2.3 ns for a plain one (profiles/r05_lab_mfma_valu.log).  hipcc's SLP
# vectoriser forms them freely from scalar code; this turns the feature off for the device code.  It is a cc1 target feature (the
# driver has no -m flag for it and -Xarch_device cannot forward an -Xclang pair), so the host pass of a .hip file sees it too and
# ignores it as unknown to x86; asm_check.py::check_no_packed_fp32 fails the build if the instructions are in the device assembly
# anyway (a toolchain that stops honouring the feature must not bring them back silently).
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 to build libseer_hip.so for gfx950)")


def _digest() -> str:
    h = hashlib.sha256()
    for p in sorted(list(CSRC.glob("*")) + [ROOT / "include" / "seer_hip.h", Path(__file__).resolve(), PKG / "asm_check.py"]):
        if p.is_file():
            h.update(p.name.encode())
            h.update(p.read_bytes())
    return h.hexdigest()


def build_library(force: bool = False, verbose: bool = False) -> Path:
    LIBDIR.mkdir(exist_ok=True)
    stamp = LIBDIR / "build.sha256"
    dig = _digest()
    if not force and LIB.exists() and stamp.exists() and stamp.read_text().strip() == dig:
        return LIB
    hipcc = _hipcc()
    objdir = LIBDIR / "obj"
    objdir.mkdir(exist_ok=True)
    # -save-temps=obj: the device assembly of every source stays next to its object for asm_check (the other temporaries go)
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT / 'include'}", f"-I{CSRC}",
             "-fno-gpu-rdc", "-Wno-unused-result", "-save-temps=obj", *NO_PACKED_FP32]

    def compile_one(src: str) -> Path:
        obj = objdir / (src + ".o")
        cmd = [hipcc, *flags, *EXTRA_FLAGS.get(src, []), "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        stem = src.split(".")[0]
        for tmp in objdir.glob(f"{stem}-*"):
            if not tmp.name.endswith("gfx950.s"):
                tmp.unlink()
        for tmp in objdir.glob(f"{src}-*"):
            tmp.unlink()
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    # the assembly rules of asm_check.py: instruction forms that misbehaved on MI355X, and the register discipline of the
    # inline-asm LDS reads in attention40.hip
    from .asm_check import check_directory
    problems = check_directory(objdir, SOURCES)
    if problems:
        raise RuntimeError("assembly check failed (seervideoldm_amd/asm_check.py):\n" + "\n".join(problems[:40]))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    # hipcc can drop the HOST stub of a kernel template instantiation without a diagnostic (seen with un-cast integer arguments of
    # __builtin_amdgcn_raw_ptr_buffer_load_lds, round 4): the library then links and only fails when it is loaded
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    r = subprocess.run([nm, "-u", str(LIB)], capture_output=True, text=True)
    missing = [l.split()[-1] for l in r.stdout.splitlines() if "__device_stub__" in l]
    if missing:
        LIB.unlink()
        raise RuntimeError("kernel launch stubs missing from the library (host side of these instantiations was not emitted):\n"
                           + "\n".join(missing[:10]))
    stamp.write_text(dig)
    return LIB


def build_c_caller() -> Path:
    """tests/abi_caller.c -> build/abi_caller: a plain C99 host of the ABI, compiled with gcc against include/seer_hip.h and
    the HIP runtime's C API (git-ignored; travels to the GPU box with the gpurun snapshot)."""
    out = ROOT / "build" / "abi_caller"
    src = ROOT / "tests" / "abi_caller.c"
    out.parent.mkdir(exist_ok=True)
    if out.exists() and out.stat().st_mtime >= max(src.stat().st_mtime, LIB.stat().st_mtime):
        return out
    rocm = Path(os.environ.get("ROCM_PATH", "/opt/rocm"))
    cmd = ["gcc", "-std=c99", "-O1", "-D__HIP_PLATFORM_AMD__", f"-I{ROOT / 'include'}", f"-I{rocm / 'include'}", str(src),
           "-o", str(out), f"-L{LIBDIR}", "-lseer_hip", f"-L{rocm / 'lib'}", "-lamdhip64", "-lm",
           "-Wl,-rpath,$ORIGIN/../seervideoldm_amd/lib", f"-Wl,-rpath,{rocm / 'lib'}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"gcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    return out


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
