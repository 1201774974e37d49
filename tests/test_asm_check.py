"""The assembly rules the build enforces (seervideoldm_amd/asm_check.py): the packed-fp32 forms that lost a term on MI355X next
to a co-tenant process are recognised, the harmless ones are not, and the device assembly of the library in the tree is clean."""
from pathlib import Path

import pytest

from seervideoldm_amd import asm_check

KERNEL = "_ZN12_GLOBAL__N_11kEv:\n"
# the two forms measured failing (profiles/r03_flake_root_cause.md): the rotary epilogue of round 2, LayerNorm backward's row sums
BAD = ["\tv_pk_fma_f32 v[36:37], v[76:77], v[36:37], v[84:85] op_sel:[0,1,0] op_sel_hi:[1,0,0]\n",
       "\tv_pk_add_f32 v[66:67], v[66:67], v[92:93] op_sel:[0,1] op_sel_hi:[1,0]\n",
       "\tv_pk_add_f32 v[62:63], v[58:59], v[62:63] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n",
       "\tv_pk_mul_f32 v[38:39], v[38:39], v[50:51] op_sel:[0,1]\n",                     # hi half of src1 broadcast: fails too
       "\tv_pk_fma_f32 v[44:45], v[46:47], v[58:59], v[44:45] op_sel:[0,1,0]\n"]
# lo-broadcasts, negations, plain accumulates, src0 / src2 swapped or hi-broadcast: measured clean (scripts/lab_pkswap.cpp)
GOOD = ["\tv_pk_mul_f32 v[30:31], v[34:35], v[30:31] op_sel_hi:[0,1]\n",
        "\tv_pk_fma_f32 v[146:147], v[146:147], v[212:213], v[202:203] op_sel_hi:[1,0,1]\n",
        "\tv_pk_fma_f32 v[130:131], v[130:131], v[132:133], v[138:139] neg_lo:[1,0,0] neg_hi:[1,0,0]\n",
        "\tv_pk_fma_f32 v[10:11], v[56:57], v[32:33], v[10:11] op_sel:[1,0,0]\n",          # src0 hi-broadcast
        "\tv_pk_mul_f32 v[216:217], v[148:149], v[216:217] op_sel:[1,0] op_sel_hi:[0,1]\n",  # src0 half-swapped
        "\tv_pk_fma_f32 v[4:5], v[6:7], v[8:9], v[4:5] op_sel:[0,0,1] op_sel_hi:[1,1,0]\n",    # src2 half-swapped
        "\tv_pk_mov_b32 v[96:97], v[92:93], v[94:95] op_sel:[0,1]\n",
        "\tv_pk_mov_b32 v[74:75], v[38:39], v[42:43] op_sel:[1,0]\n",
        "\tv_pk_add_f32 v[8:9], v[8:9], 1.0 op_sel_hi:[1,0]\n"]


@pytest.mark.parametrize("line", BAD)
def test_src1_high_half_reads_are_flagged(line):
    hits = asm_check.check_pk_src1_hi([KERNEL, line], "x.s")
    assert len(hits) == 1 and "_ZN12_GLOBAL__N_11kEv" in hits[0] and "op_sel[1] = 1" in hits[0]


def test_measured_clean_forms_pass():
    assert asm_check.check_pk_src1_hi([KERNEL] + GOOD, "x.s") == []


def test_library_assembly_in_tree_is_clean():
    objdir = Path(asm_check.__file__).resolve().parent / "lib" / "obj"
    if not list(objdir.glob("*gfx950.s")):
        pytest.skip("no kept device assembly (run __graft_entry__.build())")
    assert asm_check.check_directory(objdir) == []


def test_packed_fp32_arithmetic_is_refused_in_device_assembly():
    """build.py::NO_PACKED_FP32 must have reached the device compilation: any v_pk_{fma,mul,add}_f32 fails the build"""
    from seervideoldm_amd.asm_check import check_no_packed_fp32
    arith = [l for l in GOOD + BAD if "v_pk_mov" not in l]
    assert len(check_no_packed_fp32(["k:\n"] + GOOD + BAD, "x.s")) == len(arith) == 12
    assert check_no_packed_fp32(["k:\n", "\tv_pk_mov_b32 v[2:3], v[4:5], v[6:7]\n", "\tv_fma_f32 v1, v2, v3, v4\n",
                                 "\tv_pk_mul_f16 v1, v2, v3\n"], "x.s") == []
