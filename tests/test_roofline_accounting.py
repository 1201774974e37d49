"""The accounting contract of bench.py's `roofline` object (seervideoldm_amd/profiler.py), checked without a GPU: the FULL-SIZE
kernel schedule of BASELINE config 2 (CFG batch 2 x 12 frames x 32x32 latent, SD-v1-5 widths) walked on torch's meta device with a
shape-only stand-in for the kernel library (tests/shape_ops_backend.py).

Round 4's line counted ten launches that never ran: ops.gemm(..., ln=...) returns None (nothing launched) when the library keeps the
level-0 GEGLU projection on LayerNorm + the weight-stationary kernel, and the profiler recorded 2MNK for that call anyway.  Pinned
here: a call that launches nothing leaves no record; one step = 221 launches of the GEMM / conv class (288 before ff.net.2 and proj_out
became one launch, 256 before the ten 320-channel feed-forwards became one launch each, 246 before the row-local chains in front of the
320-channel attention launches did, 231 before the feed-forward launch took the block's last to_out + residual as its prologue) and
5.20 TFLOP of EXECUTED work
(SURVEY 8(d)'s 5.85 TFLOP for the step counts the two convs behind a nearest-2x upsample at 36 tap products per source pixel; the four
2x2 phase convs that run execute 16) -- and every row names the roof its arithmetic intensity selects."""
import pytest
import torch

from seervideoldm_amd import SeerUNet, synth
from seervideoldm_amd.profiler import TimedOps
from seervideoldm_amd.unet import _Engine
from tests import shape_ops_backend as sob


class _Evt:
    def record(self):
        pass

    def elapsed_time(self, other):
        return 1e-3              # 1 us per launch: fractions are not asserted, the bookkeeping is


@pytest.fixture(scope="module")
def walk():
    return _walk_config2()


def _walk_config2():
    cfg = dict(synth.SD15_UNET_CFG)
    model = SeerUNet(**cfg).to("meta")
    timed = TimedOps(base=sob, event=_Evt)
    eng = _Engine(model, ops=timed)
    # the rotary tables are keyed by the VALUES of the frequency buffers: meta tensors have none
    eng._rotary_table = lambda tb, T: sob.rotary_table(eng.w[tb + ".attn1.rotary_emb.freqs"], T)
    x = torch.empty((2, 4, 12, 32, 32), device="meta")
    ctx = torch.empty((2, 12, 77, 768), device="meta")
    t = torch.empty((2,), dtype=torch.long, device="meta")
    eng._kv_key = None
    eng._context = lambda c: (torch.empty((2 * 12 * 77, 768), dtype=torch.bfloat16, device="meta"), 77)
    eng.run(x, t, ctx, 0)
    timed.reset()                # first evaluation: the 16 cross-attention K|V projections ran (once per prompt, not per step)
    eng.run(x, t, ctx, 0)
    return eng, timed


def test_one_step_is_221_gemm_launches_and_5p2_executed_tflop(walk):
    eng, timed = walk
    gm = timed.summary()["gemm"]
    # 288 - 32 (ff.net.2 and proj_out of every transformer block are one GEMM) - 10 (the ten 320-channel feed-forwards: norm3,
    # ff.net.0 and that GEMM are one launch, accounted with the MACs of its two GEMMs: the step's FLOPs do not move)
    # - 10 (GroupNorm -> proj_in -> norm1 -> q|k|v of the ten 320-channel transformers: one launch each, round 6) - 5 (attn1.to_out +
    # residual -> norm2 -> attn2.to_q of the five 320-channel text blocks) - 10 (the last to_out + residual of the ten 320-channel
    # blocks: the prologue of the feed-forward launch): the MACs stay
    assert gm["launches"] == 221, gm["launches"]
    assert abs(gm["flops"] / 5.20e12 - 1.0) < 0.01, gm["flops"] / 1e12
    assert "layernorm" not in timed.summary()       # every LayerNorm of the step runs inside a GEMM-class launch
    rows = {r["name"]: r for r in timed.family_rows(1, 2500.0, 8000.0)}
    assert "ff.net.0 GEGLU L0" not in rows and "ff.net.2 | proj_out +res L0" not in rows
    assert rows["fused feed-forward (to_out +res, norm3, ff.net.0 GEGLU, ff.net.2 | proj_out +res) L0"]["launches"] == 10
    assert eng.ln_folded == 55 and eng.rowchains == 15      # (70 folds before the chains took norm1 / norm2 of the 320-channel level)
    # attention: 5 spatial + 5 cross + 5 temporal blocks at each of the three attention levels + the mid block
    assert timed.summary()["attention"]["launches"] == 48


def test_rows_name_the_roof_their_arithmetic_intensity_selects(walk):
    _, timed = walk
    rows = {r["name"]: r for r in timed.family_rows(1, 2500.0, 8000.0)}
    # what is left of the family at the 320-channel level: the three conv_shortcut launches of the up path (M 24576, K 960 / 640 ->
    # N 320: 222 flop/B on average; the level's 320 x 320 projections all run inside row-owning launches): below the 312 flop/B ridge
    p0 = rows["projections / 1x1 L0"]
    assert p0["launches"] == 3 and p0["bound"] == "hbm" and 180 < p0["ai"] < 312 and p0["frac"] == p0["frac_hbm"]
    # a 1280 -> 1280 conv at the 8x8 level: 45 GFLOP over ~37 MB
    c8 = rows["conv3x3 8x8"]
    assert c8["bound"] == "mfma" and c8["ai"] > 312 and c8["frac"] == c8["frac_mfma"]
    for r in rows.values():
        assert r["bound"] in ("mfma", "hbm") and r["frac"] >= 0
        if "ai" in r:
            assert (r["ai"] >= 312.5) == (r["bound"] == "mfma")


def test_without_the_fused_feed_forward_the_step_is_241_launches_and_ten_layernorms(monkeypatch):
    """SEER_FF_FUSED=0: the ten level-0 GEGLU projections refuse the LayerNorm fold (the library keeps them on the weight-stationary
    kernel): LayerNorm launches + plain GEMMs, recorded ONCE each"""
    monkeypatch.setenv("SEER_FF_FUSED", "0")
    eng, timed = _walk_config2()
    assert timed.summary()["gemm"]["launches"] == 241          # 231 + the ten feed-forwards as two launches each
    assert abs(timed.summary()["gemm"]["flops"] / 5.20e12 - 1.0) < 0.01
    assert timed.summary()["layernorm"]["launches"] == 10
    rows = {r["name"]: r for r in timed.family_rows(1, 2500.0, 8000.0)}
    assert rows["ff.net.0 GEGLU L0"]["launches"] == 10 and rows["ff.net.2 | proj_out +res L0"]["launches"] == 10


def test_a_call_that_launches_nothing_leaves_no_record():
    timed = TimedOps(base=sob, event=_Evt)
    a = torch.empty((24576, 320), dtype=torch.bfloat16, device="meta")
    w = torch.empty((2560, 320), dtype=torch.bfloat16, device="meta")
    assert timed.gemm(a, w, geglu=True, ln=(sob.RowStats(24576), None, 1e-5)) is None
    assert not timed.records
    assert timed.gemm(a, w, geglu=True) is not None
    assert len(timed.records["gemm"]) == 1
