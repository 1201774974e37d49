"""Host-logic tests (CPU): the product's kernel schedule (`seervideoldm_amd.unet._Engine`: weight packing, fused q|k|v,
interleaved GEGLU, skip/concat wiring, window + rotary parameters, cond_frame handling) driven through a plain-torch
stand-in for the kernel library (tests/torch_ops_backend.py, same signatures and bf16 storage rounding) must reproduce
the oracle.  This is NOT a product path -- on a GPU box the same schedule runs on libseer_hip.so (tests/test_gpu_unet.py).
"""
import pytest
import torch

from oracle import seer_oracle as O
from seervideoldm_amd import SeerUNet, synth
from seervideoldm_amd.unet import _Engine
from tests import torch_ops_backend as tob

CFG_MINI = dict(block_out_channels=(320, 320, 320, 320), layers_per_block=1, cross_attention_dim=256, attention_head_dim=8)


def _randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


@pytest.fixture(scope="module")
def mini():
    sd = synth.synth_state_dict(synth.unet_param_shapes(CFG_MINI))
    m = SeerUNet(**CFG_MINI)
    m.load_state_dict(sd, strict=True)
    return sd, m


@pytest.mark.parametrize("B,Fr,H,cond_frame", [(1, 2, 16, 0), (2, 3, 8, 1)])
def test_engine_schedule_matches_oracle(mini, B, Fr, H, cond_frame):
    sd, m = mini
    eng = _Engine(m, ops=tob)
    x = _randn((B, 4, Fr, H, H), 1)
    ctx = _randn((B, Fr, 77, 256), 2)
    t = torch.tensor([501] * B)
    with torch.no_grad():
        got = eng.run(x, t, ctx, cond_frame)
        ref = O.unet_forward(sd, CFG_MINI, x, t, ctx, cond_frame=cond_frame)
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 3e-2, rel


def test_engine_folds_layernorms_into_the_consuming_gemm(mini):
    """host logic of the folded LayerNorm (ops.fold_layernorm: W' = gamma (.) W, its row sums, beta W^T + b, with the GEGLU row
    order; unet._Engine._ln_gemm): every norm1 -> q|k|v, norm2 -> to_q, norm3 -> ff.net.0 of attention.py:198-200, 231-246,
    275-277, 308-327 runs as one GEMM over the un-normalised rows and lands on the oracle like the layernorm form does"""
    sd, m = mini
    x, ctx, t = _randn((1, 4, 2, 16, 16), 1), _randn((1, 2, 77, 256), 2), torch.tensor([501])
    m.rowchain = False          # (the chains of ops.rowchain take norm1 / norm2 of 320-channel blocks: test_engine_runs_the_row_chains)
    try:
        eng = _Engine(m, ops=tob)
        plain = _Engine(m, ops=tob, fold_ln=False)
    finally:
        del m.rowchain
    n_ln = sum(1 for k in sd if ".transformer_blocks." in k and k.endswith((".norm1.weight", ".norm2.weight", ".norm3.weight")))
    with torch.no_grad():
        got = eng.run(x, t, ctx, 0)
        assert eng.ln_fold and eng.ln_folded == n_ln == len(eng.wln), (eng.ln_folded, n_ln, len(eng.wln))
        two = plain.run(x, t, ctx, 0)
        assert plain.ln_folded == 0
        ref = O.unet_forward(sd, CFG_MINI, x, t, ctx, cond_frame=0)
    e_fold, e_two = ((got - ref).norm() / ref.norm()).item(), ((two - ref).norm() / ref.norm()).item()
    assert e_fold < 3e-2 and e_fold < 1.5 * e_two + 1e-3, (e_fold, e_two)
    # cond_frame > 0: the temporal feed-forward runs on row subsets (no statistics attached): those norms keep the layernorm kernel
    with torch.no_grad():
        x3, c3 = _randn((1, 4, 3, 8, 8), 3), _randn((1, 3, 77, 256), 4)
        got3 = eng.run(x3, t, c3, 1)
        assert 0 < eng.ln_folded < n_ln
        ref3 = O.unet_forward(sd, CFG_MINI, x3, t, c3, cond_frame=1)
    assert ((got3 - ref3).norm() / ref3.norm()).item() < 3e-2


def test_fold_layernorm_identity():
    """ops.fold_layernorm in exact arithmetic: LN(x) W^T + b == rstd (x W'^T - mean wsum) + b' (fp64, W' unrounded apart from bf16)"""
    from seervideoldm_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn((37, 64), generator=g, dtype=torch.float64) * 3 + 2
    w = torch.randn((24, 64), generator=g) * 0.2
    gamma, beta, b = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1, torch.randn(24, generator=g)
    wp, wsum, bp = ops.fold_layernorm(w, gamma, beta, b)
    assert wp.dtype == torch.bfloat16 and wsum.dtype == torch.float32 and bp.dtype == torch.float32
    mean, var = x.mean(1, keepdim=True), x.var(1, unbiased=False, keepdim=True)
    rstd = (var + 1e-5).rsqrt()
    folded = rstd * (x @ wp.double().t() - mean * wsum.double()[None, :]) + bp.double()
    # against the same LayerNorm with the ROUNDED gamma (.) W: identical up to fp64 rounding -- the fold itself adds nothing
    ln = (x - mean) * rstd
    exact = ln @ wp.double().t() + bp.double()
    assert (folded - exact).abs().max().item() < 1e-9
    # ... and against nn.LayerNorm + nn.Linear in fp64: only the bf16 rounding of gamma (.) W separates them
    ref = (ln * gamma.double() + beta.double()) @ w.double().t() + b.double()
    assert (folded - ref).abs().max().item() < 3e-2 * ref.abs().max().item()


def test_engine_counts_groupnorms(mini):
    _, m = mini
    eng = _Engine(m, ops=tob)
    # lpb=1: down 4 resnets*2 + 3 levels * 2 transformers; mid 2*2 + 2; up 8 resnets*2 + 3 levels*2 blocks*2; + out
    assert eng.n_groupnorms() == 8 + 6 + 6 + 16 + 12 + 1
    full = synth.unet_param_shapes({})
    n = sum(1 for k in full if k.endswith(".weight") and (k.endswith("norm1.weight") or k.endswith("norm2.weight")
            or k.endswith(".norm.weight") or k == "conv_norm_out.weight") and "transformer_blocks" not in k)
    assert n == 77          # SURVEY finding 3: 77 cross-frame GroupNorms per forward


def test_engine_return_attn_matches_oracle(mini):
    """`return_attn=True`: 7 score tensors [b, heads, f, h, w, L] from the LAST text block of each attention-bearing container
    (unet_3d_condition.py:291-292,317-323,372-374), and the same epsilon as the plain forward"""
    sd, m = mini
    eng = _Engine(m, ops=tob)
    B, Fr, H = 1, 2, 16
    x, ctx, t = _randn((B, 4, Fr, H, H), 1), _randn((B, Fr, 77, 256), 2), torch.tensor([501])
    with torch.no_grad():
        got, attn = eng.run(x, t, ctx, 0, return_attn=True)
        plain = eng.run(x, t, ctx, 0)
        ref, ref_attn = O.unet_forward(sd, CFG_MINI, x, t, ctx, cond_frame=0, return_attn=True)
    assert torch.equal(got, plain) and len(attn) == len(ref_attn) == 7
    for a, r in zip(attn, ref_attn):
        assert a.shape == r.shape and a.dtype == torch.float32
        assert ((a - r).norm() / r.norm()).item() < 3e-2


@pytest.mark.parametrize("B,Fr,H,cond_frame", [(1, 2, 16, 0), (2, 3, 8, 1)])
def test_engine_runs_the_row_chains(mini, B, Fr, H, cond_frame):
    """host logic of the row-local chains (unet._Engine._rc_in, _chain_next; ops.rowchain): at 320 channels GroupNorm -> proj_in -> norm1
    -> q|k|v (rotary on the temporal block's q and k, the q prescale) and attn1.to_out + residual -> norm2 -> attn2.to_q are one call
    each -- the statistics come from the producers' accumulated sums (the producer of a chain's input accumulates whatever its size),
    the residual stream is updated in place -- and the schedule lands on the oracle like the separate launches do"""
    sd, m = mini
    eng = _Engine(m, ops=tob)
    assert eng.rowchain and any(k.endswith(".rc.proj_in") for k in eng.w)
    x, ctx, t = _randn((B, 4, Fr, H, H), 1), _randn((B, Fr, 77, 256), 2), torch.tensor([501] * B)
    with torch.no_grad():
        got = eng.run(x, t, ctx, cond_frame)
        # lpb = 1: 3 down + mid + 6 up attention sites, a text and a temporal block each; the text blocks run two chains.  At 8x8 the
        # two lowest levels hold fewer than 96 rows per batch element: those sites keep the separate launches
        sites = sum(1 for k in eng.w if k.endswith(".rc.proj_in"))
        assert sites == 20 and 0 < eng.rowchains <= 30, eng.rowchains
        m.rowchain = False
        try:
            sep = _Engine(m, ops=tob).run(x, t, ctx, cond_frame)
        finally:
            del m.rowchain
        ref = O.unet_forward(sd, CFG_MINI, x, t, ctx, cond_frame=cond_frame)
    e_chain, e_sep = ((got - ref).norm() / ref.norm()).item(), ((sep - ref).norm() / ref.norm()).item()
    assert e_chain < 3e-2 and e_chain < 1.5 * e_sep + 1e-3, (e_chain, e_sep)
