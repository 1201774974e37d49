"""Per-kernel parity of the training entry points (include/seer_hip.h, "training step") on a real MI355X: every backward
kernel against torch fp32 autograd of the operator it differentiates, on seeded bf16-rounded inputs.

Gradients are compared by relative L2 error per tensor (bf16 storage of P / dS and of the outputs gives ~1e-2) plus a loose
element-wise bound.
"""
import pytest
import torch
import torch.nn.functional as Fn

pytestmark = pytest.mark.gpu

bf16 = torch.bfloat16


def _rand(shape, dev, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


def _rel(got, ref, tol, what):
    got, ref = got.float(), ref.float()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite"
    err = (got - ref).norm() / (ref.norm() + 1e-12)
    assert err < tol, f"{what}: rel L2 {err.item():.4g} >= {tol}"
    return err.item()


def _attn_autograd(q, k, v, do, causal, off=0):
    """q [B,H,Sq,d] ... fp32 leaf tensors -> (o, dq, dk, dv)"""
    q, k, v = [t.detach().float().requires_grad_(True) for t in (q, k, v)]
    d = q.shape[-1]
    s = torch.einsum("bhqd,bhkd->bhqk", q, k) * d ** -0.5
    if causal:
        i = torch.arange(s.shape[-2], device=s.device)[:, None] + off
        j = torch.arange(s.shape[-1], device=s.device)[None, :]
        s = s.masked_fill(~(j <= i), float("-inf"))
    o = torch.einsum("bhqk,bhkd->bhqd", s.softmax(-1), v)
    o.backward(do.float())
    return o.detach(), q.grad, k.grad, v.grad


@pytest.mark.parametrize("d,Sq,Sk,causal", [
    (40, 128, 128, False), (40, 1024, 1024, False), (80, 256, 256, False), (160, 64, 64, False), (160, 16, 16, False),
    (40, 1024, 77, False), (80, 256, 77, False), (160, 64, 77, False),
    (40, 768, 768, True), (80, 192, 192, True), (160, 192, 192, True), (40, 100, 100, True), (80, 272, 272, True),
    (96, 77, 77, False), (96, 924, 77, False), (96, 12, 12, True), (96, 200, 200, True),
])
def test_attention_bwd(device, d, Sq, Sk, causal):
    from seervideoldm_amd import ops, train_ops
    B, Hh = 2, 8
    C = Hh * d
    q = _rand((B, Sq, Hh, d), device, 1).to(bf16)
    k = _rand((B, Sk, Hh, d), device, 2).to(bf16)
    v = _rand((B, Sk, Hh, d), device, 3).to(bf16)
    do = _rand((B, Sq, Hh, d), device, 4).to(bf16)
    out = torch.zeros((B * Sq, C), device=device, dtype=bf16)
    lse = train_ops.attn_lse_buffer(B, Hh, Sq, device)
    kw = dict(batch=B, heads=Hh, head_dim=d, Sq=Sq, Sk=Sk, causal=causal)
    q2, k2, v2, do2 = q.reshape(B * Sq, C), k.reshape(B * Sk, C), v.reshape(B * Sk, C), do.reshape(B * Sq, C)
    ops.attention(q2, k2, v2, out, lse=lse, **kw)
    dq = torch.zeros_like(q2); dk = torch.zeros_like(k2); dv = torch.zeros_like(v2)
    train_ops.attention_bwd(q2, k2, v2, out, lse, do2, dq, dk, dv, **kw)
    o_ref, dq_ref, dk_ref, dv_ref = _attn_autograd(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3),
                                                   do.permute(0, 2, 1, 3), causal)
    # lse: log2-domain log-sum-exp of the scaled scores
    s = torch.einsum("bhqd,bhkd->bhqk", q.permute(0, 2, 1, 3).float(), k.permute(0, 2, 1, 3).float()) * d ** -0.5
    if causal:
        m = torch.ones(s.shape[-2:], dtype=torch.bool, device=device).tril()
        s = s.masked_fill(~m, float("-inf"))
    lse_ref = torch.logsumexp(s, -1) * 1.4426950408889634
    assert (lse.reshape(B, Hh, Sq) - lse_ref).abs().max() < 2e-2
    back = lambda t, S: t.permute(0, 2, 1, 3).reshape(B * S, C)
    _rel(dq, back(dq_ref, Sq), 2e-2, f"dq d{d} {Sq}x{Sk} causal={causal}")
    _rel(dk, back(dk_ref, Sk), 2e-2, f"dk d{d} {Sq}x{Sk} causal={causal}")
    _rel(dv, back(dv_ref, Sk), 2e-2, f"dv d{d} {Sq}x{Sk} causal={causal}")


def test_attention_bwd_fused_qkv_window(device):
    """temporal window form on a fused [tokens, 3C] buffer, gradients into one [tokens, 3C] buffer"""
    from seervideoldm_amd import ops, train_ops
    for d, Fr, H, W, ws in [(40, 4, 16, 16, 8), (80, 6, 8, 8, 4), (160, 3, 8, 8, 4)]:
        B, Hh = 2, 8
        C = Hh * d
        T = Fr * H * W
        qkv = _rand((B * T, 3 * C), device, 11).to(bf16)
        do = _rand((B * T, C), device, 12).to(bf16)
        out = torch.zeros((B * T, C), device=device, dtype=bf16)
        kw = dict(batch=B, heads=Hh, head_dim=d, Sq=Fr * ws * ws, Sk=Fr * ws * ws, causal=True, window=(ws, Fr, H, W))
        lse = train_ops.attn_lse_buffer(B, Hh, Fr * ws * ws, device, window=(ws, Fr, H, W))
        ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, lse=lse, **kw)
        dqkv = torch.zeros_like(qkv)
        train_ops.attention_bwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, lse, do, dqkv[:, :C], dqkv[:, C:2 * C],
                                dqkv[:, 2 * C:], **kw)

        def part(t):   # [B*T, C] -> [nW*B, heads, F*ws*ws, d]
            t = t.float().reshape(B, Fr, H // ws, ws, W // ws, ws, Hh, d)
            return t.permute(2, 4, 0, 6, 1, 3, 5, 7).reshape(-1, Hh, Fr * ws * ws, d)

        def unpart(o):
            return o.reshape(H // ws, W // ws, B, Hh, Fr, ws, ws, d).permute(2, 4, 0, 5, 1, 6, 3, 7).reshape(B * T, C)
        q, k, v = [part(t) for t in qkv.split(C, dim=1)]
        _, dq, dk, dv = _attn_autograd(q, k, v, part(do), True)
        ref = torch.cat([unpart(dq), unpart(dk), unpart(dv)], 1)
        _rel(dqkv, ref, 2e-2, f"window attention bwd d{d}")


def test_attention_bwd_strided_sequences(device):
    """FSTextTransformer's attention over frames: rows ordered (frame, token), sequences read through strides"""
    from seervideoldm_amd import ops, train_ops
    Fr, L, Hh, d = 12, 77, 8, 96
    C = Hh * d
    qkv = _rand((Fr * L, 3 * C), device, 5).to(bf16)
    do = _rand((Fr * L, C), device, 6).to(bf16)
    out = torch.zeros((Fr * L, C), device=device, dtype=bf16)
    kw = dict(batch=L, heads=Hh, head_dim=d, Sq=Fr, Sk=Fr, causal=True, seq_stride_rows=L, batch_stride_rows=1)
    lse = train_ops.attn_lse_buffer(L, Hh, Fr, device)
    ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, lse=lse, **kw)
    dqkv = torch.zeros_like(qkv)
    train_ops.attention_bwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, lse, do, dqkv[:, :C], dqkv[:, C:2 * C],
                            dqkv[:, 2 * C:], **kw)
    q, k, v = [t.reshape(Fr, L, Hh, d).permute(1, 2, 0, 3) for t in qkv.split(C, dim=1)]     # [L, heads, F, d]
    _, dq, dk, dv = _attn_autograd(q, k, v, do.reshape(Fr, L, Hh, d).permute(1, 2, 0, 3), True)
    ref = torch.cat([t.permute(2, 0, 1, 3).reshape(Fr * L, C) for t in (dq, dk, dv)], 1)
    _rel(dqkv, ref, 2e-2, "strided attention bwd")


def test_attention_bwd_deterministic(device):
    from seervideoldm_amd import ops, train_ops
    B, Hh, d, S = 2, 8, 40, 384
    C = Hh * d
    qkv = _rand((B * S, 3 * C), device, 1).to(bf16)
    do = _rand((B * S, C), device, 2).to(bf16)
    out = torch.zeros((B * S, C), device=device, dtype=bf16)
    kw = dict(batch=B, heads=Hh, head_dim=d, Sq=S, Sk=S, causal=True)
    lse = train_ops.attn_lse_buffer(B, Hh, S, device)
    ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, lse=lse, **kw)
    res = []
    for _ in range(2):
        dqkv = torch.zeros_like(qkv)
        train_ops.attention_bwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, lse, do, dqkv[:, :C], dqkv[:, C:2 * C],
                                dqkv[:, 2 * C:], **kw)
        res.append(dqkv)
    assert torch.equal(res[0], res[1])


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,cols", [(64, 64), (100, 320), (924, 768), (12288, 320), (160, 1280), (77, 8)])
def test_transpose(device, rows, cols):
    from seervideoldm_amd import train_ops
    x = _rand((rows, cols + 8), device, 1).to(bf16)[:, :cols]
    y = train_ops.transpose(x)
    pad = (rows + 63) // 64 * 64
    assert y.shape == (cols, pad)
    assert torch.equal(y[:, :rows], x.t())
    assert (y[:, rows:] == 0).all()


def test_transpose_batched(device):
    """many matrices, one launch: ragged row counts, column counts that are not tile multiples, row-strided sources; a second
    run() picks up new values at the same addresses (the trainer's weights after an AdamW step)"""
    from seervideoldm_amd import train_ops
    shapes = [(64, 64), (100, 320), (924, 768), (1, 8), (2560, 320), (160, 1288), (77, 72), (3072, 768)]
    flat = _rand((sum(r * (c + 8) for r, c in shapes),), device, 3).to(bf16)
    xs, off = [], 0
    for r, c in shapes:
        xs.append(flat[off:off + r * (c + 8)].view(r, c + 8)[:, :c])
        off += r * (c + 8)
    plan = train_ops.TransposePlan(xs)
    for rnd in range(2):
        plan.run()
        for x, y in zip(xs, plan.outputs):
            rows, cols = x.shape
            assert y.shape == (cols, (rows + 63) // 64 * 64)
            assert torch.equal(y[:, :rows], x.t()), (rnd, rows, cols)
            assert (y[:, rows:] == 0).all()
        flat.mul_(-0.5)


@pytest.mark.parametrize("rows,cols", [(64, 320), (924, 768), (12288, 960), (30, 2560), (1, 8)])
def test_colsum(device, rows, cols):
    from seervideoldm_amd import train_ops
    x = _rand((rows, cols), device, 1).to(bf16)
    got = train_ops.colsum(x)
    _rel(got, x.float().sum(0), 1e-5, "colsum")
    assert torch.equal(got, train_ops.colsum(x))


@pytest.mark.parametrize("rows,C,with_w,with_res", [(64, 320, True, True), (924, 768, True, False), (3000, 640, False, True),
                                                    (160, 1280, True, True), (7, 320, True, False)])
def test_layernorm_bwd(device, rows, C, with_w, with_res):
    from seervideoldm_amd import train_ops
    x = _rand((rows, C), device, 1, 2.0).to(bf16)
    dy = _rand((rows, C), device, 2).to(bf16)
    gamma = 1 + 0.2 * _rand((C,), device, 3)
    beta = 0.1 * _rand((C,), device, 4)
    dres = _rand((rows, C), device, 5).to(bf16) if with_res else None
    dg = torch.zeros(C, device=device) if with_w else None
    db = torch.zeros(C, device=device) if with_w else None
    dx = train_ops.layernorm_bwd(x, dy, gamma, dres=dres, dgamma=dg, dbeta=db)
    xr = x.float().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    Fn.layer_norm(xr, (C,), gr, br, 1e-5).backward(dy.float())
    ref = xr.grad + (dres.float() if with_res else 0)
    _rel(dx, ref, 1e-2, "ln dx")
    if with_w:
        _rel(dg, gr.grad, 1e-3, "ln dgamma")
        _rel(db, br.grad, 1e-3, "ln dbeta")


def test_layernorm_bwd_deferred_finals_are_the_same_bits(device):
    """d gamma / d beta left as partial slabs and added for MANY norms in one launch (seer_layernorm_bwd_partials +
    seer_colfinal_grouped: the end of a backward walk) == one final launch per norm; more norms than one launch's table holds"""
    from seervideoldm_amd import train_ops
    shapes = [(924, 1024), (12288, 320), (3072, 640), (768, 1280), (100, 320), (8, 64), (9000, 1536)] + [(64 + 8 * i, 320) for i in range(70)]
    queue, want, got = [], [], []
    for i, (rows, C) in enumerate(shapes):
        x = _rand((rows, C), device, 3 * i + 1, 2.0).to(bf16)
        dy = _rand((rows, C), device, 3 * i + 2).to(bf16)
        gamma = 1 + 0.2 * _rand((C,), device, 3 * i + 3)
        dg, db = torch.zeros(C, device=device), torch.zeros(C, device=device)
        dx = train_ops.layernorm_bwd(x, dy, gamma, dgamma=dg, dbeta=db)
        dg2, db2 = torch.full((C,), float("nan"), device=device), torch.full((C,), float("nan"), device=device)
        dx2 = train_ops.layernorm_bwd(x, dy, gamma, dgamma=dg2, dbeta=db2, defer=queue)
        assert torch.equal(dx, dx2)
        want.append((dg, db))
        got.append((dg2, db2))
    assert len(queue) == len(shapes) and torch.isnan(got[0][0]).all()
    # one-value slabs (NV = 1: column sums) ride in the same launch
    slabs = _rand((37, 1, 200), device, 999)
    col = torch.full((200,), float("nan"), device=device)
    queue.append((slabs.reshape(-1), 37, 1, 200, col, None))
    train_ops.colfinal_grouped(queue)
    _rel(col, slabs.sum((0, 1)), 1e-6, "colfinal NV=1")
    queue.pop()
    for (dg, db), (dg2, db2), sh in zip(want, got, shapes):
        assert torch.equal(dg, dg2) and torch.equal(db, db2), sh


@pytest.mark.parametrize("B,rows,C1,C2,silu,with_w", [(1, 256, 320, 0, True, False), (2, 192, 640, 0, False, True),
                                                      (1, 1024, 320, 320, True, False), (2, 48, 1280, 1280, True, True),
                                                      (1, 100, 640, 320, True, True), (1, 64, 128, 0, False, True)])
def test_groupnorm_bwd(device, B, rows, C1, C2, silu, with_w):
    from seervideoldm_amd import ops, train_ops
    G = 32
    C = C1 + C2
    x1 = _rand((B * rows, C1), device, 1, 1.5).to(bf16)
    x2 = _rand((B * rows, C2), device, 2, 0.7).to(bf16) if C2 else None
    dy = _rand((B * rows, C), device, 3).to(bf16)
    gamma = 1 + 0.2 * _rand((C,), device, 4)
    beta = 0.1 * _rand((C,), device, 5)
    stats = torch.empty((B, G, 2), device=device)
    ops.groupnorm_stats(x1, x2, B, G, stats)
    count = rows * (C // G)
    dres1 = _rand((B * rows, C1), device, 6).to(bf16)
    dg = torch.zeros(C, device=device) if with_w else None
    db = torch.zeros(C, device=device) if with_w else None
    dx1, dx2 = train_ops.groupnorm_bwd(x1, x2, B, G, stats, count, 1e-5, gamma, beta, silu, dy, dres1=dres1, dgamma=dg, dbeta=db)
    xc = torch.cat([x1, x2], 1) if C2 else x1
    xr = xc.float().reshape(B, rows, C).permute(0, 2, 1).contiguous().requires_grad_(True)     # [B, C, rows]
    gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    y = Fn.group_norm(xr, G, gr, br, 1e-5)
    if silu:
        y = Fn.silu(y)
    y.backward(dy.float().reshape(B, rows, C).permute(0, 2, 1))
    ref = xr.grad.permute(0, 2, 1).reshape(B * rows, C)
    _rel(dx1, ref[:, :C1] + dres1.float(), 1e-2, "gn dx1")
    if C2:
        _rel(dx2, ref[:, C1:], 1e-2, "gn dx2")
    if with_w:
        _rel(dg, gr.grad, 2e-3, "gn dgamma")
        _rel(db, br.grad, 2e-3, "gn dbeta")


def test_geglu_fwd_bwd(device):
    from seervideoldm_amd import ops, train_ops
    from seervideoldm_amd.weights import geglu_row_order
    rows, inner = 200, 1280
    pre_ref = _rand((rows, 2 * inner), device, 1, 1.5).to(bf16)          # reference order: [values | gates]
    order = geglu_row_order(inner).to(device)
    pre = pre_ref[:, order].contiguous()                                 # interleaved (what the packed weight produces)
    dout = _rand((rows, inner), device, 2).to(bf16)
    out = train_ops.geglu_fwd(pre)
    pr = pre_ref.float().requires_grad_(True)
    val, gate = pr.chunk(2, dim=-1)
    y = val * Fn.gelu(gate)
    y.backward(dout.float())
    _rel(out, y.detach(), 1e-2, "geglu fwd")
    dpre = train_ops.geglu_bwd(pre, dout)
    _rel(dpre, pr.grad[:, order], 1e-2, "geglu bwd")


def test_add_pool_insert(device):
    from seervideoldm_amd import train_ops
    a = _rand((300, 648), device, 1).to(bf16)[:, :640]
    b = _rand((300, 640), device, 2).to(bf16)
    assert torch.equal(train_ops.add(a, b), (a.float() + b.float()).to(bf16))
    n, H, W, C = 3, 4, 6, 64
    du = _rand((n * 4 * H * W, C), device, 3).to(bf16)
    got = train_ops.sumpool2x(du, n, H, W)
    ref = du.float().reshape(n, H, 2, W, 2, C).sum((2, 4)).reshape(n * H * W, C)
    _rel(got, ref, 1e-2, "sumpool2x")
    d = _rand((n * H * W, C), device, 4).to(bf16)
    z = train_ops.zero_insert2x(d, n, H, W).reshape(n, 2 * H, 2 * W, C)
    assert torch.equal(z[:, ::2, ::2], d.reshape(n, H, W, C))
    assert (z[:, 1::2] == 0).all() and (z[:, :, 1::2] == 0).all()


def test_mse_and_conv_out_bwd(device):
    from seervideoldm_amd import train_ops
    B, Cc, Ft, cond, H, W, C0 = 2, 4, 5, 2, 8, 8, 320
    pred = _rand((B, Cc, Ft, H, W), device, 1)
    target = _rand((B, Cc, Ft - cond, H, W), device, 2)
    loss, dpred = train_ops.mse_loss_grad(pred, target, cond)
    pr = pred.clone().requires_grad_(True)
    ref = Fn.mse_loss(pr[:, :, cond:], target, reduction="none").mean([1, 2, 3, 4]).mean()
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item()))
    _rel(dpred, pr.grad, 1e-5, "mse grad")
    w = _rand((Cc, C0, 3, 3), device, 3, 0.05)
    x = torch.zeros((B * Ft, C0, H, W), device=device, requires_grad=True)
    y = Fn.conv2d(x, w, padding=1)                                       # [B*F, Cout, H, W]
    y.backward(dpred.permute(0, 2, 1, 3, 4).reshape(B * Ft, Cc, H, W))
    got = train_ops.conv_out_bwd(dpred, w.permute(0, 2, 3, 1).contiguous())
    ref = x.grad.permute(0, 2, 3, 1).reshape(B * Ft * H * W, C0)
    _rel(got, ref, 1e-2, "conv_out bwd")


def test_adamw_matches_torch(device):
    from seervideoldm_amd import train_ops
    n = 100003
    p0 = _rand((n,), device, 1)
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref_p], lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    p = p0.clone(); m = torch.zeros_like(p); v = torch.zeros_like(p)
    pb = torch.empty((n,), device=device, dtype=bf16)
    for step in range(1, 4):
        g = _rand((n,), device, 10 + step, 3.0)
        ref_p.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([ref_p], 0.3)
        opt.step()
        ss = train_ops.sumsq(g)
        assert abs(ss.item() - (g.double() ** 2).sum().item()) < 1e-4 * ss.item()
        train_ops.adamw_step(p, g, m, v, lr=1e-3, step=step, grad_sumsq=ss, max_norm=0.3, p_bf16=pb)
        assert (p - ref_p.detach()).abs().max() < 2e-6
        assert torch.equal(pb, p.to(bf16))


def test_linear_backward_through_gemm(device):
    """dX and dW of y = x W^T + b with the forward GEMM on transposed operands (the recipe in include/seer_hip.h)"""
    from seervideoldm_amd import ops, train_ops
    for M, N, K in [(924, 768, 768), (1536, 960, 320), (160, 1280, 1280)]:
        x = _rand((M, K), device, 1).to(bf16)
        w = _rand((N, K), device, 2, K ** -0.5).to(bf16)
        dy = _rand((M, N), device, 3).to(bf16)
        dx = ops.gemm(dy, train_ops.transpose(w))
        _rel(dx, dy.float() @ w.float(), 1e-2, "dX")
        dw = ops.gemm(train_ops.transpose(dy), train_ops.transpose(x), out_f32=True)
        _rel(dw, dy.float().t() @ x.float(), 1e-2, "dW")
        _rel(train_ops.colsum(dy), dy.float().sum(0), 1e-5, "dbias")


@pytest.mark.parametrize("stride,upsample", [(1, False), (2, False), (1, True)])
def test_conv3x3_input_grad(device, stride, upsample):
    """dX of the three 3x3 conv forms through the forward conv kernel with the flipped / transposed weight"""
    from seervideoldm_amd import ops, train_ops
    from seervideoldm_amd.weights import pack_conv3x3
    n, H, W, Ci, Co = 3, 8, 8, 64, 128
    x = _rand((n, Ci, H, W), device, 1).to(bf16)
    w = _rand((Co, Ci, 3, 3), device, 2, (9 * Ci) ** -0.5).to(bf16)
    xr = x.float().requires_grad_(True)
    xin = Fn.interpolate(xr, scale_factor=2.0, mode="nearest") if upsample else xr
    y = Fn.conv2d(xin, w.float(), padding=1, stride=stride)
    Ho, Wo = y.shape[-2:]
    dy = _rand((n * Ho * Wo, Co), device, 3).to(bf16)
    y.backward(dy.float().reshape(n, Ho, Wo, Co).permute(0, 3, 1, 2))
    ref = xr.grad.permute(0, 2, 3, 1).reshape(n * H * W, Ci)
    wt = pack_conv3x3(w.float().flip(2, 3).transpose(0, 1).contiguous()).to(bf16).contiguous()      # [Ci, 9*Co]
    if stride == 2:
        got = ops.conv3x3(train_ops.zero_insert2x(dy, n, Ho, Wo), wt, n, 2 * Ho, 2 * Wo)
    elif upsample:
        got = train_ops.sumpool2x(ops.conv3x3(dy, wt, n, Ho, Wo), n, H, W)
    else:
        got = ops.conv3x3(dy, wt, n, Ho, Wo)
    _rel(got, ref, 1e-2, f"conv dX stride={stride} upsample={upsample}")


def test_axpby(device):
    from seervideoldm_amd import train_ops
    y, x = _rand((100000,), device, 1), _rand((100000,), device, 2)
    ref = 0.5 * x
    train_ops.axpby(y, x, 0.5, 0.0)
    assert torch.equal(y, ref)
    train_ops.axpby(y, x, 0.25, 1.0)
    assert torch.allclose(y, ref + 0.25 * x, atol=1e-7)


@pytest.mark.parametrize("M,N,K", [(924, 768, 768), (12288, 960, 320), (1536, 320, 1280), (160, 1280, 1280), (100, 2560, 320),
                                   (3072, 640, 2560), (64, 8, 8), (1000, 136, 72)])
def test_gemm_tn(device, M, N, K):
    """dW = dY^T X with both operands in their token-major layout (row-strided views), against fp32 matmul"""
    from seervideoldm_amd import train_ops
    dy = _rand((M, N + 8), device, 1).to(bf16)[:, :N]
    x = _rand((M, K + 16), device, 2).to(bf16)[:, 8:8 + K]
    got = train_ops.gemm_tn(dy, x)
    _rel(got, dy.float().t() @ x.float(), 2e-3, f"gemm_tn {M}x{N}x{K}")
    assert torch.equal(got, train_ops.gemm_tn(dy, x))
    cs = torch.full((N,), float("nan"), device=device)
    got2 = train_ops.gemm_tn(dy, x, colsum=cs)                       # bias gradient from the same pass
    assert torch.equal(got2, got)
    _rel(cs, dy.float().sum(0), 1e-5, f"gemm_tn colsum {M}x{N}")


def test_gemm_tn_grouped(device):
    """the deferred weight gradients of a backward walk in one launch: every problem against the fp32 product, ragged shapes and
    strided views included, split and unsplit ones side by side, more problems than one launch's table holds; a problem's
    bits do not depend on its companions"""
    from seervideoldm_amd import train_ops
    shapes = [(924, 768, 768), (12288, 960, 320), (1536, 320, 1280), (160, 1280, 1280), (100, 2560, 320), (3072, 640, 2560),
              (64, 8, 8), (1000, 136, 72), (2049, 320, 320), (4096, 128, 136), (20000, 136, 128), (40000, 64, 72)]   # > 16 384 rows: K slices
    shapes = shapes + [(192 + 64 * i, 320, 320) for i in range(45)]              # 55 problems: two launches
    probs, refs = [], []
    for i, (M, N, K) in enumerate(shapes):
        dy = _rand((M, N + 8), device, 2 * i + 1).to(bf16)[:, :N]
        x = _rand((M, K + 16), device, 2 * i + 2).to(bf16)[:, 8:8 + K]
        out = torch.full((N, K), float("nan"), device=device)
        cs = torch.full((N,), float("nan"), device=device) if i % 2 == 0 else None
        probs.append((dy, x, out, cs))
        refs.append((dy.float().t() @ x.float(), dy.float().sum(0)))
    train_ops.gemm_tn_grouped(probs)
    for (dy, x, out, cs), (r, rc), sh in zip(probs, refs, shapes):
        _rel(out, r, 2e-3, f"gemm_tn_grouped {sh}")
        if cs is not None:
            _rel(cs, rc, 1e-5, f"gemm_tn_grouped colsum {sh}")
    # alone, in another order, beside other problems: the same bits
    for i in (1, 8, 0, 30, 10, 11):
        dy, x, out, cs = probs[i]
        o2 = torch.empty_like(out)
        c2 = torch.empty_like(cs) if cs is not None else None
        train_ops.gemm_tn_grouped([(dy, x, o2, c2)])
        assert torch.equal(o2, out) and (cs is None or torch.equal(c2, cs)), shapes[i]
    outs = [torch.empty_like(p[2]) for p in probs]
    train_ops.gemm_tn_grouped([(p[0], p[1], o, None) for p, o in zip(probs, outs)][::-1])
    assert all(torch.equal(o, p[2]) for p, o in zip(probs, outs))


def test_text_loss_grad(device):
    from seervideoldm_amd import train_ops
    b, Fr, L, C = 2, 5, 77, 192
    y = _rand((b * Fr * L, C), device, 1).to(bf16)
    t = _rand((b, L, C), device, 2)
    dy0 = (_rand((b * Fr * L, C), device, 3) * 3e-5).to(bf16)      # the UNet's d context has the magnitude of this gradient
    dy = dy0.clone()
    loss = train_ops.text_loss_grad(y, t, b, Fr, dy)
    yr = y.float().reshape(b, Fr, L, C).requires_grad_(True)
    ref = Fn.mse_loss(yr.mean(1), t, reduction="none").mean([1, 2]).mean()
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-5 * ref.item()
    _rel(dy.float() - dy0.float(), yr.grad.reshape(b * Fr * L, C), 2e-2, "text loss grad (bf16 accumulate into dy)")
