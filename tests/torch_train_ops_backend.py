"""TEST INFRASTRUCTURE ONLY -- a plain-torch (CPU) stand-in for `seervideoldm_amd.train_ops` with the SAME call signatures and
the same storage rounding (bf16 tensors in/out, fp32 math inside), by torch autograd of each operator.

It lets the CPU suite exercise the HOST logic of `seervideoldm_amd.trainer.SeerTrainer` (which activations are kept, the
reverse schedule, skip / residual gradient fan-in, packed parameter layouts, the optimizer bookkeeping) against autograd of
the oracle.  Never imported by the product package.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from tests import torch_ops_backend as tob

bf16 = torch.bfloat16


def attn_lse_buffer(batch, heads, Sq, device, window=None):
    nb = batch if window is None else batch * (window[2] // window[0]) * (window[3] // window[0])
    return torch.zeros((nb * heads, Sq))


def attention_bwd(q, k, v, out, lse, dout, dq, dk, dv, **kw):
    qf, kf, vf = [t.detach().float().requires_grad_(True) for t in (q, k, v)]
    heads, d = kw["heads"], kw["head_dim"]
    C = heads * d

    res = _attention_float(qf, kf, vf, **kw)          # differentiable fp32 twin of tob.attention
    res.backward(dout[:, :C].float())
    dq[:, :C].copy_(qf.grad[:, :C].to(bf16))
    dk[:, :C].copy_(kf.grad[:, :C].to(bf16))
    dv[:, :C].copy_(vf.grad[:, :C].to(bf16))


def _attention_float(q, k, v, *, batch, heads, head_dim, Sq, Sk, causal=False, scale=None, window=None, Fq=None,
                     causal_offset=0, seq_stride_rows=1, batch_stride_rows=None):
    """differentiable fp32 twin of tob.attention; returns [rows of q, C] fp32"""
    C = heads * head_dim
    scale = head_dim ** -0.5 if scale is None else scale
    if batch_stride_rows is not None:
        rq = (torch.arange(batch)[:, None] * batch_stride_rows + torch.arange(Sq)[None] * seq_stride_rows).reshape(-1)
        rk = (torch.arange(batch)[:, None] * batch_stride_rows + torch.arange(Sk)[None] * seq_stride_rows).reshape(-1)
        o = _attention_float(q[rq], k[rk], v[rk], batch=batch, heads=heads, head_dim=head_dim, Sq=Sq, Sk=Sk, causal=causal,
                             scale=scale, causal_offset=causal_offset)
        return torch.zeros((q.shape[0], C)).index_put((rq,), o)
    if window is None:
        qq = q[:, :C].reshape(batch, Sq, heads, head_dim).permute(0, 2, 1, 3)
        kk = k[:, :C].reshape(batch, Sk, heads, head_dim).permute(0, 2, 1, 3)
        vv = v[:, :C].reshape(batch, Sk, heads, head_dim).permute(0, 2, 1, 3)
    else:
        ws, Fr, H, W = window
        Fq = Fr if Fq is None else Fq
        iq, ik = tob._tok_index(batch, ws, Fq, H, W), tob._tok_index(batch, ws, Fr, H, W)
        gat = lambda t, idx, Ft: t[:, :C].reshape(batch, Ft * H * W, heads, head_dim)[:, idx]
        qq = gat(q, iq, Fq).permute(1, 0, 3, 2, 4).reshape(-1, heads, Sq, head_dim)
        kk = gat(k, ik, Fr).permute(1, 0, 3, 2, 4).reshape(-1, heads, Sk, head_dim)
        vv = gat(v, ik, Fr).permute(1, 0, 3, 2, 4).reshape(-1, heads, Sk, head_dim)
    s = torch.einsum("bhqd,bhkd->bhqk", qq, kk) * scale
    if causal:
        i = torch.arange(Sq)[:, None] + causal_offset
        j = torch.arange(Sk)[None, :]
        s = s.masked_fill(~(j <= i), float("-inf"))
    o = torch.einsum("bhqk,bhkd->bhqd", s.softmax(-1), vv)
    if window is None:
        return o.permute(0, 2, 1, 3).reshape(batch * Sq, C)
    nW = iq.shape[0]
    o = o.reshape(nW, batch, heads, Sq, head_dim).permute(1, 0, 3, 2, 4).reshape(batch, nW * Sq, C)
    res = torch.zeros(batch, Fq * H * W, C).index_put((torch.arange(batch)[:, None], iq.reshape(-1)[None, :]), o)
    return res.reshape(-1, C)


def gemm_tn(a, b, out=None, colsum=None):
    r = a.float().t() @ b.float()
    if colsum is not None:
        colsum.copy_(a.float().sum(0))
    if out is None:
        return r
    out.copy_(r)
    return out


def gemm_tn_grouped(problems):
    for a, b, out, colsum in problems:
        gemm_tn(a, b, out=out, colsum=colsum)


def transpose(x, pad_to=64):
    rows, cols = x.shape
    ld = (rows + pad_to - 1) // pad_to * pad_to
    y = torch.zeros((cols, ld), dtype=bf16)
    y[:, :rows] = x.t()
    return y


class TransposePlan:
    def __init__(self, xs):
        self.inputs = list(xs)
        self.outputs = [transpose(x) for x in xs]

    def run(self):
        for x, y in zip(self.inputs, self.outputs):
            y.copy_(transpose(x))


def colsum(x, out=None):
    s = x.float().sum(0)
    if out is None:
        return s
    out.copy_(s)
    return out


def colfinal_grouped(items):
    for ws, n, NV, Cc, o0, o1 in items:
        t = ws[:n * NV * Cc].reshape(n, NV, Cc).sum(0)
        if o0 is not None:
            o0.copy_(t[0])
        if o1 is not None and NV > 1:
            o1.copy_(t[1])


def layernorm_bwd(x, dy, gamma, *, eps=1e-5, dres=None, dx=None, dgamma=None, dbeta=None, defer=None):
    xr = x.float().requires_grad_(True)
    g = gamma.detach().clone().requires_grad_(True)
    b = torch.zeros_like(g).requires_grad_(True)
    F.layer_norm(xr, (x.shape[1],), g, b, eps).backward(dy.float())
    r = xr.grad + (dres.float() if dres is not None else 0)
    if dgamma is not None and defer is not None:         # one slab [1][2][C]: (d beta, d gamma)
        defer.append((torch.stack([b.grad, g.grad]).reshape(-1), 1, 2, x.shape[1], dbeta, dgamma))
    elif dgamma is not None:
        dgamma.copy_(g.grad)
        dbeta.copy_(b.grad)
    if dx is None:
        return r.to(bf16)
    dx.copy_(r.to(bf16))
    return dx


def groupnorm_bwd(x1, x2, batch, groups, stats, count, eps, gamma, beta, silu, dy, *, dres1=None, dres2=None, dgamma=None,
                  dbeta=None):
    xc = x1.float() if x2 is None else torch.cat([x1.float(), x2.float()], 1)
    rows, C = xc.shape[0] // batch, xc.shape[1]
    xr = xc.reshape(batch, rows, C).permute(0, 2, 1).contiguous().requires_grad_(True)
    g = gamma.detach().clone().requires_grad_(True)
    b = beta.detach().clone().requires_grad_(True)
    y = F.group_norm(xr, groups, g, b, eps)
    if silu:
        y = F.silu(y)
    y.backward(dy.float().reshape(batch, rows, C).permute(0, 2, 1))
    d = xr.grad.permute(0, 2, 1).reshape(batch * rows, C)
    C1 = x1.shape[1]
    d1 = d[:, :C1] + (dres1.float() if dres1 is not None else 0)
    d2 = None
    if x2 is not None:
        d2 = (d[:, C1:] + (dres2.float() if dres2 is not None else 0)).to(bf16)
    if dgamma is not None:
        dgamma.copy_(g.grad)
        dbeta.copy_(b.grad)
    return d1.to(bf16), d2


def geglu_fwd(pre):
    val, gate = tob._deinterleave_geglu(pre.float())
    return (val * F.gelu(gate)).to(bf16)


def geglu_bwd(pre, dout):
    p = pre.float().requires_grad_(True)
    val, gate = tob._deinterleave_geglu(p)
    (val * F.gelu(gate)).backward(dout.float())
    return p.grad.to(bf16)


def add(a, b, out=None):
    r = (a.float() + b.float()).to(bf16)
    if out is None:
        return r
    out.copy_(r)
    return out


def sumpool2x(du, n_img, H, W):
    C = du.shape[1]
    return du.float().reshape(n_img, H, 2, W, 2, C).sum((2, 4)).reshape(n_img * H * W, C).to(bf16)


def zero_insert2x(d, n_img, H, W):
    C = d.shape[1]
    z = torch.zeros((n_img, 2 * H, 2 * W, C), dtype=bf16)
    z[:, ::2, ::2] = d.reshape(n_img, H, W, C)
    return z.reshape(-1, C)


def mse_loss_grad(pred, target, cond_f):
    p = pred.detach().clone().requires_grad_(True)
    loss = F.mse_loss(p[:, :, cond_f:], target, reduction="none").mean([1, 2, 3, 4]).mean()
    loss.backward()
    return loss.detach().reshape(1), p.grad


def text_loss_grad(y, target, b, Fr, dy):
    yr = y.float().reshape(b, Fr, -1).requires_grad_(True)
    loss = F.mse_loss(yr.mean(1), target.reshape(b, -1), reduction="none").mean(1).mean()
    loss.backward()
    dy.copy_((dy.float() + yr.grad.reshape(dy.shape)).to(bf16))
    return loss.detach().reshape(1)


def conv_out_bwd(dpred, w_ohwc):
    B, Cout, Fr, H, W = dpred.shape
    C0 = w_ohwc.shape[-1]
    x = torch.zeros((B * Fr, C0, H, W), requires_grad=True)
    F.conv2d(x, w_ohwc.permute(0, 3, 1, 2), padding=1).backward(dpred.permute(0, 2, 1, 3, 4).reshape(B * Fr, Cout, H, W))
    return x.grad.permute(0, 2, 3, 1).reshape(B * Fr * H * W, C0).to(bf16)


def axpby(y, x, alpha, beta):
    y.copy_(alpha * x + (beta * y if beta != 0 else 0))          # x may alias y


def sumsq(g):
    return (g.double() ** 2).sum().float().reshape(1)


def adamw_step(p, g, m, v, *, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, step, grad_sumsq=None, max_norm=1.0,
               p_bf16=None):
    coef = 1.0
    if grad_sumsq is not None:
        coef = min(1.0, float(max_norm / (grad_sumsq.sqrt() + 1e-6)))
    b1, b2 = betas
    gi = g * coef
    p.mul_(1 - lr * weight_decay)
    m.mul_(b1).add_(gi, alpha=1 - b1)
    v.mul_(b2).addcmul_(gi, gi, value=1 - b2)
    denom = (v.sqrt() / (1 - b2 ** step) ** 0.5).add_(eps)
    p.addcdiv_(m, denom, value=-lr / (1 - b1 ** step))
    if p_bf16 is not None:
        p_bf16.copy_(p.to(bf16))
