"""Per-kernel-class timing for the roofline line of bench.py: a drop-in proxy for `seervideoldm_amd.ops` that brackets
every launch with HIP events on the launch stream (torch's current stream IS the stream the C ABI launches on) and
accounts the ALGORITHMIC work of the call (GEMM 2MNK, conv 2*M*Co*9*Ci, attention 4*BH*Sq*Sk*d dense,
norms 2 passes * 2 bytes)."""
from __future__ import annotations

from collections import defaultdict

import torch

from . import ops as _ops


class TimedOps:
    """FLOPs are the EXECUTED ones: the conv behind a nearest-2x upsample counts its four 2x2 phase convs (16 tap products per
    source pixel), not the 36 of the reference's 9-tap conv over the upsampled grid.  A call that launches nothing (ops.gemm(...,
    ln=...) returns None when the launch cannot fold its LayerNorm; the caller then runs layernorm + the plain GEMM, both
    recorded) leaves no record."""

    def __init__(self, base=_ops, event=None):
        self._base = base
        self._event = event or (lambda: torch.cuda.Event(enable_timing=True))      # tests on CPU pass a stand-in
        self.records = defaultdict(list)      # class -> [(start_evt, end_evt, flops, bytes, tag)]

    def _timed(self, cls, flops, nbytes, fn, *a, _tag=None, **k):
        s, e = self._event(), self._event()
        s.record()
        out = fn(*a, **k)
        if out is None:                       # nothing was launched: no work, no time
            return None
        e.record()
        self.records[cls].append((s, e, flops, nbytes, _tag))
        return out

    # --- MFMA GEMM family (one kernel template: seer_gemm_kernel) ---------------------------------------------
    def gemm(self, a, w, **k):
        M, N, K = a.shape[0], w.shape[0], w.shape[1]
        n_out = N // 2 if k.get("geglu") else N
        nbytes = 2 * (M * K + N * K + M * n_out) + (2 * M * n_out if k.get("residual") is not None else 0)
        tag = f"gemm M{M} N{N} K{K}" + (" geglu" if k.get("geglu") else "") + (" +res" if k.get("residual") is not None else "")
        return self._timed("gemm", 2.0 * M * N * K, nbytes, self._base.gemm, a, w, _tag=tag, **k)

    def ff_fused(self, h, x, gamma, beta, w1f, b1, wcf, bcat, **k):
        # norm3 -> ff.net.0 (GEGLU) -> [proj_out | proj_out ff.net.2] + residuals as one launch: the MACs of the two GEMMs it holds
        M, Cc = h.shape
        pre = k.get("pre") is not None          # the block's last to_out + residual as the launch's prologue: its MACs and its operand
        flops = 2.0 * M * Cc * (8 * Cc) + 2.0 * M * (5 * Cc) * Cc + (2.0 * M * Cc * Cc if pre else 0.0)
        nbytes = 2 * ((4 if pre else 3) * M * Cc + w1f.numel() + wcf.numel() + (Cc * Cc if pre else 0))
        return self._timed("gemm", flops, nbytes, self._base.ff_fused, h, x, gamma, beta, w1f, b1, wcf, bcat,
                           _tag=f"ff_fused M{M} C{Cc}" + (" pre" if pre else ""), **k)

    def rowchain(self, inp, w1f, **k):
        # one or more 320 x 320 products over the same rows (GroupNorm / LayerNorm in between) as one launch: their MACs
        M, Cc = inp.shape
        n2 = 0 if k.get("w2f") is None else k["w2f"].numel() // (Cc * Cc)
        flops = 2.0 * M * Cc * Cc * (1 + n2)
        nbytes = 2 * (M * Cc * (2 + n2 + (1 if k.get("res") is not None else 0)) + (1 + n2) * Cc * Cc)
        return self._timed("gemm", flops, nbytes, self._base.rowchain, inp, w1f, _tag=f"rowchain M{M} n{1 + n2}" + (" +res" if k.get("res") is not None else ""), **k)

    def gemm_batched(self, a, w, **k):
        Bt, M, K = a.shape
        N = w.shape[-2]
        return self._timed("gemm", 2.0 * Bt * M * N * K, 2 * Bt * (M * K + N * K + M * N), self._base.gemm_batched, a, w, **k)

    def conv3x3(self, x, w, n_img, Hin, Win, **k):
        stride, up = k.get("stride", 1), k.get("upsample", False)
        Hs, Ws = (2 * Hin, 2 * Win) if up else (Hin, Win)
        Ho, Wo = (Hs - 1) // stride + 1, (Ws - 1) // stride + 1
        M, Co, K = n_img * Ho * Wo, w.shape[0], w.shape[1]
        nbytes = 2 * (x.numel() + w.numel() + M * Co)
        tag = f"conv n{n_img} {Hin}x{Win} {x.shape[1]}->{Co} s{stride} up{int(bool(up))}"
        return self._timed("gemm", 2.0 * M * Co * K, nbytes, self._base.conv3x3, x, w, n_img, Hin, Win, _tag=tag, **k)

    def conv_up2x(self, x, w4, n_img, Hin, Win, **k):
        # EXECUTED MACs of the four 2x2 phase convs (16 tap-products per source pixel and channel pair), not the 36 of the
        # 9-tap conv over the upsampled grid that the reference runs: the roofline fraction prices what the MFMAs did
        Co, K4 = w4.shape[1], w4.shape[2]
        M = n_img * Hin * Win
        nbytes = 2 * (x.numel() + w4.numel() + 4 * M * Co)
        tag = f"conv_up2x n{n_img} {Hin}x{Win} {x.shape[1]}->{Co} (4 phases)"
        return self._timed("gemm", 2.0 * 4 * M * Co * K4, nbytes, self._base.conv_up2x, x, w4, n_img, Hin, Win, _tag=tag, **k)

    def attention(self, q, k_, v, out, **k):
        nb = k["batch"]
        if k.get("window") is not None:
            ws, F, H, W = k["window"]
            nb *= (H // ws) * (W // ws)
        flops = 4.0 * nb * k["heads"] * k["Sq"] * k["Sk"] * k["head_dim"]
        nbytes = 2 * k["heads"] * k["head_dim"] * nb * (2 * k["Sq"] + 2 * k["Sk"])
        tag = f"attn b{nb} Sq{k['Sq']} Sk{k['Sk']} d{k['head_dim']} causal{int(bool(k.get('causal')))}"
        return self._timed("attention", flops, nbytes, self._base.attention, q, k_, v, out, _tag=tag, **k)

    def _bw(self, name, passes_bytes):
        def f(*a, **k):
            return self._timed(name, 0.0, passes_bytes(*a, **k), getattr(self._base, name), *a, **k)
        return f

    def __getattr__(self, name):
        base = getattr(self._base, name)
        if name in ("groupnorm_stats",):
            return self._bw(name, lambda x1, x2, *a, **k: 2 * (x1.numel() + (0 if x2 is None else x2.numel())))
        if name == "groupnorm_stats_from_colsums":
            # the column-sum form of the same statistics: timed under the same class, bytes = the partials it reads
            def f(cs1, cs2, *a, **k):
                nb = 4 * (cs1.buf.numel() + (0 if cs2 is None else cs2.buf.numel()))
                return self._timed("groupnorm_stats", 0.0, nb, base, cs1, cs2, *a, **k)
            return f
        if name in ("groupnorm_apply_from_colsums", "groupnorm_apply_fx"):
            # statistics + apply in one launch: timed as the apply it replaces (bytes: one read + one write of the activations)
            def f(x1, x2, *a, **k):
                nb = 4 * (x1.numel() + (0 if x2 is None else x2.numel()))
                return self._timed("groupnorm_apply", 0.0, nb, base, x1, x2, *a, **k)
            return f
        if name in ("groupnorm_apply",):
            return self._bw(name, lambda x1, x2, *a, **k: 4 * (x1.numel() + (0 if x2 is None else x2.numel())))
        if name in ("layernorm",):
            return self._bw(name, lambda x, *a, **k: 4 * x.numel())
        if name in ("rotary_inplace",):
            return self._bw(name, lambda x, *a, **k: 0)
        if callable(base) and name in ("conv_in", "conv_out", "linear_smallm", "timestep_embedding", "cast_bf16"):
            return self._bw(name, lambda *a, **k: 0)
        return base

    def summary(self):
        """class -> dict(launches, ms, flops, bytes, tflops, gbps); call after torch.cuda.synchronize()."""
        out = {}
        for cls, recs in self.records.items():
            ms = sum(r[0].elapsed_time(r[1]) for r in recs)
            fl = sum(r[2] for r in recs)
            by = sum(r[3] for r in recs)
            out[cls] = dict(launches=len(recs), ms=ms, flops=fl, bytes=by,
                            tflops=fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                            gbps=by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0)
        return out

    def shape_summary(self):
        """per-shape totals: [(tag, calls, total_ms, tflops)] sorted by time (call after synchronize)."""
        agg = {}
        for cls, recs in self.records.items():
            for r in recs:
                tag = r[4] or cls
                a = agg.setdefault(tag, [0, 0.0, 0.0])
                a[0] += 1
                a[1] += r[0].elapsed_time(r[1])
                a[2] += r[2]
        rows = [(t, n, ms, (fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0)) for t, (n, ms, fl) in agg.items()]
        return sorted(rows, key=lambda x: -x[2])

    @staticmethod
    def _family(cls: str, tag, levels=None) -> str:
        """the ~15 kernel families of a denoising step (bench.py roofline.rows); `levels` = rows -> level name of the shape at hand
        (default: BASELINE config 2)"""
        if tag is None:
            return cls
        t = tag.split()
        if t[0] == "gemm":
            M, N, K = int(t[1][1:]), int(t[2][1:]), int(t[3][1:])
            lvl = (levels or {24576: "L0", 6144: "L1", 1536: "L2", 384: "L3/mid"}).get(M, f"M{M}")
            if "geglu" in t:
                return f"ff.net.0 GEGLU {lvl}"
            if "+res" in t and K == 4 * N:
                return f"ff.net.2 +res {lvl}"
            if "+res" in t and K == 5 * N:
                return f"ff.net.2 | proj_out +res {lvl}"
            if N == 3 * K:
                return f"q|k|v projection {lvl}"
            return f"projections / 1x1 {lvl}"
        if t[0] == "rowchain":
            lvl = (levels or {24576: "L0"}).get(int(t[1][1:]), t[1])
            return ("attn1.to_out + residual -> norm2 -> attn2.to_q, one launch " if "+res" in t else
                    "GroupNorm -> proj_in -> norm1 -> q|k|v, one launch ") + lvl
        if t[0] == "ff_fused":
            lvl = (levels or {24576: "L0"}).get(int(t[1][1:]), t[1])
            return f"fused feed-forward ({'to_out +res, ' if 'pre' in t else ''}norm3, ff.net.0 GEGLU, ff.net.2 | proj_out +res) {lvl}"
        if t[0] == "conv":
            return f"conv3x3 {t[2]}"
        if t[0] == "conv_up2x":
            return "conv3x3 behind nearest-2x (4 phase convs)"
        if t[0] == "attn":
            sq, sk = int(t[2][2:]), int(t[3][2:])
            kind = "temporal window" if t[5] == "causal1" else ("text cross" if sk == 77 else "spatial self")
            return f"attention {kind} {t[4]} Sq{sq}"
        return cls

    def family_rows(self, reps: int, mfma_peak_tflops: float, hbm_peak_gbs: float):
        """[{name, launches (per step), us (average per launch), ms (per step), ai, bound, frac, frac_mfma, frac_hbm}] sorted by
        time: the spread the class average of the roofline line hides.  `bound` is the roof the family's arithmetic intensity
        (algorithmic FLOPs / algorithmic bytes) selects: below the ridge (mfma peak / hbm peak = 312 flop/B) the HBM roof is the
        lower one and `frac` = algorithmic bytes/s over the HBM peak, above it `frac` = algorithmic FLOP/s over the dense bf16
        MFMA peak; both fractions are given for the MFMA kernels."""
        agg = {}
        for cls, recs in self.records.items():
            for r in recs:
                a = agg.setdefault(self._family(cls, r[4]), [0, 0.0, 0.0, 0.0])
                a[0] += 1
                a[1] += r[0].elapsed_time(r[1])
                a[2] += r[2]
                a[3] += r[3]
        ridge = mfma_peak_tflops * 1e12 / (hbm_peak_gbs * 1e9)
        rows = []
        for name, (n, ms, fl, by) in agg.items():
            if ms <= 0:
                continue
            f_mfma = fl / (ms * 1e-3) / 1e12 / mfma_peak_tflops
            f_hbm = by / (ms * 1e-3) / 1e9 / hbm_peak_gbs
            row = dict(name=name, launches=n // reps, us=round(ms / n * 1e3, 2), ms=round(ms / reps, 3))
            if fl > 0 and by > 0:
                ai = fl / by
                row.update(ai=round(ai, 1), bound="mfma" if ai >= ridge else "hbm",
                           frac=round(f_mfma if ai >= ridge else f_hbm, 4), frac_mfma=round(f_mfma, 4), frac_hbm=round(f_hbm, 4))
            elif fl > 0:
                row.update(bound="mfma", frac=round(f_mfma, 4))
            else:
                row.update(bound="hbm", frac=round(f_hbm, 4))
            rows.append(row)
        return sorted(rows, key=lambda r: -r["ms"])

    def reset(self):
        self.records.clear()
