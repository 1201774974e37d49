"""Weight repacking: reference checkpoint layouts -> the layouts libseer_hip.so consumes.

Reference layouts: nn.Linear [out, in]; conv [Co, Ci, kh, kw]; all fp32 in `pytorch_model.bin`.
Device layouts: bf16 [N][K] with K contiguous; conv3x3 K ordered (ky, kx, ci) to match the channels-last
implicit-GEMM gather; the GEGLU projection rows interleaved in groups of 16 (16 value rows, then their 16 gate
rows) so that a lane of the MFMA epilogue holds a value and its gate.
"""
from __future__ import annotations

import torch

bf16 = torch.bfloat16


def pack_conv3x3(w: torch.Tensor) -> torch.Tensor:
    """[Co, Ci, 3, 3] -> [Co, 9*Ci] with k = (ky*3 + kx)*Ci + ci."""
    Co, Ci, kh, kw = w.shape
    assert kh == 3 and kw == 3
    return w.permute(0, 2, 3, 1).reshape(Co, 9 * Ci).contiguous()


def pack_conv3x3_up_phases(w: torch.Tensor) -> torch.Tensor:
    """[Co, Ci, 3, 3] -> [4, Co, 4*Ci]: the conv behind a nearest-2x upsample (resnet.py:52-57) as four 2x2 convs over the
    SOURCE grid.  Output pixel (2y+a, 2x+b) reads upsampled rows 2y+a-1 .. 2y+a+1 = source rows y+a-1 (ty = 0) and y+a
    (ty = 1): for a = 0 the taps ky = 1, 2 fall on the same source row and are summed, for a = 1 the taps ky = 0, 1 (same in
    x).  phase = a*2 + b, k = (ty*2 + tx)*Ci + ci.  Sums are taken in fp32 before the bf16 rounding."""
    Co, Ci, kh, kw = w.shape
    assert kh == 3 and kw == 3
    wf = w.float()
    sets = (((0,), (1, 2)), ((0, 1), (2,)))          # [a][ty] -> ky values
    out = torch.empty((4, Co, 2, 2, Ci), dtype=torch.float32, device=w.device)
    for a in range(2):
        for b in range(2):
            for ty in range(2):
                for tx in range(2):
                    acc = 0
                    for ky in sets[a][ty]:
                        for kx in sets[b][tx]:
                            acc = acc + wf[:, :, ky, kx]
                    out[a * 2 + b, :, ty, tx, :] = acc
    return out.reshape(4, Co, 4 * Ci).to(w.dtype).contiguous()


def pack_conv1x1(w: torch.Tensor) -> torch.Tensor:
    """[Co, Ci, 1, 1] -> [Co, Ci]."""
    return w.reshape(w.shape[0], w.shape[1]).contiguous()


def geglu_row_order(inner: int, device=None) -> torch.Tensor:
    """row permutation of GEGLU.proj ([2*inner, C]: value rows then gate rows, attention.py:783,792)."""
    assert inner % 16 == 0
    g = torch.arange(inner // 16, device=device)[:, None] * 16
    r = torch.arange(16, device=device)[None, :]
    val = g + r
    gate = inner + g + r
    return torch.cat([val, gate], dim=1).reshape(-1)


def interleave_geglu(w: torch.Tensor, bias: torch.Tensor):
    inner = w.shape[0] // 2
    order = geglu_row_order(inner, w.device)
    return w[order].contiguous(), bias[order].contiguous()
