"""TEST INFRASTRUCTURE ONLY -- gradient sketches: a tensor's squared norm and K inner products with seeded standard-normal vectors
(a function of the tensor's NAME only), the form in which tests/golden/train_*.npz keep the reference's gradients
(oracle/make_goldens_train.py).  For an error e = g - g_ref, E[(e . r)^2] = |e|^2 over standard-normal r."""
from __future__ import annotations

import zlib
from typing import Dict, Sequence, Tuple

import numpy as np
import torch

K_SKETCH = 4


def sketch_vectors(name: str, numel: int, k: int = K_SKETCH) -> torch.Tensor:
    """[k, numel] standard-normal vectors, a function of the tensor's name only"""
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
    return torch.randn((k, numel), generator=g, dtype=torch.float32)


def sketch(name: str, grad: torch.Tensor, k: int = K_SKETCH) -> np.ndarray:
    """[1 + k] float64: |g|^2, g . r_1 .. g . r_k"""
    g = grad.detach().reshape(-1).float().cpu()
    r = sketch_vectors(name, g.numel(), k)
    return np.asarray([float(g.double().pow(2).sum()), *(r.double() @ g.double()).tolist()], dtype=np.float64)


def sketch_errors(prefix: str, keys: Sequence[str], ref_sketch: np.ndarray, grads: Dict[str, torch.Tensor]) -> Tuple[float, float, float]:
    """(estimated |g - g_ref|^2 summed over the tensors, |g_ref|^2 summed, the worst tensor's estimated |e| / (|g_ref| + 1e-3 |G_ref|))"""
    den2 = float(ref_sketch[:, 0].sum())
    num2, per = 0.0, []
    for i, k in enumerate(keys):
        mine = sketch(prefix + k, grads[k], ref_sketch.shape[1] - 1)
        e2 = float(np.mean((mine[1:] - ref_sketch[i, 1:]) ** 2))
        num2 += e2
        per.append((e2 ** 0.5) / (ref_sketch[i, 0] ** 0.5 + 1e-3 * den2 ** 0.5))
    return num2, den2, max(per)
