"""The PRODUCT's schedule tables against the reference's: `DDIMSampler.make_schedule`
(seervideoldm_amd/ddim.py; ldm/models/diffusion/ddim_video.py:27-68, ldm/modules/diffusionmodules/util.py:46-74) must equal
tests/golden/schedule_S{4,30,50}.npz -- written by oracle/make_goldens.py from the reference's own sampler -- bit for bit as
float32, which is what a step reads from `ddim_coef`.  Host arithmetic only: runs on CPU.  (The end-to-end tolerance of the sampler
tests cannot see a 1e-4 table error such as the `alphas_prev[0] = alphas_cumprod[0]` quirk; this can.)"""
from pathlib import Path

import numpy as np
import pytest

from seervideoldm_amd import DDIMSampler

G = Path(__file__).parent / "golden"


@pytest.mark.parametrize("S,n", [(4, 4), (30, 31), (50, 50)])
def test_product_schedule_tables_equal_the_reference(S, n):
    g = np.load(G / f"schedule_S{S}.npz")
    smp = DDIMSampler("cpu")
    smp.make_schedule(S, ddim_eta=0.0, verbose=False)
    assert len(smp.ddim_timesteps) == n                       # "30 steps" is 31 UNet evaluations (SURVEY finding 6)
    assert np.array_equal(np.asarray(smp.ddim_timesteps), g["ddim_timesteps"])
    assert np.array_equal(smp._t_table.numpy(), g["ddim_timesteps"])
    coef = smp.ddim_coef.numpy()
    assert coef.dtype == np.float32 and coef.shape == (n, 4)
    for col, k in enumerate(("alphas", "alphas_prev", "sigmas", "sqrt_one_minus_alphas")):
        assert np.array_equal(coef[:, col], np.float32(g[k])), k
    assert coef[0, 1] == np.float32(g["alphas_cumprod_0_999"][0])      # the quirk: a_prev[0] is alphas_cumprod[0], not 1
    assert np.array_equal(smp.alphas_cumprod.numpy()[[0, 999]], np.float32(g["alphas_cumprod_0_999"]))
    assert np.array_equal(smp.betas.numpy()[[0, 999]], np.float32(g["betas_0_999"]))


def test_product_schedule_with_eta():
    """sigma column for eta > 0 follows ddim_video.py:60-62 (float64 formula over the float32-valued alphas)"""
    smp = DDIMSampler("cpu")
    smp.make_schedule(50, ddim_eta=1.0, verbose=False)
    a, ap = smp.ddim_alphas, smp.ddim_alphas_prev
    want = np.sqrt((1 - ap) / (1 - a) * (1 - a / ap))
    assert np.array_equal(smp.ddim_coef.numpy()[:, 2], np.float32(want))
    assert smp.ddim_coef.numpy()[1:, 2].min() > 0
