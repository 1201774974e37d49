"""The GPU parity tolerance is set from a measured number (tests/golden/calibration_bf16.json: what the REAL reference loses
under CPU bf16 autocast against its own fp32 run, written by oracle/make_goldens_w320.py in the build container).  This test
re-measures the same quantity with the oracle on whatever host it runs on, so the committed number cannot go stale."""
import json
from pathlib import Path

import numpy as np
import torch

from oracle import seer_oracle as O
from seervideoldm_amd import synth

GOLD = Path(__file__).resolve().parent / "golden"
W320_UNET = dict(sample_size=16, in_channels=4, out_channels=4, block_out_channels=(320, 320, 320, 320),
                 cross_attention_dim=256, attention_head_dim=8, layers_per_block=1)


def test_bf16_autocast_error_of_the_oracle_matches_the_committed_calibration():
    calib = json.loads((GOLD / "calibration_bf16.json").read_text())["unet_w320_bf16_autocast_vs_fp32"]
    g = {k: torch.from_numpy(v) for k, v in np.load(GOLD / "unet_w320_real.npz").items()}
    sd = synth.synth_state_dict(synth.unet_param_shapes(W320_UNET))
    y32 = O.unet_forward(sd, W320_UNET, g["sample"], g["timestep"], g["context"], cond_frame=0)
    # the oracle reproduces the reference's fp32 output on this fixture ...
    assert ((y32 - g["y_cond0"]).abs().max() / g["y_cond0"].abs().max()).item() < 1e-4
    # ... and loses what the reference loses when its matrix products run in bf16
    with torch.autocast("cpu", dtype=torch.bfloat16):
        y16 = O.unet_forward(sd, W320_UNET, g["sample"], g["timestep"], g["context"], cond_frame=0).float()
    rel = ((y16 - y32).norm() / y32.norm()).item()
    print(f"oracle bf16 autocast vs fp32: rel_l2 {rel:.4g}; committed (reference): {calib['rel_l2']:.4g}")
    assert 0.5 * calib["rel_l2"] < rel < 2.0 * calib["rel_l2"]


def test_fp16_autocast_error_of_the_oracle_matches_the_committed_calibration():
    """the same for fp16 autocast (every shipped yaml: mixed_precision "fp16"): tests/golden/calibration_fp16.json sets the
    tolerance of the fp16-storage engine (tests/test_gpu_unet.py::test_full_size_step_fp16_storage_matches_the_reference)"""
    calib = json.loads((GOLD / "calibration_fp16.json").read_text())["unet_w320_fp16_autocast_vs_fp32"]
    g = {k: torch.from_numpy(v) for k, v in np.load(GOLD / "unet_w320_real.npz").items()}
    sd = synth.synth_state_dict(synth.unet_param_shapes(W320_UNET))
    y32 = O.unet_forward(sd, W320_UNET, g["sample"], g["timestep"], g["context"], cond_frame=0)
    with torch.autocast("cpu", dtype=torch.float16):
        y16 = O.unet_forward(sd, W320_UNET, g["sample"], g["timestep"], g["context"], cond_frame=0).float()
    rel = ((y16 - y32).norm() / y32.norm()).item()
    print(f"oracle fp16 autocast vs fp32: rel_l2 {rel:.4g}; committed (reference): {calib['rel_l2']:.4g}")
    assert 0.5 * calib["rel_l2"] < rel < 2.0 * calib["rel_l2"]
    bf = json.loads((GOLD / "calibration_bf16.json").read_text())["unet_w320_bf16_autocast_vs_fp32"]
    assert calib["rel_l2"] < bf["rel_l2"] / 4          # three more significand bits
