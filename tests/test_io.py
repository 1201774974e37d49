"""Checkpoint / visualisation wire formats: the accelerate checkpoint directory the reference reads
(inference_img.py:98-104) and the GIF / PNG-grid pixels of utils/ddim_sampling_utils.py:95-123 restated with einops-free
index arithmetic (the reference's imageio / torchvision encoders are not installed: parity of the encoders is unpinned,
the pixel arrays are checked against the formulas)."""
import numpy as np
import torch

from seervideoldm_amd import FSTextTransformer, SeerUNet, synth
from seervideoldm_amd import io as sio


def test_load_seer_checkpoint_roundtrip(tmp_path):
    ucfg = dict(block_out_channels=(32, 32, 64, 64), layers_per_block=1, cross_attention_dim=32, attention_head_dim=8)
    usd = synth.synth_state_dict(synth.unet_param_shapes(ucfg))
    fcfg = dict(num_frames=3, num_layers=1, channels=192, n_heads=2, cross_attention_dim=192)
    fsd = synth.synth_state_dict(synth.fstext_param_shapes(**fcfg, max_length=80))
    torch.save(usd, tmp_path / "pytorch_model.bin")
    torch.save(fsd, tmp_path / "pytorch_model_1.bin")
    unet = SeerUNet(**ucfg)
    import seervideoldm_amd.fstext as fst_mod
    old = fst_mod.MAX_LENGTH
    fst_mod.MAX_LENGTH = 80                       # keep the fixture small (the real pos_embed holds 1024 positions)
    try:
        fst = FSTextTransformer(num_frames=3, in_channels=192, out_channels=192, n_heads=2, num_layers=1, cross_attention_dim=192)
    finally:
        fst_mod.MAX_LENGTH = old
    sio.load_seer_checkpoint(str(tmp_path), unet, fst)
    assert all(torch.equal(v, usd[k]) for k, v in unet.state_dict().items())
    assert all(torch.equal(v, fsd[k]) for k, v in fst.state_dict().items())


def test_gif_frames_and_grid_pixels(tmp_path):
    g = torch.Generator().manual_seed(0)
    b, f0, f, H, W = 2, 1, 3, 8, 12
    cond = torch.rand((b, 3, f0, H, W), generator=g)
    vids = torch.rand((b, 3, f, H, W), generator=g)
    fr = sio.gif_frames(vids, cond, num_sample_rows=1)
    assert fr.shape == (f0 + f, H + 4, b * (W + 4), 3) and fr.dtype == np.uint8
    # frame t, sample j sits at columns j*(W+4)+2 ..; the 2-pixel frame is black; values are (x*255) truncated
    for t in range(f0 + f):
        src = cond[:, :, t] if t < f0 else vids[:, :, t - f0]
        for j in range(b):
            tile = fr[t, :, j * (W + 4):(j + 1) * (W + 4)]
            assert (tile[:2] == 0).all() and (tile[-2:] == 0).all() and (tile[:, :2] == 0).all() and (tile[:, -2:] == 0).all()
            want = (src[j].permute(1, 2, 0).numpy() * 255).astype("uint8")
            assert np.array_equal(tile[2:-2, 2:-2], want)
    fr2 = sio.gif_frames(vids, cond, num_sample_rows=2)            # '(i j) c f h w -> c f (i h) (j w)' with i = 2
    assert fr2.shape == (f0 + f, 2 * (H + 4), W + 4, 3) and np.array_equal(fr2[:, :H + 4], fr[:, :, :W + 4])

    grid = sio.image_grid(vids, cond)
    hp, wp = H + 4, W + 4
    assert grid.shape == (3, b * (hp + 6) + 6, f0 * wp + 4 + f * wp + 4 + 12)
    assert torch.all(grid[:, :6] == 0.5) and torch.all(grid[:, :, :6] == 0.5)
    row0 = grid[:, 6:6 + hp, 6:]
    assert torch.equal(row0[:, 2:2 + H, 2:2 + W], cond[0, :, 0])
    green = row0[:, :, f0 * wp:f0 * wp + 4]
    assert torch.all(green[1] == 1) and torch.all(green[0] == 0) and torch.all(green[2] == 0)
    assert torch.equal(row0[:, 2:2 + H, f0 * wp + 4 + 2:f0 * wp + 4 + 2 + W], vids[0, :, 0])

    # a one-sample batch: torchvision.utils.make_grid returns the single image unpadded (tensor.squeeze(0)) -- no grey border
    g1 = sio.image_grid(vids[:1], cond[:1])
    assert g1.shape == (3, hp, f0 * wp + 4 + f * wp + 4) and torch.equal(g1, grid[:, 6:6 + hp, 6:6 + g1.shape[2]])

    gif, png = sio.save_visualization_onegif(vids, cond, 7, str(tmp_path / "img.jpg"))
    from PIL import Image
    im = Image.open(gif)
    assert im.n_frames == f0 + f and im.size == (b * (W + 4), H + 4) and im.info.get("duration") == 250   # fps 4
    pg = np.asarray(Image.open(png))
    assert pg.shape == (grid.shape[1], grid.shape[2], 3) and pg[0, 0, 0] == 128                             # 0.5 grey border
