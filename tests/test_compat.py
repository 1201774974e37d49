"""The reference's scripts import the hot path by three module names (inference_img.py:29,38-39; eval.py:29,38-39;
inference.py:29,38-39; train.py:21).  seervideoldm_amd.compat answers exactly those names with the product classes, so the
scripts run unchanged; everything else of a reference checkout keeps resolving to its own files."""
import subprocess
import sys
import textwrap
from pathlib import Path

import pytest
import torch

import seervideoldm_amd
from seervideoldm_amd import compat

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture
def installed():
    compat.install()
    yield
    compat.uninstall()


def test_the_reference_import_lines_resolve_to_the_product(installed):
    ns = {}
    exec(textwrap.dedent("""
        from seer.models.unet_3d_condition import SeerUNet, FSTextTransformer          # inference_img.py:29
        from ldm.models.diffusion.ddim_video import DDIMSampler                        # inference_img.py:38
        from utils.ddim_sampling_utils import ddim_sample, save_visualization_onegif   # inference_img.py:39
        from utils.ddim_sampling_utils import ddim_sample, save_visualization          # inference.py:39
    """), ns)
    assert ns["SeerUNet"] is seervideoldm_amd.SeerUNet and ns["FSTextTransformer"] is seervideoldm_amd.FSTextTransformer
    assert ns["DDIMSampler"] is seervideoldm_amd.DDIMSampler and ns["ddim_sample"] is seervideoldm_amd.ddim_sample
    import inspect
    # the reference's signatures (utils/ddim_sampling_utils.py:46,95): accelerator and vae come first
    assert list(inspect.signature(ns["save_visualization_onegif"]).parameters)[:6] == \
        ["accelerator", "vae", "x_samples_ddim", "x0_image", "sample_id", "image_path"]
    assert list(inspect.signature(ns["save_visualization"]).parameters)[:7] == \
        ["accelerator", "vae", "x_samples_ddim", "video_latent", "video", "results_folder", "global_step"]


def test_other_modules_of_a_checkout_keep_resolving(tmp_path, installed, monkeypatch):
    """a script run from the reference checkout also imports `utils.fvd`, `ldm.util` ...: the alias parents are namespaces over
    the checkout's own directories"""
    (tmp_path / "utils").mkdir()
    (tmp_path / "utils" / "__init__.py").write_text("")
    (tmp_path / "utils" / "fvd_like.py").write_text("VALUE = 41\n")
    (tmp_path / "utils" / "ddim_sampling_utils.py").write_text("raise ImportError('the reference file must not be imported')\n")
    (tmp_path / "ldm" / "models" / "diffusion").mkdir(parents=True)
    (tmp_path / "ldm" / "util_like.py").write_text("VALUE = 42\n")
    monkeypatch.syspath_prepend(str(tmp_path))
    compat.install()          # again: drops cached parents so that the new sys.path entry is seen
    import utils.fvd_like
    import ldm.util_like
    from utils.ddim_sampling_utils import ddim_sample
    assert utils.fvd_like.VALUE == 41 and ldm.util_like.VALUE == 42 and ddim_sample is seervideoldm_amd.ddim_sample


def test_runner_executes_an_unchanged_script(tmp_path):
    script = tmp_path / "inference_like.py"
    script.write_text(textwrap.dedent("""
        import sys
        from seer.models.unet_3d_condition import SeerUNet, FSTextTransformer
        from ldm.models.diffusion.ddim_video import DDIMSampler
        from utils.ddim_sampling_utils import ddim_sample, save_visualization_onegif
        if __name__ == "__main__":
            s = DDIMSampler("cpu")
            s.make_schedule(4, verbose=False)
            print("OK", SeerUNet.__module__, sys.argv[1:], [int(t) for t in s.ddim_timesteps])
    """))
    r = subprocess.run([sys.executable, "-m", "seervideoldm_amd.compat", str(script), "--config", "x.yaml"],
                       capture_output=True, text=True, cwd=str(ROOT), timeout=300)
    assert r.returncode == 0, r.stderr
    assert "OK seervideoldm_amd.unet ['--config', 'x.yaml'] [1, 251, 501, 751]" in r.stdout


def test_validation_grid_pixels():
    """utils/ddim_sampling_utils.py:46-92: rows = samples; [cond x3 | green | original / reconstruction / prediction | red]"""
    from seervideoldm_amd import io as sio
    g = torch.Generator().manual_seed(3)
    b, f0, f, H, W = 2, 1, 2, 8, 8
    video = torch.rand((b, 3, f0 + f, H, W), generator=g)
    rec, pred = torch.rand((b, 3, f, H, W), generator=g), torch.rand((b, 3, f, H, W), generator=g)
    grid = sio.validation_grid(pred, rec, video)
    hp, wp = H + 4, W + 4
    assert grid.shape == (3, b * (3 * hp + 6) + 6, f0 * wp + 4 + f * wp + 4 + 12)
    row0 = grid[:, 6:6 + 3 * hp, 6:-6]
    for k in range(3):                                                   # the conditioning strip repeated three times
        assert torch.equal(row0[:, k * hp + 2:k * hp + 2 + H, 2:2 + W], video[0, :, 0])
    x0 = f0 * wp + 4
    assert torch.equal(row0[:, 2:2 + H, x0 + 2:x0 + 2 + W], video[0, :, f0])            # original
    assert torch.equal(row0[:, hp + 2:hp + 2 + H, x0 + 2:x0 + 2 + W], rec[0, :, 0])     # reconstruction
    assert torch.equal(row0[:, 2 * hp + 2:2 * hp + 2 + H, x0 + wp + 2:x0 + wp + 2 + W], pred[0, :, 1])   # prediction, frame 1
    green, red = row0[:, :, f0 * wp:f0 * wp + 4], row0[:, :, -4:]
    assert torch.all(green[1] == 1) and torch.all(green[0] == 0) and torch.all(red[0] == 1) and torch.all(red[1] == 0)
