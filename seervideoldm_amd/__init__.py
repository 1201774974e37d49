"""seervideoldm_amd -- MI355X-native (gfx950) implementation of Seer's DDIM denoising hot path.

Drop-in surface: `SeerUNet`, `DDIMSampler`, `ddim_sample`, `AutoencoderKL` (and, before the path,
`FSTextTransformer`) mirror the names, arguments and checkpoint key layout of the reference; every FLOP runs in libseer_hip.so (include/seer_hip.h).
The fine-tuning step of train.py lives in `seervideoldm_amd.trainer.SeerTrainer` (hand-written backward on the same library).
Importing the package never touches the GPU; the HIP library is loaded on first use and its absence is an error.
"""
from .ddim import DDIMSampler, ddim_sample  # noqa: F401
from .fstext import FSTextTransformer  # noqa: F401
from .unet import SeerUNet  # noqa: F401
from .vae import AutoencoderKL  # noqa: F401

__all__ = ["SeerUNet", "DDIMSampler", "ddim_sample", "AutoencoderKL", "FSTextTransformer"]
