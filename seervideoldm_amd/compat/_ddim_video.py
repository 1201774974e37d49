"""`ldm.models.diffusion.ddim_video` as the reference's scripts import it (inference_img.py:38)."""
from ..ddim import DDIMSampler

__all__ = ["DDIMSampler"]
