"""`seer.models.unet_3d_condition` as the reference's scripts import it (inference_img.py:29, train.py:21)."""
from ..fstext import FSTextTransformer
from ..unet import SeerUNet

__all__ = ["SeerUNet", "FSTextTransformer"]
