"""`utils.ddim_sampling_utils` with the reference's signatures (utils/ddim_sampling_utils.py:10-123): the visualisation
helpers take `(accelerator, vae, ...)` there; the gather goes through `accelerator.gather` exactly as `concat_all_gather`
does, the pixels come from seervideoldm_amd.io."""
import torch

from .. import io as _io
from ..ddim import ddim_sample

__all__ = ["ddim_sample", "concat_all_gather", "video_tensor_to_gif", "save_visualization", "save_visualization_onegif"]


@torch.no_grad()
def concat_all_gather(accelerator, tensor):
    return accelerator.gather(tensor)


def video_tensor_to_gif(tensor, path, duration=120, loop=0, optimize=True):
    """[c, f, h, w] in [0,1] -> GIF (T.ToPILImage per frame: mul(255) truncated to uint8)"""
    from PIL import Image
    frames = [Image.fromarray((fr.detach().float().cpu().permute(1, 2, 0).mul(255).to(torch.uint8)).numpy())
              for fr in tensor.unbind(dim=1)]
    frames[0].save(path, save_all=True, append_images=frames[1:], duration=duration, loop=loop, optimize=optimize)
    return frames


def save_visualization(accelerator, vae, x_samples_ddim, video_latent, video, results_folder, global_step, num_sample_rows=2):
    return _io.save_visualization(vae, x_samples_ddim, video_latent, video, results_folder, global_step, num_sample_rows,
                                  gather=accelerator.gather)


def save_visualization_onegif(accelerator, vae, x_samples_ddim, x0_image, sample_id, image_path, num_sample_rows=1):
    return _io.save_visualization_onegif(accelerator.gather(x_samples_ddim.contiguous()), accelerator.gather(x0_image.contiguous()),
                                         sample_id, image_path, num_sample_rows)
