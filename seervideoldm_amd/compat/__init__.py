"""Run the reference's scripts UNCHANGED on the MI355X path.

`inference_img.py`, `eval.py`, `inference.py` and `train.py` import the hot path by three module names
(inference_img.py:29,38-39):

    from seer.models.unet_3d_condition import SeerUNet, FSTextTransformer
    from ldm.models.diffusion.ddim_video import DDIMSampler
    from utils.ddim_sampling_utils import ddim_sample, save_visualization_onegif

`install()` puts a finder in front of `sys.meta_path` that answers exactly those three names with modules that re-export
the product classes (same names, arguments and return values: SURVEY 8(b)).  Every other module of the reference checkout
(`utils.fvd`, `ldm.util`, datasets ...) still resolves to the reference's own file, because the parent packages are
presented as namespace packages over the directories found on `sys.path`.

    python -m seervideoldm_amd.compat inference_img.py --config configs/inference_base.yaml ...

runs the script with the aliases installed (`sys.argv`, `sys.path[0]` and `__main__` as `python script.py` sets them).
`install(vae=True)` (or `SEER_COMPAT_VAE=1` with the runner) also replaces `diffusers.AutoencoderKL` by the product's VAE
when `diffusers` is importable; without it the reference keeps its fp32 torch VAE for `vae.decode` / `vae.encode`.
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import os
import sys
import types

ALIASES = {
    "seer.models.unet_3d_condition": "seervideoldm_amd.compat._unet_3d_condition",
    "ldm.models.diffusion.ddim_video": "seervideoldm_amd.compat._ddim_video",
    "utils.ddim_sampling_utils": "seervideoldm_amd.compat._ddim_sampling_utils",
}
_PARENTS = sorted({".".join(n.split(".")[:i]) for n in ALIASES for i in range(1, n.count(".") + 1)})


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname in ALIASES:
            return importlib.machinery.ModuleSpec(fullname, self)
        if fullname in _PARENTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        if spec.name in ALIASES:
            src = importlib.import_module(ALIASES[spec.name])
            mod = types.ModuleType(spec.name, src.__doc__)
            mod.__dict__.update({k: v for k, v in src.__dict__.items() if not k.startswith("__")})
            mod.__all__ = list(getattr(src, "__all__", []))
            return mod
        # a parent package: a namespace over every directory of that name on sys.path (the reference checkout's own
        # `utils/`, `ldm/`, `seer/` when the script runs from there), so the reference's other modules keep resolving
        mod = types.ModuleType(spec.name)
        rel = spec.name.replace(".", os.sep)
        mod.__path__ = [os.path.join(p or os.getcwd(), rel) for p in sys.path if os.path.isdir(os.path.join(p or os.getcwd(), rel))]
        # submodules that were imported before install() stay importable as attributes of the new package object
        for name, sub in list(sys.modules.items()):
            if name.startswith(spec.name + ".") and "." not in name[len(spec.name) + 1:] and name not in ALIASES and name not in _PARENTS:
                setattr(mod, name.rsplit(".", 1)[1], sub)
        return mod

    def exec_module(self, module):
        pass


_finder = _AliasFinder()


def install(vae: bool = False) -> None:
    if _finder not in sys.meta_path:
        sys.meta_path.insert(0, _finder)
    for name in list(ALIASES) + _PARENTS:      # modules imported before install() (the reference's own) give way
        sys.modules.pop(name, None)
    if vae:
        try:
            import diffusers
        except ImportError:
            return
        from seervideoldm_amd import AutoencoderKL
        diffusers.AutoencoderKL = AutoencoderKL


def uninstall() -> None:
    if _finder in sys.meta_path:
        sys.meta_path.remove(_finder)
    for name in list(ALIASES) + _PARENTS:
        sys.modules.pop(name, None)
