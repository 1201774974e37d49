"""TEST INFRASTRUCTURE ONLY -- imports the REAL reference modules from /root/reference on CPU.

Only usable in the build container (the reference tree does not travel to the GPU box); used by
oracle/make_goldens.py to produce tests/golden/*.npz and by tests that pin oracle/seer_oracle.py against the reference
when /root/reference is present.  Nothing in the product package imports this file.

The reference's hot path needs third-party packages that are not installed here (diffusers 0.10.2, xformers 0.0.13,
rotary-embedding-torch 0.1.5, torchvision, imageio).  The stubs below restate exactly the pieces the path touches
(SURVEY 8(c), Appendix D).  Their arithmetic is "parity unpinned": it is written from the published behaviour of the
pinned versions, no reference test pins it.

Patches applied to the reference at import time (each one is a finding of SURVEY section 0):
  * every attention module gets `_use_memory_efficient_attention_xformers = True` (finding 4: the non-xformers
    temporal path raises; the xformers path IS the defined semantics), with xformers' MEA stubbed as
    softmax(q k^T / sqrt(d) + lower-triangular mask) v in fp32;
  * `DDIMSampler.register_buffer` no longer forces .to("cuda") (finding 5).
"""
from __future__ import annotations

import math
import sys
import types
from pathlib import Path

import torch
import torch.nn as nn

REFERENCE_ROOT = Path("/root/reference")


def available() -> bool:
    return (REFERENCE_ROOT / "seer" / "models" / "unet_3d_condition.py").exists()


# ----------------------------------------------------------------------------------------------------------- stubs
class _ConfigDict(dict):
    __getattr__ = dict.get


def _register_to_config(init):
    import functools
    import inspect

    @functools.wraps(init)
    def wrapper(self, *args, **kwargs):
        sig = inspect.signature(init)
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        self._internal_dict = _ConfigDict(cfg)
        init(self, *args, **kwargs)
    return wrapper


class _ConfigMixin:
    @property
    def config(self):
        return self._internal_dict


class _ModelMixin(nn.Module):
    pass


class _BaseOutput(dict):
    """diffusers BaseOutput: a dataclass that is also indexable by field name (attention.py:143 does ['sample'])."""
    def __post_init__(self):
        import dataclasses
        for f in dataclasses.fields(self):
            dict.__setitem__(self, f.name, getattr(self, f.name))


class _Timesteps(nn.Module):
    """diffusers==0.10.2 models/embeddings.py Timesteps / get_timestep_embedding (scale=1, max_period=10000)."""
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels, self.flip_sin_to_cos, self.downscale_freq_shift = num_channels, flip_sin_to_cos, downscale_freq_shift

    def forward(self, timesteps):
        half = self.num_channels // 2
        exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
        exponent = exponent / (half - self.downscale_freq_shift)
        emb = torch.exp(exponent)
        emb = timesteps[:, None].float() * emb[None, :]
        emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
        if self.flip_sin_to_cos:
            emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
        return emb


class _TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, out_dim or time_embed_dim)

    def forward(self, sample):
        return self.linear_2(self.act(self.linear_1(sample)))


class _RotaryEmbedding(nn.Module):
    """rotary-embedding-torch==0.1.5 RotaryEmbedding(dim), freqs_for='lang', theta=10000, not learned."""
    def __init__(self, dim, theta=10000):
        super().__init__()
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: (dim // 2)].float() / dim))
        self.register_buffer("freqs", freqs)

    def rotate_queries_or_keys(self, t, seq_dim=-2):
        seq_len = t.shape[seq_dim]
        pos = torch.arange(seq_len, device=t.device)
        freqs = torch.einsum("..., f -> ... f", pos.type(self.freqs.dtype), self.freqs)
        freqs = freqs.repeat_interleave(2, dim=-1)                      # '... n -> ... (n r)', r = 2
        rot_dim = freqs.shape[-1]
        t_left, t_mid, t_right = t[..., :0], t[..., :rot_dim], t[..., rot_dim:]
        x = t_mid.reshape(*t_mid.shape[:-1], rot_dim // 2, 2)
        x1, x2 = x.unbind(dim=-1)
        rot_half = torch.stack((-x2, x1), dim=-1).reshape(*t_mid.shape)
        t_mid = (t_mid * freqs.cos()) + (rot_half * freqs.sin())
        return torch.cat((t_left, t_mid, t_right), dim=-1)


class _LowerTriangularMask:
    pass


def _memory_efficient_attention(query, key, value, attn_bias=None, p=0.0):
    """xformers==0.0.13 ops.memory_efficient_attention on [B, M, K] inputs: softmax(q k^T / sqrt(K) + bias) v."""
    scale = query.shape[-1] ** -0.5
    s = torch.baddbmm(torch.zeros(query.shape[0], query.shape[1], key.shape[1], dtype=query.dtype, device=query.device),
                      query, key.transpose(-1, -2), beta=0, alpha=scale)
    if attn_bias is not None:
        assert isinstance(attn_bias, _LowerTriangularMask)
        i, j = s.shape[-2:]
        mask = torch.ones((i, j), dtype=torch.bool, device=s.device).tril()
        s = s.masked_fill(~mask, float("-inf"))
    return torch.bmm(s.softmax(dim=-1), value)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    if "diffusers" in sys.modules and getattr(sys.modules["diffusers"], "_seer_stub", False):
        return

    class _Logging:
        @staticmethod
        def get_logger(name):
            import logging
            return logging.getLogger(name)

    _mod("diffusers", _seer_stub=True)
    _mod("diffusers.configuration_utils", ConfigMixin=_ConfigMixin, register_to_config=_register_to_config)
    _mod("diffusers.modeling_utils", ModelMixin=_ModelMixin)
    _mod("diffusers.utils", BaseOutput=_BaseOutput, logging=_Logging)
    _mod("diffusers.utils.import_utils", is_xformers_available=lambda: True)
    _mod("diffusers.models")
    _mod("diffusers.models.embeddings", Timesteps=_Timesteps, TimestepEmbedding=_TimestepEmbedding,
         ImagePositionalEmbeddings=type("ImagePositionalEmbeddings", (nn.Module,), {}))
    _mod("rotary_embedding_torch", RotaryEmbedding=_RotaryEmbedding)
    xf = _mod("xformers")
    xf.ops = _mod("xformers.ops", memory_efficient_attention=_memory_efficient_attention,
                  LowerTriangularMask=_LowerTriangularMask)
    _mod("xformers.components")
    _mod("xformers.components.attention", AttentionMask=type("AttentionMask", (), {}))
    tv = _mod("torchvision")
    tv.utils = _mod("torchvision.utils", make_grid=None, save_image=None)
    tv.transforms = _mod("torchvision.transforms")
    _mod("imageio")


_loaded = {}


def load_reference():
    """returns a namespace with the reference classes/functions of the hot path."""
    if _loaded:
        return types.SimpleNamespace(**_loaded)
    if not available():
        raise RuntimeError("/root/reference is not present (only in the build container)")
    install_stubs()
    if str(REFERENCE_ROOT) not in sys.path:
        sys.path.insert(0, str(REFERENCE_ROOT))
    import importlib
    attention = importlib.import_module("seer.models.attention")
    resnet = importlib.import_module("seer.models.resnet")
    blocks = importlib.import_module("seer.models.unet_3d_blocks")
    unet = importlib.import_module("seer.models.unet_3d_condition")
    ddim = importlib.import_module("ldm.models.diffusion.ddim_video")
    util = importlib.import_module("ldm.modules.diffusionmodules.util")
    vae = importlib.import_module("ldm.modules.diffusionmodules.model")
    glue = importlib.import_module("utils.ddim_sampling_utils")
    # finding 5: CPU-safe register_buffer
    ddim.DDIMSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)
    _loaded.update(attention=attention, resnet=resnet, blocks=blocks, unet=unet, ddim=ddim, util=util, vae=vae,
                   glue=glue)
    return types.SimpleNamespace(**_loaded)


def enable_xformers_path(module: nn.Module):
    """finding 4: the xformers path is the only one that runs for temporal blocks; select it everywhere."""
    for m in module.modules():
        if hasattr(m, "_use_memory_efficient_attention_xformers"):
            m._use_memory_efficient_attention_xformers = True
    return module
