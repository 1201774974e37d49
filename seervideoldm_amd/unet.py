"""SeerUNet -- host-side mirror of the reference's `seer.models.unet_3d_condition.SeerUNet` over libseer_hip.so.

Same constructor keywords, same `state_dict()` key layout (a reference `pytorch_model.bin` loads with strict=True), same
call: `unet(sample[B,4,F,h,w], timestep, context[B,F,77,ctx], cond_frame=0) -> Tensor[B,4,F,h,w]`
(seer/models/unet_3d_condition.py:64-84, 283-376).  The module tree only HOLDS parameters; the forward is executed by
`_Engine`, which repacks the weights once (bf16, channels-last GEMM layouts, fused q|k|v, interleaved GEGLU rows,
concatenated time-embedding projections) and then issues hand-written HIP kernels through the C ABI on torch's current
stream.  Activations are token-major bf16 [B*F*H*W, C] end to end; nothing is computed by torch ops.

There is no CPU path: calling forward with CPU tensors, or without the built library, raises.
"""
from __future__ import annotations

import json
import os
from collections import OrderedDict
from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import ops as hip_ops
from . import synth
from .weights import geglu_row_order, pack_conv1x1, pack_conv3x3, pack_conv3x3_up_phases

bf16 = torch.bfloat16
MAX_WIN_SIZE, MAX_RATIO, MIN_WIN_SIZE = 8, 4, 4      # seer/models/attention.py:31-33


class _Node(nn.Module):
    """parameter container: gives the state dict / named_modules() the reference's dotted names."""
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("container module: the forward pass lives in SeerUNet.forward")


def _build_tree(root: nn.Module, shapes: "OrderedDict[str, Tuple[int, ...]]"):
    for key, shape in shapes.items():
        parts = key.split(".")
        mod = root
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, _Node())
            mod = mod._modules[p]
        if key.endswith("rotary_emb.freqs"):
            mod.register_buffer(parts[-1], synth.synth_tensor(key, shape))
        else:
            mod.register_parameter(parts[-1], nn.Parameter(torch.zeros(shape), requires_grad=True))


class _Config(dict):
    __getattr__ = dict.get


class SeerUNet(nn.Module):
    _supports_gradient_checkpointing = True
    config_name = "config.json"

    def __init__(self, sample_size=None, in_channels=4, out_channels=4, center_input_sample=False,
                 flip_sin_to_cos=True, freq_shift=0,
                 down_block_types=("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"),
                 up_block_types=("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"),
                 block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, downsample_padding=1,
                 mid_block_scale_factor=1, act_fn="silu", norm_num_groups=32, norm_eps=1e-5,
                 cross_attention_dim=1280, attention_head_dim=8, compute_dtype=None):
        super().__init__()
        # the reference overwrites the block types / downsample padding with constants (unet_3d_condition.py:90-92)
        if len(block_out_channels) != 4:
            raise ValueError("SeerUNet is hard-wired to 4 resolution levels (3 x CrossAttnDownBlock3D + DownBlock3D)")
        if act_fn != "silu":
            raise ValueError(f"act_fn {act_fn!r}: only 'silu' exists on this path")
        self.config = _Config(sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
                              center_input_sample=center_input_sample, flip_sin_to_cos=flip_sin_to_cos,
                              freq_shift=freq_shift, down_block_types=tuple(down_block_types),
                              up_block_types=tuple(up_block_types), block_out_channels=tuple(block_out_channels),
                              layers_per_block=layers_per_block, downsample_padding=1,
                              mid_block_scale_factor=mid_block_scale_factor, act_fn=act_fn,
                              norm_num_groups=norm_num_groups, norm_eps=norm_eps,
                              cross_attention_dim=cross_attention_dim, attention_head_dim=attention_head_dim)
        self.sample_size = sample_size
        self._shapes = synth.unet_param_shapes(self.config)
        _build_tree(self, self._shapes)
        self._engine: Optional[_Engine] = None
        self._slice_size = None
        self.use_graph = False
        self._shard = None              # parallel.FrameShard when attached (seervideoldm_amd/parallel.py)
        self.gn_colsums = True          # GroupNorm statistics from the producing GEMM's column sums (read by prepare())
        self._ops_backend = hip_ops     # tests may inject tests/torch_ops_backend.py to exercise the host logic on CPU
        self._ctx_slice = None
        # 16-bit storage type of activations and weights (fp32 accumulation and statistics either way): bf16 = the reference under
        # `mixed_precision: "bf16"` (BASELINE config 2), torch.float16 = under "fp16" (what every shipped yaml says:
        # configs/inference_base.yaml:16, eval.yaml:22, train.yaml:33).  Set here, by `unet.compute_dtype = ...`, or -- None --
        # taken from the autocast state of the call (accelerate's prepare() wraps forward in torch.autocast with the configured type).
        if compute_dtype not in (None, torch.bfloat16, torch.float16):
            raise ValueError(f"compute_dtype {compute_dtype!r}: torch.bfloat16 or torch.float16 (fp32 accumulation either way)")
        self.compute_dtype = compute_dtype

    # ---- construction / weights -------------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, path, subfolder=None, revision=None, low_cpu_mem_usage=False, **kw):
        """diffusers semantics (inference_img.py:74-79): read config.json, ignore unknown keys, load weights by name,
        non-strict (temporal keys are absent from an SD-v1-5 checkpoint)."""
        import inspect
        root = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(root, cls.config_name)) as f:
            cfg = json.load(f)
        allowed = set(inspect.signature(cls.__init__).parameters) - {"self"}
        model = cls(**{k: v for k, v in cfg.items() if k in allowed})
        for fname in ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.bin", "pytorch_model.bin"):
            fp = os.path.join(root, fname)
            if os.path.exists(fp):
                if fname.endswith(".safetensors"):
                    from safetensors.torch import load_file
                    sd = load_file(fp)
                else:
                    sd = torch.load(fp, map_location="cpu")
                model.load_state_dict(sd, strict=False)
                break
        else:
            raise FileNotFoundError(f"no weight file under {root}")
        return model

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._engine = None
        return out

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    def _dtype_of_call(self):
        """the 16-bit storage type this forward runs in: compute_dtype, else the autocast type when the caller wrapped the call in
        torch.autocast (accelerate.prepare under mixed_precision fp16 / bf16), else bf16"""
        if self.compute_dtype is not None:
            return self.compute_dtype
        if torch.is_autocast_enabled():
            dt = torch.get_autocast_dtype("cuda")
            if dt in (torch.bfloat16, torch.float16):
                return dt
        return torch.bfloat16

    def prepare(self, dtype=None):
        """(re)build the packed device weights from the current parameters; call after editing parameters in place."""
        dtype = dtype or self._dtype_of_call()
        dev = next(self.parameters()).device.type
        with torch.autocast(device_type=dev if dev in ("cuda", "cpu") else "cuda", enabled=False):     # fp32 weight algebra stays fp32
            self._engine = _Engine(self, ops=self._ops_backend, shard=self._shard, dtype=dtype)
        return self

    # ---- toggles of the reference surface ---------------------------------------------------------
    def enable_xformers_memory_efficient_attention(self, *a, **k):
        return self     # the HIP flash kernels ARE the memory-efficient path (inference_img.py:85)

    def disable_xformers_memory_efficient_attention(self):
        return self

    def set_use_memory_efficient_attention_xformers(self, valid: bool = True):
        return self

    def set_attention_slice(self, slice_size):
        heads = self.config.attention_head_dim
        if isinstance(slice_size, int) and slice_size > heads:
            raise ValueError(f"size {slice_size} has to be smaller or equal to {heads}.")
        self._slice_size = slice_size      # attention never materialises S x S here: accepted, no effect

    def _set_gradient_checkpointing(self, module, value=False):
        pass

    # ---- forward ---------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, sample: torch.Tensor, timestep: Union[torch.Tensor, float, int], context: torch.Tensor,
                cond_frame: int = 0, return_attn: bool = False) -> torch.Tensor:
        if not sample.is_cuda and self._ops_backend is hip_ops:
            raise hip_ops._lib.SeerHipError("SeerUNet.forward needs ROCm tensors: the HIP kernels are the only compute path")
        dt = self._dtype_of_call()
        # the caller's autocast state has said its piece (the storage type): everything below -- the host-side weight algebra of
        # prepare() included -- runs outside it (a torch matmul under autocast would hand back 16-bit biases)
        with torch.autocast(device_type=sample.device.type if sample.device.type in ("cuda", "cpu") else "cuda", enabled=False):
            return self._forward_impl(sample, timestep, context, cond_frame, return_attn, dt)

    def _forward_impl(self, sample, timestep, context, cond_frame, return_attn, dt):
        if self._engine is None or self._engine.device != sample.device or self._engine.dt != dt:
            self.prepare(dt)
        if self.config.center_input_sample:
            sample = 2 * sample - 1.0
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.long, device=sample.device)
        elif t.dim() == 0:
            t = t[None].to(sample.device)
        t = t.to(torch.long).broadcast_to((sample.shape[0],)).contiguous()
        if return_attn:
            # (out, attn_list): the text cross-attention scores of the last text block of every attention-bearing container
            # (unet_3d_condition.py:291-292,317-323,372-374) -- an analysis path: eager launches, one process
            if self._shard is not None:
                raise NotImplementedError("return_attn is not available on a sharded model")
            out, attn = self._engine.run(sample.float().contiguous(), t, context, int(cond_frame), return_attn=True)
            return out.to(sample.dtype), [a.to(sample.dtype) for a in attn]
        if self._shard is None:
            out = self._engine.run(sample.float().contiguous(), t, context, int(cond_frame), use_graph=self.use_graph)
            return out.to(sample.dtype)
        # sharded step: every rank holds the full inputs, computes its (batch rows, frames) block, all ranks get the full eps
        sh = self._shard
        B, _, Fr = sample.shape[:3]
        if context.dim() == 3:
            context = context[:, None].expand(-1, Fr, -1, -1)
        (b0, b1), (f0, f1) = sh.plan(B, Fr)
        sh.probe_frame_group(sample.device)
        key = (context.data_ptr(), context._version, tuple(context.shape), b0, b1, f0, f1)
        if self._ctx_slice is None or self._ctx_slice[0] != key:
            # the keyed tensor is kept alive with the slice: a recycled address must not look like the same context
            self._ctx_slice = (key, context[b0:b1, f0:f1].contiguous(), context)
        local = self._engine.run(sample[b0:b1, :, f0:f1].float().contiguous(), t[b0:b1].contiguous(),
                                 self._ctx_slice[1], int(cond_frame), use_graph=self.use_graph)
        return sh.gather_output(local, B, Fr).to(sample.dtype)


# =====================================================================================================================
FX_MAX_ROWS = 100_000        # rows at the finest level up to which the LayerNorms are folded (see _Engine._forward)
FX_MAX_ROWS_PB = int(os.environ.get("SEER_FX_MAX_ROWS_PB", "4096"))        # rows per batch element up to which a GroupNorm's statistics are accumulated in fixed point (_Engine._cb)


class _Engine:
    """packed weights + the kernel schedule of one SeerUNet forward."""

    def __init__(self, model: SeerUNet, ops=hip_ops, shard=None, fold_ln=True, dtype=bf16):
        self.ops = ops
        self.dt = dtype                 # 16-bit storage type of activations and packed weights (bf16, or IEEE half under fp16 autocast)
        self.cfg = model.config
        self.shard = shard              # parallel.FrameShard or None
        p0 = next(model.parameters())
        self.device = p0.device
        sd = {k: v.detach() for k, v in model.state_dict().items()}
        self.boc = tuple(self.cfg.block_out_channels)
        self.lpb = self.cfg.layers_per_block
        self.heads = self.cfg.attention_head_dim
        self.G = self.cfg.norm_num_groups
        self.eps = self.cfg.norm_eps
        # GroupNorm statistics from the producers' column sums (ops.ColSums) instead of a pass over the activations, in every
        # engine (single-process and sharded: a shard all-reduces the same (sum, sumsq) either way).  `model.gn_colsums = False`
        # before prepare() keeps the two-stage reduction everywhere (A/B runs, tests).
        self.gn_colsums = bool(getattr(model, "gn_colsums", True))
        # statistics from column sums INSIDE the apply launch (model.gn_fused = False / SEER_GN_FUSED=0: the two-launch form, A/B runs)
        self.gn_fused = bool(getattr(model, "gn_fused", os.environ.get("SEER_GN_FUSED", "1") != "0"))
        # ... and, single-process engines: the producers ACCUMULATE the sums per batch element in fixed point (ops.ColSumsFx) and the
        # apply launch reads them directly -- no finalize launch either (model.gn_fx = False / SEER_GN_FX=0: the forms above)
        self.gn_fx = bool(getattr(model, "gn_fx", os.environ.get("SEER_GN_FX", "1") != "0"))
        self._fx_arena = None
        self._fx_retired: List[object] = []
        self._fx = None
        # LayerNorm folded into the consuming GEMM: the producers of the residual stream leave per-row (sum, sum of squares) next
        # to it (ops.RowStats), the q|k|v / to_q / ff.net.0 GEMMs normalise in their epilogue -- no LayerNorm launch
        # (model.ln_fold = False / SEER_LN_FOLD=0: the layernorm kernel everywhere)
        self.ln_fold = bool(fold_ln) and bool(getattr(model, "ln_fold", os.environ.get("SEER_LN_FOLD", "1") != "0")) and \
            hasattr(self.ops, "fold_layernorm")       # (fold_ln=False: the trainer's engine -- its weights move, it runs its own forward)
        self.ln_folded = 0
        self._ln_on = False
        # ff.net.2 and proj_out as one two-source GEMM (model.ff_fold = False / SEER_FF_FOLD=0: two launches; see _pack)
        self.ff_fold = bool(getattr(model, "ff_fold", os.environ.get("SEER_FF_FOLD", "1") != "0"))
        # ... and, at 320 channels, the whole feed-forward with it as ONE launch (model.ff_fused = False / SEER_FF_FUSED=0: off)
        self.ff_fused = self.ff_fold and bool(getattr(model, "ff_fused", os.environ.get("SEER_FF_FUSED", "1") != "0"))
        # ... and the row-local chains in front of the attention launches at 320 channels -- GroupNorm -> proj_in -> norm1 -> q|k|v, and
        # attn1.to_out + residual -> norm2 -> attn2.to_q -- as ONE launch each (ops.rowchain, csrc/rowchain.hip;
        # model.rowchain = False / SEER_ROWCHAIN=0: the separate launches)
        self.rowchain = bool(getattr(model, "rowchain", os.environ.get("SEER_ROWCHAIN", "1") != "0")) and hasattr(self.ops, "rowchain")
        self.rowchains = 0
        # ... and the block's last to_out + residual as a prologue of the fused feed-forward (SEER_FF_PRE=0: its own launch)
        self.ff_pre = self.rowchain and self.ff_fused and os.environ.get("SEER_FF_PRE", "1") != "0"
        self._ffpre_bias: Dict[str, str] = {}
        self.gn_from_colsums = 0
        self.w: Dict[str, torch.Tensor] = {}
        self._pack(sd)
        self._rot_cache: Dict[Tuple, torch.Tensor] = {}
        self._kv_cache: Dict[str, torch.Tensor] = {}
        self._kv_key = None
        self._graphs: Dict[Tuple, object] = {}     # captured steps (whole-step graphs and sampler-step graphs), LRU: see graph_get / graph_put
        self._rec = None                # _SegmentRecorder while a segmented capture is running
        self._attn_list = None          # list that collects the text cross-attention scores of a return_attn forward
        self._attn_wanted = ()

    # ---- captured graphs: ONE cache, ONE eviction policy ------------------------------------------------------------
    # Every entry owns a private graph memory pool with a full set of step activations, so the cache is small and evicts the
    # single least-recently-used entry (dict order = recency), never everything at once.
    MAX_GRAPHS = 6

    def graph_get(self, key):
        g = self._graphs.get(key)
        if g is not None:
            self._graphs[key] = self._graphs.pop(key)          # most recently used last
        return g

    def graph_put(self, key, g) -> None:
        self._graphs.pop(key, None)
        while len(self._graphs) >= self.MAX_GRAPHS:
            self._graphs.pop(next(iter(self._graphs)))         # the least recently used one
        self._graphs[key] = g

    # ---- weight packing ---------------------------------------------------------------------------------------
    def _pack(self, sd):
        w = self.w
        dev = self.device
        f32 = lambda t: t.to(dev, torch.float32).contiguous()
        b16 = lambda t: t.to(dev, torch.float32).to(self.dt).contiguous()
        self.resnets: List[str] = []
        self.temb_slices: Dict[str, Tuple[int, int]] = {}
        temb_w, temb_b, off = [], [], 0
        for k in sd:
            if k.endswith(".time_emb_proj.weight"):
                p = k[: -len(".time_emb_proj.weight")]
                n = sd[k].shape[0]
                self.temb_slices[p] = (off, n)
                temb_w.append(sd[k]); temb_b.append(sd[p + ".time_emb_proj.bias"])
                off += n
        w["temb_all.w"] = b16(torch.cat(temb_w, 0))
        w["temb_all.b"] = f32(torch.cat(temb_b, 0))
        for k, v in sd.items():
            if ".time_emb_proj." in k or k.endswith("rotary_emb.freqs"):
                if k.endswith("rotary_emb.freqs"):
                    w[k] = f32(v)
                continue
            if k in ("conv_in.weight",):
                w[k] = f32(v.permute(2, 3, 1, 0))                       # [3,3,Cin,Cout]
            elif k in ("conv_out.weight",):
                if v.shape[0] % 4 == 0:
                    w[k] = b16(pack_conv3x3(v))                         # [Cout, 9*C0]: MFMA implicit GEMM, transposed store
                else:
                    w[k] = f32(v.permute(0, 2, 3, 1))                   # [Cout,3,3,C0]: direct kernel
            elif k.endswith(".weight") and v.dim() == 4:
                w[k] = b16(pack_conv3x3(v) if v.shape[-1] == 3 else pack_conv1x1(v))
                if ".upsamplers." in k:
                    # the conv behind the nearest-2x upsample as four 2x2 phase convs (16 instead of 36 tap-products per
                    # source pixel); the 9-tap form stays for the fine-tuning backward (trainer.py)
                    w[k + "_up4"] = b16(pack_conv3x3_up_phases(v.to(torch.float32)))
            elif k.endswith(".weight") and v.dim() == 2:
                if k.endswith("ff.net.0.proj.weight"):
                    order = geglu_row_order(v.shape[0] // 2)
                    w[k] = b16(v[order])
                    w[k[:-6] + "bias"] = f32(sd[k[:-6] + "bias"][order])
                elif k.endswith((".to_q.weight", ".to_k.weight", ".to_v.weight")):
                    continue                                             # fused below
                else:
                    w[k] = b16(v)
            elif k.endswith("ff.net.0.proj.bias"):
                continue
            else:
                w[k] = f32(v)                                            # biases, norm affine
        for k in sd:
            if k.endswith(".attn1.to_q.weight"):
                p = k[: -len(".to_q.weight")]
                w[p + ".qkv"] = b16(torch.cat([sd[p + ".to_q.weight"], sd[p + ".to_k.weight"], sd[p + ".to_v.weight"]], 0))
            elif k.endswith(".attn2.to_q.weight"):
                p = k[: -len(".to_q.weight")]
                w[p + ".q"] = b16(sd[p + ".to_q.weight"])
                w[p + ".kv"] = b16(torch.cat([sd[p + ".to_k.weight"], sd[p + ".to_v.weight"]], 0))
        # ff.net.2 followed by proj_out (attention.py:742-747 then :126,141-145): two consecutive LINEAR maps with only the residual
        # add of the block between them -- x + proj_out(h + ff2(g)) = x + [Wp | Wp W2] [h | g] + (Wp b2 + bp) -- run as ONE two-source
        # GEMM over K = C + 4C (the same FLOPs: the second map's K = C rides in the first one's K loop).  The product Wp W2 is formed
        # once, in fp32, from the fp32 weights.  One launch and one round trip of the block's residual stream fewer per transformer
        # block, 32 per step.  The plain weights stay (row subsets under cond_frame > 0, return_attn, the training engine).
        if self.ff_fold:
            for k in sd:
                if not k.endswith(".proj_out.weight") or (".attentions." not in k and ".temporal_attentions." not in k):
                    continue
                pth = k[: -len(".proj_out.weight")]
                tb = pth + ".transformer_blocks.0"
                if (tb + ".ff.net.2.weight") not in sd:
                    continue
                wp = pack_conv1x1(sd[k]).to(dev, torch.float32)
                w2, b2 = f32(sd[tb + ".ff.net.2.weight"]), f32(sd[tb + ".ff.net.2.bias"])
                w[pth + ".ffproj.w"] = torch.cat([wp, wp @ w2], dim=1).to(self.dt).contiguous()           # [C, C + 4C]
                w[pth + ".ffproj.b"] = (wp @ b2 + f32(sd[pth + ".proj_out.bias"])).contiguous()
                # 320 channels: norm3, ff.net.0, GEGLU and this GEMM run as ONE launch (ops.ff_fused, csrc/ff_fused.hip), which reads
                # both matrices in its own fragment order
                if self.ff_fused and wp.shape[0] == getattr(self.ops, "FF_FUSED_C", -1) and (tb + ".norm3.weight") in sd:
                    w[pth + ".ff_fused.w1f"], w[pth + ".ff_fused.wcf"] = self.ops.ff_fused_pack(
                        w[tb + ".ff.net.0.proj.weight"], w[pth + ".ffproj.w"])
        # the chains of ops.rowchain read their 320 x 320 matrices in the kernel's fragment order
        if self.rowchain:
            RC = self.ops.ROWCHAIN_C
            for k in list(w):
                if k.endswith(".proj_in.weight") and w[k].shape == (RC, RC) and (k[:-len(".proj_in.weight")] + ".transformer_blocks.0.attn1.qkv") in w:
                    pth = k[: -len(".proj_in.weight")]
                    tb = pth + ".transformer_blocks.0"
                    w[pth + ".rc.proj_in"] = self.ops.rowchain_pack(w[k])
                    w[tb + ".rc.qkv"] = self.ops.rowchain_pack(w[tb + ".attn1.qkv"])
                    if (tb + ".attn2.q") in w:
                        w[tb + ".rc.to_out"] = self.ops.rowchain_pack(w[tb + ".attn1.to_out.0.weight"])
                        w[tb + ".rc.q"] = self.ops.rowchain_pack(w[tb + ".attn2.q"])
                    # the LAST to_out of the block (text: attn2, temporal: attn1) as the fused feed-forward's prologue (ops.ff_fused(pre=))
                    last = tb + (".attn2" if (tb + ".attn2.q") in w else ".attn1") + ".to_out.0"
                    if (pth + ".ff_fused.w1f") in w and self.ff_pre:
                        w[tb + ".ffpre.wo"] = self.ops.rowchain_pack(w[last + ".weight"])
                        self._ffpre_bias[tb] = last + ".bias"
        # LayerNorm folded into the GEMM that consumes it (ops.fold_layernorm): W' = gamma (.) W from the fp32 weights, its row
        # sums and beta W^T + b, next to the plain weights (a launch that cannot fold runs layernorm + the plain ones)
        self.wln: Dict[str, Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = {}
        if self.ln_fold:
            fold = self.ops.fold_layernorm if self.dt == bf16 else (lambda *a: self.ops.fold_layernorm(*a, dtype=self.dt))
            for k in sd:
                if not k.endswith(".norm1.weight") or ".transformer_blocks." not in k:
                    continue
                tb = k[: -len(".norm1.weight")]
                g = lambda n: (f32(sd[f"{tb}.{n}.weight"]), f32(sd[f"{tb}.{n}.bias"]))
                qkv = torch.cat([sd[tb + ".attn1.to_q.weight"], sd[tb + ".attn1.to_k.weight"], sd[tb + ".attn1.to_v.weight"]], 0)
                self.wln[tb + ".attn1.qkv"] = fold(f32(qkv), *g("norm1"))
                if (tb + ".attn2.to_q.weight") in sd and (tb + ".norm2.weight") in sd:
                    self.wln[tb + ".attn2.q"] = fold(f32(sd[tb + ".attn2.to_q.weight"]), *g("norm2"))
                if (tb + ".norm3.weight") in sd:
                    v = sd[tb + ".ff.net.0.proj.weight"]
                    order = geglu_row_order(v.shape[0] // 2)
                    self.wln[tb + ".ff.net.0.proj.weight"] = fold(f32(v[order]), *g("norm3"), f32(sd[tb + ".ff.net.0.proj.bias"][order]))

    # ---- building blocks --------------------------------------------------------------------------------------
    def _gn(self, x1, x2, B, rows_pb, name, eps, silu):
        """GroupNorm over (C/G, F, H, W): stats (+ cross-shard reduction) then apply."""
        ops = self.ops
        stats = self._stats_arena[self._stats_i]
        self._stats_i += 1
        # statistics from the column sums the producing GEMM / conv left next to its output (ops.ColSums) when every source
        # has them: no pass over the activations; otherwise the two-stage reduction over x1 | x2
        cs1 = getattr(x1, "colsums", None)
        cs2 = getattr(x2, "colsums", None) if x2 is not None else None
        C = x1.shape[1] + (0 if x2 is None else x2.shape[1])
        FX = getattr(ops, "ColSumsFx", ())
        if self.shard is not None and self.shard.exact_stats and self._fx is not None and hasattr(ops, "groupnorm_stats_fx"):
            # frame shards (P > 1; batch groups alone exchange nothing and keep the forms below): EVERY GroupNorm normalises with exact integer statistics -- a source whose producer left no
            # accumulated sums (conv_in's output, tensors above the producers' row limit) gets them from one pass over its rows;
            # the sums stay with the tensor (a skip connection feeds a second GroupNorm with the totals already exchanged)
            if not isinstance(cs1, FX):
                cs1 = x1.colsums = ops.groupnorm_stats_fx(x1, B, self._fx)
            if x2 is not None and not isinstance(cs2, FX):
                cs2 = x2.colsums = ops.groupnorm_stats_fx(x2, B, self._fx)
        exchanged = False
        if isinstance(cs1, FX) or isinstance(cs2, FX):
            if isinstance(cs1, FX) and (x2 is None or isinstance(cs2, FX)):
                count = rows_pb * (C // self.G)
                if self.shard is not None:
                    # frame shards: the integer sums of all shards are added in place (exact: the statistics are the unsharded ones)
                    count = self.shard.reduce_fx((cs1, cs2), count, sync=self.sync_point)
                    exchanged = True
                y = ops.groupnorm_apply_fx(x1, x2, cs1, cs2, B, self.G, count, eps,
                                           self.w[name + ".weight"], self.w[name + ".bias"], silu)
                self.gn_from_colsums += 1
                if y is not None:
                    return y
                ops.groupnorm_stats_from_fx(cs1, cs2, B, self.G, stats)
            else:
                ops.groupnorm_stats(x1, x2, B, self.G, stats)
        elif self.gn_colsums and cs1 is not None and (x2 is None or cs2 is not None):
            if self.shard is None and self.gn_fused:
                # one launch: every apply block re-derives the statistics of its own groups from the column sums (no frame
                # shards: a sharded run all-reduces the statistics between the two steps)
                y = ops.groupnorm_apply_from_colsums(x1, x2, cs1, cs2, B, self.G, rows_pb * (C // self.G), eps,
                                                     self.w[name + ".weight"], self.w[name + ".bias"], silu)
                if y is not None:
                    self.gn_from_colsums += 1
                    return y
            ops.groupnorm_stats_from_colsums(cs1, cs2, B, self.G, stats)
            self.gn_from_colsums += 1
        else:
            ops.groupnorm_stats(x1, x2, B, self.G, stats)
        count = rows_pb * (C // self.G)
        if self.shard is not None:
            if exchanged:           # the statistics above came from sums that already hold every shard's share
                count = count / self.shard.local_frames * self.shard.total_frames
            else:
                count = self.shard.reduce_gn_stats(stats, count, sync=self.sync_point)
        return ops.groupnorm_apply(x1, x2, B, self.G, stats, count, eps, self.w[name + ".weight"],
                                   self.w[name + ".bias"], silu)

    def _cb(self, B, rows_pb, for_chain=False):
        """`colsum_batch` of a launch whose output (rows_pb rows per batch element) feeds a GroupNorm: (B, arena) = accumulate in
        fixed point, B = per-tile sums.  The accumulated form wins where the tensor is small (its apply launch owns slices of
        64..128 channels: 160-byte pieces of a 640-byte row at the 32x32 level, ~20 % below the full-row kernel's bandwidth, and
        eight replicas to add per block): from the 16x16 level down 7.5 / 9.6 / 5.7 / 3.8 us against 8.4 / 12.1 / 7.6 / 5.7 for
        finalize + apply, at the 32x32 level 12.6 / 27.9 against 11.3 / 25.6 (profiles/r04_gn_fx_by_level.log)."""
        if not self.gn_colsums:
            return 0
        return (B, self._fx) if (self._fx is not None and (rows_pb <= FX_MAX_ROWS_PB or for_chain)) else B

    def _chain_next(self, C, B, rows_pb):
        """will a transformer's GroupNorm -> proj_in -> norm1 -> q|k|v over [B * rows_pb, C] run as one launch (ops.rowchain)?  Its
        producer then ACCUMULATES the column sums whatever the tensor's size (_cb(for_chain=True)): the chain reads them directly and
        the statistics launch in front of it goes too (the apply launch that made the accumulated form lose at the 32x32 level is
        not run at all there)"""
        ops = self.ops
        return bool(self.rowchain and (self.shard is None or not self.shard.exact_stats) and self._fx is not None and
                    C == getattr(ops, "ROWCHAIN_C", -1) and
                    rows_pb >= ops.ROWCHAIN_ROWS and ops.rowchain_pays(B * rows_pb))

    def _gn_stats(self, x, B, rows_pb):
        """(statistics, count) of the GroupNorm over x alone, for a launch that applies the normalisation itself (ops.rowchain): the
        producer's accumulated fixed-point sums as they are, else stats [B, G, 2] from its per-tile column sums or from a pass over x.
        None: a frame-sharded engine -- the caller keeps the separate launches (batch groups alone run the single-process forms)."""
        ops = self.ops
        cs = getattr(x, "colsums", None)
        if self.shard is not None and self.shard.exact_stats:      # frame shards exchange the statistics between the two steps
            return None
        if isinstance(cs, getattr(ops, "ColSumsFx", ())):
            self.gn_from_colsums += 1
            return cs, rows_pb * (x.shape[1] // self.G)         # the producer's accumulated sums: read by the chain itself
        stats = self._stats_arena[self._stats_i]
        self._stats_i += 1
        if self.gn_colsums and cs is not None:
            ops.groupnorm_stats_from_colsums(cs, None, B, self.G, stats)
            self.gn_from_colsums += 1
        else:
            ops.groupnorm_stats(x, None, B, self.G, stats)
        return stats, rows_pb * (x.shape[1] // self.G)

    def _rc_in(self, p, tb, x, B, rows_pb, rotary, qs):
        """GroupNorm -> proj_in -> norm1 -> q|k|v of transformer `p` as one launch: (h, qkv), or None when this block / shape keeps
        the separate launches"""
        ops, w = self.ops, self.w
        if not self.rowchain or (p + ".rc.proj_in") not in w or rows_pb < ops.ROWCHAIN_ROWS or not ops.rowchain_pays(x.shape[0]):
            return None
        i0 = self._stats_i
        st = self._gn_stats(x, B, rows_pb)
        if st is None:
            return None
        r = ops.rowchain(x, w[p + ".rc.proj_in"], b1=w[p + ".proj_in.bias"],
                         gn=(st[0], st[1], 1e-6, w[p + ".norm.weight"], w[p + ".norm.bias"], rows_pb, self.G),
                         ln=(w[tb + ".norm1.weight"], w[tb + ".norm1.bias"], 1e-5), w2f=w[tb + ".rc.qkv"], col_scale=(qs, 1), rotary=rotary)
        if r is None:
            self._stats_i = i0
            return None
        self.rowchains += 1
        return r

    def _resnet(self, p, x, skip, geo, feeds_transformer=False):
        """ResnetBlock3D (resnet.py:174-208); `skip` is the channel-concat partner of unet_3d_blocks.py:596,712.  feeds_transformer:
        the output's only GroupNorm is the next transformer's (see _chain_next)."""
        ops, w = self.ops, self.w
        B, Fr, H, W = geo
        rows_pb = Fr * H * W
        cb = self._cb(B, rows_pb)          # outputs that feed a GroupNorm leave their column sums behind
        h = self._gn(x, skip, B, rows_pb, p + ".norm1", self.eps, True)
        off, n = self.temb_slices[p]
        temb = self._temb[:, off:off + n]
        h = ops.conv3x3(h, w[p + ".conv1.weight"], B * Fr, H, W, bias=w[p + ".conv1.bias"], rowvec=temb,
                        rows_per_batch=rows_pb, colsum_batch=cb)
        h = self._gn(h, None, B, rows_pb, p + ".norm2", self.eps, True)
        if (p + ".conv_shortcut.weight") in w:
            sc = ops.gemm(x, w[p + ".conv_shortcut.weight"], a2=skip, bias=w[p + ".conv_shortcut.bias"])
        else:
            assert skip is None
            sc = x
        if feeds_transformer and self._chain_next(w[p + ".conv2.weight"].shape[0], B, rows_pb):
            cb = self._cb(B, rows_pb, for_chain=True)
        return ops.conv3x3(h, w[p + ".conv2.weight"], B * Fr, H, W, bias=w[p + ".conv2.bias"], residual=sc, colsum_batch=cb)

    def _ln_gemm(self, h, tb, norm, wkey, bias_key=None, **kw):
        """LayerNorm `tb + norm` followed by the projection `wkey`: one GEMM when the producer of h left its row statistics and the
        launch can fold (ops.gemm(..., ln=)), else the layernorm kernel and the plain weights."""
        ops, w = self.ops, self.w
        rs = getattr(h, "rowstats", None)
        f = self.wln.get(wkey) if rs is not None else None
        if f is not None:
            y = ops.gemm(h, f[0], bias=f[2], ln=(rs, f[1], 1e-5), **kw)
            if y is not None:
                self.ln_folded += 1
                return y
        n = ops.layernorm(h, w[tb + norm + ".weight"], w[tb + norm + ".bias"])
        return ops.gemm(n, w[wkey], bias=(w[bias_key] if bias_key else None), **kw)

    def _rs(self):
        """extra arguments of a GEMM whose output rows feed a LayerNorm"""
        return {"rowstat": self._fx_arena if self._fx_arena is not None else True} if self._ln_on else {}

    def _ff(self, tb, h_rows):
        ops, w = self.ops, self.w
        g = self._ln_gemm(h_rows, tb, ".norm3", tb + ".ff.net.0.proj.weight", tb + ".ff.net.0.proj.bias", geglu=True)
        ops.gemm(g, w[tb + ".ff.net.2.weight"], bias=w[tb + ".ff.net.2.bias"], residual=h_rows, out=h_rows)

    def _ff_fused_rows(self, p, h):
        """will _ff_proj_out run transformer `p`'s feed-forward over the rows of h as the fused launch?"""
        return (p + ".ff_fused.w1f") in self.w and self.ops.ff_fused_pays(h.shape[0])

    def _ff_pre(self, p, tb, h):
        """will the block's last to_out + residual ride in the fused feed-forward launch (ops.ff_fused(pre=))?"""
        return (tb + ".ffpre.wo") in self.w and self._ff_fused_rows(p, h)

    def _ff_proj_out(self, p, tb, h, x, cb, a=None):
        """the feed-forward of block `tb` and the transformer's proj_out + residual x: folded into one two-source GEMM when the
        block's weights were (see _pack), else ff.net.2 + residual and proj_out + residual as two launches.  `a` (only with
        _ff_pre): the attention output whose to_out projection + residual h the fused launch computes itself."""
        ops, w = self.ops, self.w
        if self._ff_fused_rows(p, h):
            pre = None if a is None else (a, w[tb + ".ffpre.wo"], w[self._ffpre_bias[tb]])
            y = ops.ff_fused(h, x, w[tb + ".norm3.weight"], w[tb + ".norm3.bias"], w[p + ".ff_fused.w1f"],
                             w[tb + ".ff.net.0.proj.bias"], w[p + ".ff_fused.wcf"], w[p + ".ffproj.b"], colsum_batch=cb, pre=pre)
            if y is not None:
                return y
        assert a is None, "the to_out prologue exists only inside the fused feed-forward launch"
        if (p + ".ffproj.w") in w:
            g = self._ln_gemm(h, tb, ".norm3", tb + ".ff.net.0.proj.weight", tb + ".ff.net.0.proj.bias", geglu=True)
            return ops.gemm(h, w[p + ".ffproj.w"], a2=g, bias=w[p + ".ffproj.b"], residual=x, colsum_batch=cb)
        self._ff(tb, h)
        return ops.gemm(h, w[p + ".proj_out.weight"], bias=w[p + ".proj_out.bias"], residual=x, colsum_batch=cb)

    def _text_transformer(self, p, x, geo):
        """SpatialTransformer3D + BasicTextTransformerBlock3D (attention.py:129-145, 308-327)."""
        ops, w = self.ops, self.w
        B, Fr, H, W = geo
        C = x.shape[1]
        heads, d = self.heads, C // self.heads
        HW = H * W
        tb = p + ".transformer_blocks.0"
        # the q columns leave the projection as q * scale * log2(e) (one bf16 rounding): the attention kernels exponentiate
        # the raw dot products
        qs = ops.qk_prescale(d)
        rc = self._rc_in(p, tb, x, B, Fr * HW, None, qs)
        if rc is not None:
            h, qkv = rc
        else:
            hn = self._gn(x, None, B, Fr * HW, p + ".norm", 1e-6, False)
            h = ops.gemm(hn, w[p + ".proj_in.weight"], bias=w[p + ".proj_in.bias"], **self._rs())
            qkv = self._ln_gemm(h, tb, ".norm1", tb + ".attn1.qkv", col_scale=(qs, C))
        # self attention per frame
        a = torch.empty_like(h)
        ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], a, batch=B * Fr, heads=heads, head_dim=d,
                      Sq=HW, Sk=HW, q_prescaled=True)
        # text cross attention per frame (K/V depend on the context only: cached across DDIM steps)
        q = None
        if rc is not None and (tb + ".rc.q") in w and ops.rowchain_pays(a.shape[0], products=2):
            # attn1.to_out + residual (over h) -> norm2 -> attn2.to_q, one launch
            r2 = ops.rowchain(a, w[tb + ".rc.to_out"], b1=w[tb + ".attn1.to_out.0.bias"], res=h, h_out=h,
                              ln=(w[tb + ".norm2.weight"], w[tb + ".norm2.bias"], 1e-5), w2f=w[tb + ".rc.q"], col_scale=(qs, 1))
            if r2 is not None:
                q = r2[1]
                self.rowchains += 1
        if q is None:
            ops.gemm(a, w[tb + ".attn1.to_out.0.weight"], bias=w[tb + ".attn1.to_out.0.bias"], residual=h, out=h, **self._rs())
            q = self._ln_gemm(h, tb, ".norm2", tb + ".attn2.q", col_scale=(qs, C))
        kv = self._kv_cache.get(tb)
        if kv is None:
            kv = ops.gemm(self._ctx, w[tb + ".attn2.kv"])
            self._kv_cache[tb] = kv
        L = self._ctx_len
        if self._attn_list is not None and p in self._attn_wanted:
            self._attn_list.append(self._cross_scores(q, kv[:, :C], B, Fr, H, W, heads, d, L))
        ops.attention(q, kv[:, :C], kv[:, C:], a, batch=B * Fr, heads=heads, head_dim=d, Sq=HW, Sk=L, q_prescaled=True)
        # (the output's only GroupNorm is the temporal transformer's: accumulated sums when that one runs as a chain)
        cb = self._cb(B, Fr * HW, for_chain=self._chain_next(C, B, Fr * HW))
        if self._ff_pre(p, tb, h):
            return self._ff_proj_out(p, tb, h, x, cb, a=a)      # attn2.to_out + residual inside the fused feed-forward launch
        # (the fused feed-forward normalises its rows itself: no row statistics asked of their producer)
        ops.gemm(a, w[tb + ".attn2.to_out.0.weight"], bias=w[tb + ".attn2.to_out.0.bias"], residual=h, out=h,
                 **({} if self._ff_fused_rows(p, h) else self._rs()))
        return self._ff_proj_out(p, tb, h, x, cb)

    def _cross_scores(self, q, k, B, Fr, H, W, heads, d, L):
        """`attention_scores` of the text cross attention (attention.py:556-584: scale * Q K^T before the softmax) as
        [b, heads, f, h, w, L] (attention.py:320).  q carries scale * log2(e) (one bf16 rounding, as the attention kernel reads
        it): the scores are one batched MFMA GEMM over the head-split, zero-padded operands with 1 / log2(e) in its epilogue."""
        HW, Bt = H * W, B * Fr * heads
        dp, Lp = (d + 63) // 64 * 64, (L + 3) // 4 * 4
        qh = torch.zeros((Bt, HW, dp), device=q.device, dtype=q.dtype)
        kh = torch.zeros((Bt, Lp, dp), device=q.device, dtype=q.dtype)
        qh[:, :, :d] = q.reshape(B * Fr, HW, heads, d).permute(0, 2, 1, 3).reshape(Bt, HW, d)
        kh[:, :L, :d] = k.reshape(B * Fr, L, heads, d).permute(0, 2, 1, 3).reshape(Bt, L, d)
        s = self.ops.gemm_batched(qh, kh, out_f32=True, col_scale=(1.0 / self.ops.LOG2E, Lp))
        return s[:, :, :L].reshape(B, Fr, heads, H, W, L).permute(0, 2, 1, 3, 4, 5).contiguous()

    def _rotary_table(self, tb, T):
        freqs = self.w[tb + ".attn1.rotary_emb.freqs"]
        key = (freqs.data_ptr(), T)
        t = self._rot_cache.get(key)
        if t is None:
            fk = (tuple(freqs.tolist()), T)      # identical buffers share one table
            t = self._rot_cache.get(fk)
            if t is None:
                t = self.ops.rotary_table(freqs, T)
                self._rot_cache[fk] = t
            self._rot_cache[key] = t
        return t

    def _temporal_transformer(self, p, x, geo, cond_frame):
        """SpatialTransformer3D + BasicTransformerBlock3D(temporal) + WindowSTempAttention
        (attention.py:129-145, 231-248, 632-703)."""
        ops, w = self.ops, self.w
        B, Fr, H, W = geo
        C = x.shape[1]
        heads, d = self.heads, C // self.heads
        HW = H * W
        tb = p + ".transformer_blocks.0"
        F_all = Fr if self.shard is None else self.shard.total_frames
        f_off = 0 if self.shard is None else self.shard.frame_offset
        rot_dim = min(32, d)
        cs = self._rotary_table(tb, F_all * HW)
        rc = self._rc_in(p, tb, x, B, Fr * HW, (cs, Fr * HW, f_off * HW, d, rot_dim, 2), ops.qk_prescale(d))
        if rc is not None:
            h, qkv = rc
        else:
            hn = self._gn(x, None, B, Fr * HW, p + ".norm", 1e-6, False)
            h = ops.gemm(hn, w[p + ".proj_in.weight"], bias=w[p + ".proj_in.bias"], **self._rs())
            # q|k|v projection with the rotary embedding applied to the q and k columns in the GEMM epilogue
            qkv = self._ln_gemm(h, tb, ".norm1", tb + ".attn1.qkv", rotary=(cs, Fr * HW, f_off * HW, d, rot_dim, 2 * C),
                                col_scale=(ops.qk_prescale(d), C))
        a = torch.empty_like(h)
        if self.shard is not None:
            self.shard.temporal_attention(ops, qkv, a, B, heads, d, H, W, sync=self.sync_point)
        else:
            if H > MIN_WIN_SIZE:
                ws = MAX_WIN_SIZE if (H // MAX_WIN_SIZE) >= MAX_RATIO else MIN_WIN_SIZE
                ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], a, batch=B, heads=heads, head_dim=d,
                              Sq=Fr * ws * ws, Sk=Fr * ws * ws, causal=True, window=(ws, Fr, H, W), q_prescaled=True)
            else:
                ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], a, batch=B, heads=heads, head_dim=d,
                              Sq=Fr * HW, Sk=Fr * HW, causal=True, q_prescaled=True)
        # FF skips the conditioning frames (attention.py:241-246); frames are the slow index inside a batch element
        skip_f = cond_frame if self.shard is None else self.shard.local_cond_frames(cond_frame)
        if skip_f <= 0 and self._ff_pre(p, tb, h):
            return self._ff_proj_out(p, tb, h, x, self._cb(B, Fr * HW), a=a)    # attn1.to_out + residual inside the fused feed-forward launch
        ops.gemm(a, w[tb + ".attn1.to_out.0.weight"], bias=w[tb + ".attn1.to_out.0.bias"], residual=h, out=h,
                 **({} if skip_f <= 0 and self._ff_fused_rows(p, h) else self._rs()))
        if skip_f <= 0:
            return self._ff_proj_out(p, tb, h, x, self._cb(B, Fr * HW))
        elif skip_f < Fr:
            for b in range(B):
                self._ff(tb, h[b * Fr * HW + skip_f * HW:(b + 1) * Fr * HW])
        return ops.gemm(h, w[p + ".proj_out.weight"], bias=w[p + ".proj_out.bias"], residual=x,
                        colsum_batch=self._cb(B, Fr * HW))

    # ---- the schedule ---------------------------------------------------------------------------------------------
    def n_groupnorms(self):
        return sum(1 for k in self.w if k.endswith(".weight") and
                   (k.endswith("norm1.weight") or k.endswith("norm2.weight") or k.endswith(".norm.weight") or
                    k == "conv_norm_out.weight") and "transformer_blocks" not in k)

    def _forward(self, sample, t, ctx_bf16, ctx_len, cond_frame, return_attn=False):
        ops, w = self.ops, self.w
        B, Cin, Fr, H, W = sample.shape
        boc, lpb, n = self.boc, self.lpb, len(self.boc)
        self._attn_list = [] if return_attn else None
        # the last text block of each container: the reference's containers overwrite attn_map layer by layer
        self._attn_wanted = {f"down_blocks.{i}.attentions.{lpb - 1}" for i in range(n - 1)} | {"mid_block.attentions.0"} | \
                            {f"up_blocks.{i}.attentions.{lpb}" for i in range(1, n)}
        self._ctx, self._ctx_len = ctx_bf16, ctx_len
        self._stats_arena = torch.empty((self.n_groupnorms(), B, self.G, 2), device=sample.device, dtype=torch.float32)
        self._stats_i = 0
        self.gn_from_colsums = 0        # GroupNorms of this forward that took their statistics from column sums
        self.ln_folded = 0              # LayerNorms of this forward that ran inside the consuming GEMM
        self.rowchains = 0              # ops.rowchain launches of this forward
        self._fx = None
        # the folded LayerNorm trades a launch per norm for atomics in proportion to the rows: ahead up to ~100 k rows at the finest
        # level (config 2: 24 576 rows -0.19 ms; 64x64 latent, 98 304 rows: even; bridge, 131 072 rows: +0.15 ms --
        # profiles/r04_fx_ln_other_configs.log), off above
        small = B * Fr * H * W <= FX_MAX_ROWS
        fx_gn = self.gn_fx and self.gn_colsums      # (per tensor: _cb; frame shards all-reduce the integer sums: FrameShard.reduce_fx)
        self._ln_on = self.ln_fold and small
        if (fx_gn or self._ln_on) and hasattr(ops, "FxArena"):
            # fixed-point accumulators of the evaluation: a [reps, B, 2, C] slot per colsum producer, a [rows, 2] slot per producer
            # of LayerNorm rows; reset() = one fill over what the last evaluation took
            need = (self.n_groupnorms() + 16) * B * 4 * max(boc) * 2 + 5 * 16 * B * Fr * H * W * 2
            if self._fx_arena is None or self._fx_arena.buf.numel() < need:
                assert not (sample.is_cuda and torch.cuda.is_current_stream_capturing()), \
                    "the accumulator arena must exist before a graph capture"
                if self._fx_arena is not None:
                    self._fx_retired.append(self._fx_arena)     # captured steps of smaller shapes keep adding into theirs by address
                self._fx_arena = ops.FxArena(sample.device, need)
            self._fx_arena.reset()
            self._fx = self._fx_arena if fx_gn else None
        emb = ops.timestep_embedding(t, boc[0], self.cfg.flip_sin_to_cos, self.cfg.freq_shift)
        emb = ops.linear_smallm(emb, w["time_embedding.linear_1.weight"], w["time_embedding.linear_1.bias"], silu_out=True)
        emb = ops.linear_smallm(emb, w["time_embedding.linear_2.weight"], w["time_embedding.linear_2.bias"])
        self._temb = ops.linear_smallm(emb, w["temb_all.w"], w["temb_all.b"], silu_in=True)
        # (on a side stream beside conv_in -- a fork / join in the step's graph -- this chain cost the step +0.35 ms:
        #  profiles/r06_temb_fork_rejected.log)

        x = ops.conv_in(sample, w["conv_in.weight"], w["conv_in.bias"]) if self.dt == bf16 else \
            ops.conv_in(sample, w["conv_in.weight"], w["conv_in.bias"], dtype=self.dt)
        skips = [x]
        geo = (B, Fr, H, W)
        for i in range(n):
            p = f"down_blocks.{i}"
            for j in range(lpb):
                x = self._resnet(f"{p}.resnets.{j}", x, None, geo, feeds_transformer=i < n - 1)
                if i < n - 1:
                    x = self._text_transformer(f"{p}.attentions.{j}", x, geo)
                    x = self._temporal_transformer(f"{p}.temporal_attentions.{j}", x, geo, cond_frame)
                skips.append(x)
            if i < n - 1:
                x = ops.conv3x3(x, w[f"{p}.downsamplers.0.conv.weight"], B * Fr, geo[2], geo[3], stride=2,
                                bias=w[f"{p}.downsamplers.0.conv.bias"],
                                colsum_batch=self._cb(B, Fr * ((geo[2] - 1) // 2 + 1) * ((geo[3] - 1) // 2 + 1)))
                geo = (B, Fr, (geo[2] - 1) // 2 + 1, (geo[3] - 1) // 2 + 1)
                skips.append(x)
        x = self._resnet("mid_block.resnets.0", x, None, geo, feeds_transformer=True)
        x = self._text_transformer("mid_block.attentions.0", x, geo)
        x = self._temporal_transformer("mid_block.temporal_attentions.0", x, geo, cond_frame)
        x = self._resnet("mid_block.resnets.1", x, None, geo)
        for i in range(n):
            p = f"up_blocks.{i}"
            for j in range(lpb + 1):
                x = self._resnet(f"{p}.resnets.{j}", x, skips.pop(), geo, feeds_transformer=i > 0)
                if i > 0:
                    x = self._text_transformer(f"{p}.attentions.{j}", x, geo)
                    x = self._temporal_transformer(f"{p}.temporal_attentions.{j}", x, geo, cond_frame)
            if i < n - 1:
                x = ops.conv_up2x(x, w[f"{p}.upsamplers.0.conv.weight_up4"], B * Fr, geo[2], geo[3],
                                  bias=w[f"{p}.upsamplers.0.conv.bias"], colsum_batch=self._cb(B, Fr * 4 * geo[2] * geo[3]))
                geo = (B, Fr, geo[2] * 2, geo[3] * 2)
        x = self._gn(x, None, B, Fr * geo[2] * geo[3], "conv_norm_out", self.eps, True)
        out = ops.conv_out(x, w["conv_out.weight"], w["conv_out.bias"], B, Fr, geo[2], geo[3])
        if return_attn:
            attn, self._attn_list = self._attn_list, None
            return out, attn
        return out

    # ---- entry ---------------------------------------------------------------------------------------------------------
    def _context(self, context: torch.Tensor):
        """context [B, F, L, Dc] fp32 -> bf16 [B*F*L, Dc] plus the cross-attention K|V projections of the 16 text blocks, which
        depend on the context only: they run once per prompt instead of once per DDIM step.  A change is detected by the
        tensor's identity and version (the engine keeps the keyed tensor alive: a freed context's address handed to the next
        prompt must not look like a cache hit).  The bf16 context and the K|V tensors are STATIC buffers: a new prompt of the
        same shape is written into them in place, so the captured hipGraphs -- which read them by address -- stay valid and a
        new prompt costs 1 + 16 small launches, not a re-capture."""
        key = (context.data_ptr(), context._version, tuple(context.shape), context.dtype)
        if key != self._kv_key:     # (an engine is built per storage type: the cached K|V never mix types)
            c = context.reshape(-1, context.shape[-1])
            new = (self.ops.cast_bf16(c.float()) if self.dt == bf16 else self.ops.cast_bf16(c.float(), self.dt)) if c.dtype != self.dt else c
            old = getattr(self, "_ctx_bf16", None)
            if old is not None and old.shape == new.shape and self._kv_cache:
                old.copy_(new)
                for tb, kv in self._kv_cache.items():          # refresh in place, in the order the blocks first ran
                    self.ops.gemm(old, self.w[tb + ".attn2.kv"], out=kv)
            else:                                              # first prompt, or another context shape: start over
                self._ctx_bf16 = new.contiguous().clone() if new.data_ptr() == c.data_ptr() else new
                self._kv_cache = {}
                self._graphs.clear()
            self._kv_key = key
            self._kv_ctx_ref = context
        return self._ctx_bf16, context.shape[-2]

    def run(self, sample, t, context, cond_frame, use_graph=False, return_attn=False):
        if context.dim() == 3:      # [B, L, Dc] -> same text for every frame
            context = context[:, None].expand(-1, sample.shape[2], -1, -1)
        assert context.shape[0] == sample.shape[0] and context.shape[1] == sample.shape[2], \
            f"context {tuple(context.shape)} does not match sample {tuple(sample.shape)} (need [B, F, L, D])"
        H, W = sample.shape[-2:]
        if H % 8 or W % 8:
            raise ValueError("latent height/width must be multiples of 8 (three stride-2 levels + 4/8 windows)")
        ctx, L = self._context(context)
        if return_attn:
            return self._forward(sample, t, ctx, L, cond_frame, return_attn=True)
        if not use_graph or getattr(self, "_graph_broken", False):
            return self._forward(sample, t, ctx, L, cond_frame)
        return self._run_graph(sample, t, ctx, L, cond_frame)

    def sync_point(self, fn):
        """run `fn` -- a cross-rank exchange (all-reduce / all-gather on static buffers) -- at this point of the schedule.
        RCCL collectives are captured into the step's graph with everything else.  Backends that cannot be captured (gloo)
        get a segmented capture: the current hipGraph segment ends, `fn` is recorded as an eager replay step and the next
        segment opens."""
        rec = self._rec
        if rec is None or (self.shard is not None and self.shard.capture_collectives):
            fn()            # eager step, or an RCCL collective recorded into the graph being captured
            return
        rec.end_segment()
        fn()
        rec.steps.append(fn)
        rec.begin_segment()

    def _run_graph(self, sample, t, ctx, L, cond_frame):
        """hipGraph replay of the shape-static step: ~1.3k launches -> one graph launch, or -- frame-sharded -- one graph
        launch per stretch between two collectives (GroupNorm statistics / K|V exchanges stay eager torch.distributed
        calls on the same stream)."""
        key = (tuple(sample.shape), L, cond_frame, tuple(ctx.shape))
        g = self.graph_get(key)
        if g is None:
            # warm up eagerly (fills the K/V and rotary caches, creates process groups, lets allocations settle), then capture
            s_in, t_in = sample.clone(), t.clone()
            self._forward(s_in, t_in, ctx, L, cond_frame)
            torch.cuda.synchronize()
            sharded = self.shard is not None and self.shard.world > 1
            err = None
            for attempt in range(2):
                rec = _SegmentRecorder()
                self._rec = rec
                try:
                    rec.begin_segment()
                    out = self._forward(s_in, t_in, ctx, L, cond_frame)
                    rec.end_segment()
                    err = None
                except Exception as e:      # noqa: BLE001  capture refused (driver / RCCL state)
                    rec.abort()
                    err = e
                finally:
                    self._rec = None
                if not sharded:
                    break
                # Sharded: the ranks decide TOGETHER.  While collectives are captured nothing is exchanged during capture, so a
                # rank whose capture failed has not left its peers waiting; everybody learns of the failure here and repeats
                # the capture with eager exchanges between graph segments (where a failure can only be symmetric).
                if self.shard.agree(err is None, self.device):
                    break
                if not self.shard.capture_collectives or attempt == 1:
                    raise err if err is not None else RuntimeError("hipGraph capture of the sharded step failed on another rank")
                self.shard.capture_collectives = False
                import warnings
                warnings.warn("capturing the RCCL exchanges into the step graph failed on at least one rank"
                              + (f" ({type(err).__name__}: {err})" if err is not None else "")
                              + "; all ranks fall back to eager exchanges between graph segments")
            if err is not None:             # single process: stay correct, run eagerly
                self._graph_broken = True
                import warnings
                warnings.warn(f"hipGraph capture of the denoising step failed ({type(err).__name__}: {err}); running eagerly")
                return self._forward(sample, t, ctx, L, cond_frame)
            g = (rec, s_in, t_in, out)
            self.graph_put(key, g)
        rec, s_in, t_in, out = g
        s_in.copy_(sample)
        t_in.copy_(t)
        for step in rec.steps:
            step()
        # `out` is the graph's static output buffer: the caller gets its own copy (eps is only [B,4,F,h,w]); two replays of
        # one graph -- the unbatched CFG branch of p_sample_ddim calls uc then c -- must not alias each other's result
        return out.clone()


class _SegmentRecorder:
    """a step = a list of hipGraph segments (sharing one memory pool) interleaved with eager callables"""

    def __init__(self):
        self.steps = []
        self.pool = torch.cuda.graph_pool_handle()
        self._cur = None

    def begin_segment(self):
        g = torch.cuda.CUDAGraph()
        # thread_local: an RCCL watchdog thread polling its events must not invalidate this thread's capture
        ctx = torch.cuda.graph(g, pool=self.pool, capture_error_mode="thread_local")
        ctx.__enter__()
        self._cur = (g, ctx)

    def end_segment(self):
        g, ctx = self._cur
        self._cur = None
        ctx.__exit__(None, None, None)
        self.steps.append(g.replay)

    def abort(self):
        if self._cur is not None:
            g, ctx = self._cur
            self._cur = None
            try:
                ctx.__exit__(None, None, None)
            except Exception:
                pass
        self.steps = []

    @property
    def n_segments(self):
        return sum(1 for s in self.steps if getattr(s, "__self__", None).__class__ is torch.cuda.CUDAGraph)
