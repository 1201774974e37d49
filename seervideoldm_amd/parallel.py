"""Multi-GPU sharding of ONE denoising step across the GPUs of a node (one process per GPU, torch.distributed:
backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The reference has no model parallelism (DDP only).  The north-star partition shards the batch x frame
axis of the per-step UNet:

  * batch groups first: the CFG-doubled batch B = 2b splits with ZERO per-layer communication (uc half / c half);
  * frame shards inside a batch group: each rank keeps F/P frames of every activation.  Per-frame work (convs, spatial
    and text attention, LayerNorm, FF) stays local.  Two exchanges are REQUIRED for parity (SURVEY finding 3, 8(e)):
      - every GroupNorm spans all frames -> all-reduce of its (sum, sumsq) statistics, 77 per step: the producers' accumulated
        fixed-point column sums (int64 [B_local, 2, C]: exact, order-free) where the producer leaves them, B_local*32*2 floats
        otherwise;
      - temporal attention is causal over frames -> all-gather of the (rotary-applied) K|V of the frame group, 16 per step;
        the local queries then attend with `causal_offset` = sequence position of their first frame.
  * the epsilon prediction [B,4,F,h,w] (tiny) is all-gathered so every rank runs the same DDIM update.

rank r -> (batch group g = r // P, frame shard s = r % P): a frame group is P consecutive ranks.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

MAX_WIN_SIZE, MAX_RATIO, MIN_WIN_SIZE = 8, 4, 4


def split_counts(n: int, parts: int) -> List[int]:
    return [n // parts + (1 if i < n % parts else 0) for i in range(parts)]


def choose_groups(world: int, batch: int) -> Tuple[int, int]:
    """(batch_groups G, frame_shards P) with G*P == world: the largest G that divides both world and batch."""
    g = 1
    for cand in range(1, min(world, batch) + 1):
        if world % cand == 0 and batch % cand == 0:
            g = cand
    return g, world // g


class FrameShard:
    def __init__(self, world: int, rank: int, batch_groups: Optional[int] = None, capture_collectives: Optional[bool] = None):
        self.world, self.rank = world, rank
        self.forced_G = batch_groups
        self._pgs = {}
        self._xbuf = {}                 # static K|V exchange buffers (_gather_frames)
        self._probed = set()
        self.G = self.P = None
        self.total_frames = self.frame_offset = self.local_frames = 0
        self.debug_boundaries = False   # tests: hit every sync point (with a no-op exchange) even when P == 1
        self.force_exact_stats = False  # tests: the exact-statistics form of every GroupNorm even when P == 1 (see exact_stats)
        # RCCL collectives can be captured into the step's hipGraph (scripts/exp_nccl_capture.py: capture + replay of
        # all_reduce / all_gather_into_tensor verified on torch 2.10 + RCCL 2.26): the whole sharded step then replays as
        # ONE graph instead of ~94 segments with an eager exchange and a host round trip between each pair.  gloo (CPU tests,
        # several ranks on one device) cannot be captured and keeps the segmented replay.  The choice is made for all ranks
        # together and can only fall back, never diverge: the capturability probe runs on the WORLD communicator at attach()
        # and on the frame group at the first plan(), the verdicts are combined by a MIN all-reduce, and a capture that fails
        # on any rank sends every rank to the segmented replay (unet._Engine._run_graph).
        # capture_collectives: None = capture when the backend is RCCL and the probes pass; False = always segmented.
        self.capture_collectives = False
        if capture_collectives is not False:
            try:
                if dist.is_available() and dist.is_initialized():
                    self.capture_collectives = dist.get_backend() == "nccl"
            except Exception:       # noqa: BLE001  (no default group yet)
                pass

    @property
    def exact_stats(self) -> bool:
        """does every GroupNorm of this engine need exact integer statistics (unet._Engine._gn)?  Only frame shards exchange
        anything: with batch groups alone (a CFG half per rank, P == 1) the engine keeps the single-process forms -- no extra
        pass over the activations, nothing to add up across ranks.  The one-rank references of the bit-identity tests ask for the
        sharded arithmetic explicitly."""
        return (self.P or 1) > 1 or self.debug_boundaries or self.force_exact_stats

    # ---- geometry --------------------------------------------------------------------------------------------
    def plan(self, B: int, F: int):
        G, P = (self.forced_G, self.world // self.forced_G) if self.forced_G else choose_groups(self.world, B)
        if G * P != self.world or B % G:
            raise ValueError(f"cannot split batch {B} over {G} batch groups x {P} frame shards = {self.world} ranks")
        if P > F:
            raise ValueError(f"{P} frame shards but only {F} frames")
        self.G, self.P = G, P
        self.g, self.s = self.rank // P, self.rank % P
        self.frame_counts = split_counts(F, P)
        self.frame_starts = [sum(self.frame_counts[:i]) for i in range(P)]
        self.total_frames = F
        self.frame_offset = self.frame_starts[self.s]
        self.local_frames = self.frame_counts[self.s]
        bpg = B // G
        self.batch_range = (self.g * bpg, (self.g + 1) * bpg)
        self.frame_range = (self.frame_offset, self.frame_offset + self.local_frames)
        return self.batch_range, self.frame_range

    def frame_group(self):
        """process group of the P ranks that share this rank's batch rows (every rank creates every group)."""
        key = (self.G, self.P)
        if key not in self._pgs:
            groups = [dist.new_group(list(range(g * self.P, (g + 1) * self.P))) for g in range(self.G)]
            self._pgs[key] = groups
        return self._pgs[key][self.g]

    def probe_frame_group(self, device) -> None:
        """the capturability probe on THIS partition's frame-group communicator (attach() only saw WORLD); all ranks call it
        at the same point (first sharded step of a partition), the verdict is the minimum over all ranks"""
        key = (self.G, self.P)
        if not self.capture_collectives or self.P == 1 or key in self._probed:
            return
        self._probed.add(key)
        if not probe_capture(device, group=self.frame_group()):
            self.capture_collectives = False

    def control_group(self):
        """a communicator of all ranks that NO step collective uses: a rank that failed alone and votes while its peers are still
        inside the step's own all-reduce / all-gather (same ranks when P == world) must not meet them on their communicator --
        a size-1 MIN against a GroupNorm all-reduce is a mismatched collective, i.e. a hang or a garbage flag.  Created by
        every rank at the same point (attach())."""
        if "control" not in self._pgs:
            self._pgs["control"] = dist.new_group(list(range(self.world)))
        return self._pgs["control"]

    def agree(self, ok: bool, device) -> bool:
        """True iff `ok` on every rank (eager MIN all-reduce over the control group): how the ranks take a fallback together"""
        if self.world == 1 or not (dist.is_available() and dist.is_initialized()):
            return ok
        flag = torch.tensor([1.0 if ok else 0.0], device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.control_group())
        return bool(flag.item() > 0.5)

    def describe(self) -> str:
        if not self.G:
            return "unplanned"
        how = "" if self.P == 1 else (", collectives captured in the step graph" if self.capture_collectives
                                      else ", eager collectives between graph segments")
        return f"batch_groups{self.G}xframe_shards{self.P}{how}"

    # ---- hooks used by unet._Engine ----------------------------------------------------------------------------
    def reduce_gn_stats(self, stats: torch.Tensor, count_local: float, sync=None) -> float:
        """all-reduce the (sum, sumsq) statistics of one GroupNorm over the frame group; `sync` is the engine's
        sync_point (the exchange is an eager collective between two hipGraph segments)."""
        run = sync if sync is not None else (lambda fn: fn())
        if self.P > 1:
            grp = self.frame_group()
            run(lambda: dist.all_reduce(stats, group=grp))
        elif self.debug_boundaries:
            run(lambda: None)
        return count_local / self.local_frames * self.total_frames

    def reduce_fx(self, sums, count_local: float, sync=None) -> float:
        """all-reduce the ACCUMULATED fixed-point column sums (ops.ColSumsFx, int64) of a GroupNorm's sources over the frame
        group -- integer addition is exact and order-free, so every shard normalises with bit for bit the statistics the
        unsharded step computes from the same activations, and the apply launch reads them directly: no finalize launch, no fp32
        statistics tensor.  A producer's sums are exchanged once (an output that feeds two GroupNorms keeps the totals).
        Shards may hold different replica counts (uneven frame counts pick different tiles): replicas are folded first, so every
        rank sends [1, B, 2, C]."""
        run = sync if sync is not None else (lambda fn: fn())
        if self.P > 1:
            grp = self.frame_group()
            for cs in sums:
                if cs is None or cs.reduced:
                    continue
                if cs.buf.shape[0] > 1:
                    cs.buf = cs.buf.sum(dim=0, keepdim=True)
                buf = cs.buf
                run(lambda buf=buf: dist.all_reduce(buf, group=grp))
                cs.reduced = True
        elif self.debug_boundaries:
            run(lambda: None)
        return count_local / self.local_frames * self.total_frames

    def local_cond_frames(self, cond_frame: int) -> int:
        return max(0, min(self.local_frames, cond_frame - self.frame_offset))

    def _gather_frames(self, x: torch.Tensor, B: int, rows_per_frame: int, sync=None) -> torch.Tensor:
        """x [B*F_local*rows_per_frame, C] of this rank -> [B*F_total*rows_per_frame, C] in global frame order.
        The result lives in a buffer allocated here (static under graph capture); the exchange itself runs through `sync`."""
        run = sync if sync is not None else (lambda fn: fn())
        if self.P == 1:
            if self.debug_boundaries:
                run(lambda: None)
            return x
        C = x.shape[1]
        fmax = max(self.frame_counts)
        rows_l = self.local_frames * rows_per_frame
        even = all(c == fmax for c in self.frame_counts)
        grp = self.frame_group()
        # exchange buffers: one set per (site shape) for the life of the partition -- no allocation per call outside a capture
        # (inside one the graph's pool owns them anyway, and buffers of an eager run must not be baked into a graph)
        capturing = x.is_cuda and torch.cuda.is_current_stream_capturing()
        # (the frame geometry is part of the key: plan() runs on every forward, and a second clip length at the same latent size --
        # 12 then 14 frames -- must not inherit buffers sized for the first)
        key = (self.G, self.P, self.total_frames, tuple(self.frame_counts), B, rows_per_frame, C, x.dtype, x.device)
        bufs = None if capturing else self._xbuf.get(key)
        if bufs is None:
            out = torch.empty((B * self.total_frames * rows_per_frame, C), device=x.device, dtype=x.dtype)
            send = recv = None
            if not (even and B == 1):
                send = torch.zeros((B, fmax * rows_per_frame, C), device=x.device, dtype=x.dtype)
                recv = [torch.empty_like(send) for _ in range(self.P)]
            bufs = (out, send, recv)
            if not capturing:
                self._xbuf[key] = bufs
        out, send, recv = bufs
        if even and B == 1:
            # [P][F_l*rows, C] in rank order IS the global frame order: gather straight into the output
            run(lambda: dist.all_gather_into_tensor(out, x, group=grp))
            return out
        send[:, :rows_l] = x.reshape(B, rows_l, C)
        outv = out.reshape(B, self.total_frames * rows_per_frame, C)
        starts, counts = self.frame_starts, self.frame_counts

        def exchange():
            dist.all_gather(recv, send, group=grp)
            for i in range(self.P):
                n = counts[i] * rows_per_frame
                outv[:, starts[i] * rows_per_frame: starts[i] * rows_per_frame + n] = recv[i][:, :n]
        run(exchange)
        return out

    def temporal_attention(self, ops, qkv: torch.Tensor, out: torch.Tensor, B: int, heads: int, d: int, H: int, W: int,
                           sync=None):
        """causal (window) attention of the local frames' queries over the gathered K|V of frames [0, F_total).
        The q columns of `qkv` carry scale * log2(e) (unet._temporal_transformer)."""
        C = heads * d
        HW = H * W
        Fl, Ft, f0 = self.local_frames, self.total_frames, self.frame_offset
        kv = self._gather_frames(qkv[:, C:].contiguous(), B, HW, sync=sync)   # [B*Ft*HW, 2C]
        if H > MIN_WIN_SIZE:
            ws = MAX_WIN_SIZE if (H // MAX_WIN_SIZE) >= MAX_RATIO else MIN_WIN_SIZE
            ops.attention(qkv[:, :C], kv[:, :C], kv[:, C:], out, batch=B, heads=heads, head_dim=d, Sq=Fl * ws * ws,
                          Sk=Ft * ws * ws, causal=True, window=(ws, Ft, H, W), Fq=Fl, causal_offset=f0 * ws * ws,
                          q_prescaled=True)
        else:
            ops.attention(qkv[:, :C], kv[:, :C], kv[:, C:], out, batch=B, heads=heads, head_dim=d, Sq=Fl * HW,
                          Sk=Ft * HW, causal=True, causal_offset=f0 * HW, q_prescaled=True)

    # ---- whole-step plumbing --------------------------------------------------------------------------------------
    def gather_output(self, local: torch.Tensor, B: int, F: int) -> torch.Tensor:
        """local eps [B_l, C, F_l, h, w] -> full [B, C, F, h, w] on every rank"""
        if self.world == 1:
            return local
        Bl, Cc, _, h, w = local.shape
        fmax = max(self.frame_counts)
        send = torch.zeros((Bl, Cc, fmax, h, w), device=local.device, dtype=local.dtype)
        send[:, :, : self.local_frames] = local
        recv = [torch.empty_like(send) for _ in range(self.world)]
        dist.all_gather(recv, send)
        full = torch.empty((B, Cc, F, h, w), device=local.device, dtype=local.dtype)
        bpg = B // self.G
        for r in range(self.world):
            g, s = r // self.P, r % self.P
            f0, n = self.frame_starts[s], self.frame_counts[s]
            full[g * bpg:(g + 1) * bpg, :, f0:f0 + n] = recv[r][:, :, :n]
        return full


def probe_capture(device, group=None) -> bool:
    """Can collectives of `group` (default: WORLD) be recorded into a hipGraph and replayed?  One tiny all-reduce + all-gather
    is captured and replayed by every rank of the group; the verdicts of ALL ranks are combined by an (eager) MIN all-reduce
    over WORLD so that everybody takes the same path.  A stack that cannot capture falls back to eager exchanges between graph
    segments instead of failing the step."""
    ok = 1.0
    try:
        n = dist.get_world_size(group)
        x = torch.ones(64, device=device)
        out = torch.empty(64 * n, device=device)
        dist.all_reduce(x, group=group)         # communicator creation must not happen under capture
        dist.all_gather_into_tensor(out, x, group=group)
        torch.cuda.synchronize(device)
        s = torch.cuda.Stream(device=device)
        s.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(s):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                y = x * 2
                dist.all_reduce(y, group=group)
                dist.all_gather_into_tensor(out, y, group=group)
        torch.cuda.current_stream(device).wait_stream(s)
        x.fill_(1.0)
        g.replay()
        torch.cuda.synchronize(device)
        ok = 1.0 if bool((out == 2.0 * n).all()) else 0.0
    except Exception:       # noqa: BLE001  (capture unsupported / refused: every failure mode means "do not capture")
        ok = 0.0
    flag = torch.tensor([ok], device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item() > 0.5)


def attach(model, world: int, rank: int, batch_groups: Optional[int] = None, capture_collectives: Optional[bool] = None) -> FrameShard:
    """make `model.forward` run sharded: every rank passes the FULL (sample, timestep, context) and gets the FULL output.
    capture_collectives=False keeps the eager exchanges between hipGraph segments (the conservative replay)."""
    shard = FrameShard(world, rank, batch_groups, capture_collectives)
    if world > 1 and dist.is_available() and dist.is_initialized():
        shard.control_group()                # every rank, here: new_group is itself collective
    if shard.capture_collectives:
        dev = next(model.parameters()).device
        if dev.type != "cuda" or not probe_capture(dev):
            shard.capture_collectives = False
    model._shard = shard
    model._engine = None
    return shard
