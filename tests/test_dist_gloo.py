"""N > 1 path on CPU: 2 processes over gloo run the sharded step (seervideoldm_amd/parallel.py) -- batch-group split and
frame sharding with the GroupNorm statistics all-reduce and the K|V all-gather -- and must reproduce the unsharded
schedule.  Kernels are replaced by the plain-torch stand-in (tests/torch_ops_backend.py); the collectives, the shard
geometry, `causal_offset`, rotary `pos_offset` and the cond-frame bookkeeping are the product's."""
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]
CFG_MINI = dict(block_out_channels=(320, 320, 320, 320), layers_per_block=1, cross_attention_dim=256, attention_head_dim=8)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn(worker, world, *args):
    """mp.spawn(worker, (world, port, *args)); one retry on a rendezvous failure (the probed port can be taken in between)"""
    for attempt in range(2):
        try:
            mp.spawn(worker, args=(world, _free_port(), *args), nprocs=world, join=True)
            return
        except Exception as e:      # noqa: BLE001
            msg = str(e)
            if attempt == 0 and any(k in msg for k in ("Address already in use", "Connection refused", "connect() timed out",
                                                       "Connection reset", "Socket Timeout")):
                continue
            raise


def _worker(rank, world, port, batch_groups, B, Fr, H, cond_frame, out_path):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from seervideoldm_amd import SeerUNet, parallel, synth
        from tests import torch_ops_backend as tob
        sd = synth.synth_state_dict(synth.unet_param_shapes(CFG_MINI))
        m = SeerUNet(**CFG_MINI)
        m.load_state_dict(sd, strict=True)
        m._ops_backend = tob
        tob.EXACT = True        # float64 accumulation: a row's result does not depend on how many other rows a call holds
        # LayerNorm as its own launch everywhere: with cond_frame > 0 the one-rank run feeds the temporal FF a row SUBSET (layernorm
        # kernel + plain weights) where a shard that holds no conditioning frame feeds it whole rows (folded into the GEMM) -- two
        # valid arithmetics of the same operator, not one
        m.ln_fold = False
        m.ff_fold = False       # likewise ff.net.2 + proj_out as one GEMM (whole rows only) against the two launches of a row subset
        g = torch.Generator().manual_seed(7)
        x = torch.randn((B, 4, Fr, H, H), generator=g)
        ctx = torch.randn((B, Fr, 77, 256), generator=g)
        t = torch.tensor([501] * B)
        ref = ref1 = None
        if rank == 0:
            ref = m(x, t, ctx, cond_frame=cond_frame)          # the plain single-process engine
            parallel.attach(m, 1, 0).force_exact_stats = world // batch_groups > 1   # the sharded engine's code path on ONE rank: no exchange at all
            ref1 = m(x, t, ctx, cond_frame=cond_frame)
        shard = parallel.attach(m, world, rank, batch_groups=batch_groups)
        got = m(x, t, ctx, cond_frame=cond_frame)
        got2 = m(x, t, ctx, cond_frame=cond_frame)             # second call: cached context slice / groups
        if rank == 0:
            torch.save(dict(ref=ref, ref1=ref1, got=got, got2=got2, desc=shard.describe()), out_path)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("batch_groups,B,Fr,cond_frame,H", [
    (2, 2, 2, 0, 8),         # CFG halves on two ranks: no per-layer communication
    (1, 1, 4, 0, 8),         # 2 frame shards of 2 frames: GN statistics all-reduce + K|V all-gather
    (1, 2, 3, 2, 8),         # uneven frame shards (2 + 1) with conditioning frames crossing the shard boundary
    (1, 1, 3, 1, 16),        # windowed temporal attention (ws = 4 at 16 and 8) over uneven shards, causal offset inside a window
])
def test_sharded_step_matches_unsharded(tmp_path, batch_groups, B, Fr, cond_frame, H):
    out = tmp_path / "res.pt"
    _spawn(_worker, 2, batch_groups, B, Fr, H, cond_frame, str(out))
    r = torch.load(out)
    assert r["desc"].startswith(f"batch_groups{batch_groups}xframe_shards{2 // batch_groups}")
    # BIT FOR BIT the one-rank run of the same engine: the GroupNorm statistics are exact integer sums on both sides (an int64
    # all-reduce adds the shards' sums in any order), the K|V exchange moves bf16 rows, everything else is row-local -- a wrong
    # causal offset, rotary position, count, or a missed or doubled exchange cannot hide under a tolerance
    assert torch.equal(r["got"], r["ref1"]), ((r["got"] - r["ref1"]).abs().max().item())
    assert torch.equal(r["got"], r["got2"])
    # ... and that engine against the plain one (per-tile fp32 statistics where the tensors are large): same arithmetic up to the
    # statistics' rounding, which ~100 bf16 layers amplify to the 1-2e-2 two bf16 runs of this network sit apart
    rel = ((r["ref1"] - r["ref"]).norm() / r["ref"].norm()).item()
    assert rel < 3e-2, rel


def _two_lengths_worker(rank, world, port, out_path):
    """one attached model, two clip lengths at the same latent size (BASELINE configs 2 and its 14-frame reading): the static
    K|V exchange buffers of the first length must not serve the second"""
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from seervideoldm_amd import SeerUNet, parallel, synth
        from tests import torch_ops_backend as tob
        m = SeerUNet(**CFG_MINI)
        m.load_state_dict(synth.synth_state_dict(synth.unet_param_shapes(CFG_MINI)), strict=True)
        m._ops_backend = tob
        tob.EXACT = True
        res = {}
        g = torch.Generator().manual_seed(11)
        ins = {Fr: (torch.randn((1, 4, Fr, 8, 8), generator=g), torch.randn((1, Fr, 77, 256), generator=g)) for Fr in (4, 6, 5)}
        t = torch.tensor([501])
        if rank == 0:
            parallel.attach(m, 1, 0).force_exact_stats = True
            for Fr, (x, ctx) in ins.items():
                res[f"ref{Fr}"] = m(x, t, ctx, cond_frame=0)
        shard = parallel.attach(m, world, rank, batch_groups=1)
        for Fr in (4, 6, 5, 4):                                # even, even (longer), uneven, and back
            x, ctx = ins[Fr]
            res[f"got{Fr}"] = m(x, t, ctx, cond_frame=0)
        res["n_buffer_sets"] = len(shard._xbuf)
        if rank == 0:
            torch.save(res, out_path)
    finally:
        dist.destroy_process_group()


def test_clip_length_changes_on_an_attached_model(tmp_path):
    out = tmp_path / "res.pt"
    _spawn(_two_lengths_worker, 2, str(out))
    r = torch.load(out)
    for Fr in (4, 6, 5):
        assert torch.equal(r[f"got{Fr}"], r[f"ref{Fr}"]), Fr
    assert r["n_buffer_sets"] >= 3          # a set per frame geometry (and site shape), none shared across lengths


def test_shard_geometry():
    from seervideoldm_amd.parallel import FrameShard, choose_groups, split_counts
    assert choose_groups(8, 2) == (2, 4) and choose_groups(4, 2) == (2, 2) and choose_groups(2, 2) == (2, 1)
    assert choose_groups(8, 8) == (8, 1) and choose_groups(1, 2) == (1, 1)
    assert split_counts(12, 4) == [3, 3, 3, 3] and split_counts(12, 8) == [2, 2, 2, 2, 1, 1, 1, 1]
    covered = set()
    for r in range(8):
        sh = FrameShard(8, r)
        (b0, b1), (f0, f1) = sh.plan(2, 12)
        covered |= {(b, f) for b in range(b0, b1) for f in range(f0, f1)}
        assert sh.local_cond_frames(2) == max(0, min(f1 - f0, 2 - f0))
    assert covered == {(b, f) for b in range(2) for f in range(12)}


# ---- data-parallel training step (seervideoldm_amd/trainer.py): one all-reduce of the flat gradient buffers -----------------
CFG_TRAIN = dict(block_out_channels=(320, 320, 320, 320), layers_per_block=1, cross_attention_dim=192, attention_head_dim=8)
FS_TRAIN = dict(num_frames=16, num_layers=1, channels=192, n_heads=2, cross_attention_dim=192)


def _train_models():
    from seervideoldm_amd import FSTextTransformer, SeerUNet, synth
    unet = SeerUNet(**CFG_TRAIN)
    unet.load_state_dict(synth.synth_state_dict(synth.unet_param_shapes(CFG_TRAIN)), strict=True)
    fst = FSTextTransformer(num_frames=16, in_channels=192, out_channels=192, n_heads=2, num_layers=1, cross_attention_dim=192)
    fst.load_state_dict(synth.synth_state_dict(synth.fstext_param_shapes(**FS_TRAIN)), strict=True)
    fst.set_numframe(2)
    return unet, fst


def _train_batch(seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn((1, 4, 2, 8, 8), generator=g), torch.randn((1, 4, 1, 8, 8), generator=g),
            torch.tensor([300 + seed]), torch.randn((1, 77, 192), generator=g))


def _train_worker(rank, world, port, out_path):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from seervideoldm_amd.trainer import SeerTrainer
        from tests import torch_ops_backend as tob
        from tests import torch_train_ops_backend as ttob
        unet, fst = _train_models()
        tr = SeerTrainer(unet, fst, lr=1e-3, max_grad_norm=0.3, ops=tob, tops=ttob, process_group=dist.group.WORLD)
        x, noise, t, text = _train_batch(rank)                    # every rank its own micro-batch
        seen = []
        loss = tr.forward_backward(x, noise, t, text, 1, on_unet_grads=lambda: seen.append(tr.pu.g.clone()))
        local = (tr.pu.g.clone(), tr.pf.g.clone())
        assert len(seen) == 1 and torch.equal(seen[0], local[0])      # the UNet segment was final when the hook ran
        tr.optimizer_step()
        torch.save(dict(loss=loss, gu=local[0], gf=local[1], pu=tr.pu.p.clone(), pf=tr.pf.p.clone()), f"{out_path}.{rank}")
    finally:
        dist.destroy_process_group()


def test_data_parallel_train_step(tmp_path):
    """2 ranks, one micro-batch each: after the step both hold the SAME parameters, equal to a single process stepping on the
    mean of the two gradients (DDP semantics of accelerate, train.py:265-266,382)."""
    out = tmp_path / "tr"
    _spawn(_train_worker, 2, str(out))
    r0, r1 = torch.load(f"{out}.0"), torch.load(f"{out}.1")
    assert torch.equal(r0["pu"], r1["pu"]) and torch.equal(r0["pf"], r1["pf"])
    assert not torch.equal(r0["gu"], r1["gu"])
    from seervideoldm_amd.trainer import SeerTrainer
    from tests import torch_ops_backend as tob
    from tests import torch_train_ops_backend as ttob
    unet, fst = _train_models()
    tr = SeerTrainer(unet, fst, lr=1e-3, max_grad_norm=0.3, ops=tob, tops=ttob)
    tr.pu.g.copy_((r0["gu"] + r1["gu"]) * 0.5)
    tr.pf.g.copy_((r0["gf"] + r1["gf"]) * 0.5)
    tr.optimizer_step()
    assert torch.allclose(tr.pu.p, r0["pu"], atol=1e-7) and torch.allclose(tr.pf.p, r0["pf"], atol=1e-7)


def test_partition_choice_per_config():
    """the CFG-doubled batch splits first (no per-layer communication), frames only inside a batch group.
    config 2 (b = 1 -> CFG batch 2) on 8 GPUs = 2 batch groups x 4 frame shards; config 3 (b = 4 -> CFG batch 8) on 8 GPUs = 8
    batch groups and NO frame sharding (zero data-path collectives)."""
    from seervideoldm_amd.parallel import FrameShard, choose_groups
    assert choose_groups(8, 2) == (2, 4)
    assert choose_groups(8, 8) == (8, 1)
    assert choose_groups(4, 2) == (2, 2) and choose_groups(2, 2) == (2, 1) and choose_groups(2, 1) == (1, 2)
    sh = FrameShard(8, 5)
    assert sh.plan(8, 16) == ((5, 6), (0, 16)) and sh.P == 1
    assert not sh.capture_collectives              # no RCCL process group here: collectives stay eager between graph segments
    assert sh.describe() == "batch_groups8xframe_shards1"
    sh = FrameShard(8, 5)
    assert sh.plan(2, 12) == ((1, 2), (3, 6)) and (sh.G, sh.P) == (2, 4)
    assert sh.describe().startswith("batch_groups2xframe_shards4, eager collectives")


def _agree_worker(rank, world, port, out_path):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from seervideoldm_amd.parallel import FrameShard
        sh = FrameShard(world, rank)
        cpu = torch.device("cpu")
        res = [sh.agree(True, cpu),                  # everybody fine
               sh.agree(rank != 1, cpu),             # rank 1 failed: EVERY rank must learn of it
               sh.agree(False, cpu)]
        explicit = FrameShard(world, rank, capture_collectives=False)
        torch.save(dict(res=res, cap=sh.capture_collectives, cap_off=explicit.capture_collectives), f"{out_path}.{rank}")
    finally:
        dist.destroy_process_group()


def test_ranks_take_a_fallback_together(tmp_path):
    """a hipGraph capture that fails on ONE rank must send every rank to the segmented replay (unet._Engine._run_graph decides
    through FrameShard.agree): the verdict is the minimum over the ranks, identical everywhere"""
    out = tmp_path / "agree"
    _spawn(_agree_worker, 2, str(out))
    r0, r1 = torch.load(f"{out}.0"), torch.load(f"{out}.1")
    assert r0["res"] == r1["res"] == [True, False, False]
    assert not r0["cap"] and not r0["cap_off"]           # gloo: never captured
