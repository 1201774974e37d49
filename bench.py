#!/usr/bin/env python
"""bench.py -- UNet denoising steps/sec of the Seer DDIM hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one `DDIMSampler.p_sample_ddim`: one CFG-batched SeerUNet forward (B = 2b), CFG combine and the DDIM
update, on synthetic latents that are already resident in HBM.  Workload = BASELINE config 2 (Sthv2): b=1 (CFG batch 2),
12 frames total (2 conditioning + 10 predicted), 32x32 latent (256^2 pixels), full-width SD-v1-5-shaped SeerUNet
(1.08 G parameters, closed-form synthetic weights: there is no network for checkpoints), bf16 storage / fp32 accumulate.
N > 1: the headline `value` is the north star's partition of ONE step over all GPUs (seervideoldm_amd/parallel.py: batch x CFG
groups first -- no communication -- then frame shards inside a group, with their GroupNorm-statistics all-reduces and K|V
all-gathers over RCCL), "scaling": "strong": the same work, N GPUs.  N GPUs denoising N independent samples (the path's natural
units, no data-path collective; linear by construction) is timed in the same run and reported as `weak_scaling`.

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel class (the MFMA GEMM / implicit-GEMM conv template,
93 % of the step's FLOPs) from HIP-event timings taken inside this process; `cpu_baseline` times the CPU oracle
(oracle/seer_oracle.py, a port of the reference algorithm) on the host cores: one warm-up + three timed FULL steps.
`python bench.py --gpus N` without a launcher starts its N ranks itself; `--dtype fp16` times the fp16-storage engine (an extra).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0                     # same guide: HBM3E 8 TB/s (spec)
MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0      # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16
PMC_TRAFFIC_FILE = "r06_pmc_traffic.json"
WORKLOAD = dict(b=1, cond_frames=2, frames=12, latent=32, ddim_steps=50, scale=7.5)
# BASELINE.json configs: [1] is the bench line (default); the others are parity-test cases that can be timed on request
WORKLOADS = {
    "sthv2": dict(WORKLOAD, name="Sthv2 config: CFG batch 2 x 12 frames (2 cond + 10 predicted) x 32x32 latent"),
    "bridge": dict(b=4, cond_frames=1, frames=16, latent=32, ddim_steps=50, scale=7.5,
                   name="Bridge config: CFG batch 8 x 16 frames (1 cond + 15 predicted) x 32x32 latent"),
    # SURVEY 8(d)'s literal readings of "2 ref + 12 frames" and "1 ref + 16 frames"
    "sthv2_14": dict(b=1, cond_frames=2, frames=14, latent=32, ddim_steps=50, scale=7.5,
                     name="Sthv2 config, literal reading: CFG batch 2 x 14 frames (2 cond + 12 predicted) x 32x32 latent"),
    "bridge_17": dict(b=4, cond_frames=1, frames=17, latent=32, ddim_steps=50, scale=7.5,
                      name="Bridge config, literal reading: CFG batch 8 x 17 frames (1 cond + 16 predicted) x 32x32 latent"),
    "sthv2_512": dict(b=1, cond_frames=2, frames=12, latent=64, ddim_steps=50, scale=7.5,
                      name="Sthv2 512^2 config: CFG batch 2 x 12 frames x 64x64 latent (4096-token spatial attention)"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--cpu-budget-s", type=float, default=45.0,
                    help="CPU baseline: three timed full steps run when they fit this budget, else fewer frames scaled linearly")
    ap.add_argument("--no-train", action="store_true", help="skip the extra fine-tuning step measurement (config 5)")
    ap.add_argument("--collective-timeout-s", type=float, default=300.0,
                    help="N > 1: a phase with collectives (the partitioned step, the data-parallel fine-tuning step) that has not "
                         "finished after this long is abandoned: rank 0 prints the line with what was measured so far, every rank exits")
    ap.add_argument("--no-capture-collectives", action="store_true",
                    help="N > 1: keep the RCCL exchanges of the partitioned step eager between hipGraph segments")
    ap.add_argument("--dtype", choices=("bf16", "fp16"), default="bf16",
                    help="16-bit storage type of the UNet engine: bf16 = BASELINE config 2 (the headline line); fp16 = the mixed_precision every "
                         "shipped yaml names (SeerUNet(compute_dtype=torch.float16)), an extra measurement")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="sthv2",
                    help="default = BASELINE.json's metric configuration; the others are extra measurements")
    return ap.parse_args()


def build_inputs(device, seed=0):
    g = torch.Generator().manual_seed(seed)
    w = WORKLOAD
    b, f1, F, h = w["b"], w["cond_frames"], w["frames"], w["latent"]
    x_T = torch.randn((b, 4, F - f1, h, h), generator=g)
    x0_emb = torch.randn((b, 4, f1, h, h), generator=g) * 0.18215 * 5
    c = torch.randn((b, F, 77, 768), generator=g)
    uc = torch.randn((b, 1, 77, 768), generator=g).expand(-1, F, -1, -1).contiguous()
    return [t.to(device) for t in (x_T, x0_emb, c, uc)]


def gpu_busy(ms: float, device):
    """queue ~ms of GPU work so that the host runs ahead of the device during event-bracketed launches"""
    from seervideoldm_amd import ops
    a = torch.randn(8192, 8192, device=device).to(torch.bfloat16)
    out = torch.empty(8192, 8192, device=device, dtype=torch.bfloat16)
    for _ in range(max(1, int(ms / 1.3))):
        ops.gemm(a, a, out=out, tile=1)


def time_attention_block(device):
    """the north star's named kernel target: spatial self-attention of the 32x32 level, [B*F*heads = 192, 1024 tokens, d = 40]
    out of the fused q|k|v projection; back-to-back launches (one replayed HIP graph) between two HIP events on the launch stream"""
    from seervideoldm_amd import ops
    B, S, H, d = 24, 1024, 8, 40
    C = H * d
    qkv = torch.randn((B * S, 3 * C), device=device).to(torch.bfloat16)
    out = torch.empty((B * S, C), device=device, dtype=torch.bfloat16)
    run = lambda: ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, batch=B, heads=H, head_dim=d, Sq=S, Sk=S)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    # 25 launches captured in one HIP graph and replayed, as the denoising step runs them: launched one by one from Python the
    # loop is host-bound (the ctypes call costs more than the ~42 us kernel; rocprofv3 shows the kernel at 41.9 us while this
    # function used to print 54)
    per_graph, replays = 25, 4
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(per_graph):
            run()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (per_graph * replays) * 1e3
    tf = 4.0 * B * H * S * S * d / us * 1e-6
    return dict(shape="[192, 1024, 40] bf16, non-causal", us_per_launch=round(us, 2), achieved=round(tf, 1), unit="TFLOP/s",
                frac=round(tf / MFMA_BF16_DENSE_PEAK_TFLOPS, 4), flops="4*BH*Sq*Sk*d")


def host_threads() -> int:
    """threads for the CPU baseline: the CPUs this process may actually use (affinity and cgroup quota), then the count
    that runs a probe matmul fastest (256 logical CPUs oversubscribed by a small quota run an order of magnitude slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    a = torch.randn(1536, 1536)
    best, best_t = 1, float("inf")
    for c in (4, 8, 16, 32, 64, 128):
        if c > n:
            break
        torch.set_num_threads(c)
        a @ a
        t0 = time.perf_counter()
        for _ in range(3):
            a @ a
        dt = time.perf_counter() - t0
        if dt < best_t * 0.9:
            best, best_t = c, dt
    return best


def cpu_baseline(sd_cpu, cfg, budget_s):
    """time the CPU oracle on a bounded sample of the same workload (rank 0, N=1 only)."""
    from oracle import seer_oracle as O
    ncores = host_threads()
    torch.set_num_threads(ncores)
    w = WORKLOAD
    h = w["latent"]

    def one(F):
        g = torch.Generator().manual_seed(1)
        x = torch.randn((2, 4, F, h, h), generator=g)
        c = torch.randn((2, F, 77, 768), generator=g)
        t0 = time.perf_counter()
        with torch.no_grad():
            O.unet_forward(sd_cpu, cfg, x, torch.tensor([981, 981]), c, 0)
        return time.perf_counter() - t0

    # 1 warm-up + >= 3 timed.  The warm-up IS a full step (GroupNorm couples the frames and temporal attention is
    # quadratic in them: a short clip scaled linearly is not the step), and so are the timed runs whenever three of them fit the
    # budget (default: they do, ~9 s each on the GPU boxes' hosts); only a host too slow for that falls back to fewer frames, scaled.
    Fw = w["frames"] if budget_s >= 20.0 else 1          # (a test-sized budget warms up on one frame)
    t_w = one(Fw) * w["frames"] / Fw
    F = w["frames"] if 3.0 * t_w <= budget_s else int(max(1, w["frames"] * budget_s / (3.0 * t_w)))
    ts = sorted(one(F) for _ in range(3))
    tF = ts[1]                                    # median of 3
    est_full = tF * w["frames"] / F
    how = "the full step, nothing scaled" if F == w["frames"] else "scaled linearly in F to the full step"
    return dict(value=1.0 / est_full, unit="steps/s", cores=ncores, kind="port",
                sample=f"CFG-batched UNet forward (B=2, 32x32 latent, fp32) at F={F} of {w['frames']} frames: 1 warm-up at F={Fw} "
                       f"({t_w * Fw / w['frames']:.1f} s) + 3 timed ({ts[0]:.1f} / {ts[1]:.1f} / {ts[2]:.1f} s, median used), {how}")


class Watchdog:
    """N > 1 only.  A hung collective cannot be caught as an exception (the RCCL watchdog would abort the process, and the driver
    would get no line at all): while a phase with collectives runs, a timer thread on every rank waits; if the phase is still
    running at the deadline, rank 0 prints the JSON line built from what HAS been measured (`fallback()`) and every rank leaves
    with exit code 3 (`os._exit`: no destructor waits on the hung stream) -- the line is there for the driver, the status tells
    torchrun / CI that the job did not finish.  No in-process retry: a wedged stream is not recoverable; start a fresh process.
    `disarm()` and the firing timer take the same lock, so a phase that ends at the deadline prints either its own line or the
    fallback line, never both."""

    EXIT_CODE = 3

    def __init__(self, rank: int):
        import threading
        self.rank = rank
        self._ev = None
        self._lock = threading.Lock()

    def arm(self, seconds: float, fallback) -> None:
        import threading
        ev = threading.Event()
        self._ev = ev

        def run():
            if ev.wait(seconds):
                return
            with self._lock:
                if ev.is_set():              # disarmed while we were waiting for the lock: the phase finished
                    return
                if self.rank == 0:
                    try:
                        print(json.dumps(fallback()), flush=True)
                    finally:
                        os._exit(self.EXIT_CODE)
                time.sleep(5.0)              # rank 0 prints first
                os._exit(self.EXIT_CODE)
        threading.Thread(target=run, daemon=True).start()

    def disarm(self) -> None:
        with self._lock:
            if self._ev is not None:
                self._ev.set()
                self._ev = None


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` WITHOUT a launcher: start the N ranks ourselves (the driver's own launch line: one process per GPU
    under torch.distributed.run, rendezvous on 127.0.0.1) as a CHILD process and hand back its exit status.  Nothing in this
    process has touched the GPU yet (device_count() does not initialise it), so the box's rule against replacing or forking an
    initialised process is not in play.  Fewer than N devices is an error, never a silent single-GPU line filed under N."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and os.environ.get("SEER_BENCH_SAME_DEVICE") != "1":
        print(f"bench.py: --gpus {n} but this node shows {have} GPU(s); refusing to print a line for {n}", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    return subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")).returncode


def main():
    args = parse()
    WORKLOAD.update({k: v for k, v in WORKLOADS[args.workload].items() if k != "name"})
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))          # before any GPU call in this process
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line must be measured on the GPUs it names")
    assert torch.cuda.is_available(), "bench.py needs a ROCm device"
    # flow-check knobs (tests on a 1-GPU box): all ranks on cuda:0 over gloo -- RCCL refuses two ranks on one device
    if os.environ.get("SEER_BENCH_SAME_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("SEER_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            # rehearsal of the N > 1 control flow on a 1-GPU box (tests/test_bench_multi.py): gloo has no device all-gather,
            # so the two gather flavours go through host memory; everything else is the code path the driver runs over RCCL
            dist.init_process_group(backend)
            ag, agt = dist.all_gather, dist.all_gather_into_tensor

            def all_gather(recv, send, group=None):
                host = [torch.empty(r.shape, dtype=r.dtype) for r in recv]
                ag(host, send.cpu(), group=group)
                for r, h in zip(recv, host):
                    r.copy_(h)

            def all_gather_into_tensor(out, x, group=None):
                host = torch.empty(out.shape, dtype=out.dtype)
                agt(host, x.cpu(), group=group)
                out.copy_(host)

            dist.all_gather, dist.all_gather_into_tensor = all_gather, all_gather_into_tensor

    from seervideoldm_amd import DDIMSampler, SeerUNet, synth
    from seervideoldm_amd.profiler import TimedOps

    cfg = dict(synth.SD15_UNET_CFG)
    model = SeerUNet(**cfg, compute_dtype=torch.float16 if args.dtype == "fp16" else torch.bfloat16).to(device)
    sd = synth.synth_state_dict(synth.unet_param_shapes(cfg), device=device)
    model.load_state_dict(sd, strict=True)
    sd_cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sd_cpu = {k: v.cpu() for k, v in sd.items()}
    del sd
    model.eval()
    # N > 1: first N independent samples (one per GPU, no collective: `weak_scaling`), then the headline: ONE step partitioned
    # over all GPUs (batch x CFG groups, then frame shards with GroupNorm-statistics all-reduces + K|V all-gathers).
    model.use_graph = not args.no_graph

    x_T, x0_emb, c, uc = build_inputs(device)
    w = WORKLOAD
    sampler = DDIMSampler(device)
    sampler.make_schedule(ddim_num_steps=w["ddim_steps"], ddim_eta=0.0, verbose=False)
    nidx = len(sampler.ddim_timesteps)
    # as inside DDIMSampler.ddim_sampling: a step hands the next one its latent without a copy (the loops below keep no step's
    # output past the next step)
    sampler.static_step_outputs = True

    def step(i, x):
        index = nidx - 1 - (i % nidx)
        ts = sampler._t_table[index].expand(w["b"])
        x, _ = sampler.p_sample_ddim(model, x, c, ts, index=index, x0_emb=x0_emb, unconditional_guidance_scale=w["scale"],
                                     unconditional_conditioning=uc)
        return x

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def timed_steps():
        """W untimed + exactly K timed steps between barrier + synchronize pairs; MAX over ranks (ms per step)"""
        x = x_T
        for i in range(args.warmup):
            x = step(i, x)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            x = step(args.warmup + i, x)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            tt = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        assert torch.isfinite(x).all(), "non-finite latent after the timed steps"
        return dt / args.steps * 1e3

    ms_per_step = timed_steps()
    graph_live = bool(model._engine is not None and model._engine._graphs and not getattr(model._engine, "_graph_broken", False))

    # ---- N > 1: the headline is ONE step partitioned over all GPUs (the north star's batch x frame sharding)
    weak, parallelism = None, "single"
    dog = Watchdog(rank)

    def make_line(ms, par, graph, roofline, cpu, clip, train, weak_obj):
        line = {
            "metric": "UNet denoising steps/sec (12-frame 256^2 latent, 50-step DDIM)" if args.workload == "sthv2"
                      else f"UNet denoising steps/sec ({args.workload} workload, 50-step DDIM)",
            "value": round((world if "independent samples" in par else 1) * 1e3 / ms, 3), "unit": "steps/s",
            "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "strong" if (world > 1 and "independent samples" not in par) else "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": WORKLOADS[args.workload]["name"] + ", "
                                   "full-width SeerUNet 1.08G params, 50-step DDIM, scale 7.5",
                       "global_batch": WORKLOADS[args.workload]["b"], "parallelism": par,
                       "hip_graph": graph},
            "roofline": roofline, "cpu_baseline": cpu, "end_to_end": clip, "train_step": train,
        }
        if weak_obj is not None:
            line["weak_scaling"] = weak_obj
        return line

    if world > 1:
        from seervideoldm_amd import parallel
        weak = {"value": round(world * 1e3 / ms_per_step, 3), "unit": "steps/s", "ms_per_step": round(ms_per_step, 3),
                "scaling": "weak", "what": f"{world} independent samples, one CFG-batched sample per GPU, no data-path collective"}
        # the sharded step must never cost the line: a failure that reaches every rank (capture refused, a partition the frame
        # count does not allow) is reported in the JSON and the independent-samples number stands as `value`.  A rank that
        # HANGS inside a collective cannot be caught as an exception: the Watchdog prints the same fallback line at the deadline.
        sharded_error = None
        dog.arm(args.collective_timeout_s, lambda: make_line(
            ms_per_step, f"{world} independent samples (the partitioned step did not finish within "
                         f"{args.collective_timeout_s:.0f} s: abandoned)", graph_live, None, None, None, None, weak))
        shard = parallel.attach(model, world, rank, capture_collectives=False if args.no_capture_collectives else None)
        try:
            ms_sharded = timed_steps()
        except Exception as e:      # noqa: BLE001
            sharded_error = f"{type(e).__name__}: {e}"[:300]
        ok = shard.agree(sharded_error is None, device)
        dog.disarm()
        if ok:
            ms_per_step = ms_sharded
            parallelism = f"{shard.describe()} over {world} RCCL ranks"
            graph_live = bool(model._engine._graphs and not getattr(model._engine, "_graph_broken", False))
        else:
            parallelism = f"{world} independent samples (the partitioned step failed: {sharded_error or 'on another rank'})"
        model._shard = None
        model._engine = None

    # ---- roofline of the dominant kernel class: event-bracketed launches, device kept ahead of the host
    roofline = None
    if rank == 0 and world == 1:      # (N > 1: every step is collective -- no rank-0-only extra steps)
        eng = model._engine
        timed = TimedOps()
        eng.ops = timed
        model.use_graph = False
        reps = 3
        step(0, x_T)                       # warm (eager path)
        torch.cuda.synchronize()
        timed.reset()
        gpu_busy(60.0, device)
        for r in range(reps):
            step(r, x_T)
        torch.cuda.synchronize()
        summ = timed.summary()
        from seervideoldm_amd import ops as plain_ops
        eng.ops = plain_ops
        gm = summ["gemm"]
        per_launch_flops = gm["flops"] / gm["launches"]
        avg_launch_ms = gm["ms"] / gm["launches"]
        # HBM-side bytes per launch from the separate rocprofv3 --pmc passes of the same step (scripts/pmc_step.py ->
        # scripts/pmc_summary.py; PMC cannot be collected from inside this process)
        # (scripts/pmc_step.py -> scripts/pmc_summary.py; PMC cannot be collected from inside this process).  The file carries
        # the digest of the library build it was taken on: a stale file is refused, not quoted.
        traffic, traffic_note, ff_traffic, traffic_families = None, None, None, None
        try:
            if args.workload != "sthv2":
                raise OSError("the PMC passes were taken on the default workload")
            pmc = json.load(open(ROOT / "profiles" / PMC_TRAFFIC_FILE))
            built = (ROOT / "seervideoldm_amd" / "lib" / "build.sha256").read_text().strip()
            if pmc.get("build_sha256") != built:
                raise ValueError(f"{PMC_TRAFFIC_FILE} was taken on library build {str(pmc.get('build_sha256'))[:12]}, this is "
                                 f"{built[:12]}")
            # like for like with `achieved` / `algorithmic_bytes_per_launch`: the launch-weighted mean over EVERY kernel family of the
            # class (the tile kernels, the 256 x 320 tile, the weight-stationary pair, the fused feed-forward, the row chains), not the tile kernels alone
            fams = {k: v for k, v in pmc["kernels"].items() if k.startswith(("seer_gemm", "seer_ff_fused", "seer_rowchain"))}
            traffic = round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in fams.values()) / sum(v["launches"] for v in fams.values()))
            traffic_families = sorted(fams)
            ff_traffic = pmc["kernels"].get("seer_ff_fused_c320_kernel", {}).get("hbm_bytes_per_launch")
        except (OSError, KeyError, ValueError) as e:
            traffic_note = str(e)[:200]
        attn_block = time_attention_block(device)
        # the same kernel on the step's own activations (event-bracketed eager launches of the roofline passes above): the
        # random-normal q|k|v of time_attention_block toggle more bits per MFMA operand than the synthetic-weight activations
        # and the kernel runs at lower clocks on them (rocprofv3: 59 us in that loop, 42 us inside the step, same binary:
        # profiles/r02_attn40_inputs.log)
        for tag, calls, ms, _tf in timed.shape_summary():
            if tag.startswith("attn b24 Sq1024 Sk1024 d40"):
                attn_block["in_step_us"] = round(ms / calls * 1e3, 2)
        roofline = dict(bound="mfma", kernel="seer_gemm_kernel (bf16 MFMA GEMM / implicit-GEMM conv3x3, all tiles; the class also holds the "
                                             "ten seer_ff_fused_c320_kernel and the fifteen seer_rowchain_c320_kernel launches, accounted "
                                             "with the MACs of the GEMMs each contains)",
                        achieved=round(gm["tflops"], 2), peak=MFMA_BF16_DENSE_PEAK_TFLOPS, unit="TFLOP/s",
                        frac=round(gm["tflops"] / MFMA_BF16_DENSE_PEAK_TFLOPS, 4), traffic=traffic,
                        traffic_unit="bytes beyond L2 per launch (2*FETCH_SIZE + WRITE_SIZE, PMC), launch-weighted over the class's kernel families",
                        traffic_families=traffic_families,
                        algorithmic_bytes_per_launch=round(gm["bytes"] / gm["launches"]),
                        launches_per_step=gm["launches"] // reps, avg_launch_us=round(avg_launch_ms * 1e3, 2),
                        algorithmic_gflop_per_launch=round(per_launch_flops / 1e9, 3),
                        flops="executed: 2MNK per launched GEMM / conv; the convs behind a nearest-2x upsample count their four "
                              "2x2 phase convs (16 tap products per source pixel, not the reference's 36); a call that launches "
                              "nothing is not counted",
                        step_breakdown_ms={k: round(v["ms"] / reps, 3) for k, v in summ.items()},
                        attention_tflops=round(summ.get("attention", {}).get("tflops", 0.0), 2),
                        groupnorms_from_colsums=f"{eng.gn_from_colsums} of {eng.n_groupnorms()}",
                        spatial_attention_block=attn_block,
                        rows=timed.family_rows(reps, MFMA_BF16_DENSE_PEAK_TFLOPS, HBM_PEAK_GBS))
        if traffic_note:
            roofline["traffic_note"] = traffic_note
        for r in roofline["rows"]:
            if r["name"].startswith("fused feed-forward"):
                # norm3 -> ff.net.0 (GEGLU) -> [proj_out | proj_out ff.net.2] + both residuals as one launch (csrc/ff_fused.hip)
                roofline["fused_feed_forward"] = dict(
                    kernel="seer_ff_fused_c320_kernel", launches_per_step=r["launches"], us_per_launch=r["us"], frac_mfma=r["frac_mfma"],
                    algorithmic_bytes_per_launch=2 * (3 * 24576 * 320 + 2560 * 320 + 320 * 1600) if r["name"].endswith("L0") else None,
                    traffic=round(ff_traffic) if ff_traffic else None,
                    replaces="layernorm + ff.net.0 GEGLU GEMM + [proj_out | proj_out ff.net.2] GEMM: 9.7 + 53.0 + 39.4 us per block "
                             "(profiles/r05_bench_before_ff_fused.json.log)")

    # ---- end-to-end clip latency: 50 DDIM steps + frozen VAE decode of the 10 predicted frames
    clip = None
    if rank == 0 and world == 1:
        from seervideoldm_amd import AutoencoderKL, ddim_sample
        from seervideoldm_amd.vae import ldm_to_diffusers_vae
        vae = AutoencoderKL().to(device)
        vae.load_state_dict(ldm_to_diffusers_vae(synth.synth_state_dict(synth.vae_param_shapes(), device=device), 4))
        model.use_graph = not args.no_graph
        shape = (w["b"], 4, w["frames"] - w["cond_frames"], w["latent"], w["latent"])
        for _ in range(2):          # first call builds the VAE's packed weights / warms allocations
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = ddim_sample(sampler, model, vae, shape, c, x_T, x0_emb, ddim_steps=w["ddim_steps"], scale=w["scale"], uc=uc)
            torch.cuda.synchronize()
            t_clip = time.perf_counter() - t0
        z = torch.randn((shape[0] * shape[2], 4, w["latent"], w["latent"]), device=device)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        vae.decode(z)
        torch.cuda.synchronize()
        t_dec = time.perf_counter() - t0
        assert out.shape == (w["b"], 3, shape[2], 8 * w["latent"], 8 * w["latent"]) and torch.isfinite(out).all()
        # the step before the path, once per sample: CLIP [1,77,768] -> context [1,12,77,768] (FSTextTransformer, 8 layers)
        from seervideoldm_amd import FSTextTransformer
        fst = FSTextTransformer(num_frames=16, num_layers=8).to(device)
        fst.load_state_dict(synth.synth_state_dict(synth.fstext_param_shapes(), device=device))
        fst.set_numframe(w["frames"])
        clip_txt = torch.randn((w["b"], 77, 768), device=device)
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx_out = fst(context=clip_txt)
            torch.cuda.synchronize()
            t_fst = time.perf_counter() - t0
        assert ctx_out.shape == (w["b"], w["frames"], 77, 768) and torch.isfinite(ctx_out).all()
        # ... and the VAE encode of the conditioning frames (2 x 256x256 -> x0_emb latents), also once per sample
        vae.load_state_dict(ldm_to_diffusers_vae(synth.synth_state_dict(synth.vae_encoder_param_shapes(), device=device), 4))
        frames = torch.tanh(torch.randn((w["b"] * w["cond_frames"], 3, 8 * w["latent"], 8 * w["latent"]), device=device))
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lat = vae.encode(frames).latent_dist.sample() * 0.18215
            torch.cuda.synchronize()
            t_enc = time.perf_counter() - t0
        assert lat.shape == (w["b"] * w["cond_frames"], 4, w["latent"], w["latent"]) and torch.isfinite(lat).all()
        clip = dict(clip_latency_ms=round(t_clip * 1e3, 2), vae_decode_ms=round(t_dec * 1e3, 2),
                    fstext_ms=round(t_fst * 1e3, 2), vae_encode_ms=round(t_enc * 1e3, 2),
                    what="50-step ddim_sample incl. decode of 10 frames to 256x256 (full SD-VAE decoder, synthetic weights); "
                         "once per sample before it: fstext_ms = FSTextTransformer (8 layers, 182.6M params) -> context "
                         "[1,12,77,768]; vae_encode_ms = SD-VAE encoder on the 2 conditioning frames -> x0_emb")

    # ---- extra: one fine-tuning step (BASELINE config 5: b=1 per GPU, 12 frames, 2 conditioning frames).
    # Data parallel for N > 1: one micro-batch per rank and one RCCL all-reduce of the flat fp32 gradient buffers.
    train = None
    if not args.no_train and args.workload == "sthv2":
        # an extra must never cost the headline line: any failure here is reported inside the JSON instead of raised
        # (every rank takes the same path: the collectives below stay matched)
        try:
            from scripts.bench_train import time_train
            pg = None
            if world > 1:
                import torch.distributed as dist
                pg = dist.group.WORLD
                dog.arm(args.collective_timeout_s, lambda: make_line(
                    ms_per_step, parallelism, graph_live, roofline, None, clip,
                    {"error": f"the data-parallel fine-tuning step did not finish within {args.collective_timeout_s:.0f} s"}, weak))
            train = time_train(device, steps=5, warmup=2, unet=model, process_group=pg)
            if world > 1:
                tms = torch.tensor([train["ms_per_step"]], device=device)
                dist.all_reduce(tms, op=dist.ReduceOp.MAX)
                train["ms_per_step"] = float(tms)
            train["samples_per_s"] = round(world * 1e3 / train["ms_per_step"], 3)
            train["parallelism"] = "single" if world == 1 else f"dp{world}"
            if world == 1:
                # the reference's micro-batch of 1 is a 24 GB-card setting (configs/train.yaml:11); with 288 GB per GPU the same
                # step at micro-batch 8 fills the GEMMs (same kernels, same code path: train_batch_size is a config value)
                big = time_train(device, steps=3, warmup=1, b=8, unet=model)
                train["micro_batch_8"] = {"ms_per_step": round(big["ms_per_step"], 2), "peak_mem_gb": round(big["peak_mem_gb"], 1),
                                          "samples_per_s": round(8e3 / big["ms_per_step"], 2)}
        except Exception as e:      # noqa: BLE001
            train = {"error": f"{type(e).__name__}: {e}"[:300]}
        dog.disarm()

    cpu = None
    if sd_cpu is not None:
        cpu = cpu_baseline(sd_cpu, cfg, args.cpu_budget_s)

    if rank == 0:
        print(json.dumps(make_line(ms_per_step, parallelism, graph_live, roofline, cpu, clip, train, weak)), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
