"""Does a side-stream branch inside ONE captured HIP graph overlap with the main chain on this platform?  The ResnetBlock3D
case: the 1x1 shortcut GEMM over x|skip is independent of norm1 -> conv1 -> norm2; captured on a forked stream it could hide
behind the (memory-bound) GroupNorm launches.  Prints the replay time of the chain alone, the shortcut alone, both serial and
both forked.

    python scripts/exp_graph_branch.py > gpurun_out/graph_branch.log
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seervideoldm_amd import ops  # noqa: E402

bf16 = torch.bfloat16


def replay_us(build, iters=30):
    build()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(5):
            build()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (iters * 5)


def main():
    dev = torch.device("cuda:0")
    B, G = 2, 32
    print("level: rows, C(x)+C(skip) -> Cout      chain us   shortcut us   serial us   forked us")
    for rows, C1, C2, Co, H in [(12288, 320, 320, 320, 32), (12288, 640, 320, 320, 32), (3072, 640, 640, 640, 16), (768, 1280, 1280, 1280, 8)]:
        x = torch.randn((B * rows, C1), device=dev).to(bf16)
        skip = torch.randn((B * rows, C2), device=dev).to(bf16)
        Ct = C1 + C2
        gm, bt = torch.ones((Ct,), device=dev), torch.zeros((Ct,), device=dev)
        stats = torch.zeros((B, G, 2), device=dev)
        wsc = (torch.randn((Co, Ct), device=dev) / Ct ** 0.5).to(bf16)
        wcv = (torch.randn((Co, 9 * Ct), device=dev) / (9 * Ct) ** 0.5).to(bf16)
        sc_out = torch.empty((B * rows, Co), device=dev, dtype=bf16)
        side = torch.cuda.Stream()

        def chain():
            ops.groupnorm_stats(x, skip, B, G, stats)
            h = ops.groupnorm_apply(x, skip, B, G, stats, rows * (Ct // G), 1e-5, gm, bt, True)
            return ops.conv3x3(h, wcv, B * rows // (H * H), H, H)

        def shortcut():
            ops.gemm(x, wsc, a2=skip, out=sc_out)

        def serial():
            shortcut()
            chain()

        def forked():
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                shortcut()
            chain()
            cur.wait_stream(side)

        print(f"{rows:6d}, {C1}+{C2} -> {Co}:   {replay_us(chain):9.1f}   {replay_us(shortcut):9.1f}   {replay_us(serial):9.1f}   {replay_us(forked):9.1f}")


if __name__ == "__main__":
    main()
