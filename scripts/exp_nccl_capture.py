"""Can RCCL collectives be captured into a hipGraph on this stack (torch 2.10 + ROCm 7)?  world_size 1 on one GPU: checks the
mechanism (capture, replay, result), not the transport."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.ones(64, device=dev)
parts = torch.zeros(64, device=dev)
out = torch.empty(64, device=dev)
dist.all_reduce(x)                      # warm-up: communicator creation must not happen under capture
dist.all_gather_into_tensor(out, x)
torch.cuda.synchronize()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        y = x * 2
        dist.all_reduce(y)
        dist.all_gather_into_tensor(out, y)
        z = out + 1
torch.cuda.current_stream().wait_stream(s)
x.fill_(3.0)
g.replay()
torch.cuda.synchronize()
print("captured all_reduce + all_gather replayed:", z[:4].tolist(), "(expect 7.0)")
t0 = time.perf_counter()
for _ in range(200):
    g.replay()
torch.cuda.synchronize()
print(f"replay of [mul, all_reduce, all_gather, add]: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us")
t0 = time.perf_counter()
for _ in range(200):
    y = x * 2
    dist.all_reduce(y)
    dist.all_gather_into_tensor(out, y)
    z = out + 1
torch.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us")
dist.destroy_process_group()
