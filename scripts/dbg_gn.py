import torch, torch.nn.functional as Fn, sys
sys.path.insert(0, ".")
from seervideoldm_amd import ops
dev = torch.device("cuda:0")
B, rows, C, G = 2, 1024, 128, 32
g = torch.Generator().manual_seed(1)
x32 = (torch.randn((B * rows, C), generator=g) * 3).to(dev)
gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
for dt in (torch.bfloat16, torch.float16):
    x = x32.to(dt)
    stats = torch.zeros((B, G, 2), device=dev)
    ops.groupnorm_stats(x, None, B, G, stats)
    v = x.double().reshape(B, rows, G, C // G)
    ref = torch.stack([v.sum(dim=(1, 3)), (v * v).sum(dim=(1, 3))], -1)
    print(dt, "stats err", (stats.double() - ref).abs().max().item(), "ref max", ref.abs().max().item())
    y = ops.groupnorm_apply(x, None, B, G, stats, rows * (C // G), 1e-6, gamma, beta, False)
    r = Fn.group_norm(x.float().reshape(B, rows, C).permute(0, 2, 1), G, gamma, beta, 1e-6).permute(0, 2, 1).reshape(B * rows, C)
    print(dt, "apply err", (y.float() - r).abs().max().item(), y.dtype, y[:2, :4].tolist(), r[:2, :4].tolist())
