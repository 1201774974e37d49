import sys, torch
sys.path.insert(0, '/root/repo')
from scripts.bench_train import build
from seervideoldm_amd.trainer import SeerTrainer, cosine_lr
dev = torch.device('cuda:0')
unet, fst = build(dev)
fst.set_numframe(12)
tr = SeerTrainer(unet, fst, lr=1.28e-5 * 8, max_grad_norm=0.3, gradient_accumulation_steps=2)
g = torch.Generator().manual_seed(0)
acp = torch.cumprod(1 - torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000) ** 2, 0).to(dev)
data = [(torch.randn((1, 4, 2, 32, 32), generator=g).to(dev), torch.randn((1, 4, 10, 32, 32), generator=g).to(dev),
         torch.randn((1, 77, 768), generator=g).to(dev)) for _ in range(4)]
losses = []
for it in range(60):
    x0, lat, text = data[it % 4]
    noise = torch.randn(lat.shape, generator=g).to(dev)
    t = torch.randint(0, 1000, (1,), generator=g).to(dev)
    loss = tr.train_step(x0, lat, noise, t, text, acp, lr=cosine_lr(tr.step_count, 1.28e-5 * 8, 5, 40), use_graph=True)
    losses.append(float(loss))
print("losses first 6:", [round(l, 4) for l in losses[:6]])
print("losses last 6:", [round(l, 4) for l in losses[-6:]])
print("optimizer steps:", tr.step_count, "finite:", all(l == l for l in losses), "mean first 10 / last 10:", sum(losses[:10]) / 10, sum(losses[-10:]) / 10)
assert torch.isfinite(tr.pu.p).all() and torch.isfinite(tr.pf.p).all()
