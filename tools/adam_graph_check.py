"""Developer check: version counters under torch's fused Adam (the caches of GaussianModel key on them)."""
import torch
for fused in (False, True):
    p = torch.nn.Parameter(torch.randn(10, 3, device="cuda"))
    opt = torch.optim.Adam([{"params": [p], "lr": 1e-3}], lr=0.0, eps=1e-15, capturable=True, fused=fused)
    v0 = p._version
    p.grad = torch.ones_like(p)
    opt.step()
    print("fused", fused, "version before/after step", v0, p._version)
