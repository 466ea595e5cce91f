import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokenpacker_amd import TokenPacker
B, s, D, dtype = 32, 2, 4096, torch.bfloat16
torch.manual_seed(0)
m = TokenPacker(hidden_size=D, scale_factor=s).to(device="cuda", dtype=dtype)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(B, 576, 1024, generator=g, device="cuda").to(dtype)
xm = torch.randn(B, 576, 4096, generator=g, device="cuda").to(dtype)
w = torch.randn(B, 144, D, generator=g, device="cuda").to(dtype)
opt = torch.optim.SGD(m.parameters(), lr=1e-6)
for it in range(4):
    m.zero_grad(set_to_none=True)
    y = m((x, xm))
    print(it, "y finite", bool(torch.isfinite(y.float()).all()), float(y.float().abs().max()))
    (y * w).sum().backward()
    bad = [k for k, p in m.named_parameters() if not torch.isfinite(p.grad.float()).all()]
    print(it, "non-finite grads:", bad, "max|g|", max(float(p.grad.float().abs().max()) for p in m.parameters()))
    opt.step()
    badw = [k for k, p in m.named_parameters() if not torch.isfinite(p.float()).all()]
    print(it, "non-finite weights:", badw)
with torch.no_grad():
    y = m((x, xm))
print("inference after training finite:", bool(torch.isfinite(y.float()).all()))
