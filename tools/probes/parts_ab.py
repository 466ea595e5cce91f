import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from tokenpacker_amd import TokenPacker, synth, tower
B, D, dtype = 256, 4096, torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(1)
hs = [None] * 25
for i in (-2 % 25, 12, 16, 22, 23):
    pass
hs = [torch.randn(B, 577, 1024, generator=g, device="cuda").to(dtype) if i in (12, 16, 22, 23) else torch.empty(0) for i in range(25)]
try:
    x, parts = tower.select_features(hs)
    x_ref, xm_ref = tower.concat_reference(hs)
except Exception as e:
    print("select_features failed:", e); raise
m = TokenPacker(hidden_size=D, scale_factor=2); m.load_state_dict(synth.make_params(2, D)); m = m.to(device="cuda", dtype=dtype).eval().requires_grad_(False)
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    ya = m((x, parts)); yb = m((x_ref, xm_ref)); print("equal", torch.equal(ya, yb))
    for rep in range(3):
        print("rep", rep, "parts %.4f ms   concatenated (tower layout view) %.4f ms   concat itself %.4f ms" % (
            timed(lambda: m((x, parts))), timed(lambda: m((x_ref, xm_ref))), timed(lambda: tower.concat_reference(hs))))
