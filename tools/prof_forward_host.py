import cProfile, pstats, time, torch, sys
sys.path.insert(0, '.')
from tokenpacker_amd import TokenPacker, synth
m = TokenPacker(hidden_size=4096, scale_factor=2).to(device="cuda", dtype=torch.bfloat16).eval().requires_grad_(False)
x = torch.randn(1, 576, 1024, device="cuda").to(torch.bfloat16); xm = torch.randn(1, 576, 4096, device="cuda").to(torch.bfloat16)
with torch.no_grad():
    for _ in range(50): m((x, xm))
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(2000): m((x, xm))
    t_enq = time.perf_counter() - t
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t
    print("enqueue %.1f us/fwd, incl. drain %.1f us/fwd" % (t_enq / 2000 * 1e6, t_all / 2000 * 1e6))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(2000): m((x, xm))
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
