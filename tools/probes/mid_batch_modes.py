"""Which of {fresh allocations, a light load just before, idle time just before} flips a 32-image forward into its slow mode
(profiles/r03u_mid_batch_anomaly.txt)?  One process, one box; every block = 30 warm-up + 200 timed forwards."""
import glob, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench


def sclk():
    out = []
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            out += [l.strip() for l in open(f) if "*" in l]
        except OSError:
            pass
    return out[:1]


def block(m, x, xm, n=200):
    with torch.no_grad():
        for _ in range(30):
            m((x, xm))
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n):
            m((x, xm))
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def inputs(B):
    return torch.randn(B, 576, 1024, device="cuda").bfloat16(), torch.randn(B, 576, 4096, device="cuda").bfloat16()


m = bench.build_model(4096, 2, torch.bfloat16, "cuda")
x32, xm32 = inputs(32); x1, xm1 = inputs(1); x256, xm256 = inputs(256)
log = lambda tag, v: print(f"{tag:58s} {v:.4f} ms   sclk {sclk()}", flush=True)
log("1 same model, first block B=32", block(m, x32, xm32))
log("2 again, back to back", block(m, x32, xm32))
time.sleep(0.2)
log("3 after 0.2 s idle", block(m, x32, xm32))
block(m, x1, xm1, 300)
log("4 right after 300 one-image forwards", block(m, x32, xm32))
log("5 again, back to back", block(m, x32, xm32))
block(m, x256, xm256, 60)
log("6 right after 60 forwards of 256 images", block(m, x32, xm32))
m2 = bench.build_model(4096, 2, torch.bfloat16, "cuda")
log("7 NEW model (new packed image + workspace)", block(m2, x32, xm32))
xb, xmb = inputs(32)
log("8 new model, NEW inputs", block(m2, xb, xmb))
log("9 first model again", block(m, x32, xm32))
for i in range(3):
    mi = bench.build_model(4096, 2, torch.bfloat16, "cuda")
    block(mi, x1, xm1, 100)
    log(f"10.{i} fresh model, 100 one-image forwards, then B=32", block(mi, x32, xm32))
