"""A/B of two builds of the library in ONE process (same box, same clocks): times B = 32 / 64 forwards with each."""
import sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tokenpacker_amd import _capi
import bench
print("CUs", torch.cuda.get_device_properties(0).multi_processor_count, flush=True)
def run(tag):
    m = bench.build_model(4096, 2, torch.bfloat16, "cuda")
    for B in (32, 64, 1):
        x = torch.randn(B, 576, 1024, device="cuda").bfloat16(); xm = torch.randn(B, 576, 4096, device="cuda").bfloat16()
        with torch.no_grad():
            for _ in range(30): m((x, xm))
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(200): m((x, xm))
            torch.cuda.synchronize(); print(tag, "B=%d" % B, "%.4f ms" % ((time.perf_counter() - t) / 200 * 1e3), flush=True)
run("new")
_capi._lib = None
_capi.load_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "old", "libtokenpacker_hip_old.so"))    # built by hand from an earlier tp_gemm.hip (not kept in the tree)
run("old")
_capi._lib = None
_capi.load_library()
run("new-again")
