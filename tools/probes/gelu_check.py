import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
from tokenpacker_amd import _capi
import lib_ab
libs = {"old": lib_ab.open_lib(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + "/tokenpacker_amd/libtokenpacker_hip.so"), "new": lib_ab.open_lib(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + "/tokenpacker_amd/libtokenpacker_gfast.so")}
stream = torch.cuda.current_stream().cuda_stream
M, N, K = 4096, 2048, 1024
for scale in (1.0, 4.0, 16.0):
    A = lib_ab.rand((M, K), torch.float16, 1); W = lib_ab.rand((N, K), torch.float16, 2, scale * K ** -0.5); b = lib_ab.rand((N,), torch.float32, 3)
    ref = torch.nn.functional.gelu(A.double() @ W.double().t() + b.double())
    for k, lib in libs.items():
        for odt in (torch.float16, torch.float32):
            C = torch.empty(M, N, dtype=odt, device="cuda")
            assert lib.tp_linear(ctypes.byref(lib_ab.make_args(A, W, b, C, lib_ab.G, 256)), stream) == 0
            torch.cuda.synchronize()
            d = (C.double() - ref).abs()
            rel = d / ref.abs().clamp_min(1e-6)
            print(f"scale {scale:5.1f} {k} out {str(odt)[6:]:8s} max abs err {float(d.max()):.3e}  max rel err (|ref| > 1e-3) {float(rel[ref.abs() > 1e-3].max()):.3e}  rms {float((d*d).mean().sqrt()):.3e}")
