import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from tokenpacker_amd import _capi
import solo_ab
lib = ctypes.CDLL(os.path.join(ROOT, "tokenpacker_amd", os.environ.get("SOLO_LIB", "libtokenpacker_exp.so")))
lib.tp_linear.restype = ctypes.c_int; lib.tp_linear.argtypes = [ctypes.POINTER(_capi.tp_linear_args), ctypes.c_void_p]
lib.tp_exp_gemm4.restype = ctypes.c_int; lib.tp_exp_gemm4.argtypes = [ctypes.POINTER(_capi.tp_linear_args), ctypes.c_void_p, ctypes.c_int]
stream = torch.cuda.current_stream().cuda_stream
fetch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for (M, N, K) in [(256, 256, 128), (256, 256, 256), (256, 256, 384), (256, 256, 512), (256, 256, 1024), (512, 512, 256)]:
    dtype = torch.float16
    # A = one-hot structure: make it easy to see which K-tile / k-half goes wrong: A[r, k] = 1 for all, W[n, k] = (k // 32 + 1) * 2^-? -> C = sum
    A = torch.ones(M, K, dtype=dtype, device="cuda")
    W = torch.zeros(N, K, dtype=dtype, device="cuda")
    for h in range(K // 32):
        W[:, 32 * h: 32 * h + 1] = float(2 ** (h % 11))     # k-half h contributes bit h
    for rep in range(2):
        C0 = torch.zeros(M, N, dtype=torch.float32, device="cuda"); C1 = torch.zeros_like(C0)
        a0 = solo_ab.make_args(A, W, None, C0, 0); a1 = solo_ab.make_args(A, W, None, C1, 0)
        assert lib.tp_linear(ctypes.byref(a0), stream) == 0
        assert lib.tp_exp_gemm4(ctypes.byref(a1), stream, fetch) == 0
        torch.cuda.synchronize()
        d = (C0 != C1)
        print((M, N, K), "rep", rep, "mismatch", int(d.sum()), "of", d.numel())
        if d.any():
            blk = d.view(M // 16, 16, N // 16, 16).any(3).any(1)
            print(" bad 16x16 blocks:", int(blk.sum()), "rows(frag) bad:", blk.any(1).nonzero().flatten().tolist()[:40], "cols bad:", blk.any(0).nonzero().flatten().tolist()[:40])
            idx = d.nonzero()[0].tolist()
            print(" first bad", idx, "ref", float(C0[idx[0], idx[1]]), "got", float(C1[idx[0], idx[1]]))
            vals = torch.unique(C1[d])[:10].tolist(); print(" got values:", vals, " ref:", torch.unique(C0)[:4].tolist())
    # random data
    A = solo_ab.rand((M, K), dtype, 1); W = solo_ab.rand((N, K), dtype, 2, K ** -0.5)
    C0 = torch.zeros(M, N, dtype=torch.float32, device="cuda"); C1 = torch.zeros_like(C0)
    lib.tp_linear(ctypes.byref(solo_ab.make_args(A, W, None, C0, 0)), stream); lib.tp_exp_gemm4(ctypes.byref(solo_ab.make_args(A, W, None, C1, 0)), stream, fetch)
    torch.cuda.synchronize(); d = C0 != C1
    print((M, N, K), "random: mismatch", int(d.sum()), "max|d|", float((C0 - C1).abs().max()), "max|C|", float(C0.abs().max()))
    if d.any():
        cnt = d.view(M // 16, 16, N // 16, 16).sum(3).sum(1)
        print(" mismatches per 16x16 block (first 16 x 16 blocks):"); print(cnt[:16, :16].tolist())
        blk = d.view(M // 16, 16, N // 16, 16).any(3).any(1)
        print(" rows(frag) bad:", blk.any(1).nonzero().flatten().tolist()[:40], "cols bad:", blk.any(0).nonzero().flatten().tolist()[:40])
