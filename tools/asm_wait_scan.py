#!/usr/bin/env python3
"""Static check of the persistent GEMM kernels' tile seam (tokenpacker_amd/csrc: `make asm` first).

For every persistent instantiation of gemm8_kernel in build_asm/tp_gemm8-*.s, print what follows the last MFMA of the K loop
in program order:  D = LDS-DMA instruction, g = ordinary global load, s = store, [An] = an s_waitcnt vmcnt(n) written in the
source (inline asm), Wn = one the COMPILER inserted.  A `W` between the parameter loads / the DMA prologue and the stores
means hipcc found a (false) dependency on an in-flight load and the tile seam stalls for a memory round trip — that is how
the 24-bit offset multiply and the DMA-staged parameters of the attention variants were found.  Also
prints VGPR count and scratch bytes (must be 0).

    python tools/asm_wait_scan.py [path/to/tp_gemm8-hip-amdgcn-amd-amdhsa-gfx950.s]
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tokenpacker_amd", "csrc", "build_asm",
                                                           "tp_gemm8-hip-amdgcn-amd-amdhsa-gfx950.s")
s = open(path).read()
bad = 0
for m in re.finditer(r"\n(_ZN2tp12gemm8_kernel\S+):", s):
    name = m.group(1)
    # gemm8_kernel<TI, TO, AMODE, TRAIN_EPI, HALF, XMODE, PROBE, T192> (always persistent since round 4): the inference
    # instantiations (TRAIN_EPI = false, PROBE = 0) are the ones whose seam must be free of compiler-inserted waits; the training
    # epilogues load their saved pre-activations with ordinary loads on purpose
    mm = re.search(r"Li(\d+)ELb([01])ELb([01])ELi(\d+)ELi(\d+)ELb([01])E", name)
    if not mm or mm.group(2) == "1" or mm.group(5) != "0":
        continue
    i = m.start()
    j = s.index(".Lfunc_end", i)
    body = s[i:j].split("\n")
    mf = [k for k, l in enumerate(body) if "v_mfma" in l]
    if not mf:
        continue
    ev = []
    for k, l in enumerate(body):
        t = l.strip()
        if k < mf[-1]:
            continue
        if "vmcnt" in t:
            n = re.search(r"vmcnt\((\d+)\)", t).group(1)
            ev.append(f"[A{n}]" if "ASM" in body[k - 1] else f"W{n}")
        elif " lds" in t and "buffer_load" in t:
            ev.append("D")
        elif "global_load" in t:
            ev.append("g")
        elif "buffer_store" in t or "global_store" in t:
            ev.append("s")
    tail = s[j:j + 3000]
    vg = re.search(r"; NumVgprs: (\d+)", tail).group(1)
    sc = re.search(r"; ScratchSize: (\d+)", tail).group(1)
    seq = "".join(ev)
    first_store = seq.find("s")
    seam = seq[:first_store] if first_store >= 0 else seq
    flag = "W" in seam and "Lb1ELb1E" not in name              # (training epilogues load Z / pre-activations on purpose)
    bad += flag or sc != "0"
    print(f"{name[20:66]:46s} vgpr {vg:>3s} scratch {sc:>3s}  {seq[:64]}{'   <-- compiler wait at the seam' if flag else ''}")
print("compiler waits at a tile seam or scratch use:", bad)
sys.exit(1 if bad else 0)
