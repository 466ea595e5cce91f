#!/bin/bash
# round-5 session d: operand-fetch probe by address pattern; the gemm8 loop probes (DMA only, MFMA only, ...) with 3 and with 4 DMA groups in flight
OUT=gpurun_out/r05d; mkdir -p $OUT; export TMPDIR=/tmp
echo "== operand fetch probe =="
timeout 300 build_probe/fetch_probe 2>&1 | tee $OUT/operand_fetch_probe.txt | tail -16
echo "== loop probe, DEPTH 3 (product build) =="
timeout 300 python tools/loop_probe.py --out $OUT/loop_probe_d3.json 2>&1 | grep -v amdgpu.ids | cut -c1-200
echo "== loop probe, DEPTH 4 (variant s1d4) =="
TP_LIB_VARIANT=s1d4 timeout 300 python tools/loop_probe.py --out $OUT/loop_probe_d4.json 2>&1 | grep -v amdgpu.ids | cut -c1-200
