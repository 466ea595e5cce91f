#!/bin/bash
# round 6, session x: the solo kernel with the SPREAD schedule (fetch 2 / 3, tp_gemm4.hip header) against the ping-pong kernel, the
# round-5 solo schedule and the vendor's kernel (torch.matmul) on the probe shapes: bit-identity, per-K-tile / per-tile fit; then the probe
# builds (no DMA / no fragment reads / MFMAs only / no MFMAs) of the spread schedule
TAG=${TAG:-r06x}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python tools/solo_ab.py --rounds 5 --vendor --arms ${ARMS:-solo_dma,solo_spread} --out $OUT/solo_ab.json 2>&1 | tail -14
for D in 1 2 3 4; do
  echo "== probe build dbg=$D"; timeout 300 python tools/solo_ab.py --rounds 3 --dbg $D --arms ${ARMS:-solo_dma,solo_spread} --out $OUT/solo_ab_dbg$D.json 2>&1 | tail -3
done
