#!/bin/bash
# round 6, session x2: the epilogue's store bursts — variants of tp_gemm8.hip against the shipped build (tools/lib_ab.py, bit-identity + fit):
#   ladj     = output stores with adjacent lanes adjacent in memory (TP_EPI_LANE_ADJ)
#   st3      = the workgroups of an XCD start 0.3 us apart (TP_G8_STAGGER_NS=300)
#   ladjst3 / ladjst10 = both (0.3 / 1.0 us)
TAG=${TAG:-r06x}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for V in ${VARIANTS:-ladj st3 ladjst3 ladjst10}; do
  echo "== $V"; timeout 300 python tools/lib_ab.py --old tokenpacker_amd/libtokenpacker_hip.so --new tokenpacker_amd/libtokenpacker_$V.so --rounds 5 --out $OUT/lib_ab_$V.json 2>&1 | grep -v amdgpu.ids | cut -c1-230
done
