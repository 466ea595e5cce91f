#!/bin/bash
# round 6, session x3: the whole-line non-temporal output stores (TP_EPI_FULL_LINE=2 in all three GEMM sources) in whole steps:
# forward bench and the training step, default library vs the variant, interleaved
TAG=${TAG:-r06x}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
V=${V:-flnt}
for rep in 1 2; do
  for L in "" $V; do
    echo "== rep $rep lib=${L:-default}"
    TP_LIB_VARIANT=$L python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d.get('stages_ms'))"
    TP_LIB_VARIANT=$L python tools/train_bench.py --batches 32 256 --out $OUT/train_${L:-default}_$rep.json 2>/dev/null | grep -E "hip|ms" | head -6
  done
done
