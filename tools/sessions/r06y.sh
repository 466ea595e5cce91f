#!/bin/bash
# round-6 FINAL session on the frozen sources: the whole GPU suite, the bench line, sweeps, rocprof + PMC (traffic.json), training (both
# gradient chains), e2e, HD, the N = 1 line through a one-rank nccl group, the 128-seed parity sweep
TAG=${TAG:-r06y}
bash tools/gpu_round.sh $TAG smoke tests gemm bench sweep prof pmc small e2e train prof3 trainprof
OUT=gpurun_out/$TAG; export TMPDIR=/tmp
echo "== training step with the bf16 gradient chain of rounds 1-5 (TP_TUNE_BWD_CHAIN = 1) =="
timeout 600 python tools/train_bench.py --tune BWD_CHAIN=1 --out $OUT/train_bench_bf16_chain.json 2>&1 | grep "^{" | cut -c1-260
echo "== bench through the one-rank nccl group (--force-dist) =="
timeout 300 python bench.py --gpus 1 --force-dist --no-cpu-baseline --no-extras > $OUT/bench_force_dist.json 2>> $OUT/bench.err; tail -c 600 $OUT/bench_force_dist.json
echo "== parity, 128 seeds, every configuration, on these sources =="
S=$(date +%s); timeout 900 python tools/parity_sweep.py --seeds 128 --workers 16 --out $OUT/parity_seed_sweep.json 2>&1 | grep parity-sweep; echo "sweep wall $(( $(date +%s)-S )) s"
find $OUT -name "*kernel_trace.csv" -size +5M -delete; du -sh $OUT
