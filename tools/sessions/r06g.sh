#!/bin/bash
# round-6 full session on the current sources: GPU suite, bench line, sweeps, rocprof + PMC (traffic.json), training, e2e, HD, 128-seed parity
TAG=${TAG:-r06g}
bash tools/gpu_round.sh $TAG smoke tests gemm bench sweep prof pmc small e2e train prof3
OUT=gpurun_out/$TAG; export TMPDIR=/tmp
echo "== bench through the one-rank nccl group (--force-dist) =="
timeout 300 python bench.py --gpus 1 --force-dist --no-cpu-baseline --no-extras > $OUT/bench_force_dist.json 2>> $OUT/bench.err; tail -c 1500 $OUT/bench_force_dist.json
echo "== parity, 128 seeds, every configuration, on these sources =="
S=$(date +%s); timeout 900 python tools/parity_sweep.py --seeds 128 --workers 16 --out $OUT/parity_seed_sweep.json 2>&1 | grep parity-sweep; echo "sweep wall $(( $(date +%s)-S )) s"
find $OUT -name "*kernel_trace.csv" -size +5M -delete; du -sh $OUT
