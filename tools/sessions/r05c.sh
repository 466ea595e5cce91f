#!/bin/bash
# round-5 session c: operand-fetch micro-probe (LDS-DMA vs register path), kv_layer0 tile-order A/B with clocks, RCCL with two ranks on one device
OUT=gpurun_out/r05c; mkdir -p $OUT; export TMPDIR=/tmp
echo "== operand fetch probe =="
timeout 300 build_probe/fetch_probe 2>&1 | tee $OUT/operand_fetch_probe.txt | tail -14
echo "== kv_layer0 tile order: XCD_SWIZZLE 1 (default) vs 2 (W-half-resident), wall + clocks =="
for v in 1 2 1 2; do timeout 300 python bench.py --tune XCD_SWIZZLE=$v --no-cpu-baseline --no-extras --steps 40 --warmup 10 > $OUT/bench_swz$v.json 2>> $OUT/bench.err; python - "$OUT/bench_swz$v.json" "swizzle=$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d["ms_per_step"], "kv_layer0", d["stages_ms"]["kv_layer0_gelu"], "mlp2", d["stages_ms"]["mlp2"], d["clocks"].get("during"))
PY
done
echo "== RCCL, two ranks on ONE device (expected: refused as duplicate GPU; what does the self-check / init say?) =="
NCCL_DEBUG=WARN timeout 180 python bench.py --gpus 2 --single-device --batch 8 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/rccl_one_device.out 2> $OUT/rccl_one_device.err; echo "exit $?"; tail -3 $OUT/rccl_one_device.out | cut -c1-600; grep -v "amdgpu.ids" $OUT/rccl_one_device.err | tail -12 | cut -c1-400
