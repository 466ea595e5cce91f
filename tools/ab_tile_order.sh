export TMPDIR=/tmp
R=$(pwd); OUT=gpurun_out/r02h; mkdir -p $OUT
for sw in 1 2 1 2; do timeout 200 python bench.py --tune XCD_SWIZZLE=$sw --no-cpu-baseline | python -c "import json,sys; d=json.load(sys.stdin); print('swizzle', d['config']['tuning']['TP_TUNE_XCD_SWIZZLE'], d['value'], d['ms_per_step'], d['stages_ms']['kv_layer0_gelu'], d['stages_ms']['mlp2'])"; done
for sw in 1 2; do
 ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/sw$sw/pmc_FETCH -o pmc -- python $R/bench.py --tune XCD_SWIZZLE=$sw --steps 3 --warmup 2 --no-cpu-baseline > $R/$OUT/pmc_sw$sw.log 2>&1 ); echo "pmc sw$sw exit $?"
 python tools/pmc_summary.py $OUT/sw$sw | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if 'Li1ELb1ELb0ELb0' in k: print('swizzle $sw kv_layer0: duration us', round(v['duration_ns']/1e3,1), 'FETCH x2 GB', round(v['hbm_read_bytes_corrected']/1e9,3))
"
done
find $OUT -name "*kernel_trace.csv" -size +5M -delete
