#!/bin/bash
# round 6, session f: kernel timeline of the B = 256 forward (s = 2 and s = 3): where the query side sits, gaps between launches
TAG=${TAG:-r06f}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}
for sf in 2 3; do
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/trace_s$sf -o t -- python $R/bench.py --scale-factor $sf --steps 12 --warmup 4 --no-cpu-baseline --no-extras --min-seconds 0 > $R/$OUT/trace_s$sf.log 2>&1 ); echo "trace s=$sf exit $?"
python - $sf <<'PY'
import csv, glob, sys
sf = sys.argv[1]
f = glob.glob(f"gpurun_out/r06f/trace_s{sf}/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "fillBufferAligned" in r["Kernel_Name"]]
i0, i1 = idx[-4], idx[-3]
t0 = int(rows[i0]["Start_Timestamp"])
print(f"== s={sf} B=256: kernels from one forward's memset to the next (start us, end us, dur us, queue, name)")
for r in rows[i0:i1 + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:8.1f} {e/1e3:8.1f} {(e-s)/1e3:7.1f}  q{r.get('Queue_Id','?')}  {r['Kernel_Name'][:100]}")
PY
done
find $OUT -name "*kernel_trace.csv" -size +5M -delete
