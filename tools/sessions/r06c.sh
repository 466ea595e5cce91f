#!/bin/bash
# round 6, session c: kernel timeline of a 32-image forward with the K launch on the side stream (DECOUPLE_K=0) / on the caller's (2)
TAG=${TAG:-r06c}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}
for k in 0 2; do
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/trace_k$k -o t -- python $R/bench.py --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --min-seconds 0 --tune DECOUPLE_K=$k > $R/$OUT/trace_k$k.log 2>&1 ); echo "trace k=$k exit $?"
done
python - <<'PY'
import csv, glob, os
for k in (0, 2):
    f = glob.glob(f"gpurun_out/r06c/trace_k{k}/**/*kernel_trace.csv", recursive=True)
    if not f: print("no trace", k); continue
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last complete forward: find the last 'point_queries' kernel and print 14 kernels from there
    idx = [i for i, r in enumerate(rows) if "point_queries" in r["Kernel_Name"]]
    i0 = idx[-12]
    t0 = int(rows[i0]["Start_Timestamp"])
    print(f"== DECOUPLE_K={k}: kernels of one forward (start us, end us, dur us, queue, name)")
    for r in rows[i0:idx[-11]]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print(f"{s/1e3:8.1f} {e/1e3:8.1f} {(e-s)/1e3:7.1f}  q{r.get('Queue_Id','?')}  {r['Kernel_Name'][:90]}")
PY
find $OUT -name "*kernel_trace.csv" -size +5M -delete
