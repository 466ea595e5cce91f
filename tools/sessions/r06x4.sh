#!/bin/bash
# round 6, session x4: (1) the solo kernel's spread schedule after the two hazard fixes (bit-identity + fit, vendor in the same process);
# (2) the ping-pong loop's probe builds on the whole chip and on a quarter of it (tools/loop_probe.py --reserve 24): clock or contention?
TAG=${TAG:-r06x}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python tools/solo_ab.py --rounds 5 --vendor --arms solo_dma,solo_spread --out $OUT/solo_ab_final.json 2>&1 | tail -13 | cut -c1-330
for R in 0 24 16; do echo "== loop probes, reserve $R"; timeout 400 python tools/loop_probe.py --rounds 3 --reserve $R --out $OUT/loop_probe_reserve$R.json 2>&1 | grep "^{" | cut -c1-200; done
