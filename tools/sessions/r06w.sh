#!/bin/bash
# round 6, session w: the experimental one-wave-per-SIMD kernel (csrc/experimental/tp_gemm4.hip, both fetch modes) under the same stall-counter
# split as the shipped kernel and the vendor's (r06a): where does the HIP-compiled solo layout lose?
TAG=${TAG:-r06w}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 600 python tools/solo_ab.py --rounds 5 --out $OUT/solo_ab.json 2>&1 | tail -12
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/pmc_gemm$i -o pmc -- python $R/tools/solo_ab.py --rounds 1 --out $R/$OUT/solo_ab_pmc.json > $R/$OUT/pmc_gemm$i.log 2>&1 ); echo "pmc pass $i exit $?"
done
python tools/pmc_summary.py $OUT --all > $OUT/pmc_solo_summary.json 2> $OUT/pmc_summary.err; tail -2 $OUT/pmc_summary.err
find $OUT -name "*kernel_trace.csv" -size +5M -delete
