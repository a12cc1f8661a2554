#!/bin/bash
# round-4 diagnostic session: LDS-DMA soak tests, PMC counters of every hot kernel, dense-forward ablations
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04_s1; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_2_stress.py -q -x > $OUT/stress.log 2>&1; echo "stress rc=$?" >> $OUT/stress.log; tail -4 $OUT/stress.log
for LIB in "" $(ls tools/ab/*.so); do
  echo "== lib=${LIB:-HEAD}" >> $OUT/ab.txt
  VIL_ATTN_LIB=${LIB:+$PWD/$LIB} timeout 200 python tools/kernel_bench.py meddeep_s3_dense,small_s3_dense --fwd-only --reps 20 >> $OUT/ab.txt 2>&1
done
cat $OUT/ab.txt
bash tools/pmc_all.sh $OUT/pmc > $OUT/pmc.txt 2>&1; tail -60 $OUT/pmc.txt
