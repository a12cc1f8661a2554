#!/bin/bash
# HBM traffic of every library kernel INSIDE THE REAL TRAINING STEP (not an isolated micro-benchmark): rocprofv3 --pmc
# passes (counters only, one group per run, as the MI355X guide prescribes) over `bench.py --graph off`, for ViL-Small@224
# and ViL-Medium-Deep@384 -- round 5: and the 384 recipe configurations (f8 / f12, Base-Deep random shift), CONFIGS in the
# environment overrides the list.  usage (on the GPU box): tools/pmc_step.sh <outdir>   -> <outdir>/pmc_traffic.json
set -u
OUT=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
rm -f "$OUT/tags.json"
for CFG in ${CONFIGS:-vil_small_224 vil_medium_deep_384 vil_medium_deep_384_f8f12 vil_base_deep_384_rs}; do
  ARGS="bench.py --graph off --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-eval --no-tertiary --config $CFG --detail $OUT/detail_$CFG.json"
  VIL_BENCH_DUMP_TAGS=$PWD/$OUT/tags.json timeout 600 python $ARGS > "$OUT/plain_$CFG.json" 2> "$OUT/plain_$CFG.err"
  timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_$CFG" -o pmc -- python $ARGS > "$OUT/fetch_$CFG.log" 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/write_$CFG" -o pmc -- python $ARGS > "$OUT/write_$CFG.log" 2>&1
done
python tools/pmc_step_summary.py "$OUT" "$OUT/pmc_traffic.json"
