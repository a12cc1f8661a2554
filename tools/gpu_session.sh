#!/bin/bash
# One gpurun call of the build/measure loop:  tools/gpu_session.sh <tag> [steps...]
#   steps: test  ab  bench  pmc_<shape>  meddeep
# Everything lands under gpurun_out/<tag>/.
set -u
TAG=$1; shift
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
rm -f gpurun_out/parity_report.txt
for STEP in "$@"; do
  case $STEP in
    test)
      timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1
      echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"; tail -5 "$OUT/pytest_gpu.log"
      cp gpurun_out/parity_report.txt "$OUT/parity_report.txt" 2>/dev/null ;;
    testall)      # no -x: every failure listed
      timeout 900 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1
      echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"; tail -15 "$OUT/pytest_gpu.log"
      cp gpurun_out/parity_report.txt "$OUT/parity_report.txt" 2>/dev/null ;;
    ab)
      for SH in ${AB_SHAPES:-small_s1 small_s2 meddeep_s1_f7 small_s3_dense basedeep_s1_f6_rs}; do
        for LIB in $(ls tools/ab/*.so) ""; do
          echo "== $SH lib=${LIB:-HEAD}" >> "$OUT/ab.txt"
          VIL_ATTN_LIB=${LIB:+$PWD/$LIB} timeout 200 python tools/kernel_bench.py $SH --reps 10 >> "$OUT/ab.txt" 2>&1
        done
      done
      cat "$OUT/ab.txt" ;;
    bench)
      timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; tail -c 3000 "$OUT/bench.json" ;;
    bench2)      # the driver's multi-GPU command line, two ranks SHARING this box's one GPU (gloo): exercises the whole N > 1 flow
      VIL_SHARE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
        bench.py --gpus 2 --steps 5 --warmup 2 > "$OUT/bench2.json" 2> "$OUT/bench2.err"
      tail -c 600 "$OUT/bench2.err"; python -c "import json; d=json.loads(open('$OUT/bench2.json').read().strip().splitlines()[-1]); print('2 ranks on 1 GPU:', d['value'], d['ms_per_step'], d['config']['launch'], d['comm'], d['secondary']['value'])" ;;
    configs)      # the other BASELINE configurations (parity-test cases, not bench lines): one short run each
      for CFG in vil_tiny_224 vil_medium_deep_384_f8f12 vil_base_deep_384_rs; do
        timeout 600 python bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_$CFG.json" 2> "$OUT/bench_$CFG.err"
        python -c "import json; d=json.load(open('$OUT/bench_$CFG.json')); print('$CFG', d['value'], d['ms_per_step'], d['hot_path_ms_per_step'], d['roofline'] and (d['roofline']['shape'], d['roofline']['frac']))"
      done ;;
    meddeep)
      timeout 600 python bench.py --config vil_medium_deep_384 --no-cpu-baseline > "$OUT/bench_meddeep.json" 2> "$OUT/bench_meddeep.err"
      tail -c 1500 "$OUT/bench_meddeep.json" ;;
    pmc_*)
      SH=${STEP#pmc_}
      bash tools/pmc.sh $SH "$OUT/pmc_$SH" > "$OUT/pmc_$SH.txt" 2>&1; grep -A40 "k_mfma_fwd\|k_mfma_bwd" "$OUT/pmc_$SH.txt" | grep -- "--\|^k_" ;;
    trafficab)      # isolated HBM traffic of the three MFMA kernels at one shape, for every A/B library
      cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
      for LIB in $(ls tools/ab/*.so) ""; do
        N=$(basename "${LIB:-HEAD}" .so)
        for P in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
          D="$OUT/traffic_$N/p_${P%% *}"; mkdir -p "$OUT/traffic_$N"
          VIL_ATTN_LIB=${LIB:+$PWD/$LIB} timeout 300 rocprofv3 --pmc $P --output-format csv -d "$D" -o pmc -- python tools/kernel_bench.py ${TRAFFIC_SHAPE:-small_s1} --reps 2 > "$D.log" 2>&1
        done
        echo "== $N" >> "$OUT/trafficab.txt"; python tools/pmc_summary.py "$OUT/traffic_$N" 2>/dev/null | grep -A3 "^k_mfma_fwd\|^k_mfma_bwd" | grep "^k_\|HBM" >> "$OUT/trafficab.txt"
      done
      cat "$OUT/trafficab.txt" ;;
    pmcstep)
      bash tools/pmc_step.sh "$OUT/pmcstep" > "$OUT/pmcstep.txt" 2>&1; tail -30 "$OUT/pmcstep.txt" ;;
    profile)      # rocprofv3 kernel trace + stats of the bench command (graph mode) and its steady-state summary
      cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
      for CFG in vil_small_224 vil_medium_deep_384; do
        timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$CFG" -o t -- python bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-eval --no-tertiary > "$OUT/trace_$CFG.json" 2> "$OUT/trace_$CFG.err"
        KT=$(find "$OUT/trace_$CFG" -name "*kernel_trace.csv" | head -1)
        ST=$(find "$OUT/trace_$CFG" -name "*kernel_stats.csv" | head -1)
        MARK="void k_mfma_fwd<bf16,2>"
        python tools/trace_summary.py "$KT" "$MARK" 1 60 > "$OUT/steady_$CFG.txt" 2>&1
        cp "$ST" "$OUT/kernel_stats_$CFG.csv" 2>/dev/null
        gzip -c "$KT" > "$OUT/kernel_trace_$CFG.csv.gz"; rm -rf "$OUT/trace_$CFG"            # (the raw trace is tens of MB)
        head -14 "$OUT/steady_$CFG.txt"
      done ;;
    opbench)
      timeout 900 python tools/op_benchmark.py > "$OUT/op_benchmark.jsonl" 2> "$OUT/op_benchmark.err"; cat "$OUT/op_benchmark.jsonl" ;;
    smoke)
      python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; tail -2 "$OUT/smoke.log" ;;
    *) echo "unknown step $STEP" ;;
  esac
done
