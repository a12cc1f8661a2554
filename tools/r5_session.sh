#!/bin/bash
# Round-5 gpurun sessions: tools/r5_session.sh <tag> [ubench|test|testall|ab|bench|benchfull|smoke|pmcstep|profile|dense|wgrad]...
# Everything lands under gpurun_out/<tag>/ ; AB_SHAPES / REPS / ROUNDS tune the A/B step.
set -u
TAG=$1; shift
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
rm -f gpurun_out/parity_report.txt
for STEP in "$@"; do
  T0=$(date +%s)
  case $STEP in
    ubench)
      timeout 120 tools/ubench/valu_rate > "$OUT/valu_rate.txt" 2>&1; cat "$OUT/valu_rate.txt" ;;
    test)
      timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1
      echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"; tail -8 "$OUT/pytest_gpu.log"
      cp gpurun_out/parity_report.txt "$OUT/parity_report.txt" 2>/dev/null ;;
    testall)
      timeout 900 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1
      echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"; grep -E "^FAILED|^ERROR|passed|failed" "$OUT/pytest_gpu.log" | tail -40
      cp gpurun_out/parity_report.txt "$OUT/parity_report.txt" 2>/dev/null ;;
    ab)
      bash tools/attn_ab.sh "$OUT/ab.txt" "${AB_SHAPES:-small_s1,meddeep_s1_f7,small_s2}" "${ROUNDS:-2}" > "$OUT/ab_summary.txt" 2>&1; cat "$OUT/ab_summary.txt" ;;
    dense)
      for LIB in "" $(ls tools/ab/libvilattn_*.so 2>/dev/null); do
        echo "== lib=${LIB:-HEAD}" >> "$OUT/dense.txt"
        VIL_ATTN_LIB=${LIB:+$PWD/$LIB} timeout 200 python tools/kernel_bench.py small_s3_dense,meddeep_s3_dense,small_s4_dense,meddeep_s4_dense --reps 20 >> "$OUT/dense.txt" 2>&1
      done; cat "$OUT/dense.txt" ;;
    bench)
      timeout 600 python bench.py --detail "$OUT/bench_detail.json" > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$? bytes=$(tail -1 "$OUT/bench.json" | wc -c)"; tail -1 "$OUT/bench.json" ;;
    benchquick)
      timeout 600 python bench.py --no-tertiary --no-eval --no-cpu-baseline --detail "$OUT/benchq_detail.json" > "$OUT/benchq.json" 2> "$OUT/benchq.err"; tail -1 "$OUT/benchq.json" ;;
    smoke)
      python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; tail -2 "$OUT/smoke.log" ;;
    pmcstep)
      bash tools/pmc_step.sh "$OUT/pmcstep" > "$OUT/pmcstep.txt" 2>&1; tail -30 "$OUT/pmcstep.txt" ;;
    profile)
      cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
      for CFG in vil_small_224 vil_medium_deep_384; do
        timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$CFG" -o t -- python bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-eval --no-tertiary --detail "$OUT/trace_${CFG}_detail.json" > "$OUT/trace_$CFG.json" 2> "$OUT/trace_$CFG.err"
        KT=$(find "$OUT/trace_$CFG" -name "*kernel_trace.csv" | head -1)
        ST=$(find "$OUT/trace_$CFG" -name "*kernel_stats.csv" | head -1)
        python tools/trace_summary.py "$KT" "void k_mfma_fwd<bf16,2>" 1 60 > "$OUT/steady_$CFG.txt" 2>&1
        cp "$ST" "$OUT/kernel_stats_$CFG.csv" 2>/dev/null
        rm -rf "$OUT/trace_$CFG"
        head -14 "$OUT/steady_$CFG.txt"
      done ;;
    *) echo "unknown step $STEP" ;;
  esac
  echo "[step $STEP: $(( $(date +%s) - T0 )) s]"
done
