# end-of-round measurement call (round 6): full GPU suite, in-step PMC traffic of the four bench configurations, rocprofv3
# kernel traces of both headline configurations, the bench line, per-kernel A/B of the sliding-chunk layer against the
# round-5 library and against this tree without the chunk-workgroup forward, the chunk-workgroup forward against the
# wave-per-chunk forward per shape, PMC pipe utilisation, the micro-benchmarks DESIGN 4.9 cites.
# Results -> gpurun_out/<tag>/ ; tools/collect_profiles.sh <tag> r06 copies what is to be judged into profiles/.
set -u
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r06_final}; PART=${2:-all}
OUT=gpurun_out/$TAG; mkdir -p $OUT
if [ $PART = all ] || [ $PART = 1 ]; then
rm -f gpurun_out/parity_report.txt
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest_gpu.log | tail -20
cp gpurun_out/parity_report.txt $OUT/parity_report.txt 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 900 python bench.py --detail $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.err
tail -2 $OUT/bench.err; echo "bench line bytes: $(tail -1 $OUT/bench.json | wc -c)"; tail -1 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for CFG in vil_small_224 vil_medium_deep_384; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$CFG" -o t -- python bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-eval --no-tertiary --detail "$OUT/trace_${CFG}_detail.json" > "$OUT/trace_$CFG.json" 2> "$OUT/trace_$CFG.err"
  KT=$(find "$OUT/trace_$CFG" -name "*kernel_trace.csv" | head -1)
  ST=$(find "$OUT/trace_$CFG" -name "*kernel_stats.csv" | head -1)
  python tools/trace_summary.py "$KT" "k_cw_prep(" auto 60 > "$OUT/steady_$CFG.txt" 2>&1
  cp "$ST" "$OUT/kernel_stats_$CFG.csv" 2>/dev/null
  rm -rf "$OUT/trace_$CFG"
  head -14 "$OUT/steady_$CFG.txt"
done
S=small_s1,small_s2,meddeep_s1_f7,meddeep_s2_f7,meddeep_s1_f8,meddeep_s2_f12,basedeep_s1_f6_rs,basedeep_s2_f8_rs
bash tools/attn_ab.sh $OUT/ab.txt "$S" 2 > $OUT/ab_summary.txt 2>&1; cat $OUT/ab_summary.txt
timeout 600 python tools/cw_check.py small_s1,small_s2,meddeep_s1_f7,meddeep_s2_f7,meddeep_s1_f8,basedeep_s1_f6_rs,basedeep_s2_f8_rs --reps 20 > $OUT/cw_vs_wave.txt 2>&1; cut -c1-400 $OUT/cw_vs_wave.txt
fi
if [ $PART = all ] || [ $PART = 2 ]; then
bash tools/pmc_step.sh $OUT/pmcstep > $OUT/pmcstep.txt 2>&1; tail -12 $OUT/pmcstep.txt
timeout 200 python tools/dense_bench.py --dense-only > $OUT/dense_bench.txt 2>&1
bash tools/pmc_all.sh $OUT/pmc --only sc_56x56_m32,sc_28x28_m64,sc_96x96_m32,sc_48x48_m64,sc_48x48_w12_m64,sc_96x96_w6_m32_rs,dense_14x14,dense_24x24 > $OUT/pmc.txt 2>&1; tail -40 $OUT/pmc.txt
timeout 120 tools/ubench/dma_rate > $OUT/dma_rate.txt 2>&1; tail -30 $OUT/dma_rate.txt
timeout 120 tools/ubench/lds_atomic > $OUT/lds_atomic.txt 2>&1; tail -12 $OUT/lds_atomic.txt
timeout 900 python tools/ab_bench.py LIB HEAD tools/ab/libvilattn_nocw.so --rounds 3 > $OUT/ab_bench_nocw.txt 2>&1; cat $OUT/ab_bench_nocw.txt
timeout 900 python tools/ab_bench.py LIB HEAD tools/ab/libvilattn_round5.so --rounds 2 > $OUT/ab_bench_round5.txt 2>&1; cat $OUT/ab_bench_round5.txt
fi
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2
