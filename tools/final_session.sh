# end-of-round measurement call (round 5): full GPU suite, in-step PMC traffic of the four bench configurations, rocprofv3
# kernel traces of both headline configurations, the bench line, per-kernel A/B against the round-4 library.
# Results -> gpurun_out/<tag>/ ; tools/collect_profiles.sh <tag> r05 copies what is to be judged into profiles/.
set -u
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r05_final}
OUT=gpurun_out/$TAG; mkdir -p $OUT
bash tools/r5_session.sh $TAG testall pmcstep profile smoke > $OUT/session.log 2>&1
timeout 900 python bench.py --detail $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.err
tail -2 $OUT/bench.err; echo "bench line bytes: $(tail -1 $OUT/bench.json | wc -c)"; tail -1 $OUT/bench.json
AB_SHAPES=small_s1,small_s2,meddeep_s1_f7,meddeep_s2_f7,meddeep_s1_f8,meddeep_s2_f12,basedeep_s1_f6_rs,basedeep_s2_f8_rs bash tools/attn_ab.sh $OUT/ab.txt "small_s1,small_s2,meddeep_s1_f7,meddeep_s2_f7,meddeep_s1_f8,meddeep_s2_f12,basedeep_s1_f6_rs,basedeep_s2_f8_rs" 1 > $OUT/ab_summary.txt 2>&1
cat $OUT/ab_summary.txt
timeout 200 python tools/dense_bench.py --dense-only > $OUT/dense_bench.txt 2>&1
bash tools/pmc_all.sh $OUT/pmc --only sc_56x56_m32,sc_28x28_m64,sc_96x96_m32,sc_48x48_m64,sc_48x48_w12_m64,sc_96x96_w6_m32_rs,dense_14x14,dense_24x24 > $OUT/pmc.txt 2>&1; tail -40 $OUT/pmc.txt
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2
tail -12 $OUT/session.log
