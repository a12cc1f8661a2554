# end-of-round measurement call: full GPU suite, in-step PMC traffic, rocprofv3 traces of both configurations, the bench
# line, counters of every hot kernel, the reference's operator protocol.  Results -> gpurun_out/r04_final/ (copy what is
# to be judged into profiles/ afterwards: tools/collect_profiles.sh r04_final r04)
set -u
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r04_final}
OUT=gpurun_out/$TAG; mkdir -p $OUT
bash tools/gpu_session.sh $TAG testall pmcstep profile > $OUT/session.log 2>&1
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -2 $OUT/bench.err
python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['hot_path_ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['secondary']['value'], d['secondary']['ms_per_step'], [t['value'] for t in d['tertiary']], {k: v['images_per_s'] for k, v in d['eval'].items()}, d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
bash tools/pmc_all.sh $OUT/pmc > $OUT/pmc.txt 2>&1; tail -45 $OUT/pmc.txt
timeout 300 python tools/op_benchmark.py > $OUT/op_benchmark.jsonl 2> $OUT/op_benchmark.err; cut -c1-200 $OUT/op_benchmark.jsonl
timeout 120 python tools/dense_bench.py --dense-only > $OUT/dense_bench.txt 2>&1
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2
tail -6 $OUT/session.log
