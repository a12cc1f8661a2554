set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_final; mkdir -p $OUT
bash tools/gpu_session.sh r03_final test pmcstep profile > $OUT/session.log 2>&1
cp $OUT/pmcstep/pmc_traffic.json profiles/r03_pmc_traffic.json 2>/dev/null
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -2 $OUT/bench.err
python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['hot_path_ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['secondary']['value'], d['secondary']['ms_per_step'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
timeout 300 python tools/op_benchmark.py > $OUT/op_benchmark.jsonl 2> $OUT/op_benchmark.err; cat $OUT/op_benchmark.jsonl | cut -c1-200
timeout 120 python tools/skinny_probe.py > $OUT/skinny_probe.txt 2>&1
timeout 120 python tools/wgrad2_probe.py --sweep > $OUT/wgrad2_sweep.txt 2>&1
timeout 120 python tools/mlp_bench.py > $OUT/mlp_bench.txt 2>&1
timeout 120 python tools/dense_bench.py > $OUT/dense_bench.txt 2>&1
tail -3 $OUT/skinny_probe.txt $OUT/mlp_bench.txt
tail -6 $OUT/session.log
