set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_final2; mkdir -p $OUT
timeout 150 python -m pytest tests/test_gpu_2_glue.py tests/test_gpu_3_engine.py -q > $OUT/pytest_a.log 2>&1; tail -3 $OUT/pytest_a.log
timeout 60 python -m pytest tests/test_gpu_1_parity.py -q -k dense > $OUT/pytest_b.log 2>&1; tail -2 $OUT/pytest_b.log
bash tools/pmc_step.sh $OUT/pmcstep > $OUT/pmcstep.txt 2>&1; tail -3 $OUT/pmcstep.txt
cp $OUT/pmcstep/pmc_traffic.json profiles/r03_pmc_traffic.json 2>/dev/null
timeout 200 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['hot_path_ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['secondary']['value'], d['secondary']['ms_per_step'], d['cpu_baseline']['value'])"
