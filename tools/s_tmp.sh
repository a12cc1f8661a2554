cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_1_parity.py -m gpu -x -q 2>&1 | tail -2
for LIB in "" tools/ab/libvilattn_wg_oldswz.so; do
  echo "== lib=${LIB:-HEAD}"
  VIL_ATTN_LIB=${LIB:+$PWD/$LIB} timeout 300 python tools/kernel_bench.py small_s2,meddeep_s2,small_s1 --reps 20 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['shape'], {k.replace('k_mfma_',''):round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items() if 'prep' not in k and 'post' not in k})"
done
