cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06c
bash tools/safe_launch_trace.sh gpurun_out/r06c/safe_trace.txt small_s1 | head -40
for R in 1 2; do for LIB in "" tools/ab/libvilattn_nopair.so; do echo "== round $R lib=${LIB:-HEAD(pair order)}"; VIL_ATTN_LIB=${LIB:+$PWD/$LIB} timeout 300 python tools/cw_check.py small_s1,meddeep_s1_f7,meddeep_s1_f8 --reps 20 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'], 'cw', d['us_cw']['k_mfma_fwd'], 'wave', d['us_wave']['k_mfma_fwd'], 'diff', d['diff_vs_wave']['out']['max'])
"; done; done 2>&1 | tee gpurun_out/r06c/pair_order_ab.txt
timeout 600 python -m pytest tests/test_gpu_1_cw.py -q 2>&1 | tail -3
