cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_2_glue.py tests/test_gpu_2_stress.py -m gpu -x -q -k "wgrad" 2>&1 | tail -3
python tools/ab_bench.py LIB HEAD tools/ab/libvilattn_wg_oldswz.so --no-eval --no-tertiary --rounds 2
