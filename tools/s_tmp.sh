cd "$GRAFT_REPO_ROOT"
python tools/ab_bench.py LIB HEAD tools/ab/libvilattn_gf4.so --no-eval --no-tertiary --rounds 2
