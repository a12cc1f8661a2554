cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -m gpu -x -q -k "launch_shapes or soak_dense" 2>&1 | tail -2
