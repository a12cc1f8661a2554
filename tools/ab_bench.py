"""A/B of a module-level switch inside ONE process on ONE box (box-to-box variance of this pool is +-4 %, larger than
most single-kernel effects): runs bench.py's timed loop for each setting, alternating, and prints ms per step.
    python tools/ab_bench.py vision_longformer_amd.linear._GELU_EPILOGUE True False [--model small|meddeep] [--rounds 3]
    python tools/ab_bench.py LIB HEAD tools/ab/libvilattn_<name>.so [--rounds 3]        (whole libraries: HEAD = the in-tree build)"""
import importlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(target, value, extra):
    if target == "LIB":
        pre = "" if value == "HEAD" else f"from vision_longformer_amd import _lib; _lib.use_library_for_ab({os.path.join(ROOT, value)!r}); "
        code = (f"import sys; sys.path.insert(0, {ROOT!r}); {pre}import bench; "
                f"sys.argv = ['bench.py', '--no-cpu-baseline'] + {extra!r}; bench.main()")
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            raise SystemExit(out.stderr[-2000:])
        d = json.loads(line[-1])
        return d["ms_per_step"], d.get("secondary", {}).get("ms_per_step")
    mod, attr = target.rsplit(".", 1)
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import importlib; m = importlib.import_module({mod!r}); "
            f"setattr(m, {attr!r}, {value}); import bench; sys.argv = ['bench.py', '--no-cpu-baseline'] + {extra!r}; bench.main()")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        raise SystemExit(out.stderr[-2000:])
    d = json.loads(line[-1])
    return d["ms_per_step"], d.get("secondary", {}).get("ms_per_step")


if __name__ == "__main__":
    target, a, b = sys.argv[1:4]
    rest = sys.argv[4:]
    rounds = 2
    if "--rounds" in rest:
        i = rest.index("--rounds"); rounds = int(rest[i + 1]); del rest[i:i + 2]
    for r in range(rounds):
        for v in (a, b):
            ms, ms2 = one(target, v, rest)
            print(f"round {r} {target} = {v}: {ms:.3f} ms/step" + (f", secondary {ms2:.3f} ms/step" if ms2 else ""), flush=True)
