"""Steady-state per-step summary of a rocprofv3 kernel trace of bench.py (last 3 steps)."""
import csv, collections, sys
path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
import re
def pretty(n):
    """rocprofv3 leaves names whose template arguments are __bf16 / _Float16 mangled: _Z10k_mfma_fwdIDF16bLi2EEv... -> k_mfma_fwd<bf16,2>"""
    m = re.match(r"_Z(\d+)", n)
    if not m:
        return n
    name = n[m.end():m.end() + int(m.group(1))]
    args = n[m.end() + int(m.group(1)):].split("Ev")[0].lstrip("I")
    args = args.replace("DF16b", "bf16,").replace("DF16_", "f16,")
    args = re.sub(r"Li(\d+)E", r"\1,", args)
    args = re.sub(r"Lb(\d)E", r"\1,", args).rstrip("E,").rstrip(",")
    return f"void {name}<{args}>"
for r in rows:
    r["Kernel_Name"] = pretty(r["Kernel_Name"])
marker = sys.argv[2] if len(sys.argv) > 2 else "void k_mfma_fwd<bf16,2>"
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(marker)]
def period(idx):
    """marker launches per step, from the trace itself: the smallest p for which the launch counts between consecutive
    marker launches repeat with period p over the last 3 periods"""
    gaps = [b - a for a, b in zip(idx, idx[1:])]
    for p in range(1, len(gaps) // 4):
        if all(gaps[-1 - i] == gaps[-1 - i - p] for i in range(3 * p)):
            return p
    print("# WARNING: no launch period found for marker %s; assuming 1 launch per step" % marker)
    return 1
per = period(idx) if len(sys.argv) > 3 and sys.argv[3] == "auto" else int(sys.argv[3]) if len(sys.argv) > 3 else 1   # marker launches per step
nsteps = 3
s0, s1 = idx[-1 - nsteps * per], idx[-1]
t0, t1 = int(rows[s0]["Start_Timestamp"]), int(rows[s1]["Start_Timestamp"])
def classify(n):
    if n.startswith("Cijk"): return "gemm(hipblaslt)"
    if "k_mfma" in n or "k_cw_" in n or "k_gq_" in n or "k_delta" in n or "k_reduce" in n or "k_scalar" in n or "k_glo_" in n or "k_dense_" in n: return "vil hot path"
    if "k_wgrad" in n or "k_colsum" in n: return "vil weight gradient"
    if "k_dgrad_dgelu" in n or "k_fwd_gelu" in n or "k_skinny" in n: return "vil GEMM (fused epilogue / weights in registers)"
    if "k_optim" in n: return "vil optimizer"
    if "k_patchify" in n or "k_sc2d" in n: return "vil glue"
    if "k_ln_" in n: return "vil layernorm"
    if "layer_norm" in n or "cuCompute" in n: return "torch layernorm"
    if "attn_fwd" in n or "bwd_kernel" in n or "attn_bwd" in n: return "sdpa (dense attn)"
    if "bfloat16_copy" in n or "bfloat16tofloat32" in n or "copy_kernel" in n.lower(): return "casts/copies"
    if "reduce_kernel" in n: return "reduce"
    if "multi_tensor" in n or "adam" in n.lower(): return "optimizer"
    if "conv" in n.lower() or "Im2d2Col" in n or "igemm" in n or "_ZN2ck" in n or "ck::" in n: return "conv"
    if "elementwise" in n: return "elementwise"
    if "Cat" in n or "index" in n.lower() or "gather" in n.lower(): return "cat/index"
    return "other"
agg = collections.defaultdict(lambda: [0, 0.0]); cat = collections.defaultdict(lambda: [0, 0.0])
for r in rows[s0:s1]:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg[r["Kernel_Name"]]; a[0] += 1; a[1] += d
    c = cat[classify(r["Kernel_Name"])]; c[0] += 1; c[1] += d
tot = sum(v[1] for v in agg.values())
print("# last %d steady-state steps: wall %.2f ms/step, kernel-busy %.2f ms/step, %d launches/step" % (
    nsteps, (t1 - t0) / 1e6 / nsteps, tot / 1e6 / nsteps, (s1 - s0) / nsteps))
for k, v in sorted(cat.items(), key=lambda kv: -kv[1][1]):
    print("%-20s %5d calls/step %8.3f ms/step %5.1f%%" % (k, v[0] / nsteps, v[1] / 1e6 / nsteps, 100 * v[1] / tot))
print("%-100s %6s %10s %9s %6s" % ("kernel", "calls", "us/call", "ms/step", "%"))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[4]) if len(sys.argv) > 4 else 40]:
    print("%-100s %6d %10.1f %9.3f %6.1f" % (k[:100], v[0] / nsteps, v[1] / v[0] / 1e3, v[1] / 1e6 / nsteps, 100 * v[1] / tot))
