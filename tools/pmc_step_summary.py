"""Builds the PMC traffic file bench.py attaches to its roofline objects (profiles/r06_pmc_traffic.json) from the
rocprofv3 --pmc passes of tools/pmc_step.sh.  Dispatches are matched to problem shapes by launch order: every eager
step launches the library kernels in the same order, which bench.py dumped (VIL_BENCH_DUMP_TAGS) together with each
launch's algorithmic bytes.  FETCH_SIZE (KB) is doubled (gfx950: the counter reports half of a wide coalesced stream --
MI355X guide, re-calibrated with tools/fetch_calib.py), WRITE_SIZE (KB) is used as is.  The file is stamped with the
fingerprint of the kernel sources; bench.py ignores it for any other build."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vision_longformer_amd import _lib  # noqa: E402

SINK = {"k_cw_prep": "k_mfma_table", "k_cw_fwd": "k_mfma_fwd", "k_mfma_prep": "k_mfma_table", "k_mfma_prep_bwd": "k_mfma_table", "k_mfma_fwd": "k_mfma_fwd", "k_mfma_delta": "k_delta", "k_mfma_bwd_dq": "k_mfma_bwd_dq",
        "k_mfma_bwd_dkdv": "k_mfma_bwd_dkdv", "k_mfma_post_bwd": "k_reduce_glo",
        "k_glo_fwd": "k_glo_fwd", "k_gq_merge": "k_glo_fwd", "k_glo_bwd": "k_glo_bwd", "k_dense_fwd": "k_dense_fwd", "k_dense_bwd_dq": "k_dense_bwd_dq",
        "k_dense_bwd_dkdv": "k_dense_bwd_dkdv", "k_dense_reduce": "k_dense_reduce", "k_wgrad": "k_wgrad", "k_wgrad_reduce": "k_wgrad_reduce",
        "k_wgrad2": "k_wgrad", "k_wgrad2_reduce": "k_wgrad_reduce"}


def base(name):
    import re
    m = re.match(r"_Z(\d+)", name)          # rocprofv3 leaves names with __bf16 / _Float16 template arguments mangled
    if m:
        n = name[m.end():m.end() + int(m.group(1))]
    else:
        n = name.replace("void ", "").split("(")[0].split("<")[0].strip()
    if n == "k_cw_fwd" and "Lb0E" not in name:
        # the exact kernel's launch behind the fast one (redo_only: returns at once unless a column was flagged).  bench.py runs
        # bf16, whose fast kernels keep their mangled names (...Lb0EEv...) in rocprofv3's output; the exact twins appear as
        # ...Lb1EEv... or, mis-demangled, as "void k_cw_fwd<bool _Accum, int, EL, ...>"
        return None
    return SINK.get(n)


def load(pattern):
    """{sink name: [ {counter: value} per dispatch, in dispatch order ]}"""
    rows = []
    for f in glob.glob(pattern, recursive=True):
        rows += list(csv.DictReader(open(f)))
    per = collections.OrderedDict()
    for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
        k = base(r["Kernel_Name"])
        if k is None:
            continue
        d = per.setdefault(k, collections.OrderedDict()).setdefault(int(r["Dispatch_Id"]), {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return {k: list(v.values()) for k, v in per.items()}


def main():
    out_dir, out_file = sys.argv[1], sys.argv[2]
    tags = json.load(open(os.path.join(out_dir, "tags.json")))
    res = {"source_fingerprint": _lib.source_fingerprint(),
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (separate passes, tools/pmc_step.sh) over "
                     "`bench.py --graph off --steps 2 --warmup 1` (the real eager training step); FETCH_SIZE x2 (gfx950 "
                     "correction), WRITE_SIZE as is; mean over the last step's dispatches of each (kernel, shape)",
           "configs": {}}
    for cfg, t in tags.items():
        seq = collections.OrderedDict()
        for name, label, alg in t["launches"]:
            seq.setdefault(name, []).append((label, alg))
        fetch = load(os.path.join(out_dir, f"fetch_{cfg}", "**", "*counter_collection.csv"))
        write = load(os.path.join(out_dir, f"write_{cfg}", "**", "*counter_collection.csv"))
        kern = {}
        for name, labels in seq.items():
            acc = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0, 0.0])
            for src, idx in ((fetch.get(name, []), 1), (write.get(name, []), 2)):
                # the LAST step's dispatches: the first eager step also holds the launches of the one-off plan selection
                # (vil_linear_wgrad_tune), which would shift a from-the-start alignment
                src = src[-len(labels):] if len(src) >= len(labels) else []
                for j, d in enumerate(src):
                    lab, alg = labels[j % len(labels)]
                    a = acc[lab]
                    if idx == 1:
                        a[0] += 1; a[1] += d.get("FETCH_SIZE", 0.0) * 2 * 1024; a[5] = alg
                    else:
                        a[2] += d.get("WRITE_SIZE", 0.0) * 1024; a[3] += d.get("TCC_HIT_sum", 0.0); a[4] += d.get("TCC_MISS_sum", 0.0)
            kern[name] = {}
            for lab, (n, fb, wb, hit, miss, alg) in acc.items():
                if n == 0:
                    continue
                hb = (fb + wb) / n
                kern[name][lab] = {"dispatches": n, "hbm_bytes_per_launch": hb, "read_bytes": fb / n, "write_bytes": wb / n,
                                   "algorithmic_bytes": alg, "traffic_over_algorithmic": round(hb / alg, 3) if alg else None,
                                   "l2_hit_rate": round(hit / (hit + miss), 4) if hit + miss else None}
        res["configs"][cfg] = {"per_gpu_batch": t["per_gpu_batch"], "kernels": kern}
    json.dump(res, open(out_file, "w"), indent=1)
    for cfg, e in res["configs"].items():
        for k in ("k_mfma_fwd", "k_mfma_bwd_dq", "k_mfma_bwd_dkdv"):
            for lab, v in e["kernels"].get(k, {}).items():
                print(f"{cfg:22s} {k:18s} {lab:24s} hbm {v['hbm_bytes_per_launch'] / 1e6:8.1f} MB  alg {v['algorithmic_bytes'] / 1e6:8.1f} MB  "
                      f"x{v['traffic_over_algorithmic']}  L2 hit {v['l2_hit_rate']}")


if __name__ == "__main__":
    main()
