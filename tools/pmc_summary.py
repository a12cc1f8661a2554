"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch)."""
import csv, glob, sys, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        import re
        m = re.match(r"_Z(\d+)", name)      # mangled (template arguments __bf16 / _Float16): keep name + raw argument list
        if m:
            rest = name[m.end() + int(m.group(1)):]
            name = name[m.end():m.end() + int(m.group(1))] + "<" + rest.split("Ev")[0].lstrip("I") + ">"
        if not ("k_mfma" in name or "k_scalar" in name or "k_delta" in name or "k_cw" in name):
            continue
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in agg.items():
    res[k] = {c: sum(v) / len(v) for c, v in d.items()}
    res[k]["_dispatches"] = len(next(iter(d.values())))
for k, d in sorted(res.items()):
    print(k)
    for c in sorted(d):
        print("   %-28s %16.1f" % (c, d[c]))
    g = d.get
    if g("SQ_WAVE_CYCLES"):
        wc = g("SQ_WAVE_CYCLES")
        print("   -- active_any/wave_cycles %.3f  valu %.3f  wait_any %.3f  wait_inst %.3f  mfma_busy/busy_cycles*4 %.3f" % (
            g("SQ_ACTIVE_INST_ANY", 0) / wc, g("SQ_ACTIVE_INST_VALU", 0) / wc, g("SQ_WAIT_ANY", 0) / wc,
            g("SQ_WAIT_INST_ANY", 0) / wc, g("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(g("SQ_BUSY_CYCLES", 1), 1)))
    if g("FETCH_SIZE") is not None:
        print("   -- HBM read MB (FETCH_SIZE KB x2 gfx950 correction) %.1f  write MB %.1f" % (
            g("FETCH_SIZE") * 2 / 1024, g("WRITE_SIZE", 0) / 1024))
json.dump(res, open(out + "/summary.json", "w"), indent=1)
