"""Cuts the rocprofv3 --pmc dispatch streams of tools/pmc_all.py into cases (marker = the process's only `sign` kernel,
grid size growing with the case index), averages every counter per (case, kernel) and derives the pipe utilisation:

  occupancy   = SQ_WAVE_CYCLES * 4 / (1024 SIMDs * cycles)            [SQ_WAVE_CYCLES counts quad-cycles]
  MFMA busy   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * cycles)             [counts cycles: 16 per 16x16x32, 32 per 32x32x16]
  VALU issue  = SQ_ACTIVE_INST_VALU * 4 / (1024 * cycles)              [quad-cycles of VALU-class issue incl. MFMA, exp = 2 quads]
  LDS busy    = SQ_LDS_IDX_ACTIVE / (256 CUs * cycles), conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  wave time   = active (SQ_ACTIVE_INST_ANY) / issue-stalled (SQ_WAIT_INST_ANY) / parked on s_waitcnt or barrier (SQ_WAIT_ANY), of SQ_WAVE_CYCLES
  cycles      = GRBM_GUI_ACTIVE / 8 XCDs (collected in every pass)

usage: tools/pmc_all_summary.py <outdir> [<profiles prefix, e.g. profiles/r04_pmc_>]"""
import collections
import csv
import glob
import json
import os
import re
import sys


def kname(name):
    m = re.match(r"_Z(\d+)", name)
    if m:
        rest = name[m.end() + int(m.group(1)):]
        return name[m.end():m.end() + int(m.group(1))] + "<" + rest.split("Ev")[0].lstrip("I") + ">"
    return name.replace("void ", "").split("(")[0].strip()


def load_pass(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    disp = collections.OrderedDict()
    for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
        e = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"]), "c": {},
                                                    "vgpr": int(r.get("VGPR_Count", 0) or 0) + int(r.get("Accum_VGPR_Count", 0) or 0),
                                                    "lds": int(r.get("LDS_Block_Size", 0) or 0), "wg": int(r.get("Workgroup_Size", 0) or 0)})
        e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return list(disp.values())


def main():
    out = sys.argv[1]
    prefix = sys.argv[2] if len(sys.argv) > 2 else None
    cases = None
    for lg in sorted(glob.glob(os.path.join(out, "p*.log"))):
        for line in open(lg):
            if line.startswith('{"cases"'):
                cases = json.loads(line)
    assert cases, "no case list in the pass logs"
    names = cases["cases"]
    agg = collections.OrderedDict()          # (case, kernel) -> {counter: [values]}
    meta = {}
    for pd in sorted(glob.glob(os.path.join(out, "p[0-9]"))):
        disp = load_pass(pd)
        marks = sorted({d["grid"] for d in disp if "sign_kernel" in d["name"]})
        cur = None
        for d in disp:
            if "sign_kernel" in d["name"]:
                cur = marks.index(d["grid"])
                continue
            k = kname(d["name"])
            if cur is None or cur >= len(names) or not k.startswith("k_"):
                continue
            a = agg.setdefault((names[cur], k), collections.defaultdict(list))
            for c, v in d["c"].items():
                a[c + ("@" + os.path.basename(pd) if c == "GRBM_GUI_ACTIVE" else "")].append(v)
            meta[(names[cur], k)] = dict(grid=d["grid"], workgroup=d["wg"], vgpr=d["vgpr"], lds_bytes=d["lds"])
    res = collections.OrderedDict()
    lines = []
    hdr = "%-20s %-34s %8s %5s %6s %6s %6s %6s %6s | %6s %6s %6s | %5s %6s" % (
        "case", "kernel", "us", "occ", "MFMA%", "VALUi%", "LDS%", "confl%", "VMEM%", "activ%", "stall%", "wait%", "vgpr", "lds KB")
    lines.append(hdr)
    for (case, k), a in agg.items():
        m = {c: sum(v) / len(v) for c, v in a.items()}
        g = lambda c, p=None: m.get(c + ("@" + p if p else ""), 0.0)
        e = dict(meta[(case, k)])
        e["counters"] = {c: round(v, 1) for c, v in m.items()}
        e["dispatches"] = len(next(iter(a.values())))
        d = {}
        cy1, cy2, cy3 = (g("GRBM_GUI_ACTIVE", p) / 8 for p in ("p1", "p2", "p3"))
        if cy1:
            wc = max(g("SQ_WAVE_CYCLES"), 1.0)
            d.update(cycles=round(cy1), occupancy_waves_per_simd=round(wc * 4 / (1024 * cy1), 2),
                     mfma_busy=round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * cy1), 4),
                     valu_issue_busy=round(g("SQ_ACTIVE_INST_VALU") * 4 / (1024 * cy1), 4),
                     wave_active=round(g("SQ_ACTIVE_INST_ANY") / wc, 4), wave_issue_stalled=round(g("SQ_WAIT_INST_ANY") / wc, 4),
                     wave_waiting=round(g("SQ_WAIT_ANY") / wc, 4))
        if cy2:
            d.update(lds_busy=round(g("SQ_LDS_IDX_ACTIVE") / (256 * cy2), 4),
                     lds_conflict_share=round(g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1.0), 4),
                     insts_per_wave=dict(valu=g("SQ_INSTS_VALU"), mfma=g("SQ_INSTS_MFMA"), trans=g("SQ_INSTS_VALU_TRANS_F32"),
                                         lds=g("SQ_INSTS_LDS"), salu=g("SQ_INSTS_SALU"), vmem_rd=g("SQ_INSTS_VMEM_RD")))
        if cy3:
            d.update(vmem_issue_busy=round(g("SQ_ACTIVE_INST_VMEM") * 4 / (1024 * cy3), 4),
                     lds_issue_busy=round(g("SQ_ACTIVE_INST_LDS") * 4 / (1024 * cy3), 4))
        if cy1 and g("SQ_WAVES"):
            nw = g("SQ_WAVES")
            if "insts_per_wave" in d:
                d["insts_per_wave"] = {kk: round(v / nw, 1) for kk, v in d["insts_per_wave"].items()}
        e["derived"] = d
        res.setdefault(case, collections.OrderedDict())[k] = e
        us = cy1 / 2.1e3 if cy1 else 0.0         # nominal 2.1 GHz under load: for orientation only
        lines.append("%-20s %-34s %8.1f %5.2f %6.1f %6.1f %6.1f %6.1f %6.1f | %6.1f %6.1f %6.1f | %5d %6.1f" % (
            case, k[:34], us, d.get("occupancy_waves_per_simd", 0), 100 * d.get("mfma_busy", 0), 100 * d.get("valu_issue_busy", 0),
            100 * d.get("lds_busy", 0), 100 * d.get("lds_conflict_share", 0), 100 * d.get("vmem_issue_busy", 0),
            100 * d.get("wave_active", 0), 100 * d.get("wave_issue_stalled", 0), 100 * d.get("wave_waiting", 0), e["vgpr"], e["lds_bytes"] / 1024))
    json.dump({"cases": cases, "kernels": res}, open(os.path.join(out, "pmc_all.json"), "w"), indent=1)
    txt = "\n".join(lines) + "\n"
    open(os.path.join(out, "pipe_utilisation.txt"), "w").write(__doc__.split("usage:")[0] + "\n" + txt)
    print(txt)
    if prefix:
        for case, ks in res.items():
            for k, e in ks.items():
                kk = re.sub(r"[^A-Za-z0-9_]+", "_", k).strip("_")
                json.dump({"case": case, "kernel": k, **e}, open(f"{prefix}{kk}_{case}.json", "w"), indent=1)


if __name__ == "__main__":
    main()
