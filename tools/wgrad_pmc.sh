#!/bin/bash
# PMC counters of k_wgrad on one shape: tools/wgrad_pmc.sh T CI CO outdir
set -u
T=$1; CI=$2; CO=$3; OUT=$4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
cat > /tmp/wg_one.py <<PY
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from vision_longformer_amd.linear import _wgrad
dev = torch.device("cuda:0")
x = torch.randn($T, $CI, device=dev, dtype=torch.bfloat16); dy = torch.randn($T, $CO, device=dev, dtype=torch.bfloat16)
for _ in range(3): _wgrad(dy, x, True)
torch.cuda.synchronize()
PY
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
P3="SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC"
P4="FETCH_SIZE"
P5="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
i=1
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  timeout 300 rocprofv3 --pmc $P --output-format csv -d "$OUT/p$i" -o pmc -- python /tmp/wg_one.py > "$OUT/p$i.log" 2>&1
  i=$((i+1))
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("k_wgrad("):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
d = {k: sum(v) / len(v) for k, v in agg.items()}
for k in sorted(d): print("%-28s %16.1f" % (k, d[k]))
wc = d.get("SQ_WAVE_CYCLES", 1)
print("active_any/wave %.3f valu %.3f lds %.3f wait_any %.3f wait_inst %.3f" % (d.get("SQ_ACTIVE_INST_ANY",0)/wc, d.get("SQ_ACTIVE_INST_VALU",0)/wc, d.get("SQ_ACTIVE_INST_LDS",0)/wc, d.get("SQ_WAIT_ANY",0)/wc, d.get("SQ_WAIT_INST_ANY",0)/wc))
print("mfma busy frac of (gui/8*1024 SIMD-cycles): %.3f" % (d.get("SQ_VALU_MFMA_BUSY_CYCLES",0) / (d.get("GRBM_GUI_ACTIVE",1)/8*1024)))
print("HBM read MB %.1f write MB %.1f ; L2 hit %.0f miss %.0f" % (d.get("FETCH_SIZE",0)*2/1024, d.get("WRITE_SIZE",0)/1024, d.get("TCC_HIT_sum",0), d.get("TCC_MISS_sum",0)))
print("bank conflict / lds idx active %.3f" % (d.get("SQ_LDS_BANK_CONFLICT",0)/max(d.get("SQ_LDS_IDX_ACTIVE",1),1)))
PY
