#!/bin/bash
# PMC counter passes (rocprofv3 --pmc only, one group per run) over tools/kernel_bench.py
# usage: tools/pmc.sh <shape> <outdir> [extra kernel_bench args]
set -u
SHAPE=$1; OUT=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"
P4="FETCH_SIZE"
P5="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
i=1
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  timeout 300 rocprofv3 --pmc $P --output-format csv -d "$OUT/p$i" -o pmc -- python tools/kernel_bench.py $SHAPE --reps 2 "$@" > "$OUT/p$i.log" 2>&1
  i=$((i+1))
done
python tools/pmc_summary.py "$OUT"
