#!/bin/bash
# PMC counter passes (rocprofv3 --pmc only, one counter group per run -- never combined with tracing) over
# tools/pmc_all.py: every hand-written hot kernel at its bench shapes in ONE process per pass.
# usage (GPU box): tools/pmc_all.sh <outdir> [extra pmc_all.py args]   ->  <outdir>/pmc_all.json + pipe_utilisation.txt
set -u
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
P3="SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
i=1
for P in "$P1" "$P2" "$P3"; do
  [ -n "${PMC_PASSES:-}" ] && [[ ! " $PMC_PASSES " =~ " $i " ]] && { i=$((i+1)); continue; }
  timeout 400 rocprofv3 --pmc $P --output-format csv -d "$OUT/p$i" -o pmc -- python tools/pmc_all.py "$@" > "$OUT/p$i.log" 2>&1
  tail -1 "$OUT/p$i.log" | cut -c1-400
  i=$((i+1))
done
python tools/pmc_all_summary.py "$OUT"
