#!/bin/bash
# durations, in launch order, of the exact chunk-workgroup kernel's redo_only launch (normally empty) next to its fast twin:
# rocprofv3 --kernel-trace over tools/attn_ab.py <shape>.   usage (GPU box): tools/safe_launch_trace.sh <out.txt> [shape]
OUT=$1; SHAPE=${2:-small_s1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/_safe_trace; rm -rf $D
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python tools/attn_ab.py $SHAPE --reps 12 > /dev/null 2>&1
python - "$(find $D -name '*kernel_trace.csv' | head -1)" > "$OUT" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
for r in rows:
    n = r["Kernel_Name"]
    if "k_cw_fwd" in n:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        gap = (int(r["Start_Timestamp"]) - prev_end) / 1e3 if prev_end else 0.0
        print("%-12s %8.1f us   gap to previous kernel %7.1f us   grid %s wg %s lds %s" % (
            "fast" if "Lb0E" in n else "exact/redo", d, gap, r.get("Grid_Size"), r.get("Workgroup_Size"), r.get("LDS_Block_Size")))
    prev_end = int(r["End_Timestamp"])
PY
rm -rf $D
cat "$OUT"
