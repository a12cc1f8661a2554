#!/bin/bash
# A/B of whole libraries on ONE box: HEAD and every tools/ab/libvilattn_*.so, alternating, ROUNDS times.
#   tools/attn_ab.sh <out.txt> "<shape,shape>" [rounds]
OUT=$1; SHAPES=$2; ROUNDS=${3:-2}
cd "$(dirname "$0")/.."
for R in $(seq 1 $ROUNDS); do
  for LIB in "" $(ls tools/ab/libvilattn_*.so 2>/dev/null); do
    VIL_ATTN_LIB=${LIB:+$PWD/$LIB} timeout 300 python tools/attn_ab.py $SHAPES --reps ${REPS:-10} >> "$OUT" 2>&1
  done
done
python - "$OUT" <<'PY'
import json, sys, collections
rows = collections.defaultdict(list)
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); rows[(d["shape"], d["lib"])].append(d)
for (sh, lib), ds in sorted(rows.items()):
    ks = sorted(ds[0]["us"])
    print(f"{sh:18s} {lib:34s} total " + "/".join(f"{d['total_us']:.0f}" for d in ds) + "  " +
          " ".join(f"{k[2:]}={min(d['us'].get(k, 0) for d in ds):.0f}" for k in ks) + f"  chk {ds[0]['checksums']}")
PY
