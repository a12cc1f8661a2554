#!/bin/bash
# per-shape k_wgrad / k_wgrad_reduce durations of tools/wgrad_probe.py from a rocprofv3 kernel trace
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_wg && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_wg -o t -- python $GRAFT_REPO_ROOT/tools/wgrad_probe.py > /tmp/wg.log 2>&1
python - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob("/tmp/prof_wg/**/*kernel_trace.csv",recursive=True)[0])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
seq=[(r["Kernel_Name"].split("(")[0], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3) for r in rows if "k_wgrad" in r["Kernel_Name"]]
names=["s3 qkv","s3 proj","s3 fc1","s3 fc2","s4 fc1","s4 fc2","s2 fc1","s2 fc2","s2 kv","s1 fc1","s1 fc2","s1 kv"]
i=0; j=0
while i < len(seq):
    chunk=seq[i:i+48]
    a=[d for n,d in chunk if n=="k_wgrad"]; b=[d for n,d in chunk if n=="k_wgrad_reduce"]
    print("%-8s k_wgrad %6.1f us  reduce %5.1f us" % (names[j] if j < len(names) else "?", sum(a[3:])/len(a[3:]), sum(b[3:])/len(b[3:])))
    i+=48; j+=1
PY
