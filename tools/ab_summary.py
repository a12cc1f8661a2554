import sys,json,re
cur=None
for line in open(sys.argv[1]):
    m=re.match(r"== (\S+) lib=(\S*)",line)
    if m: cur=(m.group(1),m.group(2).split("libvilattn_")[-1].replace(".so","") or "HEAD"); continue
    if not line.startswith("{"): continue
    d=json.loads(line)["kernels"]
    print("%-18s %-10s fwd %.1f dq %.1f dkdv %.1f delta %.1f table %.1f"%(cur[0],cur[1],d["k_mfma_fwd"]["avg_ms"]*1e3,d["k_mfma_bwd_dq"]["avg_ms"]*1e3,d["k_mfma_bwd_dkdv"]["avg_ms"]*1e3,d["k_delta"]["avg_ms"]*1e3,d["k_mfma_table"]["avg_ms"]*1e3))
