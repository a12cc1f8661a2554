"""Build profiles/r01_pmc_traffic.json from the per-shape PMC summaries (tools/pmc.sh -> summary.json):
HBM bytes per launch of every hot-path kernel, averaged over the launch mix of one ViL-Small step
(stage 1 x1, stage 2 x2, stage 3 (dense) x8, stage 4 (dense) x1).
FETCH_SIZE (KB) is doubled (gfx950: the counter reports half of a wide coalesced stream; re-calibrated with
tools/fetch_calib.py on a 256 MiB copy), WRITE_SIZE (KB) is used as is."""
import json, sys
root = sys.argv[1]
mix = {"small_s1": 1, "small_s2": 2, "small_s3_dense": 8, "small_s4_dense": 1}
out = {"config": "vil_small_224", "per_gpu_batch": 128, "launch_mix": mix,
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (separate passes, tools/pmc.sh) over "
                 "tools/kernel_bench.py at the four hot-path shapes of ViL-Small, B=128; FETCH_SIZE x2 (gfx950 correction, "
                 "re-calibrated on a 256 MiB copy), WRITE_SIZE as is; per-launch average weighted by the step's launch mix",
       "kernels": {}}
per = {}
for shape in mix:
    d = json.load(open(f"{root}/pmc_{shape}/summary.json"))
    for k, v in d.items():
        if "FETCH_SIZE" not in v:
            continue
        name = k.split("<")[0]
        per.setdefault(name, {})[shape] = (v["FETCH_SIZE"] * 2 + v.get("WRITE_SIZE", 0.0)) * 1024.0
for name, sh in per.items():
    if set(sh) != set(mix):
        continue
    tot = sum(mix.values())
    out["kernels"][name] = {**{s + "_bytes": round(b) for s, b in sh.items()},
                            "hbm_bytes_per_launch": sum(mix[s] * sh[s] for s in mix) / tot}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
