"""Turns the two rocprofv3 --pmc passes of scripts/collect_profiles.sh into profiles/<tag>_pmc_traffic.json.
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md: 128-byte read requests are
tallied at 64 bytes).  Per launch = mean over the profiled launches of each kernel."""
import csv, json, sys, collections
tag = sys.argv[1]
def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter:
            name = row["Kernel_Name"]
            key = "fwd_tile3_kernel" if "fwd_tile3_kernel" in name else "bwd_mv_tile3_kernel" if "bwd_mv_tile3_kernel" in name else None
            if key:
                acc[key].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
fetch = per_kernel("gpurun_out/%s_fetch/fetch_counter_collection.csv" % tag, "FETCH_SIZE")
write = per_kernel("gpurun_out/%s_write/write_counter_collection.csv" % tag, "WRITE_SIZE")
B, N, d, p = 1024, 4000, 2, 3
tile = d * p * (p + 1) * 8
alg = {"fwd_tile3_kernel": B * (N + 1) * tile, "bwd_mv_tile3_kernel": 2 * B * (N - 1) * tile + B * tile}
out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (python3 bench.py --steps 3 --warmup 1, "
               "scripts/collect_profiles.sh); counter unit KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies "
               "128-B read requests at 64 B); per launch, C2 workload", "kernels": {}}
for k in alg:
    out["kernels"][k] = {"FETCH_SIZE_KiB_raw": fetch[k], "WRITE_SIZE_KiB": write[k],
                         "hbm_bytes_corrected": (2 * fetch[k] + write[k]) * 1024, "algorithmic_bytes": alg[k]}
json.dump(out, open("profiles/%s_pmc_traffic.json" % tag, "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
