#!/usr/bin/env python3
"""gpurun_out/<tag>_* (scripts/collect_r04.sh, parts a and b) -> profiles/r04_*<suffix>: the bench line, rocprofv3 kernel
statistics, the PMC traffic of the headline kernels and of the dense kernels (FETCH_SIZE doubled as MI355X_MICROARCH.md
prescribes for gfx950), MFMA counters, the full-size lines of the other configurations, phase stamps.
    python3 scripts/r04_to_profiles.py <tag> [suffix]"""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1]
suf = sys.argv[2] if len(sys.argv) > 2 else ""
G, P = "gpurun_out", "profiles"


def first(pattern):
    f = sorted(glob.glob(pattern, recursive=True))
    return f[0] if f else None


def copy(src_pattern, dst):
    f = first(src_pattern)
    if f and os.path.getsize(f) > 0:
        shutil.copy(f, os.path.join(P, dst))
        print("  ", dst)
        return True
    return False


def counters(d, names):
    """sum of each counter per kernel over the run, and the number of dispatches"""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for f in glob.glob(os.path.join(G, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] in names:
                k = row["Kernel_Name"]
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                disp[k].add(row["Dispatch_Id"])
    return acc, {k: len(v) for k, v in disp.items()}


# ---- part a -------------------------------------------------------------------------------------------------------
copy(f"{G}/{tag}_bench.json", f"r04_bench{suf}.json")
copy(f"{G}/{tag}_stats/**/*kernel_stats.csv", f"r04_tile_kernel_stats{suf}.csv")
for c in ("c3", "c4"):
    copy(f"{G}/{tag}_{c}_stats/**/*kernel_stats.csv", f"r04_{c}_kernel_stats{suf}.csv")
if os.path.isdir(f"{G}/{tag}_fetch"):
    fe, nf = counters(f"{tag}_fetch", {"FETCH_SIZE"})
    wr, nw = counters(f"{tag}_write", {"WRITE_SIZE"})
    B, N, d, p = 1024, 4000, 2, 3
    tile = d * p * (p + 1) * 8
    alg = {"fwd_tile3_kernel": B * (N + 1) * tile, "bwd_mv_tile3_kernel": 2 * B * (N - 1) * tile + B * tile}
    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (python3 bench.py --steps 3 --warmup 1 "
                   "--no-cpu-baseline, scripts/collect_profiles.sh); counter unit KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md "
                   "(gfx950 tallies 128-B read requests at 64 B); per launch, C2 workload", "kernels": {}}
    for key in alg:
        kf = [k for k in fe if key in k]
        kw = [k for k in wr if key in k]
        if kf and kw:
            f_ = fe[kf[0]]["FETCH_SIZE"] / nf[kf[0]]
            w_ = wr[kw[0]]["WRITE_SIZE"] / nw[kw[0]]
            out["kernels"][key] = {"FETCH_SIZE_KiB_raw": f_, "WRITE_SIZE_KiB": w_, "hbm_bytes_corrected": (2 * f_ + w_) * 1024,
                                   "algorithmic_bytes": alg[key]}
    json.dump(out, open(f"{P}/r04{suf}_pmc_traffic.json", "w"), indent=1)
    print("  ", f"r04{suf}_pmc_traffic.json", json.dumps(out["kernels"]))
    copy(f"{G}/{tag}_fetch/**/*counter_collection.csv", f"r04{suf}_pmc_fetch_size_counter_collection.csv")
    copy(f"{G}/{tag}_write/**/*counter_collection.csv", f"r04{suf}_pmc_write_size_counter_collection.csv")
for c, (B, N, d, p, per) in {"c3": (512, 20000, 3, 4, 3 * 3 * 4 * 5 * 8), "c4": (1024, 800, 2, 3, (2 * 2 * 3 * 4 + 2 * 3) * 8)}.items():
    if os.path.isdir(f"{G}/{tag}_{c}_fetch"):
        fe, nf = counters(f"{tag}_{c}_fetch", {"FETCH_SIZE"})
        wr, nw = counters(f"{tag}_{c}_write", {"WRITE_SIZE"})
        o = {"note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of python3 scripts/bench_configs.py {c}; per launch; "
                     "FETCH_SIZE doubled (gfx950); algorithmic bytes of the whole solve = B N x the per-trajectory-step figure of DESIGN.md section 4",
             "algorithmic_bytes_solve": B * N * per, "kernels": {}}
        for k in fe:
            if "tile" in k and k in wr:
                f_, w_ = fe[k]["FETCH_SIZE"] / nf[k], wr[k]["WRITE_SIZE"] / nw[k]
                o["kernels"][k[:70]] = {"FETCH_SIZE_KiB_raw": f_, "WRITE_SIZE_KiB": w_, "hbm_bytes_corrected": (2 * f_ + w_) * 1024}
        json.dump(o, open(f"{P}/r04_{c}_pmc_traffic{suf}.json", "w"), indent=1)
        print("  ", f"r04_{c}_pmc_traffic{suf}.json")
lines = []
for f in (f"{tag}_configs_c3_c4.jsonl", f"{tag}_c3_quad.jsonl", f"{tag}_c4_unfused.jsonl", f"{tag}_c5_standard_full.json", f"{tag}_c5_sqrt_full.json",
          f"{tag}_c5_N50.json", f"{tag}_c5_regs_lu_N50.json"):
    fp = os.path.join(G, f)
    if os.path.exists(fp):
        for ln in open(fp):
            if ln.startswith("{"):
                o = json.loads(ln)
                o["source"] = f
                lines.append(json.dumps(o))
if lines:
    open(f"{P}/r04_other_configs_c3_c4_c5_full_size{suf}.jsonl", "w").write("\n".join(lines) + "\n")
    print("  ", f"r04_other_configs_c3_c4_c5_full_size{suf}.jsonl", len(lines), "lines")
copy(f"{G}/{tag}_nderiv_times.jsonl", f"r04_nderiv_times{suf}.jsonl")
copy(f"{G}/{tag}_block_vs_dense.jsonl", f"r04_block_vs_dense_32var_ring{suf}.jsonl")
# C4 forward step: cycles per step of the chkrebtii forward kernel with parts removed
rows = []
for name, f in (("shipped kernel", f"{tag}_configs_c3_c4.jsonl"), ("no generator (z constant)", f"{tag}_c4_RK_T3_ABLATE_1.jsonl"),
                ("no generator, no square root", f"{tag}_c4_RK_T3_ABLATE_2.jsonl"),
                ("sampler without ITS generator", f"{tag}_c4_RK_T3_ABLATE_SIMZ.jsonl")):
    fp = os.path.join(G, f)
    if os.path.exists(fp):
        for ln in open(fp):
            if ln.startswith("{") and "C4" in ln:
                o = json.loads(ln)
                ms = o["kernels_ms"].get("fwd_tile3_kernel")
                if ms:
                    rows.append((name, ms, o["kernels_ms"].get("bwd_sim_tile3_kernel"), o["ms"]))
if rows:
    with open(f"{P}/r04_c4_fwd_step_cycles{suf}.txt", "w") as fh:
        fh.write("C4 (FitzHugh-Nagumo, n_deriv 3, N = 800, 1024 draws, solve_sim + interrogate_chkrebtii + log-posterior): the forward step\n"
                 "of fwd_tile3_kernel's chkrebtii branch (solve_tile3_kernels.hpp), HIP events of scripts/bench_configs.py c4; experiment\n"
                 "builds of scripts/c4_ablation.sh (results wrong by construction).  cycles = ms / 800 steps x 2.4 GHz nominal.\n\n")
        fh.write("%-34s %10s %12s %12s %10s\n" % ("build", "fwd ms", "cycles/step", "bwd_sim ms", "wall ms"))
        for name, ms, b, w in rows:
            fh.write("%-34s %10.4f %12.0f %12.4f %10.4f\n" % (name, ms, ms * 1e-3 / 800 * 2.4e9, b or 0, w))
    print("  ", f"r04_c4_fwd_step_cycles{suf}.txt")

# ---- part b -------------------------------------------------------------------------------------------------------
copy(f"{G}/{tag}_c5_stats/**/*kernel_stats.csv", f"r04_c5_standard_N200_kernel_stats{suf}.csv")
copy(f"{G}/{tag}_c5sq_stats/**/*kernel_stats.csv", f"r04_c5_square_root_N200_kernel_stats{suf}.csv")
forms = {}
for key, d in (("standard, panel loop over memory (default)", "c5"), ("standard, register-resident elimination (RK_DENSE_LU=regs)", "c5regs"), ("square_root", "c5sq")):
    if os.path.isdir(f"{G}/{tag}_{d}_FETCH_SIZE"):
        fe, _ = counters(f"{tag}_{d}_FETCH_SIZE", {"FETCH_SIZE"})
        wr, _ = counters(f"{tag}_{d}_WRITE_SIZE", {"WRITE_SIZE"})
        forms[key] = {}
        for k in fe:
            if "dense" in k and ("fwd" in k or "bwd" in k) and k in wr:
                f_, w_ = fe[k]["FETCH_SIZE"], wr[k]["WRITE_SIZE"]
                forms[key][k[:60]] = {"FETCH_SIZE_KiB_raw": f_, "WRITE_SIZE_KiB": w_, "hbm_bytes_corrected": (2 * f_ + w_) * 1024,
                                      "MB_per_trajectory_step": (2 * f_ + w_) * 1024 / (256 * 50) / 1e6,
                                      "MB_per_trajectory_step_fetch_uncorrected": (f_ + w_) * 1024 / (256 * 50) / 1e6}
if forms:
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 scripts/bench_configs.py c5 --c5-steps 50 "
                       "[--c5-kalman square-root], B = 256, p = 160, m = 32; one launch of each kernel per run (per trajectory-step = bytes / (256 x 50)); counter "
                       "unit KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md.  Algorithmic bytes of a trajectory-step: 3 x 206 KB = 0.62 MB.",
               "forms": forms}, open(f"{P}/r04_c5_pmc_traffic_dense{suf}.json", "w"), indent=1)
    print("  ", f"r04_c5_pmc_traffic_dense{suf}.json")
mf = {}
for key, d in (("standard", "c5_mfma"), ("square_root", "c5sq_mfma")):
    if os.path.isdir(f"{G}/{tag}_{d}"):
        acc, nd = counters(f"{tag}_{d}", {"SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_WAVE_CYCLES"})
        mf[key] = {k[:60]: dict(v, launches=nd[k]) for k, v in acc.items() if "dense" in k and ("fwd" in k or "bwd" in k)}
times = {}
for form, f in (("standard", f"{tag}_c5_N50.json"),):
    fp = os.path.join(G, f)
    if os.path.exists(fp):
        for ln in open(fp):
            if ln.startswith("{"):
                times[form] = json.loads(ln)["kernels_ms"]
for form, ks in mf.items():
    for k, v in ks.items():
        v["mfma_flop"] = v["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512
        for name, ms in times.get(form, {}).items():
            if name in k:
                v["kernel_ms_N50_unprofiled"] = ms
                v["mfma_busy_frac_of_simd_cycles"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (ms * 1e-3 * 2.4e9 * 1024)
if mf:
    json.dump({"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES -- python3 "
                       "scripts/bench_configs.py c5 --c5-steps 50; sums over launches (one timed solve per run).  MOPS_F64 counts add or mul operations / 512; MFMA_BUSY is summed over the SIMDs; mfma_busy_frac_of_simd_cycles = MFMA_BUSY / (kernel time x 2.4 GHz x 1024 SIMDs), kernel time = HIP events of the unprofiled run of the same command", "forms": mf},
              open(f"{P}/r04_c5_mfma_counters{suf}.json", "w"), indent=1)
    print("  ", f"r04_c5_mfma_counters{suf}.json")
txt = []
for name, f in (("covariance form", f"{tag}_c5_stamps.txt"), ("square-root form", f"{tag}_c5sq_stamps.txt")):
    fp = os.path.join(G, f)
    if os.path.exists(fp):
        for ln in open(fp):
            if "phase cycles" in ln:
                txt.append(name + ": " + ln.strip())
if txt:
    open(f"{P}/r04_c5_phase_cycles{suf}.txt", "w").write(
        "C5 dense kernels, s_memtime stamps per phase (make -C rodeo_amd/csrc stamps; RK_DENSE_STAMPS=1 RK_LIB_PATH=rodeo_amd/librodeo_kalman_stamps.so\n"
        "python3 scripts/bench_configs.py c5 --c5-steps 50 [--c5-kalman square-root]); cycles per step of workgroup 0, B = 256.  Every stamp costs a\n"
        "workgroup barrier (~2 k cycles): the sum exceeds the unstamped step.  LU slots (panel loop over memory): 'LU panel' = the first panel; 'LU swaps /\n"
        "gather' = wave 0 of every later panel up to its factorisation, 'LU trsm' = wave 0 including it (the chain), 'LU: wave 1 strips' = wave 1's column\n"
        "strip of every panel, 'LU gemm' = the whole panel loop.\n\n" + "\n".join(txt) + "\n")
    print("  ", f"r04_c5_phase_cycles{suf}.txt")
