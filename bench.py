#!/usr/bin/env python3
"""
bench.py -- headline benchmark of BASELINE.json: trajectory-steps per second (ODE steps/s x batch) of the batched
probabilistic ODE solve, FitzHugh-Nagumo n_vars=2 q(n_deriv)=3, 4000 steps, `solve_mv` + `interrogate_kramer`,
1024 trajectories per GPU (BASELINE.json configs[1]; SURVEY.md section 8d "C2").

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: spawns its own N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W                  (the driver's launch: RANK / WORLD_SIZE from the env)

One "step" = one complete solve (forward filter kernel + backward smoother kernel) of this rank's batch with all
inputs and outputs resident in HBM.  The path shards by independent trajectories: rank r owns global trajectories
[r*B, (r+1)*B) ("weak" scaling, no data-path collective).  Timing: barrier + device sync on both sides of exactly K
steps, MAX over ranks; rank 0 prints ONE JSON line.

No torch in these processes: the ranks meet over rodeo_amd.hostgroup (standard-library TCP star; carries the 128-byte
ncclUniqueId and the max-over-ranks), the device work goes through librodeo_kalman.so (ctypes), and the barriers
around the timed region run over RCCL inside that library when the communicator comes up on every rank (else over the
host channel; `--require-rccl` turns that fallback into a non-zero exit).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
# HBM traffic of the two headline kernels from this command's rocprofv3 --pmc passes (scripts/collect_profiles.sh): the
# committed file `roofline.traffic` is read from -- named, so that counter files of other workloads can never be picked up
PMC_TRAFFIC_FILE = "r04_pmc_traffic.json"
N_STEPS, N_TRAJ, T_MAX, P, D = 4000, 1024, 40.0, 3, 2


def make_problem(ra, rank):
    """theta_b = (0.2,0.2,3) exp(0.1 eps_b), x0_b = (-1,1) + 0.1 eps'_b, seed 20240; global index."""
    rng = np.random.default_rng(20240)
    n_tot = N_TRAJ * (rank + 1)
    eps = rng.standard_normal((n_tot, 5))[rank * N_TRAJ:(rank + 1) * N_TRAJ]
    theta = np.array([0.2, 0.2, 3.0]) * np.exp(0.1 * eps[:, :3])
    x0v = np.array([-1.0, 1.0]) + 0.1 * eps[:, 3:]
    if rank == 0:
        theta[0], x0v[0] = [0.2, 0.2, 3.0], [-1.0, 1.0]       # b = 0 is the unperturbed README problem
    W, init = ra.utils.first_order_pad(ra.ode.fitzhugh_nagumo, D, P)
    x0 = init(x0v, 0.0, theta=theta)
    prior = ra.ibm_init(T_MAX / N_STEPS, P, np.array([0.1, 0.1]))
    return W, x0, theta, prior


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_quota():
    """CPUs this process may use at once according to its cgroup (the GPU boxes cap a job's CPU share), or None."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = float(f.read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_baseline(W, x0, theta, prior, budget_s=14.0):
    """
    The reference algorithm on the host cores (SURVEY.md section 8d "CPU baseline"; the protocol of the reference's own
    examples/timings.py:28-46 -- warm-up, then the mean over repeated runs), as a bounded sample of the SAME workload:
      port   -- plain-C restatement (oracle/c, -O3 -march=native, compile-time block sizes, OpenMP over trajectories).
                Outputs and per-thread scratch are allocated and paged in ONCE, outside the timing; only the C passes are
                timed (omp_get_wtime inside the library).  Thread sweep {1, 8, 16, 32, 64, 128, all}: `value` is the best
                (the GPU boxes cap a job's CPU share by cgroup, reported as `cgroup_cpu_quota`: more threads than that only
                contend, which is what a falling sweep shows).
      numpy  -- the batch-vectorised NumPy restatement (oracle/scan.py: einsum / batched solves over the 1024
                trajectories, what vmap + XLA do), a few hundred steps of the same problem.
    Both stand in for rodeo's JAX-CPU path, which cannot run here (no jax).
    """
    out = {"value": None, "unit": "trajectory-steps/s", "cores": 0, "kind": "port", "sample": ""}
    try:
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle", "c"), "native"], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        from oracle import c_port
        native = os.path.join(ROOT, "oracle", "c", "librk_oracle_native.so")
        if os.path.exists(native):
            c_port._PATH = native
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        counts = sorted({t for t in (1, 8, 16, 32, 64, 128, avail) if t <= avail})
        ts = c_port.TimedSolve("fitzhugh_nagumo", "kramer", W, x0, 0.0, T_MAX, N_STEPS, prior, theta, max_threads=avail)
        per_leg = budget_s / (len(counts) + 1)
        sweep, spent = {}, 0.0
        for nt in counts:
            n_traj = N_TRAJ if nt > 1 else 128                   # one thread: a 128-trajectory sample of the batch
            warm = ts.run(1, nt, n_traj)                         # warm-up pass (also calibrates)
            reps = int(min(max(per_leg / max(warm, 1e-4), 1), 400))
            sec = ts.run(reps, nt, n_traj)
            spent += warm + sec
            sweep[str(nt)] = reps * n_traj * N_STEPS / sec
        best_nt = max(sweep, key=sweep.get)
        one = sweep.get("1")
        out.update({
            "value": sweep[best_nt], "cores": int(best_nt), "value_best": sweep[best_nt], "threads_best": int(best_nt),
            "value_1thread": one, "threads_available": avail, "cgroup_cpu_quota": _cpu_quota(),
            "cpu_model": _cpu_model(),
            "scaling_efficiency_at_best": (sweep[best_nt] / (int(best_nt) * one)) if one else None,
            "thread_sweep": sweep,
            "sample": f"solve_mv + kramer on the C2 problem ({N_STEPS} steps): repeated passes over the {N_TRAJ} "
                      f"trajectories (128 of them on one thread), ~{per_leg:.1f} s per thread count, {spent:.1f} s in "
                      f"all; plain C -O3 -march=native restatement of the reference algorithm (oracle/c), outputs and "
                      f"scratch preallocated and paged in outside the timing"})
    except Exception as e:                                   # the baseline is a report, never a reason to fail
        out["sample"] = f"failed: {e}"
        return out
    try:                                                     # NumPy leg: the same problem, first `n_np` steps
        from oracle import scan, odes, interrogations as oi
        n_np = 300
        t_end = T_MAX * n_np / N_STEPS
        scan.solve_mv(None, odes.fitzhugh_nagumo, W, x0, 0.0, t_end / 10, n_np // 10, oi.interrogate_kramer, prior,
                      theta=theta)                           # warm-up
        t0 = time.perf_counter()
        scan.solve_mv(None, odes.fitzhugh_nagumo, W, x0, 0.0, t_end, n_np, oi.interrogate_kramer, prior, theta=theta)
        dt = time.perf_counter() - t0
        out["numpy_batched"] = {"value": N_TRAJ * n_np / dt, "unit": "trajectory-steps/s",
                                "sample": f"oracle/scan.py solve_mv, {N_TRAJ} trajectories x the first {n_np} steps "
                                          f"(dt as in C2), batch-vectorised NumPy, {dt:.1f} s"}
    except Exception as e:                                   # noqa: BLE001
        out["numpy_batched"] = {"value": None, "sample": f"failed: {e}"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--require-rccl", action="store_true", help="N > 1: exit non-zero if RCCL does not come up on every rank")
    ap.add_argument("--comm", choices=["rccl", "host"], default="rccl", help="N > 1: what carries the barriers")
    ap.add_argument("--allow-host-fallback", action="store_true",
                    help="N > 1 on a box with N GPUs: keep going on the host channel when RCCL does not come up (default: exit 3)")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # own launcher: N rank processes of this script, started before anything here has touched a GPU
        from rodeo_amd import hostgroup
        sys.exit(hostgroup.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus,
                                       timeout=float(os.environ.get("RK_BENCH_TIMEOUT", "1500"))))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world

    import rodeo_amd as ra
    from rodeo_amd import _lib, hostgroup, shard
    import ctypes as C
    lib = _lib.load()
    n_dev = C.c_int(0)
    _lib.check(lib.rk_device_count(C.byref(n_dev)))
    if local_rank >= n_dev.value:       # rehearsal on a box with fewer GPUs than ranks: share devices (RCCL then falls back)
        print(f"[rank {rank}] only {n_dev.value} GPU(s) visible; sharing device {local_rank % n_dev.value}", file=sys.stderr)
    dev = ra.Device(local_rank % max(n_dev.value, 1))

    group = hostgroup.HostGroup.from_env(timeout=float(os.environ.get("RK_BENCH_RDZV_TIMEOUT", "180")))
    rccl, comm, hard_exit = None, "none", False
    if world > 1:
        comm = "host-tcp"
        if args.comm == "rccl":
            sys.stdout.flush()
            saved_stdout = os.dup(1)
            os.dup2(2, 1)                 # RCCL prints diagnostics on stdout; keep stdout for the one JSON line
            try:
                rccl, hard_exit, why = shard.init_rccl_or_fail(
                    dev, group, deadline=float(os.environ.get("RK_BENCH_RCCL_TIMEOUT", "90")))
            finally:
                sys.stdout.flush()
                os.dup2(saved_stdout, 1)
                os.close(saved_stdout)
            if rccl is not None:
                comm = "rccl"
            else:
                # A box with one GPU per rank MUST bring RCCL up: a scaling run that quietly measured the host channel would
                # be worthless.  The fallback is for the shared-GPU rehearsal only (fewer GPUs than ranks) or on request.
                enough_gpus = n_dev.value >= world
                fatal = args.require_rccl or (enough_gpus and not args.allow_host_fallback)
                print(f"[rank {rank}] RCCL communicator unavailable ({why}); "
                      + ("exiting with code 3 (--allow-host-fallback to run on the host channel)" if fatal
                         else "host channel carries the barriers (config.comm = host-tcp)"), file=sys.stderr)
                if fatal:
                    sys.stdout.flush(); sys.stderr.flush()
                    os._exit(3)            # every rank takes this branch (the outcome was agreed over the host channel)

    def barrier():
        dev.sync()
        if rccl is not None:
            rccl.barrier()
        else:
            group.barrier()

    W, x0, theta, prior = make_problem(ra, rank)
    plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, W, x0, 0.0, T_MAX, N_STEPS, ra.interrogate.interrogate_kramer, prior,
                        device=dev, traj_offset=rank * N_TRAJ, theta=theta)
    # the warm-up solves run with the per-launch event brackets ON so that the handle's event pool is filled before the timed
    # region (no hipEventCreate inside it); re-enabling clears the recorded entries and keeps the pool
    dev.profile_enable(True, keep=True)
    for _ in range(args.warmup):
        plan.mv(None)
    dev.sync()
    # ---- timed region: exactly K solves, inputs/outputs resident in HBM.  HIP events bracket every kernel launch on
    # the library's stream during these K solves (two hipEventRecord per launch, no synchronisation); they are read
    # after the closing barrier and give the per-kernel durations the roofline figures use ----
    dev.profile_enable(True, keep=True)
    barrier()
    t0 = time.perf_counter()
    dev.timer_start()
    for _ in range(args.steps):
        plan.mv(None)
    dev_ms = dev.timer_stop()                  # HIP events on the library's stream (also synchronises)
    barrier()
    wall = time.perf_counter() - t0
    wall = float(group.allreduce(wall, "max"))
    acc = {}
    for name, ms in dev.profile_last(cap=8 * args.steps + 16):
        acc.setdefault(name, []).append(ms)
    dev.profile_enable(False)
    kern_ms = {k: float(np.mean(v)) for k, v in acc.items()}

    # sanity on the result of the timed computation (cheap): finite, end conditions
    last = plan.var_state.slice0_host(-1)                      # state at t_max (tile layout: [Sigma | mu] rows)
    ok = bool(np.all(np.isfinite(last)))

    if rank == 0:
        units = N_TRAJ * N_STEPS                                  # trajectory-steps per launch per rank
        a_fwd, a_bwd = D * P * (P + 1) * 8, 2 * D * P * (P + 1) * 8    # write filt | read filt + write smooth
        dom = max(kern_ms, key=kern_ms.get)
        a_dom = a_fwd if dom.startswith("fwd") else a_bwd
        layout = {0: "batch-minor", 1: "tile3 (3x4 [Sigma|mu] rows per block, 96 B)"}.get(plan.layout, str(plan.layout))
        achieved = a_dom * units / (kern_ms[dom] * 1e-3) / 1e9
        solve_ms = sum(kern_ms.values())
        # HBM bytes per launch from separate rocprofv3 --pmc passes (FETCH_SIZE corrected x2, WRITE_SIZE), if recorded
        # (not measured by this run: counters need their own rocprofv3 pass -- scripts/collect_profiles.sh; the file is named)
        traffic, traffic_src = None, None
        try:
            # the file is NAMED (PMC_TRAFFIC_FILE at the top): the newest set of this command's counter passes, not whatever
            # sorts last among the profiles of other workloads
            traffic = json.load(open(os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)))["kernels"][dom]["hbm_bytes_corrected"]
            traffic_src = "profiles/" + PMC_TRAFFIC_FILE + " (separate rocprofv3 --pmc passes of this command, committed)"
        except Exception:
            traffic, traffic_src = None, None
        out = {
            "metric": "ODE steps/sec x batch (trajectory-steps/s), FitzHugh-Nagumo q=3, 4000 steps, solve_mv",
            "value": world * units * args.steps / wall,
            "unit": "trajectory-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C2: FitzHugh-Nagumo n_vars=2 n_deriv=3, t in [0,40], 4000 steps, "
                                   "1024 trajectories per GPU (perturbed theta/x0, IBM prior sigma=0.1), "
                                   "solve_mv + interrogate_kramer, kalman_type=standard",
                       "n_traj_per_gpu": N_TRAJ, "n_steps": N_STEPS, "n_block": D, "n_bstate": P,
                       "sharding": f"batch split over {world} rank(s), no data-path collective", "comm": comm,
                       "hbm_layout": layout, "result_finite": ok},
            "device_ms_per_step": dev_ms / args.steps,
            "kernels_ms": kern_ms,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_traj_step": a_dom},
            # the whole solve against the HBM roof: `frac` divides by the wall-clock ms_per_step of the timed region (what the
            # driver's clock sees); `frac_kernel_sum` by the sum of the kernels' HIP-event durations (no launch gaps)
            "roofline_solve": {"bound": "hbm", "achieved": (a_fwd + a_bwd) * units / (wall / args.steps) / 1e9,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": (a_fwd + a_bwd) * units / (wall / args.steps) / 1e9 / HBM_PEAK_GBS,
                               "frac_kernel_sum": (a_fwd + a_bwd) * units / (solve_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "algorithmic_bytes_per_traj_step": a_fwd + a_bwd},
        }
        # why the dominant kernel sits far below the HBM roof: one dependent chain per wave
        fwd = next((k for k in kern_ms if k.startswith("fwd")), None)
        if fwd is not None:
            tiles_per_wave = 4
            out["latency_bound"] = {
                "kernel": fwd, "ns_per_dependent_step": kern_ms[fwd] * 1e6 / N_STEPS,
                "waves": -(-N_TRAJ * D // tiles_per_wave), "simds": 1024,
                "note": "the filter recursion is sequential in time: one wave per SIMD at most, its time is n_steps x the "
                        "step's dependent instruction chain (7 fp64 MFMAs + 13 VALU ~ 240 cycles) whatever the batch up to "
                        "2048 trajectories (profiles/r01_c2_kernel_times_vs_batch.log)"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(W, x0, theta, prior)
        print(json.dumps(out), flush=True)

    if rccl is not None:
        rccl.close()
    group.barrier()
    group.close()
    if hard_exit:                                    # a helper thread is still inside the RCCL bootstrap (agreed by all ranks)
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
