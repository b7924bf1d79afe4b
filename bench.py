#!/usr/bin/env python3
"""
bench.py -- headline benchmark of BASELINE.json: trajectory-steps per second (ODE steps/s x batch) of the batched
probabilistic ODE solve, FitzHugh-Nagumo n_vars=2 q(n_deriv)=3, 4000 steps, `solve_mv` + `interrogate_kramer`,
1024 trajectories per GPU (BASELINE.json configs[1]; SURVEY.md section 8d "C2").

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one complete solve (forward filter kernel + backward smoother kernel) of this rank's batch with all
inputs and outputs resident in HBM.  The path shards by independent trajectories: rank r owns global trajectories
[r*B, (r+1)*B) ("weak" scaling, no data-path collective).  Timing: barrier + device sync on both sides of exactly K
steps, MAX over ranks; rank 0 prints ONE JSON line.

torch is used only as CPU-side rendezvous plumbing (gloo) when N > 1; it never touches the GPU.  The device work goes
through librodeo_kalman.so (ctypes); barrier / max-reduce between ranks run over RCCL inside that library when the
communicator comes up, else over gloo.
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


def cpu_baseline(W, x0, theta, prior):
    """Plain-C restatement of the reference algorithm (oracle/c, 'port') on the host cores, bounded sample."""
    try:
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle", "c"), "native"], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        from oracle import c_port
        native = os.path.join(ROOT, "oracle", "c", "librk_oracle_native.so")
        if os.path.exists(native):
            c_port._PATH = native
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        # calibrate on one pass over the batch, then repeat the pass until ~12 s of CPU work have been timed
        n = N_TRAJ
        t0 = time.perf_counter()
        c_port.solve_mv("fitzhugh_nagumo", "kramer", W, x0[:n], 0.0, T_MAX, N_STEPS, prior, theta[:n], cores)
        dt1 = max(time.perf_counter() - t0, 1e-3)
        reps = int(min(max(12.0 / dt1, 1), 200))
        t0 = time.perf_counter()
        for _ in range(reps):
            c_port.solve_mv("fitzhugh_nagumo", "kramer", W, x0[:n], 0.0, T_MAX, N_STEPS, prior, theta[:n], cores)
        dt = time.perf_counter() - t0
        return {"value": reps * n * N_STEPS / dt, "unit": "trajectory-steps/s", "cores": cores, "kind": "port",
                "sample": f"{reps} passes over the {n} trajectories x {N_STEPS} steps ({dt:.1f} s), solve_mv+kramer, "
                          f"plain C -O3 -march=native restatement of the reference algorithm (oracle/c), OpenMP over "
                          f"trajectories on {cores} host threads"}
    except Exception as e:                                   # the baseline is a report, never a reason to fail
        return {"value": None, "unit": "trajectory-steps/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N with N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import rodeo_amd as ra                      # loads librodeo_kalman.so BEFORE torch's bundled ROCm libraries
    from rodeo_amd import _lib
    import ctypes as C
    lib = _lib.load()
    n_dev = C.c_int(0)
    _lib.check(lib.rk_device_count(C.byref(n_dev)))
    if local_rank >= n_dev.value:       # rehearsal on a box with fewer GPUs than ranks: share devices (RCCL then falls back)
        print(f"[rank {rank}] only {n_dev.value} GPU(s) visible; sharing device {local_rank % n_dev.value}", file=sys.stderr)
    dev = ra.Device(local_rank % max(n_dev.value, 1))

    dist = None
    comm = "none"
    hard_exit = False
    if world > 1:
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)                     # gloo and RCCL print diagnostics on stdout; keep stdout for the one JSON line
        try:
            import torch.distributed as dist
            dist.init_process_group("gloo", rank=rank, world_size=world)
            comm = "gloo"
            # RCCL communicator inside the library (only used for the barriers around the timed region: the data path has
            # no collective).  Set up in a helper thread with a deadline, and used only if EVERY rank got it -- a rank
            # stuck in the bootstrap must not take the scaling run down; gloo then carries the barriers.
            import threading
            import torch
            state = {"ok": False, "err": None}

            uid = (C.c_char * _lib.COMM_UID_BYTES)()
            if rank == 0:
                try:
                    _lib.check(lib.rk_comm_uid(uid))
                except Exception as e:                       # noqa: BLE001
                    state["err"] = e
            box = [bytes(uid), state["err"] is None]
            dist.broadcast_object_list(box, src=0)           # (gloo, main thread)
            uid2 = (C.c_char * _lib.COMM_UID_BYTES).from_buffer_copy(box[0])

            def _init():
                try:
                    _lib.check(lib.rk_comm_init(dev.h, rank, world, uid2))
                    state["ok"] = True
                except Exception as e:                       # noqa: BLE001
                    state["err"] = e
            stuck = False
            if box[1]:
                th = threading.Thread(target=_init, daemon=True)
                th.start()
                th.join(timeout=float(os.environ.get("RK_BENCH_RCCL_TIMEOUT", "90")))
                stuck = th.is_alive()
            flag = torch.tensor([1 if (state["ok"] and not stuck) else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)      # every rank takes part, stuck helper thread or not
            if int(flag.item()) == 1:
                comm = "rccl"
            else:
                why = "timed out" if stuck else (state["err"] or "another rank failed")
                print(f"[rank {rank}] RCCL communicator unavailable ({why}); using gloo for barriers", file=sys.stderr)
                if state["ok"] and not stuck:
                    lib.rk_comm_destroy(dev.h)
                hard_exit = stuck
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    def barrier():
        dev.sync()
        if comm == "rccl":
            _lib.check(lib.rk_comm_barrier(dev.h))
        elif dist is not None:
            dist.barrier()

    W, x0, theta, prior = make_problem(ra, rank)
    plan = ra.SolvePlan(ra.ode.fitzhugh_nagumo, W, x0, 0.0, T_MAX, N_STEPS, ra.interrogate.interrogate_kramer, prior,
                        device=dev, traj_offset=rank * N_TRAJ, theta=theta)
    for _ in range(args.warmup):
        plan.mv(None)
    # ---- timed region: exactly K solves, inputs/outputs resident in HBM ----
    barrier()
    t0 = time.perf_counter()
    dev.timer_start()
    for _ in range(args.steps):
        plan.mv(None)
    dev_ms = dev.timer_stop()                  # HIP events on the library's stream (also synchronises)
    barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    # ---- per-kernel device time (HIP events around each launch), outside the timed region ----
    dev.profile_enable(True)
    acc = {}
    reps = max(3, min(args.steps, 10))
    for _ in range(reps):
        plan.mv(None)
        for name, ms in dev.profile_last():
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
        traffic = None
        try:
            pm = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("pmc_traffic.json"))
            if pm:
                traffic = json.load(open(os.path.join(ROOT, "profiles", pm[-1])))["kernels"][dom]["hbm_bytes_corrected"]
        except Exception:
            traffic = None
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
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_traj_step": a_dom},
            "roofline_solve": {"bound": "hbm", "achieved": (a_fwd + a_bwd) * units / (solve_ms * 1e-3) / 1e9,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": (a_fwd + a_bwd) * units / (solve_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
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

    if comm == "rccl":
        lib.rk_comm_destroy(dev.h)
    if dist is not None and not hard_exit:
        dist.barrier()
        dist.destroy_process_group()
    if hard_exit:                                    # a helper thread is still inside the RCCL bootstrap
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
