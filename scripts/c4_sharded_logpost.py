#!/usr/bin/env python3
"""
BASELINE config 4 across GPUs: 8192 parameter draws of the FitzHugh-Nagumo pseudo-marginal log-posterior
(solve_sim + interrogate_chkrebtii, N = 800, 41 observations), sharded contiguously over the ranks, draws keyed by the
GLOBAL draw index (traj_offset), per-draw scalars all-gathered over RCCL/xGMI inside librodeo_kalman.so
(rk_allgather_f64).  No torch: the ranks meet over rodeo_amd.hostgroup (standard library), which also carries the
gather when RCCL cannot be set up (a rehearsal with several ranks on one GPU) unless --require-rccl is given.

    python scripts/c4_sharded_logpost.py --gpus G [--check] [--require-rccl]       (spawns its own G ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 scripts/c4_sharded_logpost.py
Rank 0 prints one bench-style JSON line; --check re-evaluates the first draws of every shard on rank 0 alone and
compares bit for bit (draws are keyed by the global index, so the sharding must not change them).
"""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--draws", type=int, default=8192)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--require-rccl", action="store_true")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        from rodeo_amd import hostgroup
        sys.exit(hostgroup.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus, timeout=900))
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import rodeo_amd as ra
    from rodeo_amd import _lib, shard, hostgroup
    from rodeo_amd.inference.pseudo_marginal import FitzLogPosterior
    lib = _lib.load()
    n_dev = C.c_int(0)
    _lib.check(lib.rk_device_count(C.byref(n_dev)))
    dev = ra.Device(local_rank % max(n_dev.value, 1))
    group = hostgroup.HostGroup.from_env(timeout=180)
    comm, hard_exit = None, False
    if world > 1:
        saved = os.dup(1); os.dup2(2, 1)                # RCCL chatter off stdout
        try:
            comm, hard_exit, why = shard.init_rccl_or_fail(dev, group)
        finally:
            sys.stdout.flush(); os.dup2(saved, 1); os.close(saved)
        if comm is None:
            print(f"[rank {rank}] RCCL unavailable ({why}); gathering over the host channel", file=sys.stderr)
            if args.require_rccl:
                sys.stderr.flush()
                os._exit(3)

    # the draws (docs/examples/parameter.md:372-373, 400-401): identical on every rank, each takes its slice
    n_total = args.draws
    rng = np.random.default_rng(20242)
    u0 = np.concatenate([np.log([.2, .2, 3.]), [-1., 1.], [.1, .1]])
    upars = u0 + np.array([0.01, 0.1, 0.01, 0.01, 0.01, 0.01, 0.01]) * rng.standard_normal((n_total, 7))
    obs_t = np.linspace(0, 40, 41)
    Y = np.array([-1., 1.]) + rng.standard_normal((41, 2))
    lo, hi = shard.partition(n_total, rank, world)
    lp_fun = FitzLogPosterior(Y, obs_t, 0., 40., 800, np.sqrt(0.005), hi - lo, device=dev, traj_offset=lo)
    seed = 20242
    mine, _ = lp_fun(upars[lo:hi], seed)                # warm-up (JIT-free: built-in ODE), also the checked values
    dev.sync(); group.barrier()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        if comm is not None:
            # the log-posteriors never leave the device on the way in: padded to the common width in a device staging buffer,
            # rk_allgather_f64 over RCCL / xGMI, one download of all n_total values (ragged shards included)
            full = shard.gather_scalars_device(lp_fun.device(upars[lo:hi], seed), n_total, rank, world, comm, dev)
        else:
            mine, _ = lp_fun(upars[lo:hi], seed)
            full = shard.gather_scalars(mine, n_total, rank, world, group)
    dev.sync(); group.barrier()
    dt = float(group.allreduce((time.perf_counter() - t0) / args.reps, "max"))
    ok = None
    if args.check and rank == 0:
        ok = True
        for r in range(world):                          # the first 16 draws of every shard, evaluated here with their global offsets
            a, b = shard.partition(n_total, r, world)
            k = min(16, b - a)
            ref = FitzLogPosterior(Y, obs_t, 0., 40., 800, np.sqrt(0.005), k, device=dev, traj_offset=a)(upars[a:a + k], seed)[0]
            ok = ok and bool(np.array_equal(ref, full[a:a + k]))
    if rank == 0:
        a_sim = (2 * 2 * 3 * 4 + 2 * 3) * 8              # A_sim = (2 d p (p+1) + d p) 8 = 432 B per draw-step
        val = n_total * 800 / dt
        print(json.dumps({
            "metric": "draw-steps/s of the C4 pseudo-marginal log-posterior (solve_sim + interrogate_chkrebtii + Gaussian "
                      "observation log-density), all ranks", "value": val, "unit": "draw-steps/s", "n_gpus": world,
            "steps": args.reps, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "strong",
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C4: FitzHugh-Nagumo, 8192 parameter draws, N=800, 41 observations", "draws": n_total,
                       "comm": "rccl" if comm is not None else ("host-tcp" if world > 1 else "none"),
                       "gather": "rk_allgather_f64 (RCCL, device buffers, padded width)" if comm is not None else "host"},
            "roofline_solve": {"bound": "hbm", "achieved": a_sim * val / 1e9, "peak": 8000.0 * world, "unit": "GB/s",
                               "frac": a_sim * val / 1e9 / (8000.0 * world), "algorithmic_bytes_per_draw_step": a_sim},
            "finite": bool(np.all(np.isfinite(full))), "matches_single_rank": ok}), flush=True)
    if comm is not None:
        comm.close()
    group.barrier(); group.close()
    code = 4 if (args.check and rank == 0 and not ok) else 0
    sys.stdout.flush(); sys.stderr.flush()
    if hard_exit:                                       # a helper thread is still inside the RCCL bootstrap: no interpreter teardown
        os._exit(code)
    sys.exit(code)


if __name__ == "__main__":
    main()
