#!/usr/bin/env python3
"""
BASELINE config 4 across GPUs: 8192 parameter draws of the FitzHugh-Nagumo pseudo-marginal log-posterior
(solve_sim + interrogate_chkrebtii, N = 800, 41 observations), sharded contiguously over the ranks, draws keyed by the
GLOBAL draw index (traj_offset), per-draw scalars all-gathered over RCCL/xGMI inside librodeo_kalman.so
(rk_allgather_f64) -- or over gloo when RCCL cannot be set up (e.g. a rehearsal with several ranks on one GPU).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 scripts/c4_sharded_logpost.py
Rank 0 prints one JSON line; --check re-evaluates the first draws of every shard on rank 0 alone and compares.
"""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--draws", type=int, default=8192)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import rodeo_amd as ra                              # before torch: our library's ROCm libraries first
    from rodeo_amd import _lib, shard
    from rodeo_amd.inference.pseudo_marginal import FitzLogPosterior
    lib = _lib.load()
    n_dev = C.c_int(0)
    _lib.check(lib.rk_device_count(C.byref(n_dev)))
    dev = ra.Device(local_rank % max(n_dev.value, 1))
    dist, comm = None, None
    if world > 1:
        saved = os.dup(1); os.dup2(2, 1)                # gloo / RCCL chatter off stdout
        try:
            import torch.distributed as dist
            dist.init_process_group("gloo", rank=rank, world_size=world)

            def bcast(b):
                box = [b]
                dist.broadcast_object_list(box, src=0)
                return box[0]
            try:
                comm = shard.RcclComm(dev, rank, world, bcast=bcast)
            except Exception as e:
                print(f"[rank {rank}] RCCL unavailable ({e}); gathering over gloo", file=sys.stderr)
        finally:
            sys.stdout.flush(); os.dup2(saved, 1); os.close(saved)

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
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        mine, _ = lp_fun(upars[lo:hi], seed)
        if comm is not None and (hi - lo) * world == n_total:
            send = dev.to_device(np.ascontiguousarray(mine))
            recv = dev.empty((n_total,))
            comm.allgather(send, recv, hi - lo)
            full = recv.to_host()
        else:
            full = shard.gather_scalars(mine, n_total, rank, world, dist)
    if dist is not None:
        dist.barrier()
    dt = (time.perf_counter() - t0) / args.reps
    ok = None
    if args.check and rank == 0:
        ok = True
        for r in range(world):                          # the first 16 draws of every shard, evaluated here with their global offsets
            a, b = shard.partition(n_total, r, world)
            k = min(16, b - a)
            ref = FitzLogPosterior(Y, obs_t, 0., 40., 800, np.sqrt(0.005), k, device=dev, traj_offset=a)(upars[a:a + k], seed)[0]
            ok = ok and bool(np.array_equal(ref, full[a:a + k]))
    if rank == 0:
        print(json.dumps({"config": "C4: FitzHugh-Nagumo pseudo-marginal log-posterior, solve_sim + chkrebtii, N=800",
                          "draws": n_total, "n_gpus": world, "comm": "rccl" if comm is not None else ("gloo" if world > 1 else "none"),
                          "ms_per_evaluation": dt * 1e3, "draw_steps_per_s": n_total * 800 / dt,
                          "finite": bool(np.all(np.isfinite(full))), "matches_single_rank": ok}), flush=True)
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
