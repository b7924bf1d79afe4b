"""bench.py end to end on the GPU: the one JSON line the driver reads (contract of the task brief: metric of BASELINE.json on
its C2 workload, whole-job value, roofline of the dominant kernel from HIP events of the timed solves, cpu_baseline on a
bounded sample), for the single-process launch and for the self-spawned two-rank launch on one GPU (RCCL refuses a shared
device there: the host channel carries the barriers and the line says so)."""
import json
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, timeout=600):
    env = dict(os.environ, RK_BENCH_RCCL_TIMEOUT="20")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout,
                         env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]                     # exactly ONE line on stdout
    return json.loads(lines[0])


def _common(d, n_gpus, steps, warmup, shared_gpu=False):
    assert d["unit"] == "trajectory-steps/s" and "FitzHugh-Nagumo" in d["metric"] and "4000 steps" in d["metric"]
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "1024 trajectories per GPU" in d["config"]["workload"]
    assert abs(d["value"] - n_gpus * 1024 * 4000 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["kernel"] in d["kernels_ms"] and 0.05 < r["frac"] < 1.0
    assert r["traffic"] is None or 0.9 < r["traffic"] / (r["algorithmic_bytes_per_traj_step"] * 1024 * 4000) < 1.5
    # (two ranks rehearsing on ONE card halve each rank's rate: ~0.21 nominal, seen between 0.18 and 0.22)
    assert (0.1 if shared_gpu else 0.2) < d["roofline_solve"]["frac"] < 1.0
    # the per-kernel times come from the timed solves themselves and add up to (a bit less than) the wall time per solve
    assert 0.8 * d["ms_per_step"] < sum(d["kernels_ms"].values()) <= 1.02 * d["ms_per_step"]


def test_single_process_line_with_cpu_baseline():
    d = _run("--steps", "5", "--warmup", "2")
    _common(d, 1, 5, 2)
    assert d["value"] > 2e9                                        # (5.8e9 at the end of round 2)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == d["unit"] and c["cores"] >= 1 and c["value"] > 1e6 and c["sample"]
    assert c["value_1thread"] > 1e6 and c["value_best"] >= c["value_1thread"] and str(c["threads_best"]) in c["thread_sweep"]


def test_two_ranks_on_one_gpu_fall_back_to_the_host_channel():
    d = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    _common(d, 2, 3, 1, shared_gpu=True)
    assert d["config"]["comm"] in ("host-tcp", "rccl") and "cpu_baseline" not in d
