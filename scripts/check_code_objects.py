#!/usr/bin/env python3
"""
Build-time check of the gfx950 code objects inside rodeo_amd/librodeo_kalman.so (run by `make check` in
rodeo_amd/csrc, by __graft_entry__.build() and by tests/test_abi_and_host.py): every kernel's AMDGPU metadata is read
back with llvm-readelf and must satisfy what the launches assume, so that a resource problem shows up as a failed
build here and not as a fault on the GPU box:

  * no dynamic stack (`.uses_dynamic_stack: false`): the kernels call real device functions (dense path), whose frames
    must be known to the compiler -- an indirect or recursive call would make the runtime guess a stack size;
  * private segment (scratch: spills + frames of called functions) at most SCRATCH_LIMIT bytes per lane, except the
    kernels listed in SCRATCH_ALLOW with their own bound (runtime-sized local arrays of the unit-parity path);
  * LDS at most 160 KB (MI355X_MICROARCH.md);
  * registers: VGPRs + AGPRs of a kernel fit the waves its own launch bound puts on one SIMD
    (512 per SIMD lane / ceil(max_flat_workgroup_size / 256) waves);
  * DPP hazards: the disassembly of every code object goes through scripts/check_dpp_hazards.py (the inline-assembly DPP
    FMAs of the blocked-tile, dense and LU kernels are invisible to the compiler's own hazard recogniser).

    python3 scripts/check_code_objects.py [path/to/librodeo_kalman.so]      exit code 0 = all kernels pass
"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_dpp_hazards                                             # noqa: E402

LLVM = os.environ.get("RK_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
DPP_RESULTS = [0, 0]            # DPP instructions scanned / hazards found in the shipped code objects (None: skip the scan)
SCRATCH_LIMIT = 2048
SCRATCH_ALLOW = {"kalman_op_kernel": 32768,        # kalman_batched.hip: runtime (n_state <= 16) local matrices, unit-parity ops
                 "13bwd_mv_kernel": 6144,          # solve_small.hip, one lane per (trajectory, block): the p x p register matrices
                 "14bwd_sim_kernel": 6144,         # spill from n_bstate = 7 on (the n_bmeas > 1 / non-block path, p <= 9)
                 "10fwd_kernelINS_8Lorenz63ELi6": 4096,   # lane-per-trajectory forward, three blocks of 6 x 6 per lane (fenrir / _solve_filter at p = 6)
                 "15fwd_sqrt_kernel": 6144,       # solve_sqrt.hip, lane-per-trajectory square-root filter: the stacked (2p x p)
                 "15bwd_sqrt_kernel": 6144,
                 "21bwd_sqrt_chain_kernel": 4096,  # the two-kernel form's chain: three p x p record matrices + the 3p x p stack at n_bstate = 8
                 "16sqrt_gain_kernel": 6144,
                 "17fenrir_bwd_kernel": 4096}       # lane per (block, trajectory) on the blocked tile records at n_bstate = 8       # Householder inputs of every block spill from n_bstate = 7 on (functional path)
LDS_LIMIT = 160 * 1024
FIELDS = ("agpr_count", "group_segment_fixed_size", "max_flat_workgroup_size", "private_segment_fixed_size",
          "sgpr_spill_count", "uses_dynamic_stack", "vgpr_count", "vgpr_spill_count")


def kernels_of(so_path):
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "lib.so")
        os.symlink(os.path.abspath(so_path), local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, cwd=tmp,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], check=True,
                                 capture_output=True, text=True).stdout
            if DPP_RESULTS is not None:
                # inline-assembly DPP operands are invisible to the compiler's hazard recogniser: scan the disassembly of the
                # SHIPPED code (scripts/check_dpp_hazards.py: VGPR write -> DPP read, VALU write of EXEC -> DPP, labels)
                dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", os.path.join(tmp, f)],
                                     check=True, capture_output=True, text=True).stdout
                n_dpp, n_bad = check_dpp_hazards.scan(dis.splitlines(), f, verbose=True)
                DPP_RESULTS[0] += n_dpp
                DPP_RESULTS[1] += n_bad
            for block in re.split(r"\n  - (?=\.agpr_count|\.args)", txt)[1:]:
                k = {}
                m = re.search(r"^\s+\.name:\s+(\S+)", block, re.M)
                if not m:
                    continue
                k["name"] = m.group(1)
                for fld in FIELDS:
                    mm = re.search(r"^\s+\.%s:\s+(\S+)" % fld, block, re.M)
                    if mm:
                        v = mm.group(1)
                        k[fld] = (v == "true") if v in ("true", "false") else int(v)
                out.append(k)
    return out


def demangled(name):
    try:
        out = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt"), name], capture_output=True, text=True).stdout.strip()
        return out or name
    except OSError:
        return name


# Producer/consumer kernels whose design counts on TWO workgroups per CU (the chain wave of one fills the stalls of the
# other): a few registers too many make the two workgroups of a CU run one after the other without any other symptom
# (found with per-workgroup time stamps on bwd_mv_tile4_kernel, which had crept to 260 registers: DESIGN.md).
MIN_WORKGROUPS_PER_CU = {"19bwd_mv_tile3_kernel": 2, "19bwd_mv_tile4_kernel": 2, "20bwd_sim_tile3_kernel": 2,
                         "23fenrir_bwd_tile3_kernel": 2}


def workgroups_per_cu(k):
    """Resident workgroups per CU allowed by registers (512 per lane and SIMD, granule 8, 8 waves per SIMD at most)
    and by LDS (160 KiB)."""
    wg = k.get("max_flat_workgroup_size", 1024)
    regs = k.get("vgpr_count", 0) + k.get("agpr_count", 0)
    alloc = max(8, -(-regs // 8) * 8)
    waves_per_simd = min(8, 512 // alloc)
    waves = -(-wg // 64)
    by_regs = (waves_per_simd * 4) // waves
    lds = k.get("group_segment_fixed_size", 0)
    by_lds = LDS_LIMIT // lds if lds else 32
    return min(by_regs, by_lds, 32 // waves)


def check(kernels):
    problems = []
    for k in kernels:
        nm = demangled(k["name"])
        if k.get("uses_dynamic_stack", False):
            problems.append(f"{nm}: uses a dynamic stack")
        lim = max([v for key, v in SCRATCH_ALLOW.items() if key in k["name"]] + [SCRATCH_LIMIT])
        if k.get("private_segment_fixed_size", 0) > lim:
            problems.append(f"{nm}: {k['private_segment_fixed_size']} B of scratch per lane (limit {lim})")
        if k.get("group_segment_fixed_size", 0) > LDS_LIMIT:
            problems.append(f"{nm}: {k['group_segment_fixed_size']} B of LDS (limit {LDS_LIMIT})")
        for key, need in MIN_WORKGROUPS_PER_CU.items():
            if key in k["name"] and k.get("max_flat_workgroup_size", 1024) > 256:
                need = 1                       # the 8-wave two-set form of the kernel (SETS = 2): one workgroup per CU by design
            if key in k["name"] and workgroups_per_cu(k) < need:
                problems.append(f"{nm}: {workgroups_per_cu(k)} workgroup(s) per CU (registers "
                                f"{k.get('vgpr_count', 0) + k.get('agpr_count', 0)}, LDS {k.get('group_segment_fixed_size', 0)} B); "
                                f"the kernel is built for {need}")
        wg = k.get("max_flat_workgroup_size", 1024)
        waves_per_simd = max(1, -(-wg // 256))
        regs = k.get("vgpr_count", 0) + k.get("agpr_count", 0)
        if regs > 512 // waves_per_simd:
            problems.append(f"{nm}: {regs} VGPRs+AGPRs do not fit {waves_per_simd} wave(s) per SIMD "
                            f"(workgroup {wg})")
    return problems


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "rodeo_amd", "librodeo_kalman.so")
    ks = kernels_of(so)
    if not ks:
        print("check_code_objects: no gfx950 kernels found in", so)
        return 2
    problems = check(ks)
    if DPP_RESULTS is not None:
        print(f"check_code_objects: {DPP_RESULTS[0]} DPP instructions in the shipped code objects, {DPP_RESULTS[1]} hazard(s)")
        if DPP_RESULTS[1]:
            problems.append(f"{DPP_RESULTS[1]} DPP hazard(s) (VGPR / EXEC write too close in front of a DPP read, see above)")
    worst = max(ks, key=lambda k: k.get("private_segment_fixed_size", 0))
    print(f"check_code_objects: {len(ks)} kernels; largest scratch {worst.get('private_segment_fixed_size', 0)} B "
          f"({demangled(worst['name']).split('(')[0][:60]}); {len(problems)} problem(s)")
    for p in problems:
        print("  ", p)
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
