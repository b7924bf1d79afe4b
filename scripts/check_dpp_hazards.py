#!/usr/bin/env python3
"""
Inline-assembly DPP instructions are invisible to the compiler's hazard recogniser.  On gfx950
  * a VALU write of a VGPR needs TWO wait states before a DPP operand (src0) reads it;
  * a VALU write of EXEC (v_cmpx_*, v_readfirstlane / v_readlane do not count) needs FIVE wait states before any DPP
    instruction.
This scans either the saved assembly of a translation unit (hipcc --save-temps=obj ... ->
build/<name>-hip-amdgcn-amd-amdhsa-gfx950.s) or the llvm-objdump -d listing of a code object (what
scripts/check_code_objects.py feeds it for every code object of the shipped library) and reports every DPP instruction
whose hazard distance is not covered.  Basic-block boundaries: at a label (a branch target) the instructions in front are
unknown, so the history is reset CONSERVATIVELY -- a label counts as a VALU write of every VGPR and of EXEC.
    python3 scripts/check_dpp_hazards.py rodeo_amd/csrc/build/solve_tilen-hip-amdgcn-amd-amdhsa-gfx950.s [kernel-name-substring]
"""
import re, sys

ALL = "ALL"          # a label: unknown predecessor


def regs(tok):
    tok = tok.strip()
    m = re.fullmatch(r"-?\|?v\[(\d+):(\d+)\]\|?", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"-?\|?v(\d+)\|?", tok)
    return {int(m.group(1))} if m else set()


DPP_MARKS = ("row_newbcast", "quad_perm", "row_shr", "row_shl", "row_ror", "row_bcast", "row_mirror", "row_half_mirror",
             "wave_shr", "wave_shl", "wave_ror", "wave_rol")


def scan(lines, path="<stdin>", only="", verbose=True):
    """lines: assembly text (.s or objdump -d).  Returns (n_dpp, n_hazards)."""
    bad = n_dpp = 0
    kernel, hist = None, []
    for ln, line in enumerate(lines, 1):
        m = re.match(r"^(?:[0-9a-f]+ <)?(_Z\w+)>?:", line)
        if m:
            kernel, hist = m.group(1), []
            continue
        if re.match(r"^(?:[0-9a-f]+ <)?[.\w$]+>?:\s*(;.*)?$", line.strip()) or re.match(r"^<?\.?L\w+>?:", line.strip()):
            hist.append((ALL, None))                                # a branch target inside the kernel
            continue
        t = re.split(r";|//", line)[0].strip()
        if not t or t.startswith("."):
            continue
        parts = t.replace(",", " ").split()
        op = parts[0]
        if only and (kernel is None or only not in kernel):
            continue
        is_dpp = "_dpp" in op or any(k in t for k in DPP_MARKS)
        if is_dpp:
            n_dpp += 1
            src = regs(parts[2]) if len(parts) > 2 else set()
            wait = 0
            for pop, pdst in reversed(hist[-8:]):
                if pop.startswith("s_nop"):
                    wait += pdst + 1
                    continue
                if wait >= 5:
                    break
                hazard = None
                if pop == ALL:
                    hazard = "a branch target (unknown predecessor)" if wait < 2 else None      # (exec writes by VALU do not sit in front of labels here)
                elif pop.startswith("v_cmpx") and wait < 5:
                    hazard = f"{pop} (VALU write of EXEC, {wait} wait states)"
                elif pop.startswith("v_") and wait < 2 and pdst & src:
                    hazard = f"{pop} writing v{sorted(pdst & src)} {wait} wait states earlier"
                if hazard:
                    if verbose:
                        print(f"{path}:{ln}: {kernel}: `{t}` behind {hazard}")
                    bad += 1
                    break
                wait += 1
        if op.startswith("s_nop"):
            hist.append((op, int(parts[1], 0)))
        else:
            dst = regs(parts[1]) if len(parts) > 1 and op.startswith("v_") and not op.startswith("v_cmp") else set()
            hist.append((op, dst))
    return n_dpp, bad


def main(path, only=""):
    n_dpp, bad = scan(open(path), path, only)
    print(f"{n_dpp} DPP instructions checked, {bad} hazards")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""))
