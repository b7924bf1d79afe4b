#!/usr/bin/env python3
"""
Inline-assembly DPP instructions are invisible to the compiler's hazard recogniser: on gfx950 a VALU write of a VGPR needs
two wait states before a DPP operand (src0) reads it.  This scans the saved assembly of a translation unit
(hipcc --save-temps=obj ... -> build/<name>-hip-amdgcn-amd-amdhsa-gfx950.s) and reports every DPP instruction whose DPP
source was written by one of the two instructions in front of it without an s_nop covering the distance.
    python3 scripts/check_dpp_hazards.py rodeo_amd/csrc/build/solve_tilen-hip-amdgcn-amd-amdhsa-gfx950.s [kernel-name-substring]
"""
import re, sys


def regs(tok):
    m = re.fullmatch(r"-?\|?v\[(\d+):(\d+)\]\|?", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"-?\|?v(\d+)\|?", tok)
    return {int(m.group(1))} if m else set()


def main(path, only=""):
    bad = n_dpp = 0
    kernel, hist = None, []
    for ln, line in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel, hist = m.group(1), []
            continue
        t = line.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        parts = t.replace(",", " ").split()
        op = parts[0]
        if only and (kernel is None or only not in kernel):
            continue
        if "dpp" in op or "row_newbcast" in t or "quad_perm" in t or "row_shr" in t or "row_ror" in t or "row_bcast" in t:
            n_dpp += 1
            src = regs(parts[2]) if len(parts) > 2 else set()
            wait = 0
            for pop, pdst in reversed(hist[-4:]):
                if pop.startswith("s_nop"):
                    wait += pdst + 1
                    continue
                if wait >= 2:
                    break
                if pop.startswith("v_") and pdst & src:
                    print(f"{path}:{ln}: {kernel}: `{t}` reads v{sorted(pdst & src)} written {wait} wait states earlier by {pop}")
                    bad += 1
                    break
                wait += 1
        if op.startswith("s_nop"):
            hist.append((op, int(parts[1])))
        else:
            dst = regs(parts[1]) if len(parts) > 1 and op.startswith("v_") and not op.startswith("v_cmp") else set()
            hist.append((op, dst))
    print(f"{n_dpp} DPP instructions checked, {bad} hazards")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""))
