"""Generates probe17.hip: random topological orders of the 17 statements of fwd_tile4_kernel's Lorenz step (solve_tile4_kernels.hpp), pinned
with empty-asm dependencies like probe9 did for the n_deriv = 3 step (gen_probe9.py).  python3 gen_probe17.py [seed] [count]"""
import random, sys
deps = {"U": [], "VO": [], "MP": [], "SPT": ["U"], "SP": ["U"], "DPP": ["VO"], "AM": ["DPP"], "Z": ["SPT"], "WS": ["SP"],
        "YH": ["MP", "AM"], "SC": ["Z"], "RCP": ["SC"], "E": ["RCP"], "Y": ["E"], "K": ["Y", "Z"], "SN": ["K", "WS", "SP"],
        "MN": ["K", "YH", "MP", "SN"]}
stmt = {"U": "U = MF(S, Qt, 0.0); sink[(size_t)n * 128 + l] = S; sink[(size_t)n * 128 + 64 + l] = m;",
        "VO": "v_own = MF(Y0, m, 0.0);", "MP": "mp = MF(Qt, m, 0.0);", "SPT": "SpT = MF(Qt, U, RtT);", "SP": "Sp = MF(U, Qt, Rt);",
        "DPP": "n1 = dpp64<0x12C>(v_own); p1 = dpp64<0x124>(v_own); p2 = dpp64<0x128>(v_own);",
        "AM": "a_meas = fma(c4, p2 * p1, fma(c3, p1 * n1, fma(c2, p1, fma(c1, n1, c0 * v_own))));",
        "Z": "Z = MF(SpT, Xw, 0.0);", "WS": "WS = MF(Xw, Sp, 0.0);", "YH": "yhat = MF(Xw, mp, a_meas);", "SC": "Sc = MF(Z, Xw, 0.0);",
        "RCP": "y0 = __builtin_amdgcn_rcp(Sc);", "E": "e = fma(-Sc, y0, 1.0);", "Y": "y = fma(y0, fma(e, e, e), y0);",
        "K": "K = -Z * y;", "SN": "S = fma(K, WS, Sp);", "MN": "m = fma(K, yhat, mp);"}
inp = {"U": "S", "VO": "m", "MP": "m", "SPT": "U", "SP": "U", "DPP": "v_own", "AM": "n1", "Z": "SpT", "WS": "Sp", "YH": "mp", "SC": "Z",
       "RCP": "Sc", "E": "y0", "Y": "e", "K": "y", "SN": "WS", "MN": "yhat"}
outv = {"U": "U", "VO": "v_own", "MP": "mp", "SPT": "SpT", "SP": "Sp", "DPP": "p2", "AM": "a_meas", "Z": "Z", "WS": "WS", "YH": "yhat",
        "SC": "Sc", "RCP": "y0", "E": "e", "Y": "y", "K": "K", "SN": "S", "MN": "m"}
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
# V0: hipcc's own order of the shipped kernel (from the assembly), V1: the source order of the shipped kernel
orders = [["U", "VO", "SPT", "SP", "DPP", "AM", "Z", "MP", "WS", "SC", "YH", "RCP", "E", "Y", "K", "SN", "MN"],
          ["U", "VO", "MP", "SPT", "SP", "DPP", "AM", "Z", "WS", "YH", "SC", "RCP", "E", "Y", "K", "SN", "MN"]]
# hand-made orders by the "a VALU instruction behind an MFMA waits for it" model: few groups of back-to-back MFMAs
orders += [["U", "MP", "VO", "DPP", "AM", "SPT", "SP", "Z", "WS", "YH", "SC", "RCP", "E", "Y", "K", "SN", "MN"],
           ["U", "VO", "MP", "DPP", "AM", "SPT", "SP", "Z", "WS", "YH", "SC", "RCP", "E", "Y", "K", "SN", "MN"],
           ["U", "VO", "SPT", "MP", "SP", "Z", "DPP", "AM", "WS", "SC", "YH", "RCP", "E", "Y", "K", "SN", "MN"],
           ["U", "VO", "SPT", "SP", "MP", "Z", "SC", "DPP", "AM", "WS", "YH", "RCP", "E", "Y", "K", "SN", "MN"],
           ["U", "SPT", "VO", "Z", "MP", "SC", "SP", "DPP", "AM", "WS", "YH", "RCP", "E", "Y", "K", "SN", "MN"],
           ["U", "SPT", "VO", "Z", "SP", "SC", "MP", "WS", "RCP", "DPP", "AM", "E", "Y", "YH", "K", "SN", "MN"]]
seen = {tuple(o) for o in orders}
while len(orders) < count:
    done, o = set(), []
    while len(o) < len(deps):
        ready = [k for k in deps if k not in done and all(d in done for d in deps[k])]
        # bias: the covariance chain first half of the time (U, SPT, Z, SC as early as they are ready)
        pri = [k for k in ready if k in ("U", "SPT", "Z", "SC", "RCP", "E", "Y")]
        k = random.choice(pri) if pri and random.random() < 0.5 else random.choice(ready)
        o.append(k); done.add(k)
    if tuple(o) not in seen:
        seen.add(tuple(o)); orders.append(o)
def body(o, pinned=True):
    out = []
    for i, k in enumerate(o):
        if i and pinned: out.append('asm("" : "+v"(%s) : "v"(%s));' % (inp[k], outv[o[i - 1]]))
        out.append(stmt[k])
    return " ".join(out)
src = open("probe8.hip").read()
head = src[:src.index("#define SB")]
kern = '''template <int V>
__global__ void __launch_bounds__(64) k(double* out, long long* cyc, int iters, double seed, double* sink) {
    const int l = threadIdx.x, r = l >> 4, c = l & 3;
    const double Qt = r == c ? 1.0 : (c > r ? 0.0 : 1e-2), Rt = 1e-6 * (1 + r + c) + (r == c ? 1e-4 : 0.0), RtT = Rt;
    const double Y0 = r == 0 ? 1.0 : 1e-2, Xw = r == 1 ? 1.0 : (r == 0 ? 0.3 : 0.0);
    const double c0 = -0.1 * seed, c1 = 0.2, c2 = -0.05, c3 = 0.01, c4 = -0.02;
    double S = r == c ? 1e-3 : 1e-5, m = seed + 0.1 * r;
    double U, v_own, mp, SpT, Sp, n1, p1, p2, a_meas, Z, WS, yhat, Sc, y0, e, y, K;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int n = 0; n < iters; ++n) {
''' + "".join("        if constexpr (V == %d) { %s }\n" % (i, body(o, i != 1)) for i, o in enumerate(orders)) + '''    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[l] = S + m;
    if (l == 0) cyc[0] = t1 - t0;
}
template <int V> void run(const char* name) {
    const int iters = 2000;
    double *out, *sink; long long* cyc; CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&cyc, 8)); CK(hipMalloc(&sink, (size_t)iters * 128 * 8));
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL((k<V>), dim3(1), dim3(64), 0, 0, out, cyc, iters, 0.5, sink); CK(hipDeviceSynchronize()); }
    long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%7.2f  V%d  %s\\n", (double)h / iters, V, name);
    CK(hipFree(out)); CK(hipFree(cyc)); CK(hipFree(sink));
}
int main() {
''' + "".join('    run<%d>("%s");\n' % (i, " ".join(o) + (" (unpinned source order)" if i == 1 else "")) for i, o in enumerate(orders)) + "    return 0;\n}\n"
open("probe17.hip", "w").write(head.replace("Probe 8", "Probe 17 (generated by gen_probe17.py): the Lorenz step of fwd_tile4_kernel;").replace("fwd_tile3_kernel's", "the") + kern)
