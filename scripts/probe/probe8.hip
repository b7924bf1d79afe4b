// Probe 8: cycles per forward step of fwd_tile3_kernel's instruction stream for different issue orders (one wave,
// synthetic operands).  The arithmetic is the kernel's; only the order of the statements differs between variants.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ double MF(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
template <int CTRL> __device__ __forceinline__ double dpp64(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double mirror(double x) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0x141, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0x141, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
#define SB __builtin_amdgcn_sched_barrier(0)
#define D_U    const double U = MF(M, Qt, 0.0);
#define D_B0   const double B0 = MF(Y0, M, 0.0);
#define D_MPT  const double MpT = MF(Qt0, U, RtT);
#define D_MP   const double Mp = MF(U, Qt, Rt);
#define D_DPP  const double v_own = dpp64<0xFF>(B0); const double v_oth = mirror(v_own);
#define D_POLY const double Xw = fma(fma(fma(c3, v_own, c2), v_own, c1), v_own, fma(co, v_oth, c0));
#define D_Z0   const double Z0 = MF(MpT, Xw, 0.0);
#define D_WS   const double WS = MF(Xw, Mp, 0.0);
#define D_S    const double S = MF(Z0, Xw, 0.0);
#define D_RCP  const double y0 = __builtin_amdgcn_rcp(S);
#define D_PW   const double PW = Z0 * WS;
#define D_E    const double e = fma(-S, y0, 1.0);
#define D_Y    const double y = fma(y0, fma(e, e, e), y0);
#define D_M    M = fma(-PW, y, Mp);
template <int V>
__global__ void __launch_bounds__(64) k(double* out, long long* cyc, int iters, double seed) {
    const int l = threadIdx.x, r = l >> 4, c = l & 3;
    const bool in3 = r < 3 && c < 3;
    const double Qt = in3 ? (r == c ? 1.0 : (c > r ? 0.0 : 1e-2)) : (r == 3 && c == 3 ? 1.0 : 0.0);
    const double Qt0 = in3 ? Qt : 0.0, Rt = in3 ? 1e-6 * (1 + r + c) : 0.0, RtT = Rt;
    const double Y0 = r < 3 ? (r == 0 ? 1.0 : 1e-2) : 0.0;
    const double c3 = r == 3 ? -2.0 : 0.0, c2 = r == 0 ? 3.0 : 0.0, c1 = 0.0, co = r == 3 ? -3.0 : 0.0, c0 = r == 1 ? 1.0 : (r == 0 ? -3.0 : 0.0);
    double M = r < 3 ? (c == 3 ? seed : (r == c ? 1e-3 : 0.0)) : (c == 3 ? 1.0 : 0.0);
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int n = 0; n < iters; ++n) {
        if constexpr (V == 0) { D_U D_B0 D_MP D_MPT D_DPP D_POLY D_WS D_Z0 D_S D_RCP D_PW D_E D_Y D_M }       // compiler's order
        if constexpr (V == 1) { D_B0 SB; D_U SB; D_DPP SB; D_MPT SB; D_POLY SB; D_Z0 SB; D_MP SB; D_S SB; D_WS SB; D_RCP D_PW D_E D_Y D_M SB; }
        if constexpr (V == 2) { D_U SB; D_B0 SB; D_MPT SB; D_MP SB; D_DPP D_POLY SB; D_Z0 SB; D_WS SB; D_S SB; D_RCP D_PW D_E D_Y D_M SB; }
        if constexpr (V == 3) { D_B0 SB; D_U SB; D_DPP D_POLY SB; D_MPT SB; D_MP SB; D_Z0 SB; D_WS SB; D_S SB; D_RCP D_PW D_E D_Y D_M SB; }
        if constexpr (V == 4) { D_B0 SB; D_U SB; D_DPP SB; D_MPT SB; D_POLY SB; D_Z0 SB; D_MP SB; D_S SB; D_WS SB; D_RCP SB; D_E SB; D_PW SB; D_Y SB; D_M SB; }
        if constexpr (V == 5) { D_B0 SB; D_U SB; D_DPP SB; D_MPT SB; D_POLY SB; D_Z0 SB; D_S SB; D_MP SB; D_WS SB; D_RCP D_PW D_E D_Y D_M SB; }
        if constexpr (V == 6) { D_B0 SB; D_U SB; D_DPP SB; D_MPT SB; D_POLY SB; D_Z0 SB; D_MP SB; D_S SB; D_RCP SB; D_WS SB; D_E SB; D_Y SB; D_PW SB; D_M SB; }
        if constexpr (V == 7) { D_B0 SB; D_U SB; D_MPT SB; D_DPP D_POLY SB; D_Z0 SB; D_MP SB; D_S SB; D_WS SB; D_RCP D_PW D_E D_Y D_M SB; }
        if constexpr (V == 8) { D_B0 SB; D_U SB; D_DPP SB; D_MPT SB; D_MP SB; D_POLY SB; D_Z0 SB; D_WS SB; D_S SB; D_RCP D_PW D_E D_Y D_M SB; }
        if constexpr (V == 10) { D_U D_B0 D_MPT SB; D_DPP D_POLY SB; D_MP D_Z0 D_S D_WS SB; D_RCP D_PW D_E D_Y D_M SB; }
        if constexpr (V == 11) { D_U SB; D_B0 SB; D_MPT SB; D_DPP D_POLY SB; D_MP SB; D_Z0 SB; D_S SB; D_WS SB; D_RCP D_PW D_E D_Y D_M SB; }
        if constexpr (V == 12) { D_U SB; D_B0 SB; D_MPT SB; D_DPP D_POLY SB; D_MP SB; D_Z0 SB; D_WS SB; D_S SB; D_RCP D_PW D_E D_Y D_M SB; }
        if constexpr (V == 13) { D_B0 SB; D_U SB; D_MPT SB; D_DPP D_POLY SB; D_MP SB; D_Z0 SB; D_WS SB; D_S SB; D_PW D_RCP D_E D_Y D_M SB; }
        if constexpr (V == 20 || V == 21) {   // order pinned with empty-asm dependencies instead of sched_barriers
#define TIE(a, b) asm("" : "+v"(a) : "v"(b))
            double U = MF(M, Qt, 0.0);
            TIE(M, U);                                    // B0 after U
            double B0 = MF(Y0, M, 0.0);
            TIE(U, B0);                                   // MpT after B0
            double MpT = MF(Qt0, U, RtT);
            TIE(B0, MpT);                                 // the VALU block after MpT
            const double v_own = dpp64<0xFF>(B0); const double v_oth = mirror(v_own);
            D_POLY
            TIE(U, Xw);                                   // Mp after the VALU block
            double Mp = MF(U, Qt, Rt);
            TIE(MpT, Mp);                                 // Z0 after Mp
            double Z0 = MF(MpT, Xw, 0.0);
            TIE(Mp, Z0);                                  // WS after Z0
            double WS = MF(Xw, Mp, 0.0);
            TIE(Z0, WS);                                  // S after WS
            double S = MF(Z0, Xw, 0.0);
            TIE(WS, S);                                   // PW after S
            const double PW = Z0 * WS;
            TIE(S, PW);                                   // rcp after PW
            const double y0 = __builtin_amdgcn_rcp(S);
            D_E D_Y D_M
            if (V == 21) out[64 + l] = M;                 // perturbation: a store per step
        }
        if constexpr (V == 9) {   // S's reciprocal first, M- and W~Sigma- inside the reciprocal chain
            D_B0 SB; D_U SB; D_DPP SB; D_MPT SB; D_POLY SB; D_Z0 SB; D_MP SB; D_S SB; D_WS SB; D_RCP SB; D_E SB; D_Y SB; D_PW SB; D_M SB; }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[l] = M;
    if (l == 0) cyc[0] = t1 - t0;
}
template <int V> void run(const char* name) {
    double* out; long long* cyc; CK(hipMalloc(&out, 128 * 8)); CK(hipMalloc(&cyc, 8));
    const int iters = 2000;
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL((k<V>), dim3(1), dim3(64), 0, 0, out, cyc, iters, 0.5); CK(hipDeviceSynchronize()); }
    long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    double ho[64]; CK(hipMemcpy(ho, out, 512, hipMemcpyDeviceToHost));
    printf("V%d %-100s %7.2f cycles/step   (M[3] = %.6g)\n", V, name, (double)h / iters, ho[3]);
}
int main() {
    run<0>("compiler order");
    run<1>("B0 U | dpp | MpT | poly | Z0 | Mp | S | WS | rcp PW e y M");
    run<2>("U B0 MpT Mp | dpp poly | Z0 WS S | rcp ...");
    run<3>("B0 U | dpp poly | MpT Mp Z0 WS S | rcp ...");
    run<4>("as V1, reciprocal chain pinned: rcp e PW y M");
    run<5>("as V1 with S before Mp");
    run<6>("as V1 with WS inside the reciprocal chain");
    run<7>("B0 U MpT | dpp poly | Z0 Mp S WS | rcp ...");
    run<8>("B0 U | dpp | MpT Mp | poly | Z0 WS S | rcp ...");
    run<9>("as V1, reciprocal chain pinned: rcp e y PW M");
    run<20>("U B0 MpT | dpp poly | Mp Z0 WS S | PW rcp e y M, pinned by empty-asm dependencies");
    run<21>("same + a global store per step");
    run<10>("{U B0 MpT} {dpp poly} {Mp Z0 S WS} {rcp..}: compiler free inside groups");
    run<11>("U|B0|MpT|{dpp poly}|Mp|Z0|S|WS|{rcp..}");
    run<12>("U|B0|MpT|{dpp poly}|Mp|Z0|WS|S|{rcp..}");
    run<13>("B0|U|MpT|{dpp poly}|Mp|Z0|WS|S|{PW rcp..}");
    return 0;
}
