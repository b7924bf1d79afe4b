// User-supplied ODE right-hand sides: hiprtc builds of the forward / interrogation kernels.
//
// rodeo's `ode_fun` is an arbitrary Python callable evaluated inside the scan (src/rodeo/solve.py:70-78) and
// differentiated by jax.jacfwd (src/rodeo/interrogate.py:76).  Here the time loop lives in one GPU kernel, so a new ODE
// arrives as HIP source for a small struct (the interface of csrc/rhs.hpp, or a scalar-generic `rhs` wrapped by
// rk::AutoJac of csrc/dual.hpp for the Jacobian) and the kernel templates of solve_small_kernels.hpp are
// instantiated for it at run time, once per (n_bstate, interrogation) actually used.
#include <hip/hiprtc.h>
#include <dlfcn.h>
#include <link.h>
#include <limits.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>
#include "common.hpp"
#include "solve_args.hpp"
#include "solve_dense_itg_kernels.hpp"
#include "build/embedded_sources.inc"

namespace rk {

struct UserRhs {
    std::string type_name;      // C++ type inside namespace rk, e.g. "MyOde" or "AutoJac<MyOde>"
    std::string source;
    int n_block, n_theta, n_bmeas;
};

static std::mutex g_mu;
static std::vector<UserRhs> g_rhs;                                   // id = RK_RHS_USER_BASE + index
struct JitEntry { hipModule_t mod; hipFunction_t fn; };
static std::map<std::tuple<int, int, int, int, int>, JitEntry> g_cache;   // (device, rhs, P, itg, kind)

// Are BOTH the libhiprtc behind hiprtcCompileProgram and the compiler library it drives (libamd_comgr) the ones under
// /opt/rocm (the toolchain of this build)?  Evaluated at the first compilation, i.e. after whatever the process has loaded
// by then: torch imported BEFORE this library brings its own libhiprtc, torch imported AFTER it (but before the first
// build) its own libamd_comgr, which the system's hiprtc then picks up -- either way the option must stay away.
// RK_JIT_BACKEND_OPTIONS = 0 / 1 overrides.
static bool under_opt_rocm(const char* path) {
    char real[PATH_MAX];
    const char* r = path && realpath(path, real) ? real : path;
    return r && strncmp(r, "/opt/rocm", 9) == 0;
}
static int find_comgr(struct dl_phdr_info* info, size_t, void* out) {
    if (info->dlpi_name && strstr(info->dlpi_name, "libamd_comgr")) {
        *(std::string*)out = info->dlpi_name;
        return 1;
    }
    return 0;
}
static bool hiprtc_takes_backend_options() {
    static const int v = [] {
        if (const char* e = getenv("RK_JIT_BACKEND_OPTIONS")) return atoi(e) != 0 ? 1 : 0;
        Dl_info info;
        if (!dladdr((void*)&hiprtcCompileProgram, &info) || !under_opt_rocm(info.dli_fname)) return 0;
        std::string comgr;
        dl_iterate_phdr(find_comgr, &comgr);
        if (comgr.empty()) {                              // not loaded yet: take the one next to this hiprtc, now
            std::string dir(info.dli_fname);
            dir.erase(dir.find_last_of('/') + 1);
            if (!dlopen((dir + "libamd_comgr.so.3").c_str(), RTLD_NOW | RTLD_GLOBAL)) return 0;
            dl_iterate_phdr(find_comgr, &comgr);
        }
        return !comgr.empty() && under_opt_rocm(comgr.c_str()) ? 1 : 0;
    }();
    return v != 0;
}

static std::string kernel_expr(const UserRhs& u, int P, int itg, int kind) {
    char buf[512];
    (void)u;   // the user's type is aliased to rk::UserRhsT inside the translation unit (it may be a template-id)
    if (kind == 2) snprintf(buf, sizeof buf, "rk::interrogate_kernel<rk::UserRhsT, %d, %d>", P, itg);
    else if (kind == 10) snprintf(buf, sizeof buf, "rk::interrogate_kernel_m<rk::UserRhsT, %d, %d>", P, itg);      // n_bmeas > 1, standalone
    else if (kind == 3) snprintf(buf, sizeof buf, "rk::fwd_tile3_kernel<rk::UserRhsT, %d>", itg);      // MFMA-tile forward, p = 3
    else if (kind == 4) snprintf(buf, sizeof buf, "rk::fwd_tile4_kernel<rk::UserRhsT, %d>", itg);      // MFMA-tile forward, p = 4
    else if (kind == 7 || kind == 8)                                                                        // n_bmeas > 1
        snprintf(buf, sizeof buf, "rk::fwd_kernel_m<rk::UserRhsT, %d, %d, %s>", P, itg, kind == 8 ? "true" : "false");
    else if (kind == 6) snprintf(buf, sizeof buf, "rk::fwd_sqrt_kernel<rk::UserRhsT, %d, %d>", P, itg);   // square-root filter
    else if (kind == 9) snprintf(buf, sizeof buf, "rk::dense_interrogate_kernel<rk::UserRhsT::Inner, %d, %d>", P, itg);   // dense path
    else if (kind == 5) snprintf(buf, sizeof buf, "rk::fwd_tilen_kernel<rk::UserRhsT, %d, %d>", itg, P); // blocked tiles, P here = NB
    else snprintf(buf, sizeof buf, "rk::fwd_kernel<rk::UserRhsT, %d, %d, %s>", P, itg, kind == 1 ? "true" : "false");
    return buf;
}

// compile one instantiation; returns code object in `code` and the mangled name in `lowered`
static int jit_compile(const UserRhs& u, int P, int itg, int kind, std::vector<char>& code, std::string& lowered) {
    const std::string src = std::string("#include \"solve_small_kernels.hpp\"\n#include \"dual.hpp\"\n"
                                        "#include \"solve_tile3_kernels.hpp\"\n#include \"solve_tile4_kernels.hpp\"\n"
                                        "#include \"solve_tilen_kernels.hpp\"\n#include \"solve_sqrt_kernels.hpp\"\n"
                                        "#include \"solve_small_m_kernels.hpp\"\n#include \"solve_dense_itg_kernels.hpp\"\nnamespace rk {\n") +
                            u.source + "\nusing UserRhsT = " + u.type_name + ";\n}  // namespace rk\n";
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "rk_user_rhs.hip", kJitNumHeaders, kJitHeaderSources, kJitHeaderNames) !=
        HIPRTC_SUCCESS) {
        set_error("hiprtcCreateProgram failed");
        return RK_ERR_HIP;
    }
    const std::string expr = kernel_expr(u, P, itg, kind);
    hiprtcAddNameExpression(prog, expr.c_str());
    // (Makefile: why aligned loops; MFMA results in VGPRs like the ahead-of-time build -- without it every MFMA result of the
    // tile kernels goes through an AGPR and two v_accvgpr_read, 12 extra instructions per step of the p = 3 forward kernel:
    // 0.79 against 0.67 ms on the headline shape.)  The -mllvm option exists in the ROCm compiler this library was built
    // with; an unknown -mllvm option makes LLVM call exit(), and in a process that loaded another ROCm first (import torch:
    // its wheel carries its own libhiprtc / comgr under the same soname) the calls below land in THAT one -- so the
    // option is passed only when the hiprtc that serves us is the system's.
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-falign-loops=64", "-mllvm", "-amdgpu-mfma-vgpr-form"};
    const hiprtcResult r = hiprtcCompileProgram(prog, hiprtc_takes_backend_options() ? 6 : 4, opts);
    if (r != HIPRTC_SUCCESS) {
        size_t ls = 0;
        hiprtcGetProgramLogSize(prog, &ls);
        std::string log(ls, '\0');
        if (ls) hiprtcGetProgramLog(prog, &log[0]);
        if (log.size() > 800) log.resize(800);
        set_error("hiprtc could not compile the user right-hand side '%s': %s", u.type_name.c_str(), log.c_str());
        hiprtcDestroyProgram(&prog);
        return RK_ERR_INVALID;
    }
    const char* low = nullptr;
    if (hiprtcGetLoweredName(prog, expr.c_str(), &low) != HIPRTC_SUCCESS || !low) {
        set_error("hiprtcGetLoweredName failed for %s", expr.c_str());
        hiprtcDestroyProgram(&prog);
        return RK_ERR_HIP;
    }
    lowered = low;
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    code.resize(cs);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    return RK_OK;
}

// compiled code objects, device independent: (rhs, P, itg, kind) -> (return code, code, lowered name).  A failed
// compilation is remembered too (the tile kernels are tried first and simply do not exist for some right-hand sides).
struct JitCode { int rc; std::vector<char> code; std::string lowered; std::string error; };
static std::map<std::tuple<int, int, int, int>, JitCode> g_code;

static const JitCode& jit_code_locked(int rhs_id, int P, int itg, int kind) {
    const auto key = std::make_tuple(rhs_id, P, itg, kind);
    auto it = g_code.find(key);
    if (it == g_code.end()) {
        JitCode c;
        c.rc = jit_compile(g_rhs[rhs_id - RK_RHS_USER_BASE], P, itg, kind, c.code, c.lowered);
        if (c.rc) c.error = rk_last_error();
        it = g_code.emplace(key, std::move(c)).first;
    }
    return it->second;
}

static int jit_get(rk_handle h, int rhs_id, int P, int itg, int kind, hipFunction_t* fn) {
    std::lock_guard<std::mutex> lk(g_mu);
    const int idx = rhs_id - RK_RHS_USER_BASE;
    RK_REQUIRE(idx >= 0 && idx < (int)g_rhs.size(), RK_ERR_INVALID, "unknown user rhs_id %d", rhs_id);
    const auto key = std::make_tuple(h->device, rhs_id, P, itg, kind);
    auto it = g_cache.find(key);
    if (it == g_cache.end()) {
        const JitCode& c = jit_code_locked(rhs_id, P, itg, kind);
        if (c.rc) { set_error("%s", c.error.c_str()); return c.rc; }
        JitEntry e;
        RK_HIP(hipModuleLoadData(&e.mod, c.code.data()));
        RK_HIP(hipModuleGetFunction(&e.fn, e.mod, c.lowered.c_str()));
        if (getenv("RK_JIT_VERBOSE")) {                   // resources of the kernel hiprtc built (a spilled dual copy of a big system shows here)
            int regs = 0, scratch = 0, lds = 0;
            (void)hipFuncGetAttribute(&regs, HIP_FUNC_ATTRIBUTE_NUM_REGS, e.fn);
            (void)hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, e.fn);
            (void)hipFuncGetAttribute(&lds, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, e.fn);
            fprintf(stderr, "[rk] jit kernel %s: %d registers, %d B scratch per lane, %d B LDS\n", c.lowered.c_str(), regs, scratch, lds);
        }
        it = g_cache.emplace(key, e).first;
    }
    *fn = it->second.fn;
    return RK_OK;
}

// Does the MFMA-tile forward kernel exist for this user right-hand side and configuration?  (It needs NDEP == 1 and a
// block count the tile kernels support; decided by compiling it once -- cached -- so that rk_solve_layout and the
// solve agree.)  which = 3 / 4 for n_bstate = 3 / 4.
bool user_tile_available(const rk_solve_cfg* c, int which) {
    std::lock_guard<std::mutex> lk(g_mu);
    const int idx = c->rhs_id - RK_RHS_USER_BASE;
    if (idx < 0 || idx >= (int)g_rhs.size()) return false;
    const int nb = g_rhs[idx].n_block;
    if (c->n_block != nb || c->n_bmeas != 1 || g_rhs[idx].n_bmeas != 1 || c->kalman_type != RK_KALMAN_STANDARD) return false;
    if (nb < 1 || nb > (which == 4 ? 4 : 64)) return false;      // p = 3 and blocked tiles: up to 64 blocks (4 per wave, LDS exchange); p = 4: one wave
    // which = 5: the blocked tile kernel (solve_tilen_kernels.hpp), instantiated per NB = 1 (p = 4) / 2 (p = 5 .. 8)
    const int pkey = which == 5 ? (c->n_bstate <= 4 ? 1 : 2) : which;
    const JitCode& jc = jit_code_locked(c->rhs_id, pkey, c->interrogate, which);
    if (jc.rc && getenv("RK_JIT_VERBOSE")) fprintf(stderr, "[rk] tile kernel not available for user rhs %d (p = %d): %s\n", c->rhs_id, which, jc.error.c_str());
    return jc.rc == RK_OK;
}

int user_forward_tile(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a, double* tiles, int which) {
    hipFunction_t fn;
    const int pkey = which == 5 ? (c->n_bstate <= 4 ? 1 : 2) : which;
    int rc = jit_get(h, c->rhs_id, pkey, c->interrogate, which, &fn);
    if (rc) return rc;
    SolveArgs args = a;
    int P = c->n_bstate;
    void* params[] = {&args, &tiles, &P};                           // (the p = 3 / p = 4 kernels take the first two)
    const int tpw = c->n_block == 3 ? 3 : 4;
    const int nw = c->n_block <= 4 ? 1 : (c->n_block + 3) / 4;     // waves per workgroup (TileWaves<D>)
    const int grid = nw == 1 ? div_up(a.B * c->n_block, tpw) : a.B;
    launch_placement_primer(h, dim3(grid), dim3(64 * nw));
    LaunchTimer t(h, which == 3 ? "fwd_tile3_kernel<user>" : (which == 4 ? "fwd_tile4_kernel<user>" : "fwd_tilen_kernel<user>"));
    RK_HIP(hipModuleLaunchKernel(fn, grid, 1, 1, 64 * nw, 1, 1, 0, h->stream, params, nullptr));
    t.stop();
    return RK_OK;
}

bool is_user_rhs(int rhs_id) { return rhs_id >= RK_RHS_USER_BASE; }

int user_forward_sqrt(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a) {
    {
        std::lock_guard<std::mutex> lk(g_mu);
        const int idx = c->rhs_id - RK_RHS_USER_BASE;
        RK_REQUIRE(idx >= 0 && idx < (int)g_rhs.size(), RK_ERR_INVALID, "unknown user rhs_id %d", c->rhs_id);
        RK_REQUIRE(c->n_block == g_rhs[idx].n_block && c->n_bmeas == 1 && g_rhs[idx].n_bmeas == 1, RK_ERR_UNSUPPORTED,
                   "square-root solver: user rhs %d needs n_block=%d, n_bmeas=1 (got %d, %d)", c->rhs_id, g_rhs[idx].n_block,
                   c->n_block, c->n_bmeas);
    }
    hipFunction_t fn;
    int rc = jit_get(h, c->rhs_id, c->n_bstate, c->interrogate, 6, &fn);
    if (rc) return rc;
    SolveArgs args = a;
    void* params[] = {&args};
    LaunchTimer t(h, "fwd_sqrt_kernel<user>");
    RK_HIP(hipModuleLaunchKernel(fn, div_up(a.B, 64 / c->n_block), 1, 1, 64, 1, 1, 0, h->stream, params, nullptr));     // 64 / D trajectories per wave
    t.stop();
    return RK_OK;
}

int user_rhs_check(const rk_solve_cfg* c) {
    std::lock_guard<std::mutex> lk(g_mu);
    const int idx = c->rhs_id - RK_RHS_USER_BASE;
    RK_REQUIRE(idx >= 0 && idx < (int)g_rhs.size(), RK_ERR_INVALID, "unknown user rhs_id %d", c->rhs_id);
    RK_REQUIRE(c->n_block == g_rhs[idx].n_block && c->n_bmeas == g_rhs[idx].n_bmeas, RK_ERR_UNSUPPORTED,
               "user rhs %d needs n_block=%d, n_bmeas=%d (got %d, %d)", c->rhs_id, g_rhs[idx].n_block, g_rhs[idx].n_bmeas,
               c->n_block, c->n_bmeas);
    const int pmax = c->n_bmeas > 1 ? 9 : 6;                     // (the backward kernels exist up to n_bstate = 9)
    RK_REQUIRE(c->n_bstate >= 2 && c->n_bstate <= pmax, RK_ERR_UNSUPPORTED, "lane-per-trajectory path supports n_bstate in [2, %d] "
               "here, got %d", pmax, c->n_bstate);
    RK_REQUIRE(c->n_bmeas <= c->n_bstate, RK_ERR_INVALID, "n_bmeas = %d exceeds n_bstate = %d", c->n_bmeas, c->n_bstate);
    return RK_OK;
}

static int user_n_bmeas(int rhs_id) {
    std::lock_guard<std::mutex> lk(g_mu);
    const int idx = rhs_id - RK_RHS_USER_BASE;
    return idx >= 0 && idx < (int)g_rhs.size() ? g_rhs[idx].n_bmeas : 1;
}

// ---- dense ("non-block") path: the interrogation kernel around the user's right-hand side (solve_dense_itg_kernels.hpp)
// Taken for one block with several measurements that the lane kernels do not serve (n_bstate > 9 or n_bmeas > 4).
bool user_dense_wanted(const rk_solve_cfg* c) {
    std::lock_guard<std::mutex> lk(g_mu);
    const int idx = c->rhs_id - RK_RHS_USER_BASE;
    if (idx < 0 || idx >= (int)g_rhs.size()) return false;
    const UserRhs& u = g_rhs[idx];
    return u.n_block == 1 && c->n_block == 1 && u.n_bmeas > 1 && c->n_bmeas == u.n_bmeas && (c->n_bstate > 9 || c->n_bmeas > 4);
}

int user_dense_interrogate(rk_handle h, const rk_solve_cfg* c, const DenseItgArgs& a) {
    hipFunction_t fn;
    int rc = jit_get(h, c->rhs_id, c->n_bstate, c->interrogate, 9, &fn);
    if (rc) return rc;
    DenseItgArgs args = a;
    void* params[] = {&args};
    RK_HIP(hipModuleLaunchKernel(fn, a.B, 1, 1, 256, 1, 1, 0, h->stream, params, nullptr));
    return RK_OK;
}

int user_forward(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a) {
    int rc = user_rhs_check(c);
    if (rc) return rc;
    hipFunction_t fn;
    const bool sp = (c->flags & RK_FLAG_STORE_PRED) != 0;
    rc = jit_get(h, c->rhs_id, c->n_bstate, c->interrogate, user_n_bmeas(c->rhs_id) > 1 ? (sp ? 8 : 7) : (sp ? 1 : 0), &fn);
    if (rc) return rc;
    SolveArgs args = a;
    void* params[] = {&args};
    LaunchTimer t(h, "fwd_kernel<user>");
    RK_HIP(hipModuleLaunchKernel(fn, div_up(a.B, 64), 1, 1, 64, 1, 1, 0, h->stream, params, nullptr));
    t.stop();
    return RK_OK;
}

int user_interrogate(rk_handle h, const rk_solve_cfg* c, const SolveArgs& a, double t, int step, const double* mp,
                     const double* vp, double* wm, double* mm_, double* vm) {
    int rc = user_rhs_check(c);
    if (rc) return rc;
    const bool multi = c->n_bmeas > 1;                     // several measurements per block: interrogate_kernel_m
    RK_REQUIRE(!multi || c->kalman_type == RK_KALMAN_STANDARD, RK_ERR_UNSUPPORTED,
               "rk_interrogate_batched: n_bmeas > 1 with kalman_type = square-root is fused into the solvers only");
    hipFunction_t fn;
    rc = jit_get(h, c->rhs_id, c->n_bstate, c->interrogate, multi ? 10 : 2, &fn);
    if (rc) return rc;
    SolveArgs args = a;
    int sqrt_mode = c->kalman_type == RK_KALMAN_SQRT ? 1 : 0;
    void* params[] = {&args, &t, &step, &mp, &vp, &wm, &mm_, &vm, &sqrt_mode};
    RK_HIP(hipModuleLaunchKernel(fn, div_up(a.B, 64), 1, 1, 64, 1, 1, 0, h->stream, params, nullptr));
    return RK_OK;
}

}  // namespace rk

using namespace rk;

extern "C" {

int rk_register_rhs_source_m(const char* type_name, const char* source, int32_t n_block, int32_t n_bmeas, int32_t n_theta,
                             int32_t* rhs_id) {
    RK_REQUIRE(type_name && source && rhs_id, RK_ERR_INVALID, "rk_register_rhs_source: null argument");
    // (n_bmeas > 4: one block holding all variables, served by the dense path -- solve_dense.hip)
    RK_REQUIRE(n_block >= 1 && n_block <= 64 && n_theta >= 0 && n_bmeas >= 1 && (n_bmeas <= 4 || (n_block == 1 && n_bmeas <= 256)),
               RK_ERR_INVALID, "rk_register_rhs_source: bad n_block / n_bmeas / n_theta");
    std::lock_guard<std::mutex> lk(g_mu);
    g_rhs.push_back(UserRhs{type_name, source, n_block, n_theta, n_bmeas});
    *rhs_id = RK_RHS_USER_BASE + (int)g_rhs.size() - 1;
    return RK_OK;
}

int rk_register_rhs_source(const char* type_name, const char* source, int32_t n_block, int32_t n_theta, int32_t* rhs_id) {
    return rk_register_rhs_source_m(type_name, source, n_block, 1, n_theta, rhs_id);
}

int rk_rhs_compile_check(int32_t rhs_id, int32_t n_bstate, int32_t interrogate) {
    UserRhs u;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        const int idx = rhs_id - RK_RHS_USER_BASE;
        RK_REQUIRE(idx >= 0 && idx < (int)g_rhs.size(), RK_ERR_INVALID, "unknown user rhs_id %d", rhs_id);
        u = g_rhs[idx];
    }
    std::vector<char> code;
    std::string lowered;
    const bool dense = u.n_block == 1 && u.n_bmeas > 1 && (n_bstate > 9 || u.n_bmeas > 4);      // (user_dense_wanted)
    int rc = jit_compile(u, n_bstate, interrogate, dense ? 9 : (u.n_bmeas > 1 ? 7 : 0), code, lowered);
    if (rc == RK_OK && !dense && u.n_bmeas > 1) rc = jit_compile(u, n_bstate, interrogate, 10, code, lowered);     // + the standalone interrogation
    return rc;
}

}  // extern "C"
