// Handle / memory / timing / RCCL entry points of librodeo_kalman.so (see include/rodeo_kalman.h).
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <rccl/rccl.h>
#include "common.hpp"

namespace rk {
constexpr size_t RK_PROFILE_KEEP_MAX = 1 << 16;

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    (void)hipGetLastError();
    return RK_ERR_HIP;
}

LaunchTimer::LaunchTimer(rk_handle h_, const char* name) : h(h_), on(h_->profile), idx(0) {
    if (!on) return;
    // keep mode (rk_profile_enable(h, 2)) accumulates over calls: bounded, so that a caller that forgets to switch it off
    // neither grows the entry list nor the event pool without limit -- launches past the bound are simply not bracketed
    if (h->profile_keep && h->prof.size() >= RK_PROFILE_KEEP_MAX) { on = false; return; }
    auto get = [&]() -> hipEvent_t {
        if (h->event_used == h->event_pool.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            h->event_pool.push_back(e);
        }
        return h->event_pool[h->event_used++];
    };
    rk_profile_entry ent{name, get(), get()};
    if (!ent.start || !ent.stop) { on = false; return; }
    idx = h->prof.size();
    h->prof.push_back(ent);
    (void)hipEventRecord(ent.start, h->stream);
}

void LaunchTimer::stop() {
    if (on) (void)hipEventRecord(h->prof[idx].stop, h->stream);
}

}  // namespace rk

using namespace rk;

extern "C" {

const char* rk_last_error(void) { return g_err; }
const char* rk_version(void) { return "rodeo_kalman 0.2.0 (gfx950)"; }

int rk_device_count(int* n) {
    RK_REQUIRE(n, RK_ERR_INVALID, "rk_device_count: null pointer");
    hipError_t e = hipGetDeviceCount(n);
    if (e != hipSuccess) { *n = 0; return hip_fail(e, "hipGetDeviceCount", __FILE__, __LINE__); }
    return RK_OK;
}

}  // extern "C"  (closed for the primer kernel, reopened below)

namespace rk {
// Placement primer.  The forward filter kernels are one dependent chain per wave: their run time is the time of the
// SLOWEST wave, and two waves on one SIMD take 1.84x as long as one.  Behind a kernel of another workgroup shape (the
// backward kernels: 256 threads, 48 KB of LDS) the hardware dispatcher does not spread 1024 single-wave workgroups over
// the 1024 SIMDs: measured per-wave HW_ID (scripts/placement_probe.py, profiles/r02_placement_probe.jsonl) -- 45 to 75
// SIMDs get two waves and as many stay empty, 0.46 -> 0.76 ms at 2048 trajectories; behind a kernel of its OWN shape the
// placement is exact.  An empty launch of that shape in front (2-3 us) restores it: 0.455 ms, 1024 distinct SIMDs.
__global__ void placement_primer_kernel() {}

void launch_placement_primer(rk_handle h, dim3 grid, dim3 block) {
    // (only when the launch has more waves than half the chip's SIMDs: with fewer -- the benchmark's 512 -- no SIMD gets
    // two either way, measured at B = 1024, and the empty launch would only cost its 5 us)
    const unsigned waves = grid.x * ((block.x + 63) / 64);
    if (waves > 2u * (unsigned)h->prop.multiProcessorCount) hipLaunchKernelGGL(placement_primer_kernel, grid, block, 0, h->stream);
}
}  // namespace rk

extern "C" {

int rk_create(int device_id, rk_handle* out) {
    RK_REQUIRE(out, RK_ERR_INVALID, "rk_create: null handle pointer");
    *out = nullptr;
    int n = 0;
    RK_HIP(hipGetDeviceCount(&n));
    RK_REQUIRE(device_id >= 0 && device_id < n, RK_ERR_INVALID, "rk_create: device %d not in [0, %d)", device_id, n);
    RK_HIP(hipSetDevice(device_id));
    rk_handle h = new rk_handle_s();
    h->device = device_id;
    h->profile = false;
    h->profile_keep = false;
    h->event_used = 0;
    h->comm = nullptr;
    h->comm_scratch = nullptr;
    h->op_scratch = nullptr;
    h->op_scratch_bytes = 0;
    h->rank = 0;
    h->nranks = 1;
    RK_HIP(hipGetDeviceProperties(&h->prop, device_id));
    RK_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    RK_HIP(hipEventCreate(&h->t0));
    RK_HIP(hipEventCreate(&h->t1));
    *out = h;
    return RK_OK;
}

int rk_destroy(rk_handle h) {
    if (!h) return RK_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->comm) { (void)ncclCommDestroy((ncclComm_t)h->comm); h->comm = nullptr; }
    if (h->comm_scratch) { (void)hipFree(h->comm_scratch); h->comm_scratch = nullptr; }
    if (h->op_scratch) { (void)hipFree(h->op_scratch); h->op_scratch = nullptr; }
    for (auto e : h->event_pool) (void)hipEventDestroy(e);
    (void)hipEventDestroy(h->t0);
    (void)hipEventDestroy(h->t1);
    (void)hipStreamDestroy(h->stream);
    delete h;
    return RK_OK;
}

int rk_device_name(rk_handle h, char* buf, size_t buflen) {
    RK_REQUIRE(h && buf && buflen, RK_ERR_INVALID, "rk_device_name: bad arguments");
    snprintf(buf, buflen, "%s (%s, %d CUs)", h->prop.name, h->prop.gcnArchName, h->prop.multiProcessorCount);
    return RK_OK;
}

int rk_alloc(rk_handle h, size_t bytes, void** dptr) {
    RK_REQUIRE(h && dptr, RK_ERR_INVALID, "rk_alloc: bad arguments");
    *dptr = nullptr;
    if (bytes == 0) return RK_OK;
    RK_HIP(hipSetDevice(h->device));
    hipError_t e = hipMalloc(dptr, bytes);
    if (e == hipErrorOutOfMemory) {
        set_error("rk_alloc: out of device memory allocating %zu bytes", bytes);
        (void)hipGetLastError();
        return RK_ERR_NOMEM;
    }
    RK_HIP(e);
    return RK_OK;
}

int rk_free(rk_handle h, void* dptr) {
    RK_REQUIRE(h, RK_ERR_INVALID, "rk_free: null handle");
    if (!dptr) return RK_OK;
    RK_HIP(hipSetDevice(h->device));
    RK_HIP(hipStreamSynchronize(h->stream));
    RK_HIP(hipFree(dptr));
    return RK_OK;
}

int rk_memset(rk_handle h, void* dptr, int value, size_t bytes) {
    RK_REQUIRE(h && (dptr || !bytes), RK_ERR_INVALID, "rk_memset: bad arguments");
    if (bytes) RK_HIP(hipMemsetAsync(dptr, value, bytes, h->stream));
    return RK_OK;
}

int rk_h2d(rk_handle h, void* dst, const void* src, size_t bytes) {
    RK_REQUIRE(h && ((dst && src) || !bytes), RK_ERR_INVALID, "rk_h2d: bad arguments");
    if (!bytes) return RK_OK;
    RK_HIP(hipSetDevice(h->device));
    RK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    RK_HIP(hipStreamSynchronize(h->stream));
    return RK_OK;
}

int rk_d2h(rk_handle h, void* dst, const void* src, size_t bytes) {
    RK_REQUIRE(h && ((dst && src) || !bytes), RK_ERR_INVALID, "rk_d2h: bad arguments");
    if (!bytes) return RK_OK;
    RK_HIP(hipSetDevice(h->device));
    RK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    RK_HIP(hipStreamSynchronize(h->stream));
    return RK_OK;
}

int rk_d2d(rk_handle h, void* dst, const void* src, size_t bytes) {
    RK_REQUIRE(h && ((dst && src) || !bytes), RK_ERR_INVALID, "rk_d2d: bad arguments");
    if (!bytes) return RK_OK;
    RK_HIP(hipSetDevice(h->device));
    RK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, h->stream));
    return RK_OK;
}

int rk_sync(rk_handle h) {
    RK_REQUIRE(h, RK_ERR_INVALID, "rk_sync: null handle");
    RK_HIP(hipStreamSynchronize(h->stream));
    return RK_OK;
}

int rk_timer_start(rk_handle h) {
    RK_REQUIRE(h, RK_ERR_INVALID, "rk_timer_start: null handle");
    RK_HIP(hipEventRecord(h->t0, h->stream));
    return RK_OK;
}

int rk_timer_stop(rk_handle h, double* ms) {
    RK_REQUIRE(h && ms, RK_ERR_INVALID, "rk_timer_stop: bad arguments");
    RK_HIP(hipEventRecord(h->t1, h->stream));
    RK_HIP(hipEventSynchronize(h->t1));
    float f = 0.f;
    RK_HIP(hipEventElapsedTime(&f, h->t0, h->t1));
    *ms = (double)f;
    return RK_OK;
}

int rk_profile_enable(rk_handle h, int on) {
    RK_REQUIRE(h, RK_ERR_INVALID, "rk_profile_enable: null handle");
    h->profile = on != 0;
    h->profile_keep = on == 2;
    h->prof.clear();
    h->event_used = 0;
    return RK_OK;
}

int rk_profile_last(rk_handle h, int cap, const char** names, double* ms, int* n) {
    RK_REQUIRE(h && n, RK_ERR_INVALID, "rk_profile_last: bad arguments");
    RK_HIP(hipStreamSynchronize(h->stream));
    *n = (int)h->prof.size();
    for (int i = 0; i < *n && i < cap; ++i) {
        float f = 0.f;
        RK_HIP(hipEventElapsedTime(&f, h->prof[i].start, h->prof[i].stop));
        if (names) names[i] = h->prof[i].name;
        if (ms) ms[i] = (double)f;
    }
    return RK_OK;
}

// ---- RCCL -------------------------------------------------------------------------------------------------
#define RK_NCCL(call)                                                                         \
    do {                                                                                      \
        ncclResult_t r__ = (call);                                                            \
        if (r__ != ncclSuccess) {                                                             \
            set_error("RCCL error %d (%s) in %s", (int)r__, ncclGetErrorString(r__), #call);  \
            return RK_ERR_RCCL;                                                               \
        }                                                                                     \
    } while (0)

int rk_comm_uid(void* uid128) {
    RK_REQUIRE(uid128, RK_ERR_INVALID, "rk_comm_uid: null pointer");
    static_assert(sizeof(ncclUniqueId) == RK_COMM_UID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    RK_NCCL(ncclGetUniqueId(&id));
    memcpy(uid128, &id, sizeof(id));
    return RK_OK;
}

int rk_comm_init(rk_handle h, int rank, int nranks, const void* uid128) {
    RK_REQUIRE(h && uid128 && nranks >= 1 && rank >= 0 && rank < nranks, RK_ERR_INVALID, "rk_comm_init: bad arguments");
    RK_REQUIRE(!h->comm, RK_ERR_INVALID, "rk_comm_init: communicator already initialised");
    RK_HIP(hipSetDevice(h->device));
    ncclUniqueId id;
    memcpy(&id, uid128, sizeof(id));
    // rk_comm_barrier's word, on h->device -- allocated BEFORE the communicator exists: a failure here must not leave a
    // communicator behind that neither rk_comm_destroy nor rk_destroy would ever see
    if (!h->comm_scratch) RK_HIP(hipMalloc((void**)&h->comm_scratch, sizeof(double)));
    ncclComm_t comm;
    RK_NCCL(ncclCommInitRank(&comm, nranks, id, rank));
    h->comm = (void*)comm;
    h->rank = rank;
    h->nranks = nranks;
    return RK_OK;
}

int rk_comm_destroy(rk_handle h) {
    RK_REQUIRE(h, RK_ERR_INVALID, "rk_comm_destroy: null handle");
    if (h->comm) {
        RK_HIP(hipStreamSynchronize(h->stream));
        RK_NCCL(ncclCommDestroy((ncclComm_t)h->comm));
        h->comm = nullptr;
    }
    if (h->comm_scratch) {
        RK_HIP(hipSetDevice(h->device));
        RK_HIP(hipFree(h->comm_scratch));
        h->comm_scratch = nullptr;
    }
    return RK_OK;
}

int rk_allgather_f64(rk_handle h, const double* send, double* recv, size_t count) {
    RK_REQUIRE(h && send && recv, RK_ERR_INVALID, "rk_allgather_f64: bad arguments");
    if (h->nranks == 1 && !h->comm) {
        if (send != recv) RK_HIP(hipMemcpyAsync(recv, send, count * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        return RK_OK;
    }
    RK_REQUIRE(h->comm, RK_ERR_INVALID, "rk_allgather_f64: rk_comm_init has not been called");
    RK_NCCL(ncclAllGather(send, recv, count, ncclDouble, (ncclComm_t)h->comm, h->stream));
    return RK_OK;
}

int rk_allreduce_max_f64(rk_handle h, const double* send, double* recv, size_t count) {
    RK_REQUIRE(h && send && recv, RK_ERR_INVALID, "rk_allreduce_max_f64: bad arguments");
    if (h->nranks == 1 && !h->comm) {
        if (send != recv) RK_HIP(hipMemcpyAsync(recv, send, count * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        return RK_OK;
    }
    RK_REQUIRE(h->comm, RK_ERR_INVALID, "rk_allreduce_max_f64: rk_comm_init has not been called");
    RK_NCCL(ncclAllReduce(send, recv, count, ncclDouble, ncclMax, (ncclComm_t)h->comm, h->stream));
    return RK_OK;
}

int rk_comm_barrier(rk_handle h) {
    RK_REQUIRE(h, RK_ERR_INVALID, "rk_comm_barrier: null handle");
    if (!h->comm) { RK_HIP(hipStreamSynchronize(h->stream)); return RK_OK; }
    // a 1-element all-reduce on the handle's scratch word (allocated by rk_comm_init on the handle's device, freed by
    // rk_comm_destroy), then wait: every rank has reached this point
    double* const scratch = h->comm_scratch;
    RK_REQUIRE(scratch, RK_ERR_INVALID, "rk_comm_barrier: communicator without scratch word");
    RK_HIP(hipMemsetAsync(scratch, 0, sizeof(double), h->stream));
    RK_NCCL(ncclAllReduce(scratch, scratch, 1, ncclDouble, ncclSum, (ncclComm_t)h->comm, h->stream));
    RK_HIP(hipStreamSynchronize(h->stream));
    return RK_OK;
}

}  // extern "C"
