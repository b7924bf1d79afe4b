// Handle, error plumbing and launch helpers shared by the translation units of librodeo_kalman.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/rodeo_kalman.h"

struct rk_profile_entry {
    const char* name;
    hipEvent_t start, stop;
};

struct rk_handle_s {
    int device;
    hipStream_t stream;
    hipEvent_t t0, t1;
    bool profile;
    bool profile_keep;                        // rk_profile_enable(h, 2): entries accumulate over calls
    std::vector<rk_profile_entry> prof;       // launches of the last rk_solve_* call
    std::vector<hipEvent_t> event_pool;       // reused events
    size_t event_used;
    void* comm;                               // ncclComm_t (opaque here)
    double* comm_scratch;                     // one device word for rk_comm_barrier, owned with the communicator
    void* op_scratch;                         // grow-only device scratch of the large-block per-step operators
    size_t op_scratch_bytes;
    int rank, nranks;
    hipDeviceProp_t prop;
};

namespace rk {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define RK_HIP(call)                                                          \
    do {                                                                      \
        hipError_t e__ = (call);                                              \
        if (e__ != hipSuccess) return rk::hip_fail(e__, #call, __FILE__, __LINE__); \
    } while (0)

#define RK_REQUIRE(cond, code, ...)      \
    do {                                 \
        if (!(cond)) {                   \
            rk::set_error(__VA_ARGS__);  \
            return (code);               \
        }                                \
    } while (0)

// RAII-less profiling bracket around one kernel launch on the handle's stream
struct LaunchTimer {
    rk_handle h;
    bool on;
    size_t idx;
    LaunchTimer(rk_handle h_, const char* name);
    void stop();
};

inline int div_up(int a, int b) { return (a + b - 1) / b; }

// An empty kernel with the grid / block shape of the latency-bound forward kernel that follows it (api.hip).
void launch_placement_primer(rk_handle h, dim3 grid, dim3 block);

}  // namespace rk
