/*
 * rodeo_kalman.h -- C ABI of librodeo_kalman.so: MI355X (gfx950) implementation of rodeo's Kalman
 * filter / smoother time-stepping core, batched over independent trajectories.
 *
 * The reference (mlysy/rodeo v1.1.3) has no FFI: its boundaries are Python call conventions.  Each entry point
 * below cites the reference function whose work it performs (paths relative to the reference tree); the Python
 * package rodeo_amd/ binds these with ctypes and re-exposes rodeo's own names and keyword arguments
 * (INTEGRATION.md shows the binding).  Plain pointers and sizes only -- no torch / JAX types.
 *
 * Conventions
 * -----------
 *  - Every function returns an int status: RK_OK (0) or a negative RK_ERR_* code; rk_last_error() returns a
 *    thread-local message for the last failure.  The library never aborts.  Non-finite values propagate
 *    silently, as in the reference (no NaN checks anywhere in src/rodeo/).
 *  - The caller owns every buffer.  Device buffers come from rk_alloc / go to rk_free; the library keeps no
 *    pointer between calls.
 *  - One handle <-> one HIP device + one stream.  Calls on a handle are asynchronous w.r.t. the host unless
 *    stated otherwise; rk_sync(h) waits.  Calls on one handle must be serialised by the caller.
 *  - All arithmetic is IEEE fp64.
 *
 * Device data layout ("batch-minor")
 * ----------------------------------
 *  A per-trajectory array of logical shape S (e.g. (n_block, n_bstate, n_bstate)) for a batch of B trajectories
 *  is stored as the array of shape (S..., B): the element e of trajectory b sits at  ptr[e * B + b].
 *  Consecutive lanes of a wavefront therefore touch consecutive 8-byte words (512 B per wave per element).
 *  Time-indexed outputs have shape (n_steps + 1, S..., B).  Inputs flagged "shared" (batched == 0) have plain
 *  shape S and are broadcast to all trajectories.
 */
#ifndef RODEO_KALMAN_H
#define RODEO_KALMAN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------------------------------------- */
#define RK_OK                 0
#define RK_ERR_INVALID       -1   /* null pointer / bad shape / inconsistent arguments                      */
#define RK_ERR_UNSUPPORTED   -2   /* combination of (rhs, n_block, n_bstate, n_bmeas, interrogate, kalman)  */
#define RK_ERR_HIP           -3   /* a HIP runtime call failed                                              */
#define RK_ERR_RCCL          -4   /* an RCCL call failed                                                    */
#define RK_ERR_NOMEM         -5

/* ---- enumerations ------------------------------------------------------------------------------------- */
/* kalman_type string of src/rodeo/solve.py:138-143 */
#define RK_KALMAN_STANDARD    0   /* "standard"    -> src/rodeo/kalmantv/standard.py    */
#define RK_KALMAN_SQRT        1   /* "square-root" -> src/rodeo/kalmantv/square_root.py */

/* interrogate callable of src/rodeo/solve.py:70-78, recognised by identity on the Python side */
#define RK_INTERROGATE_RODEO      0   /* src/rodeo/interrogate.py:87-115 */
#define RK_INTERROGATE_SCHOBER    1   /* src/rodeo/interrogate.py:50-62  */
#define RK_INTERROGATE_KRAMER     2   /* src/rodeo/interrogate.py:65-84  */
#define RK_INTERROGATE_CHKREBTII  3   /* src/rodeo/interrogate.py:13-47  */

/* built-in ODE right-hand sides (device code for the `ode_fun` callable of src/rodeo/solve.py:218) */
#define RK_RHS_FITZHUGH_NAGUMO  1   /* README.md:92-99; theta = (a, b, c); n_block = 2, n_bmeas = 1           */
#define RK_RHS_LORENZ63         2   /* docs/examples/lorenz.md:85-92; theta = (rho, sigma, beta); n_block = 3 */
#define RK_RHS_HIGHER_ORDER     3   /* docs/examples/higher_order.md:47-59; x'' = sin 2t - x; n_block = 1     */
#define RK_RHS_LINEAR_DENSE     4   /* x' = A x as one dense block (prior/indep_init.py); theta = A row-major */
#define RK_RHS_USER_BASE        1000 /* ids returned by rk_register_rhs_source start here                     */

/* flags for rk_solve_cfg.flags */
#define RK_FLAG_STORE_PRED   1    /* also write the predicted moments (solve.py:93-96 state_pred)            */
#define RK_FLAG_BATCH_MINOR  2    /* force the batch-minor kernels/layout even where the tile path exists    */

/* which call a layout query refers to */
#define RK_MODE_FILTER  0
#define RK_MODE_MV      1
#define RK_MODE_SIM     2

/* output layouts of the fused solvers (see rk_solve_layout) */
#define RK_LAYOUT_BATCH_MINOR  0  /* mean_state (N+1, d, p, B), var_state (N+1, d, p, p, B)                  */
#define RK_LAYOUT_TILE3        1  /* n_bstate = 3 only: var_state holds (N+1, B, d, 3, 4) doubles, row i of a
                                     block = [Sigma[i][0..2], mu[i]]; mean_state is not used (may be NULL)  */
#define RK_LAYOUT_TILE4        3  /* n_bstate = 4 only: var_state holds (N+1, B, d, 20) doubles per block:
                                     [Sigma row-major (16) | mu (4)]; mean_state is not used (may be NULL)   */
#define RK_LAYOUT_TILEP        4  /* blocked tile path, n_bstate = 5 .. 8: var_state holds (N+1, B, d, p*p + p) doubles per
                                     block: [Sigma row-major (p*p) | mu (p)]; mean_state is not used (may be NULL)
                                     (RK_LAYOUT_TILE4 is this format at p = 4)                                   */
#define RK_LAYOUT_TRAJ_MAJOR   2  /* dense large-block path: the reference's own layout with a leading batch
                                     axis, mean_state (B, N+1, d, p), var_state (B, N+1, d, p, p).  With
                                     kalman_type = RK_KALMAN_SQRT var_state (and var_pred) hold the LOWER factors
                                     L_n row-major, exact zeros above the diagonal (what src/rodeo/solve.py
                                     returns in that mode)                                                    */

typedef struct rk_handle_s* rk_handle;

/* ---- handle, memory, synchronisation ------------------------------------------------------------------ */
int         rk_create(int device_id, rk_handle* h);
int         rk_destroy(rk_handle h);
const char* rk_last_error(void);
const char* rk_version(void);
int         rk_device_count(int* n);
int         rk_device_name(rk_handle h, char* buf, size_t buflen);
int         rk_alloc(rk_handle h, size_t bytes, void** dptr);
int         rk_free(rk_handle h, void* dptr);
int         rk_memset(rk_handle h, void* dptr, int value, size_t bytes);
int         rk_h2d(rk_handle h, void* dst_dev, const void* src_host, size_t bytes);   /* synchronous */
int         rk_d2h(rk_handle h, void* dst_host, const void* src_dev, size_t bytes);   /* synchronous */
int         rk_d2d(rk_handle h, void* dst_dev, const void* src_dev, size_t bytes);    /* asynchronous, on the handle's stream */
int         rk_sync(rk_handle h);

/* HIP-event timing on the handle's stream (the stream every kernel of this library is launched on). */
int         rk_timer_start(rk_handle h);
int         rk_timer_stop(rk_handle h, double* elapsed_ms);     /* records, synchronises, returns elapsed   */
/* Per-kernel device time of the last rk_solve_* call on this handle, from HIP events bracketing each launch.
 * Enabled with rk_profile_enable(h, 1); names/ms arrays of capacity cap are filled, *n = number of launches.
 * rk_profile_enable(h, 2) keeps the entries of every call since it was enabled (rk_profile_last then returns all of
 * them, in launch order) instead of only those of the last solve; rk_profile_enable(h, 0) switches it off. */
int         rk_profile_enable(rk_handle h, int on);
int         rk_profile_last(rk_handle h, int cap, const char** names, double* ms, int* n);

/* ---- user-supplied ODE right-hand sides ------------------------------------------------------------------------
 * The `ode_fun(X, t, **params)` callable of src/rodeo/solve.py:218 (and its jax.jacfwd, src/rodeo/interrogate.py:76)
 * for an ODE that is not built in: HIP source defining, inside namespace rk, a struct with the interface of
 * rodeo_amd/csrc/rhs.hpp (D, NTHETA, NDEP, f<P>, fjac<P>) -- or a scalar-generic `rhs<T, P>` used through
 * "AutoJac<Name>" (forward-mode duals, rodeo_amd/csrc/dual.hpp).  type_name is that C++ type.  The kernels are
 * compiled with hiprtc on first use per (n_bstate, interrogation); n_bmeas = 1, kalman_type = standard.
 * rk_rhs_compile_check compiles without loading (needs no GPU) and reports compiler errors via rk_last_error().   */
int rk_register_rhs_source(const char* type_name, const char* source, int32_t n_block, int32_t n_theta,
                           int32_t* rhs_id);
/* the same for a right-hand side with n_bmeas > 1 measurements per block (src/rodeo/solve.py:48-51: ode_weight (n_block,
 * n_bmeas, n_bstate)), e.g. the reference's "non-block" form of a small system (prior/indep_init.py, examples/solve_nb.py):
 * `type_name` then names a type with the interface of csrc/solve_small_m_kernels.hpp (rk::AutoJacM<...> around a
 * scalar-generic rhs writing out[D][M]); such right-hand sides run on the lane-per-trajectory kernels (n_bstate <= 9,
 * n_bmeas <= 4) or, for ONE block beyond that (n_bmeas up to 256, n_bstate up to 768), on the dense MFMA path with its
 * interrogation kernel built around them (csrc/solve_dense_itg_kernels.hpp; standard filter, all solvers and interrogations). */
int rk_register_rhs_source_m(const char* type_name, const char* source, int32_t n_block, int32_t n_bmeas, int32_t n_theta,
                             int32_t* rhs_id);
int rk_rhs_compile_check(int32_t rhs_id, int32_t n_bstate, int32_t interrogate);

/* ---- whole-solve boundary ---------------------------------------------------------------------------------
 * Mirrors  solve_mv / solve_sim (key, ode_fun, ode_weight, ode_init, t_min, t_max, n_steps, interrogate,
 *                                prior_pars, kalman_type, **params)         src/rodeo/solve.py:125-129,208-212 */
typedef struct {
    int32_t  n_traj;        /* B: trajectories held by THIS handle/rank                                     */
    int32_t  n_steps;       /* N (solve.py:223)                                                             */
    int32_t  n_block;       /* d = ode_weight.shape[0]                                                      */
    int32_t  n_bstate;      /* p = ode_weight.shape[2]                                                      */
    int32_t  n_bmeas;       /* m = ode_weight.shape[1]                                                      */
    int32_t  rhs_id;        /* RK_RHS_*                                                                     */
    int32_t  interrogate;   /* RK_INTERROGATE_*                                                             */
    int32_t  kalman_type;   /* RK_KALMAN_*                                                                  */
    int32_t  n_theta;       /* doubles of ODE parameters per trajectory                                     */
    int32_t  flags;         /* RK_FLAG_*                                                                    */
    double   t_min, t_max;  /* solve.py:221-222; step n is interrogated at t_min + (t_max-t_min)(n+1)/N     */
    uint64_t seed;          /* Philox key (stands in for the jax PRNG `key`; see oracle/counter_rng.py)     */
    uint64_t traj_offset;   /* global index of local trajectory 0: makes draws independent of the sharding  */
} rk_solve_cfg;

typedef struct {
    const double* ode_weight;    int32_t ode_weight_batched;    /* W  (d, m, p [,B])   solve.py:219         */
    const double* ode_init;      int32_t ode_init_batched;      /* x0 (d, p [,B])      solve.py:220         */
    const double* prior_weight;  int32_t prior_weight_batched;  /* Q  (d, p, p [,B])   prior_pars[0]        */
    const double* prior_var;     int32_t prior_var_batched;     /* R  (d, p, p [,B])   prior_pars[1]        */
    const double* theta;         int32_t theta_batched;         /* (n_theta [,B])      **params, packed     */
} rk_solve_in;

typedef struct {
    /* filtered moments; after rk_solve_mv they hold the SMOOTHED moments (smoothing is done in place)      */
    double* mean_state;     /* (N+1, d, p, B)                                                               */
    double* var_state;      /* (N+1, d, p, p, B)                                                            */
    /* predicted moments, only written when RK_FLAG_STORE_PRED is set (may be NULL otherwise); with          */
    /* RK_LAYOUT_TRAJ_MAJOR (the dense path) they are trajectory-major like the state: (B, N+1, p[, p])     */
    double* mean_pred;      /* (N+1, d, p, B)                                                               */
    double* var_pred;       /* (N+1, d, p, p, B)                                                            */
    /* rk_solve_sim: the sample path; mean_state / var_state are then the filter workspace                  */
    double* x_state;        /* (N+1, d, p, B)                                                               */
    /* scratch for the dense large-block path: rk_solve_workspace_bytes() bytes (NULL if that returns 0)     */
    void*   workspace;
    /* size of `workspace` in bytes: a call whose configuration needs more is refused (RK_ERR_INVALID) before */
    /* any kernel is launched                                                                                */
    size_t  workspace_bytes;
} rk_solve_out;

/* Output layout the fused kernels use for this configuration and call (RK_MODE_*).  The MFMA-tile kernels
 * (n_bstate = 3 or 4, n_bmeas = 1, solve_mv / filter, kramer | schober | rodeo) write RK_LAYOUT_TILE3 / _TILE4, the dense
 * path RK_LAYOUT_TRAJ_MAJOR; everything else
 * RK_LAYOUT_BATCH_MINOR.  The caller sizes and interprets out->mean_state / var_state accordingly.             */
int rk_solve_layout(const rk_solve_cfg* cfg, int32_t mode, int32_t* layout);
/* bytes of the output arrays for a configuration in the given layout (any pointer may be NULL);
 * RK_LAYOUT_TILE3: *mean_bytes = 0, *var_bytes = ((N+1) * B * d * 12 + 128 * ceil(B * d / 8)) * 8 -- the buffer MUST
 * have this size: behind the tiles sits a scratch tail (64 doubles per wave) that lanes without an output slot use.
 * RK_LAYOUT_TILE4 likewise (20 doubles per tile + 128 per wave); always size tile buffers with this function.   */
int rk_solve_sizes(const rk_solve_cfg* cfg, int32_t layout, size_t* mean_bytes, size_t* var_bytes);

/* bytes of device scratch (rk_solve_out.workspace) the configuration needs; 0 for the standard small-block kernels.  Dense path
 * with RK_KALMAN_SQRT and mode != RK_MODE_FILTER: per-trajectory scratch PLUS the predicted factors of every step,
 * (B, N+1, p, p) doubles (src/rodeo/kalmantv/square_root.py:170-175 solves with them in the backward pass), unless
 * RK_FLAG_STORE_PRED hands the caller's var_pred to the library for that purpose.
 * Blocked tile path (n_bstate 4..8): the hand-off records of the two-kernel backward passes (solve_sim; solve_mv only with
 * RK_TILEN_BWD=split -- the default one-kernel solve_mv needs none of it, the size is still reported for that switch).
 * Small blocks with RK_KALMAN_SQRT, rk_solve_mv / rk_solve_sim: (N, d, 3p^2 + p | p^2 + 2p, B) doubles of records for the two-kernel backward pass;
 * OPTIONAL -- with workspace = NULL (or too small) the one-kernel backward pass runs, same results, about twice the time. */
int rk_solve_workspace_bytes(const rk_solve_cfg* cfg, int32_t mode, size_t* bytes);

/* forward pass only: src/rodeo/solve.py:31-122 (_solve_filter).  out->mean_state/var_state <- filtered. */
int rk_solve_filter(rk_handle h, const rk_solve_cfg* cfg, const rk_solve_in* in, const rk_solve_out* out);
/* forward + backward mean/variance smoother: src/rodeo/solve.py:208-302 (solve_mv).                     */
int rk_solve_mv(rk_handle h, const rk_solve_cfg* cfg, const rk_solve_in* in, const rk_solve_out* out);
/* forward + backward sampler: src/rodeo/solve.py:125-205 (solve_sim).  out->x_state <- one draw per traj. */
int rk_solve_sim(rk_handle h, const rk_solve_cfg* cfg, const rk_solve_in* in, const rk_solve_out* out);

/* Gather x[obs_ind[k], :, 0] and reduce the Gaussian observation log-likelihood + N(0, prior_sd^2) log-prior
 * per trajectory: the tail of the user log-posterior of docs/examples/parameter.md:188-210,331-354 (and of
 * src/rodeo/inference/basic.py:47-62 with a Gaussian obs_loglik).
 *   state: the solver output holding the path -- layout RK_LAYOUT_BATCH_MINOR: x_state / mean_state (N+1, d, p, B);
 *          layout RK_LAYOUT_TILE3 / RK_LAYOUT_TILE4: the tile buffer (the mean is column 3 / entries 16..19);
 *   obs (n_obs, d) row-major on device; obs_ind (n_obs) int32 on device (clamped to [0, N]);
 *   upars (n_prior, B) batch-minor or NULL (no prior term); out logpost (B).                              */
int rk_gauss_obs_logpost(rk_handle h, int32_t n_traj, int32_t n_steps, int32_t n_block, int32_t n_bstate,
                         int32_t layout, const double* state, const double* obs, const int32_t* obs_ind,
                         int32_t n_obs, double noise_sd, const double* upars, int32_t n_prior, double prior_sd,
                         double* logpost);

/* rk_solve_sim followed by rk_gauss_obs_logpost on its sample path, as ONE call: the body of the user-level log-posterior of
 * docs/examples/parameter.md:331-354 (constrain -> solve_sim with interrogate_chkrebtii -> Gaussian observation
 * log-likelihood at searchsorted indices + normal log-prior), what a pseudo-marginal sampler evaluates once per proposal
 * (src/rodeo/inference/pseudo_marginal.py:135-149).  On the n_bstate = 3 tile path with n_block in {1, 2, 4} the backward
 * sampler reduces the log-posterior itself (no second launch) and out->x_state may be NULL: then no path is stored at
 * all and logpost (B) is the only result.  Any other configuration runs the two kernels back to back and needs x_state.
 * upars / obs / obs_ind must be on the device BEFORE the call (upload them before launching, not between the launches).  */
int rk_solve_sim_logpost(rk_handle h, const rk_solve_cfg* cfg, const rk_solve_in* in, const rk_solve_out* out,
                         const double* obs, const int32_t* obs_ind, int32_t n_obs, double noise_sd,
                         const double* upars, int32_t n_prior, double prior_sd, double* logpost);

/* Fenrir's backward pass (src/rodeo/inference/fenrir.py:86-259; the forward pass is rk_solve_filter with the same
 * cfg / in, fenrir.py:304-313): log p(y_{0:M} | Z_{1:N}) per trajectory from the filter's output in `out` -- either
 * the RK_LAYOUT_TILE3 tiles (no flags; predicted moments are re-evaluated on the fly; n_bobs = 1) or the RK_LAYOUT_TILE4 /
 * RK_LAYOUT_TILEP records (no flags; n_bstate 4..8, any n_bobs) when rk_solve_layout reports that layout for
 * RK_MODE_FILTER, or the batch-minor filtered and predicted moments of a call with
 * RK_FLAG_STORE_PRED | RK_FLAG_BATCH_MINOR (n_bstate 2..6; required at n_bstate = 3 when n_bobs > 1).  Observations (fenrir.py:106-122), n_bobs = 1..3
 * per block: obs (n_obs, d, n_bobs), obs_weight (n_obs, d, n_bobs, p), obs_var (n_obs, d, n_bobs, n_bobs) row-major on
 * device, shared by all trajectories; obs_ind (n_obs) = searchsorted(sim_times, obs_times), ascending.  logdens (B) is
 * overwritten.  The log-density follows src/rodeo/utils.py:60-78 (eigendecomposition of the forecast variance;
 * eigenvalues with |w| <= 1e-8 contribute nothing).
 * kalman_type = RK_KALMAN_SQRT (fenrir.py:292-296): `out` holds the batch-minor filtered means and FACTORS of the
 * square-root rk_solve_filter (no predictions needed: they are re-evaluated), prior_var and obs_var are lower
 * factors; square_root.forecast squares its factor (square_root.py:343-344), so the value is the same log-likelihood.    */
int rk_fenrir_backward(rk_handle h, const rk_solve_cfg* cfg, const rk_solve_in* in, const rk_solve_out* out,
                       const double* obs, const double* obs_weight, const double* obs_var, const int32_t* obs_ind,
                       int32_t n_obs, int32_t n_bobs, double* logdens);

/* Fenrir's data-adaptive solver (src/rodeo/inference/fenrir.py:333-457 `_smooth_mv`, `solve_mv`): mean and variance of
 * p(X_{0:N} | Z_{1:N}, Y_{0:M}).  Input as for rk_fenrir_backward with RK_FLAG_STORE_PRED | RK_FLAG_BATCH_MINOR; the
 * backward filter's moments are kept in `workspace` (rk_fenrir_workspace_bytes), the smoothing pass overwrites
 * out->mean_state / out->var_state (batch-minor (N+1, d, p [,p], B)) with the result.  RK_KALMAN_SQRT (fenrir.py:421-426):
 * input as for the square-root rk_fenrir_backward, the result's var_state holds factors.                              */
int rk_fenrir_workspace_bytes(const rk_solve_cfg* cfg, size_t* bytes);
int rk_fenrir_solve_mv(rk_handle h, const rk_solve_cfg* cfg, const rk_solve_in* in, const rk_solve_out* out,
                       const double* obs, const double* obs_weight, const double* obs_var, const int32_t* obs_ind,
                       int32_t n_obs, int32_t n_bobs, void* workspace);
/* The same on the records of the blocked-tile forward pass (kalman_type standard, n_bstate 4 .. 8, rk_solve_filter WITHOUT
 * RK_FLAG_STORE_PRED | RK_FLAG_BATCH_MINOR: out->var_state = the records): predicted moments are re-evaluated, nothing but the
 * filtered records is read; the result goes to mean_out (N+1, d, p, B) and var_out (N+1, d, p, p, B), batch-minor.
 * Replaces the same reference lines as rk_fenrir_solve_mv (src/rodeo/inference/fenrir.py:333-457).                        */
int rk_fenrir_solve_mv_tiles(rk_handle h, const rk_solve_cfg* cfg, const rk_solve_in* in, const rk_solve_out* out,
                             const double* obs, const double* obs_weight, const double* obs_var, const int32_t* obs_ind,
                             int32_t n_obs, int32_t n_bobs, void* workspace, double* mean_out, double* var_out);

/* ---- per-step operator boundary -------------------------------------------------------------------------
 * Batched versions of the nine functions of src/rodeo/kalmantv/standard.py (kalman_type = RK_KALMAN_STANDARD)
 * and src/rodeo/kalmantv/square_root.py (RK_KALMAN_SQRT).  n = batch size (the reference's vmap axis);
 * every array is batch-minor: vectors (n_state, n), matrices (rows, cols, n).  A NULL optional input means 0.
 * Names of the arguments are the reference's keyword names.  Any n_state up to 768: one lane per item up to 16, one
 * 512-thread workgroup per item on the dense solver's GEMM / LU / QR blocks beyond (the reference's operators are
 * size-agnostic, standard.py:31-60); the large-block path keeps a grow-only device scratch on the handle.              */
typedef struct {
    int32_t n;            /* batch                                                                         */
    int32_t n_state;
    int32_t n_meas;
    int32_t kalman_type;
} rk_op_cfg;

/* standard.py:31-60 / square_root.py:30-58 */
int rk_kalman_predict_batched(rk_handle h, const rk_op_cfg* c,
        const double* mean_state_past, const double* var_state_past, const double* mean_state,
        const double* wgt_state, const double* var_state,
        double* mean_state_pred, double* var_state_pred);
/* standard.py:63-103 / square_root.py:61-101 */
int rk_kalman_update_batched(rk_handle h, const rk_op_cfg* c,
        const double* mean_state_pred, const double* var_state_pred, const double* x_meas,
        const double* mean_meas, const double* wgt_meas, const double* var_meas,
        double* mean_state_filt, double* var_state_filt);
/* standard.py:106-157 / square_root.py:104-155 */
int rk_kalman_filter_batched(rk_handle h, const rk_op_cfg* c,
        const double* mean_state_past, const double* var_state_past, const double* mean_state,
        const double* wgt_state, const double* var_state, const double* x_meas,
        const double* mean_meas, const double* wgt_meas, const double* var_meas,
        double* mean_state_pred, double* var_state_pred, double* mean_state_filt, double* var_state_filt);
/* standard.py:180-217 / square_root.py:178-219 (var_state = R factor is required for RK_KALMAN_SQRT) */
int rk_kalman_smooth_mv_batched(rk_handle h, const rk_op_cfg* c,
        const double* mean_state_next, const double* var_state_next,
        const double* mean_state_filt, const double* var_state_filt,
        const double* mean_state_pred, const double* var_state_pred,
        const double* wgt_state, const double* var_state,
        double* mean_state_smooth, double* var_state_smooth);
/* standard.py:220-255 / square_root.py:222-261 */
int rk_kalman_smooth_sim_batched(rk_handle h, const rk_op_cfg* c,
        const double* x_state_next,
        const double* mean_state_filt, const double* var_state_filt,
        const double* mean_state_pred, const double* var_state_pred,
        const double* wgt_state, const double* var_state,
        double* mean_state_sim, double* var_state_sim);
/* standard.py:258-305 / square_root.py:264-314 */
int rk_kalman_smooth_batched(rk_handle h, const rk_op_cfg* c,
        const double* x_state_next, const double* mean_state_next, const double* var_state_next,
        const double* mean_state_filt, const double* var_state_filt,
        const double* mean_state_pred, const double* var_state_pred,
        const double* wgt_state, const double* var_state,
        double* mean_state_sim, double* var_state_sim, double* mean_state_smooth, double* var_state_smooth);
/* standard.py:308-336 / square_root.py:317-345 */
int rk_kalman_forecast_batched(rk_handle h, const rk_op_cfg* c,
        const double* mean_state_pred, const double* var_state_pred,
        const double* mean_meas, const double* wgt_meas, const double* var_meas,
        double* mean_fore, double* var_fore);
/* standard.py:339-371 / square_root.py:348-385 */
int rk_kalman_smooth_cond_batched(rk_handle h, const rk_op_cfg* c,
        const double* mean_state_filt, const double* var_state_filt,
        const double* mean_state_pred, const double* var_state_pred,
        const double* wgt_state, const double* var_state,
        double* wgt_state_cond, double* mean_state_cond, double* var_state_cond);

/* ---- interrogation boundary ------------------------------------------------------------------------------
 * One interrogation for a batch of predicted states: src/rodeo/interrogate.py (all four variants; the variant,
 * rhs, dims, seed and traj_offset are taken from cfg; `step` addresses the Philox stream for chkrebtii).
 *   mean_state_pred (d, p, B), var_state_pred (d, p, p, B), ode_weight/theta as in rk_solve_in;
 *   outputs wgt_meas (d, m, p, B), mean_meas (d, m, B), var_meas (d, m, m, B); any n_bmeas m for a registered
 *   right-hand side.  cfg->kalman_type = RK_KALMAN_SQRT matters to chkrebtii only (interrogate.py:35-42, m = 1):
 *   var_state_pred is then the factor L-, var_meas = W L- has shape (d, 1, p, B) and the draw is mu- + (W L-) z.  */
int rk_interrogate_batched(rk_handle h, const rk_solve_cfg* cfg, const rk_solve_in* in, double t, int32_t step,
        const double* mean_state_pred, const double* var_state_pred,
        double* wgt_meas, double* mean_meas, double* var_meas);

/* ---- multi-GPU: batch sharding over RCCL / xGMI --------------------------------------------------------------
 * The path shards by independent trajectories (no data-path collective).  The only exchange is the all-gather of
 * per-trajectory scalars (e.g. log-posteriors).  uid is an opaque 128-byte ncclUniqueId made on rank 0 and
 * distributed by the host launcher.                                                                           */
#define RK_COMM_UID_BYTES 128
int rk_comm_uid(void* uid128);
int rk_comm_init(rk_handle h, int rank, int nranks, const void* uid128);
int rk_comm_destroy(rk_handle h);
int rk_allgather_f64(rk_handle h, const double* send_dev, double* recv_dev, size_t count_per_rank);
int rk_allreduce_max_f64(rk_handle h, const double* send_dev, double* recv_dev, size_t count);
int rk_comm_barrier(rk_handle h);

#ifdef __cplusplus
}
#endif
#endif /* RODEO_KALMAN_H */
