/*
 * C ABI of libcnmpc.so, the MI355X (gfx950) implementation of the NMPC hot
 * path.  Two groups of entry points:
 *
 *  1. The controller interface of the reference library, unchanged: same
 *     symbol names, argument types and meaning as c/cnmpc.h:39-99 of the
 *     reference in its double-precision configuration, so that the reference's
 *     Python ctypes module (python/nmpc/__init__.py:94-219) and any C caller
 *     can load this library instead.  These operate on one hidden controller
 *     instance on the GPU (device $NMPC_DEVICE, default 0, horizon 100).
 *
 *  2. An additive batch interface (nmpc_batch_*, nmpc_*_samples) that runs
 *     many independent controller instances in lockstep on one GPU.  It has no
 *     counterpart in the reference, whose state is a process-global singleton
 *     (ccs-c66x/cnmpc.c:282-302); each function cites the reference function
 *     whose per-instance behaviour it reproduces.
 *
 * All pointers are plain host or device addresses (the library tells them
 * apart itself); no ownership crosses the boundary.  Not thread-safe per
 * object, like the reference.
 */
#ifndef CNMPC_AMD_H
#define CNMPC_AMD_H

#include <stdint.h>

/* real_t is double; a build with -DNMPC_SINGLE_PRECISION (libcnmpc_f32.so) is
 * the reference's shipped single-precision configuration (include/config.h:27-28,
 * c/cnmpc.h:29-33) and reports NMPC_PRECISION_FLOAT. */
#ifndef NMPC_SINGLE_PRECISION
#define NMPC_DOUBLE_PRECISION
#endif
#define NMPC_CONTROL_DIM 3
#define NMPC_STATE_DIM 13
#define NMPC_DELTA_DIM (NMPC_STATE_DIM - 1)
#define NMPC_REFERENCE_DIM (NMPC_STATE_DIM + NMPC_CONTROL_DIM)
#define NMPC_GRADIENT_DIM (NMPC_REFERENCE_DIM - 1)
#define OCP_HORIZON_LENGTH 100
#define OCP_STEP_LENGTH ((real_t)(1.0 / 50.0))

#ifdef NMPC_SINGLE_PRECISION
typedef float real_t;
#else
typedef double real_t;
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* c/cnmpc.h:39-44 */
struct nmpc_state_t {
    real_t position[3];
    real_t velocity[3];
    real_t attitude[4]; /* x, y, z, W */
    real_t angular_velocity[3];
};

/* c/cnmpc.h:46-50 */
enum nmpc_result_t { NMPC_OK, NMPC_INFEASIBLE, NMPC_ERROR };

/* c/cnmpc.h:90-93 */
enum nmpc_precision_t { NMPC_PRECISION_FLOAT = 0, NMPC_PRECISION_DOUBLE = 1 };

/* ------------------------------------------------------------------ *
 * 1. Reference controller interface (hidden single instance).
 *    Any GPU failure in these aborts the process, as the reference's
 *    assert()s do (ccs-c66x/cnmpc.c:699,932,943,1078).
 * ------------------------------------------------------------------ */

/* c/cnmpc.h:52 -> ccs-c66x/cnmpc.c:872-948 */
void nmpc_init(void);
/* c/cnmpc.h:53 -> ccs-c66x/cnmpc.c:950-951 (empty there; empty here unless
 * nmpc_set_rti(1) has been called) */
void nmpc_preparation_step(void);
/* c/cnmpc.h:54 -> ccs-c66x/cnmpc.c:953-956 */
void nmpc_feedback_step(real_t measurement[NMPC_STATE_DIM]);
/* c/cnmpc.h:55 -> ccs-c66x/cnmpc.c:958-969 */
enum nmpc_result_t nmpc_get_controls(real_t controls[NMPC_CONTROL_DIM]);
/* c/cnmpc.h:56 -> ccs-c66x/cnmpc.c:971-986 */
void nmpc_update_horizon(real_t new_reference[NMPC_REFERENCE_DIM]);

/* c/cnmpc.h:59-63 -> ccs-c66x/cnmpc.c:988-1020; read by the next nmpc_init
 * (weights) and by every later linearisation (bounds) */
void nmpc_set_state_weights(real_t coeffs[NMPC_DELTA_DIM]);
void nmpc_set_control_weights(real_t coeffs[NMPC_CONTROL_DIM]);
void nmpc_set_terminal_weights(real_t coeffs[NMPC_DELTA_DIM]);
void nmpc_set_lower_control_bound(real_t coeffs[NMPC_CONTROL_DIM]);
void nmpc_set_upper_control_bound(real_t coeffs[NMPC_CONTROL_DIM]);
/* c/cnmpc.h:64-65 -> ccs-c66x/cnmpc.c:1022-1080 */
void nmpc_set_reference_point(real_t coeffs[NMPC_REFERENCE_DIM], uint32_t i);
/* c/cnmpc.h:68 -> ccs-c66x/cnmpc.c:1082-1086 */
void nmpc_set_wind_velocity(real_t x, real_t y, real_t z);

/* c/cnmpc.h:71-85 -> c/cnmpc.cpp:120-180 (only the reference's C++ variant
 * exports these; the plant state lives on the host, the RK4 step runs on the
 * GPU with the aircraft model of ccs-c66x/cnmpc.c:365-539) */
void nmpc_fixedwingdynamics_set_position(real_t lat, real_t lon, real_t alt);
void nmpc_fixedwingdynamics_set_velocity(real_t x, real_t y, real_t z);
void nmpc_fixedwingdynamics_set_attitude(real_t w, real_t x, real_t y, real_t z);
void nmpc_fixedwingdynamics_set_angular_velocity(real_t x, real_t y, real_t z);
void nmpc_fixedwingdynamics_set_state(struct nmpc_state_t *in);
void nmpc_fixedwingdynamics_get_state(struct nmpc_state_t *in);
void nmpc_fixedwingdynamics_integrate(float dt, real_t control_vector[NMPC_CONTROL_DIM]);

/* Not in the reference: nmpc_batch_set_rti for the hidden instance. */
void nmpc_set_rti(int enabled);

/* c/cnmpc.h:95-99 -> ccs-c66x/cnmpc.c:1088-1110 */
uint32_t nmpc_config_get_state_dim(void);
uint32_t nmpc_config_get_control_dim(void);
uint32_t nmpc_config_get_horizon_length(void);
real_t nmpc_config_get_step_length(void);
enum nmpc_precision_t nmpc_config_get_precision(void);

/* ------------------------------------------------------------------ *
 * 2. Batch interface.  Every function returns 0 on success or a negative
 *    NMPC_E_* code; nmpc_last_error() describes the last failure of the
 *    calling thread.  All work is queued on the batch's HIP stream and
 *    functions that return data to the host wait for it.  (A batch of 512
 *    instances or more runs part of nmpc_batch_feedback_step on a second
 *    stream of its own, forked from and joined back into the batch's stream:
 *    work ordered behind the batch's stream is ordered behind all of it.)
 * ------------------------------------------------------------------ */
enum {
    NMPC_E_ARG = -1,      /* bad argument (null pointer, index out of range) */
    NMPC_E_STATE = -2,    /* call order violated (e.g. not initialised) */
    NMPC_E_HIP = -3       /* a HIP runtime call failed */
};
const char *nmpc_last_error(void);

typedef struct nmpc_batch nmpc_batch_t;

/* `instances` controllers with a `horizon`-stage OCP each on GPU `device`.
 * `stream` is a hipStream_t to queue work on, or NULL for a private stream.
 * All solver state starts at zero, like the reference's zero-initialised
 * statics (ccs-c66x/cnmpc.c:285-302). */
int nmpc_batch_create(nmpc_batch_t **out, uint32_t instances, uint32_t horizon,
                      int device, void *stream);
int nmpc_batch_destroy(nmpc_batch_t *b);
int nmpc_batch_synchronize(nmpc_batch_t *b);
/* Back to the state right after nmpc_batch_create (multipliers, stage
 * variables, references, wind, results all zero; weights and bounds kept), so
 * that the next nmpc_batch_init + reference loading is a cold start.  The
 * Jacobian and Newton-system storage is not cleared: nmpc_batch_init zeroes
 * the former and the first solve rebuilds the latter in full.  Asynchronous. */
int nmpc_batch_reset(nmpc_batch_t *b);

/* nmpc_set_*_weights / nmpc_set_*_control_bound for every instance
 * (ccs-c66x/cnmpc.c:988-1020).  Host pointers. */
int nmpc_batch_set_state_weights(nmpc_batch_t *b, const real_t coeffs[12]);
int nmpc_batch_set_control_weights(nmpc_batch_t *b, const real_t coeffs[3]);
int nmpc_batch_set_terminal_weights(nmpc_batch_t *b, const real_t coeffs[12]);
int nmpc_batch_set_lower_control_bound(nmpc_batch_t *b, const real_t coeffs[3]);
int nmpc_batch_set_upper_control_bound(nmpc_batch_t *b, const real_t coeffs[3]);

/* The same for instances first .. first+count-1 alone, each with values of its
 * own: the reference keeps weights and bounds per controller
 * (ccs-c66x/cnmpc.c:988-1020), and so may the instances of a batch.  Arrays
 * [count][12] / [count][3] in host or device memory; a null array leaves that
 * setting as it is.  As in the reference, weights take effect at the next
 * nmpc_batch_init (cnmpc.c:899-925); bounds are read by nmpc_batch_init, by
 * every stage linearisation after the call and by every feedback step
 * (cnmpc.c:920-931, 643-645, 1053-1062, 675-702).  The setters above keep
 * writing every instance.  The first call allocates one record of 72 values
 * per instance. */
int nmpc_batch_set_instance_weights(nmpc_batch_t *b, uint32_t first, uint32_t count,
                                    const real_t *state_weights, const real_t *control_weights,
                                    const real_t *terminal_weights);
int nmpc_batch_set_instance_control_bounds(nmpc_batch_t *b, uint32_t first, uint32_t count,
                                           const real_t *lower, const real_t *upper);

/* nmpc_init for every instance (ccs-c66x/cnmpc.c:872-948). */
int nmpc_batch_init(nmpc_batch_t *b);

/* nmpc_set_wind_velocity per instance; wind: [instances][3]
 * (ccs-c66x/cnmpc.c:1082-1086). */
int nmpc_batch_set_wind_velocity(nmpc_batch_t *b, const real_t *wind);

/* nmpc_set_reference_point(refs[b][i], first + i) for i = 0..count-1 in
 * ascending order, for every instance b; refs: [instances][count][16]
 * (ccs-c66x/cnmpc.c:1022-1080).  Linearises stages first-1 .. first+count-2. */
int nmpc_batch_set_reference_points(nmpc_batch_t *b, const real_t *refs,
                                    uint32_t first, uint32_t count);

/* The same with the reference points made on the device from waypoints, the
 * step the reference's harness runs on the host right before
 * nmpc_set_reference_point (interpolate_reference,
 * scripts/nmpc-xplane-synthetic.py:62-78): waypoints [instances][n_waypoints][17]
 * rows (t, 16-vector reference point), sorted by t per instance; horizon
 * position first + i gets the piecewise-linear interpolation at
 * t0 + (first + i) dt with the attitude quaternion scaled back to unit length;
 * past the last waypoint the last waypoint is used unchanged. */
int nmpc_batch_set_reference_from_waypoints(nmpc_batch_t *b, const real_t *waypoints,
                                            uint32_t n_waypoints, real_t t0, real_t dt,
                                            uint32_t first, uint32_t count);

/* nmpc_batch_set_reference_from_waypoints for callers that cannot allocate
 * device memory (tight serving loops, harnesses that forbid hipMalloc, code
 * that records work for replay).  No counterpart in the reference.  The
 * interpolated points pass between the kernels through a fixed array in the
 * library's code object instead of a buffer allocated on demand: instances x
 * count must not exceed NMPC_STATIC_SCRATCH_POINTS (NMPC_E_ARG otherwise),
 * `waypoints` must be device memory, and kernel timing is not applied to its
 * launches.  The call makes no hipMalloc / hipFree and creates no event (checked:
 * free device memory unchanged over warmed-up calls).  All batches on one GPU
 * share the array; the library orders their calls with one event per GPU, made
 * by nmpc_batch_create: a call waits, on its own stream, until the call made
 * before it on that GPU (from whichever batch and stream) has read its points.
 * Calls made from several host threads at once are not ordered.  Not exercised
 * under HIP stream capture.  Results are those of
 * nmpc_batch_set_reference_from_waypoints. */
#define NMPC_STATIC_SCRATCH_POINTS 65536u   /* 8 MiB per GPU */
int nmpc_batch_set_reference_from_waypoints_noalloc(nmpc_batch_t *b, const real_t *waypoints,
                                                    uint32_t n_waypoints, real_t t0, real_t dt,
                                                    uint32_t first, uint32_t count);

/* nmpc_preparation_step: no work (ccs-c66x/cnmpc.c:950-951), unless the
 * real-time-iteration mode below is on. */
int nmpc_batch_preparation_step(nmpc_batch_t *b);

/* Real-time-iteration mode (default off; no counterpart in the reference's
 * code, whose preparation step is empty although its comments assign it this
 * work: src/ocp.cpp:245-250, 315-322).  While on, the preparation step after a
 * feedback step that ended with NMPC_OK linearises every stage again: a stage
 * that carries a solution about that solution, x = x_ref[k] (+) z_k[0:12],
 * u = u_ref[k] + z_k[12:15], with C_k = d f / d (x, u) there and
 * c_k = delta(x_ref[k+1], f(x, u)) - C_k z_k; a stage a reference point has
 * linearised since (the new last stage after nmpc_batch_update_horizon) about
 * the reference as before.  The multipliers and the constant terms of the stage
 * objectives restart from zero and every stage is re-staged against them
 * (multipliers of the old linearisation are a worse start for the dual Newton
 * iteration than none).  After a feedback step that did not converge the
 * instance is left exactly as it is.  Results then differ from the
 * reference's; oracle/nmpc_port.c defines the arithmetic.  What it buys, in a
 * closed loop of 4 096 instances against simulated vehicles over 200 control
 * steps (tools/rti_closed_loop.py): mean position error 1.4 m instead of
 * 4.7 m, at 44 % more time per control step and 88 % instead of 98 % of the
 * solves ending with NMPC_OK within the iteration limit. */
int nmpc_batch_set_rti(nmpc_batch_t *b, int enabled);

/* Re-linearisation of every stage about the stored reference trajectory with
 * the current wind estimate and control bounds: what calling
 * nmpc_set_reference_point again with the same points for i = 0..horizon does
 * (ccs-c66x/cnmpc.c:1022-1080), without moving the points.  The reference's
 * preparation step is empty (ccs-c66x/cnmpc.c:950-951, src/ocp.cpp:245-250,
 * 315-322); this is the work its comments assign to it, offered separately so
 * that nmpc_batch_preparation_step stays a drop-in. */
int nmpc_batch_relinearise(nmpc_batch_t *b);

/* nmpc_feedback_step per instance; measurement: [instances][13]
 * (ccs-c66x/cnmpc.c:953-956).  Asynchronous. */
int nmpc_batch_feedback_step(nmpc_batch_t *b, const real_t *measurement);

/* nmpc_get_controls per instance (ccs-c66x/cnmpc.c:958-969): controls
 * [instances][3], status [instances] (NMPC_OK / NMPC_INFEASIBLE), iterations
 * [instances] (qpDUNES' log.numIter, ccs-c66x/qpDUNES/dual_qp.c:53-56,105,220).
 * Any of the three may be NULL. */
int nmpc_batch_get_controls(nmpc_batch_t *b, real_t *controls, int32_t *status,
                            int32_t *iterations);

/* nmpc_update_horizon per instance; new_reference: [instances][16]
 * (ccs-c66x/cnmpc.c:971-986). */
int nmpc_batch_update_horizon(nmpc_batch_t *b, const real_t *new_reference);

uint32_t nmpc_batch_instances(const nmpc_batch_t *b);
uint32_t nmpc_batch_horizon(const nmpc_batch_t *b);

/* Introspection for parity tests, host output buffers.
 * Stage record of horizon position k (332 doubles): C[180] c[12] zLow[15]
 * zUpp[15] q[15] p z[15] y[30] qStep[15] pStep zUnc[15] dz[15] pad[3]
 * (the fields of interval_t, ccs-c66x/qpDUNES/types.h:411-474).
 * Global record: lambda[12N] deltaLambda[12N] gradient[12N] hessian[288N]
 * cholHessian[288N] ieqStatus[15(N+1)] prevIeqStatus[15(N+1)] xref[13(N+1)]
 * uref[3N] controls[3] (qpData_t, types.h:650-693; cnmpc.c:285-301). */
int nmpc_batch_get_stage(nmpc_batch_t *b, uint32_t instance, uint32_t k, real_t *out332);
uint32_t nmpc_batch_global_size(const nmpc_batch_t *b);
int nmpc_batch_get_global(nmpc_batch_t *b, uint32_t instance, real_t *out);
/* info[6] = status (qpDUNES return_t value), numIter, lastActSetChangeIdx,
 * regularised, line-search iterations, last result */
int nmpc_batch_get_solver_info(nmpc_batch_t *b, uint32_t instance, int32_t info[6],
                               real_t *alpha);

/* Plant roll-forward beside the controllers, for closed-loop studies that stay
 * on the GPU: per instance what nmpc_fixedwingdynamics_set_state / _get_state /
 * _integrate do for the one controller (c/cnmpc.cpp:140-180, with the state
 * model and RK4 step of ccs-c66x/cnmpc.c:320-400).
 *   state     [instances][13]
 *   controls  [instances][3], or NULL for the controls nmpc_batch_get_controls
 *             would return (last good controls of every instance)
 *   wind      [instances][3] acting on the plant, or NULL for the wind estimate
 *             given to nmpc_batch_set_wind_velocity
 * nmpc_batch_feedback_step_from_plant is nmpc_batch_feedback_step with the
 * plant states as measurements. */
int nmpc_batch_plant_set_state(nmpc_batch_t *b, const real_t *state);
int nmpc_batch_plant_get_state(nmpc_batch_t *b, real_t *state);
int nmpc_batch_plant_integrate(nmpc_batch_t *b, real_t dt, const real_t *controls,
                               const real_t *wind);
int nmpc_batch_feedback_step_from_plant(nmpc_batch_t *b);

/* Per-kernel timing with HIP events on the batch's stream.  kernel ids: */
enum {
    NMPC_K_LINEARISE = 0, NMPC_K_QP_BEGIN, NMPC_K_LINE_SEARCH_TAIL, NMPC_K_NEWTON_SOLVE,
    NMPC_K_NEWTON_STEP, NMPC_K_OTHER, NMPC_K_COUNT
};
int nmpc_batch_set_profiling(nmpc_batch_t *b, int enabled);
/* Restricts the timing to the kernel ids whose bit is set in `kinds` (0: all of
 * them again).  Every timed launch costs two event packets on the stream. */
int nmpc_batch_set_profiling_kinds(nmpc_batch_t *b, unsigned kinds);
/* Waits for the stream; total_ms / launches since profiling was last enabled. */
int nmpc_batch_get_kernel_time(nmpc_batch_t *b, int kernel, double *total_ms,
                               uint64_t *launches);

/* Stateless sample kernels on `device` (stream may be NULL).
 * Stage linearisation of n samples (ccs-c66x/cnmpc.c:583-667): x [n][13],
 * u [n][3], x_next [n][13], wind [n][3], lower/upper [3] (host) ->
 * jacobian [n][180] row-major 12x15, residual [n][12]. */
int nmpc_linearise_samples(int device, void *stream, uint64_t n, const real_t *x,
                           const real_t *u, const real_t *x_next, const real_t *wind,
                           const real_t lower[3], const real_t upper[3],
                           real_t *jacobian, real_t *residual);
/* One RK4 step of n samples (ccs-c66x/cnmpc.c:365-400): -> out [n][13]. */
int nmpc_rk4_samples(int device, void *stream, uint64_t n, const real_t *x,
                     const real_t *u, const real_t *wind, real_t delta, real_t *out);

/* Measured machine peaks for roofline fractions (no counterpart in the
 * reference; not on the controller path).  Each call allocates its own
 * buffers, runs a few passes on the device's null stream and reports the
 * last (vector rate) or best (triad) pass.
 * fp64 vector rate: independent x <- x * a + b chains, 8 wavefronts per SIMD;
 * fused != 0 uses one fused multiply-add per step (2 flop), fused == 0 a
 * multiply and an add (the library's own kernels are built without
 * contraction).  Stream triad: a[i] = b[i] + s * c[i] over three arrays of
 * bytes_per_array each, 16 bytes per lane and access, in three launch shapes
 * (eight and two workgroups per CU striding through the arrays, one workgroup
 * per contiguous 64 KB); reports 3 x bytes / time of the fastest pass. */
int nmpc_measure_fp64_vector_peak(int device, int fused, double *tflops);
int nmpc_measure_stream_triad(int device, uint64_t bytes_per_array, double *gbps);

#ifdef __cplusplus
}
#endif
#endif
