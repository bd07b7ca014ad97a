/*
 * TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C, scalar; fp64, or fp32
 * with -DPORT_SINGLE_PRECISION: libnmpc_port.so / libnmpc_port_f32.so) of the
 * reference NMPC hot path: the pure-C controller ccs-c66x/cnmpc.c together with
 * the parts of the vendored qpDUNES 1.0beta it reaches (diagonal-Hessian
 * clipping stage solver, reverse block-tridiagonal Newton factorisation,
 * accelerated gradient-bisection line search).
 *
 * It exists to check the HIP implementation and to serve as the CPU baseline
 * in bench.py.  Nothing in the shipped library (nmpc_amd/) includes, links or
 * calls it.  Parity of this restatement is pinned against the compiled
 * reference (oracle/_ref) and the committed golden vectors in tests/golden/.
 *
 * Unlike the reference, state lives in an explicit instance object and the
 * horizon length is a run-time parameter.
 */
#ifndef NMPC_PORT_H
#define NMPC_PORT_H

#ifdef __cplusplus
extern "C" {
#endif

#ifdef PORT_SINGLE_PRECISION
typedef float real_t;       /* the reference's shipped configuration (include/config.h:27) */
#else
typedef double real_t;
#endif

typedef struct port_s port_t;

unsigned port_real_bytes(void);             /* sizeof(real_t) of this build */

port_t *port_create(int horizon);          /* zero-filled, like fresh statics */
void port_destroy(port_t *p);

/* mirrors of the C ABI in c/cnmpc.h:52-68 */
void port_set_state_weights(port_t *p, const real_t *w12);
void port_set_control_weights(port_t *p, const real_t *w3);
void port_set_terminal_weights(port_t *p, const real_t *w12);
void port_set_lower_control_bound(port_t *p, const real_t *b3);
void port_set_upper_control_bound(port_t *p, const real_t *b3);
void port_set_wind_velocity(port_t *p, real_t x, real_t y, real_t z);
void port_init(port_t *p);
void port_set_reference_point(port_t *p, const real_t *ref16, unsigned i);
void port_feedback_step(port_t *p, const real_t *meas13);
int port_get_controls(port_t *p, real_t *u3);    /* 0 OK, 1 INFEASIBLE */
void port_update_horizon(port_t *p, const real_t *ref16);

/* Real-time-iteration preparation step (no code for it in the reference, whose
 * preparation step is empty): re-linearises every stage that carries a solution
 * of the last feedback step about that solution.  See nmpc_port.c. */
void port_preparation_step_rti(port_t *p);
void port_state_plus_delta(real_t *out13, const real_t *x13, const real_t *d12);

/* introspection, same record layout as oracle/ref_wrap.c */
int port_horizon(const port_t *p);
int port_num_iter(const port_t *p);
int port_last_result(const port_t *p);
int port_last_status(const port_t *p);           /* internal solver status */
real_t port_alpha(const port_t *p);
int port_line_search_iter(const port_t *p);
int port_hessian_regularized(const port_t *p);
/* backtracking trials, bisection trials, full steps of the last solve */
void port_trial_counts(const port_t *p, int *out3);
void port_get_stage(const port_t *p, unsigned k, real_t *out332);
unsigned port_global_size(const port_t *p);
void port_get_global(const port_t *p, real_t *out);

/* model functions (cnmpc.c:320-667) */
void port_x8_dynamics(real_t *out6, const real_t *x13, const real_t *u3,
                      const real_t *wind3);
void port_state_model(real_t *out13, const real_t *x13, const real_t *u3,
                      const real_t *wind3);
void port_rk4(real_t *out13, const real_t *x13, const real_t *u3,
              const real_t *wind3, real_t delta);
void port_state_to_delta(real_t *out12, const real_t *s1, const real_t *s2);
void port_linearise(const real_t *x13, const real_t *u3, const real_t *xnext13,
                    const real_t *wind3, const real_t *lower3,
                    const real_t *upper3, real_t *jac180, real_t *res12);

/*
 * Batched drivers for the CPU baseline: `count` independent instances spread
 * over `threads` host threads.  refs: [count][N+1][16], meas: [count][13],
 * wind: [count][3]; outputs u: [count][3], status/iters: [count].
 * Each instance is created, configured, cold-started (N+1 reference points,
 * feedback, get controls) and destroyed.  Returns wall seconds of the solve
 * part (reference loading + feedback), excluding create/destroy.
 */
double port_batch_cold_solve(int horizon, int count, int threads,
                             const real_t *weights_q12, const real_t *weights_r3,
                             const real_t *weights_t12, const real_t *lower3,
                             const real_t *upper3, const real_t *refs,
                             const real_t *meas, const real_t *wind,
                             real_t *u, int *status, int *iters);

/* port_linearise over `count` samples ([count][13], [count][3], ...) on `threads`
 * host threads; returns wall seconds. */
double port_batch_linearise(long count, int threads, const real_t *x, const real_t *u,
                            const real_t *xnext, const real_t *wind, const real_t *lower3,
                            const real_t *upper3, real_t *jac, real_t *res);

#ifdef __cplusplus
}
#endif
#endif
