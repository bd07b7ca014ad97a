/*
 * TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C, fp64, scalar) of the
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

typedef struct port_s port_t;

port_t *port_create(int horizon);          /* zero-filled, like fresh statics */
void port_destroy(port_t *p);

/* mirrors of the C ABI in c/cnmpc.h:52-68 */
void port_set_state_weights(port_t *p, const double *w12);
void port_set_control_weights(port_t *p, const double *w3);
void port_set_terminal_weights(port_t *p, const double *w12);
void port_set_lower_control_bound(port_t *p, const double *b3);
void port_set_upper_control_bound(port_t *p, const double *b3);
void port_set_wind_velocity(port_t *p, double x, double y, double z);
void port_init(port_t *p);
void port_set_reference_point(port_t *p, const double *ref16, unsigned i);
void port_feedback_step(port_t *p, const double *meas13);
int port_get_controls(port_t *p, double *u3);    /* 0 OK, 1 INFEASIBLE */
void port_update_horizon(port_t *p, const double *ref16);

/* Real-time-iteration preparation step (no code for it in the reference, whose
 * preparation step is empty): re-linearises every stage that carries a solution
 * of the last feedback step about that solution.  See nmpc_port.c. */
void port_preparation_step_rti(port_t *p);
void port_state_plus_delta(double *out13, const double *x13, const double *d12);

/* introspection, same record layout as oracle/ref_wrap.c */
int port_horizon(const port_t *p);
int port_num_iter(const port_t *p);
int port_last_result(const port_t *p);
int port_last_status(const port_t *p);           /* internal solver status */
double port_alpha(const port_t *p);
int port_line_search_iter(const port_t *p);
int port_hessian_regularized(const port_t *p);
/* backtracking trials, bisection trials, full steps of the last solve */
void port_trial_counts(const port_t *p, int *out3);
void port_get_stage(const port_t *p, unsigned k, double *out332);
unsigned port_global_size(const port_t *p);
void port_get_global(const port_t *p, double *out);

/* model functions (cnmpc.c:320-667) */
void port_x8_dynamics(double *out6, const double *x13, const double *u3,
                      const double *wind3);
void port_state_model(double *out13, const double *x13, const double *u3,
                      const double *wind3);
void port_rk4(double *out13, const double *x13, const double *u3,
              const double *wind3, double delta);
void port_state_to_delta(double *out12, const double *s1, const double *s2);
void port_linearise(const double *x13, const double *u3, const double *xnext13,
                    const double *wind3, const double *lower3,
                    const double *upper3, double *jac180, double *res12);

/*
 * Batched drivers for the CPU baseline: `count` independent instances spread
 * over `threads` host threads.  refs: [count][N+1][16], meas: [count][13],
 * wind: [count][3]; outputs u: [count][3], status/iters: [count].
 * Each instance is created, configured, cold-started (N+1 reference points,
 * feedback, get controls) and destroyed.  Returns wall seconds of the solve
 * part (reference loading + feedback), excluding create/destroy.
 */
double port_batch_cold_solve(int horizon, int count, int threads,
                             const double *weights_q12, const double *weights_r3,
                             const double *weights_t12, const double *lower3,
                             const double *upper3, const double *refs,
                             const double *meas, const double *wind,
                             double *u, int *status, int *iters);

/* port_linearise over `count` samples ([count][13], [count][3], ...) on `threads`
 * host threads; returns wall seconds. */
double port_batch_linearise(long count, int threads, const double *x, const double *u,
                            const double *xnext, const double *wind, const double *lower3,
                            const double *upper3, double *jac, double *res);

#ifdef __cplusplus
}
#endif
#endif
