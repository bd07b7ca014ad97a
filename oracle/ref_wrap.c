/*
 * Test infrastructure only -- never part of the shipped library.
 *
 * Accessor translation unit for the reference's pure-C NMPC variant.  It pulls
 * the reference's ccs-c66x/cnmpc.c into this translation unit (resolved on the
 * include path at build time; the file itself is not copied) so that its
 * file-static state and functions can be read by the parity tests and the
 * golden-vector generator.  Built by oracle/Makefile into oracle/_ref/ only
 * where the reference tree is present.
 */
#include "cnmpc.c"

#define REF_N   OCP_HORIZON_LENGTH
#define REF_NX  NMPC_DELTA_DIM
#define REF_NZ  NMPC_GRADIENT_DIM

/* per-stage record: C[180] c[12] zLow[15] zUpp[15] q[15] p z[15] y[30]
 * qStep[15] pStep zUnc[15] dz[15] */
#define REF_STAGE_REC 332

unsigned ref_horizon(void) { return REF_N; }
unsigned ref_real_bytes(void) { return (unsigned)sizeof(real_t); }
unsigned ref_stage_rec(void) { return REF_STAGE_REC; }

int ref_num_iter(void) { return ocp_qp_data.qpdata.log.numIter; }
int ref_last_result(void) { return ocp_last_result ? 1 : 0; }
real_t ref_alpha(void) { return ocp_qp_data.qpdata.alpha; }
int ref_line_search_iter(void) {
    return (int)ocp_qp_data.qpdata.log.itLog[0].numLineSearchIter;
}
int ref_hessian_regularized(void) {
    return (int)ocp_qp_data.qpdata.log.itLog[0].isHessianRegularized;
}

/* Direct access to the static model functions (cnmpc.c:320-667). */
void ref_set_wind(const real_t *w) {
    wind_velocity[0] = w[0]; wind_velocity[1] = w[1]; wind_velocity[2] = w[2];
}
void ref_x8_dynamics(real_t *out6, const real_t *state, const real_t *control) {
    _state_x8_dynamics(out6, state, control);
}
void ref_state_model(real_t *out13, const real_t *state, const real_t *control) {
    _state_model(out13, state, control);
}
void ref_rk4(real_t *out13, const real_t *state, const real_t *control,
             real_t delta) {
    _state_integrate_rk4(out13, state, control, delta);
}
void ref_state_to_delta(real_t *out12, const real_t *s1, const real_t *s2) {
    _state_to_delta(out12, s1, s2);
}
/* Uses the control bounds currently held by the library (cnmpc.c:643-645). */
void ref_solve_interval_ivp(const real_t *state, const real_t *control,
                            const real_t *next_state, real_t *jac180,
                            real_t *res12) {
    _solve_interval_ivp(state, control, jac180, next_state, res12);
}

static real_t *put(real_t *o, const real_t *src, size_t n) {
    memcpy(o, src, n * sizeof(real_t));
    return o + n;
}

void ref_get_stage(unsigned k, real_t *o) {
    const interval_t *iv = ocp_qp_data.qpdata.intervals[k];
    memset(o, 0, REF_STAGE_REC * sizeof(real_t));
    size_t nv = iv->nV;
    if (k < REF_N) {
        memcpy(o, iv->C.data, 180 * sizeof(real_t));
        memcpy(o + 180, iv->c.data, 12 * sizeof(real_t));
    }
    memcpy(o + 192, iv->zLow.data, nv * sizeof(real_t));
    memcpy(o + 207, iv->zUpp.data, nv * sizeof(real_t));
    memcpy(o + 222, iv->q.data, nv * sizeof(real_t));
    o[237] = iv->p;
    memcpy(o + 238, iv->z.data, nv * sizeof(real_t));
    memcpy(o + 253, iv->y.data, 2 * nv * sizeof(real_t));
    memcpy(o + 283, iv->qpSolverClipping.qStep.data, nv * sizeof(real_t));
    o[298] = iv->qpSolverClipping.pStep;
    memcpy(o + 299, iv->qpSolverClipping.zUnconstrained.data,
           nv * sizeof(real_t));
    memcpy(o + 314, iv->qpSolverClipping.dz.data, nv * sizeof(real_t));
    /* 329..331 spare */
}

/* lambda[12N] dLambda[12N] grad[12N] hess[288N] chol[288N]
 * ieq[15(N+1)] prevIeq[15(N+1)] xref[13(N+1)] uref[3N] u[3] */
unsigned ref_global_size(void) {
    return 3 * 12 * REF_N + 2 * 288 * REF_N + 2 * 15 * (REF_N + 1) +
           13 * (REF_N + 1) + 3 * REF_N + 3;
}

void ref_get_global(real_t *o) {
    const qpData_t *qp = &ocp_qp_data.qpdata;
    size_t k, i;
    o = put(o, qp->lambda.data, 12 * REF_N);
    o = put(o, qp->deltaLambda.data, 12 * REF_N);
    o = put(o, qp->gradient.data, 12 * REF_N);
    o = put(o, qp->hessian.data, 288 * REF_N);
    o = put(o, qp->cholHessian.data, 288 * REF_N);
    for (k = 0; k <= REF_N; k++)
        for (i = 0; i < 15; i++)
            *o++ = (i < qp->intervals[k]->nV)
                       ? (real_t)qp->log.itLog[0].ieqStatus[k][i] : 0.0;
    for (k = 0; k <= REF_N; k++)
        for (i = 0; i < 15; i++)
            *o++ = (i < qp->intervals[k]->nV)
                       ? (real_t)qp->log.itLog[0].prevIeqStatus[k][i] : 0.0;
    o = put(o, ocp_state_reference, 13 * (REF_N + 1));
    o = put(o, ocp_control_reference, 3 * REF_N);
    o = put(o, ocp_control_value, 3);
}
