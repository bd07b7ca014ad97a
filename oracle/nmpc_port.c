/*
 * TEST INFRASTRUCTURE ONLY -- see nmpc_port.h.
 *
 * Scalar restatement of the reference NMPC hot path in the reference's two
 * precisions (real_t = double, or float with -DPORT_SINGLE_PRECISION: the
 * reference's shipped configuration, include/config.h:27-28).  Every arithmetic
 * expression keeps the operand order, association and literal types of the
 * reference (float-suffixed, real_t-cast or plain double literals, and the
 * double-valued libm calls the reference makes on real_t operands), so that
 * results are bit-identical to the reference compiled without FMA contraction
 * (build with -ffp-contract=off, no -ffast-math).
 *
 * Reference locations (paths relative to the reference tree):
 *   model        ccs-c66x/cnmpc.c:168-280 (quaternion helpers, fatan2),
 *                :320-363 (kinematics), :365-400 (RK4), :402-539 (X8 aero),
 *                :541-579 (state difference), :583-667 (stage linearisation)
 *   controller   ccs-c66x/cnmpc.c:675-726, :872-1086
 *   QP solver    ccs-c66x/qpDUNES/dual_qp.c, setup_qp.c,
 *                stage_qp_solver_clipping.c, matrix_vector.c, utils.c
 *                (qpDUNES 1.0beta, vendored in the reference)
 */
#include "nmpc_port.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

enum { NS = 13, NU = 3, NX = 12, NZ = 15, NREF = 16 };

/* qpDUNES return codes used on this path (qpDUNES/types.h:211-231) */
enum {
    ST_OK = 0,
    ST_OPTIMAL = 1,
    ST_ITER_LIMIT = 8,
    ST_DIV_ZERO = 9,
    ST_LS_MAXITER = 10,
    ST_LS_MIN_STEP = 11,
    ST_LS_MAX_STEP = 12,
    ST_NO_ASCENT = 13
};

/* effective solver options: setup_qp.c:667-725 overridden at cnmpc.c:891-894 */
#define OPT_MAX_ITER 5
#define OPT_MAX_LS_ITER 19
#define OPT_MAX_LS_REFINE 40
/* the option fields are real_t (qpDUNES/types.h:487-543): every value is
 * rounded to real_t when it is stored */
#define OPT_STAT_TOL ((real_t)1e-3f)
#define OPT_EQ_TOL ((real_t)2.221e-16)
#define OPT_DIAG_REG_TOL ((real_t)1.e-10)
#define OPT_ACT_TOL ((real_t)(1e4 * OPT_EQ_TOL))
#define OPT_INFTY ((real_t)1.e12)
#define OPT_REG_PARAM ((real_t)1.e-6)
#define OPT_LS_REDUCTION ((real_t)0.1)
#define OPT_LS_INCREASE ((real_t)1.5)
#define OPT_LS_MIN_ABS OPT_EQ_TOL
#define OPT_LS_MIN_REL ((real_t)1.e-14)
#define OPT_LS_STAT_TOL ((real_t)1.e-3)
#define OPT_LS_MAX_STEP ((real_t)1.)

#ifdef PORT_SINGLE_PRECISION
#define FD_STEP ((real_t)1.857e-2)  /* cnmpc.c:152 */
#else
#define FD_STEP ((real_t)1.221e-4)  /* cnmpc.c:154 */
#endif
#define STEP_LENGTH ((real_t)(1.0 / 50.0))  /* include/config.h:60 */
#define G_ACC ((real_t)9.80665)
#define RHO_AIR ((real_t)1.225)

typedef struct {
    real_t C[NX * NZ], c[NX];
    real_t zLow[NZ], zUpp[NZ];
    real_t q[NZ], p;
    real_t z[NZ], y[2 * NZ];
    real_t qStep[NZ], pStep;
    real_t zUnc[NZ], dz[NZ];
    real_t tmp[NZ];             /* trial vector (zVecTmp) */
    int nV;
    int changed;                /* actSetHasChanged */
    int fresh;                  /* linearised about the reference since the last solve */
} stage_t;

struct port_s {
    int N;
    real_t wind[3];
    real_t *xref;               /* (N+1) x 13 */
    real_t *uref;               /* N x 3 */
    real_t lb[NU], ub[NU], wq[NX], wr[NU], wt[NX];
    real_t Hreg[NZ], Hterm[NX]; /* stage Hessian diagonals as the solver sees them */
    stage_t *rec;               /* N+1 records */
    int *slot;                  /* position -> record */
    real_t *lambda, *dlambda, *grad, *gradTry, *hess, *chol;
    int *ieq, *prev;            /* (N+1) x 15 */
    real_t alpha;
    int numIter, lastResult, lastStatus, lsIter, regularized;
    int nBacktrack, nBisect, nFullStep;   /* per-solve trial counters (diagnostics) */
    real_t u[NU];
};

/* ------------------------------------------------------------------ model */

static void quat_mul(real_t *r, const real_t *a, const real_t *b) {
    /* xyzw storage, cnmpc.c:175-178 */
    r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    r[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    r[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
}

static void quat_rotate(real_t *r, const real_t *q, const real_t *v) {
    /* cnmpc.c:213-239 */
    real_t tx, ty, tz, rx, ry, rz;
    tx = q[1] * v[2];
    ty = q[2] * v[0];
    tx -= q[2] * v[1];
    ty -= q[0] * v[2];
    tz = q[0] * v[1];
    ty *= 2.0;
    tz -= q[1] * v[0];
    tx *= 2.0;
    tz *= 2.0;
    rx = v[0]; rx += q[3] * tx; rx += q[1] * tz; rx -= q[2] * ty;
    ry = v[1]; ry += q[3] * ty; ry += q[2] * tx; ry -= q[0] * tz;
    rz = v[2]; rz += q[3] * tz; rz -= q[1] * tx; rz += q[0] * ty;
    r[0] = rx; r[1] = ry; r[2] = rz;
}

static float approx_atan2(float y, float x) {
    /* cnmpc.c:257-280: float in, float out; the constants are real_t, so the
     * intermediates are double in the fp64 configuration and float otherwise */
    const real_t pi = (real_t)3.14159265, pi_2 = (real_t)1.5707963;
    if (x == (real_t)0.0) {
        if (y > (real_t)0.0) return (float)pi_2;
        if (y == (real_t)0.0) return (float)(real_t)0.0;
        return (float)-pi_2;
    }
    float res;
    float z = y / x;
    if ((z < 0 ? -z : z) < (real_t)1.0) {
        res = (float)(z / ((real_t)1.0 + (real_t)0.28 * z * z));
        if (x < (real_t)0.0) {
            if (y < (real_t)0.0) return (float)(res - pi);
            return (float)(res + pi);
        }
    } else {
        res = (float)(pi_2 - z / (z * z + (real_t)0.28));
        if (y < (real_t)0.0) return (float)(res - pi);
    }
    return res;
}

#define FMAXM(a, b) (((a) > (b)) ? (a) : (b))
#define FABSM(x) ((x) < 0 ? -(x) : (x))

void port_x8_dynamics(real_t *out, const real_t *s, const real_t *u,
                      const real_t *wind) {
    /* cnmpc.c:402-539 */
    real_t ned[3], af[3];
    ned[0] = wind[0] - s[3];
    ned[1] = wind[1] - s[4];
    ned[2] = wind[2] - s[5];
    quat_rotate(af, &s[6], ned);

    real_t rx, ry, rz, tx, ty;
    tx = s[7] * (G_ACC * 2.0f);
    ty = -s[6] * (G_ACC * 2.0f);
    rx = s[9] * tx;
    ry = s[9] * ty;
    ry += s[8] * tx;
    rx -= s[8] * ty;
    rz = G_ACC;
    rz -= s[7] * tx;
    rz += s[6] * ty;
    out[0] = rx; out[1] = ry; out[2] = rz;

    real_t ax2 = af[0] * af[0], ay2 = af[1] * af[1], az2 = af[2] * af[2];
    real_t rpm = u[0] * 25000.0f;
    real_t ve2 = (0.0025f * 0.0025f) * rpm * rpm;
    real_t thrust = FMAXM(0.0f, ve2 - ax2) *
                    (0.26315789473684f * 0.5f * RHO_AIR * 0.025f);

    real_t hv2 = ax2 + ay2;
    real_t qbar = (RHO_AIR * 0.5f) * hv2;
    real_t v_inv = (real_t)(1.0 / sqrt(FMAXM(1.0f, hv2 + az2)));
    real_t vv = (real_t)sqrt(ax2 + az2);
    real_t vv_inv = (real_t)(1.0 / (FMAXM(1.0f, vv)));

    real_t sin_b = af[1] * v_inv;
    real_t cos_b = vv * v_inv;
    real_t alpha = approx_atan2((float)-af[2], (float)-af[0]);
    real_t a2 = alpha * alpha;
    real_t sin_a = -af[2] * vv_inv;
    real_t cos_a = -af[0] * vv_inv;

    real_t lift = (-5.0f * alpha + 1.0f) * a2 + 2.5f * alpha + 0.12f;
    real_t sca = sin_a * cos_a;
    real_t alt = 0.8f * sca;
    if ((alpha < -0.25f && lift > alt) || (alpha > 0.0f && lift < alt))
        lift = alt;

    lift = (qbar * 0.26315789473684f) * lift;
    real_t drag = (qbar * 0.26315789473684f) * (0.05f + 0.7f * sin_a * sin_a);
    real_t side = (qbar * 0.26315789473684f) * 0.3f * sin_b * cos_b;

    real_t fx = lift * sin_a - drag * cos_a - side * sin_b;
    real_t fz = lift * cos_a + drag * sin_a;
    real_t fy = side * cos_b;
    out[1] += fy;
    out[0] += fx + thrust;
    out[2] -= fz;

    real_t yaw_r = s[12], pitch_r = s[11], roll_r = s[10];
    real_t la = u[1] - 0.5f, ra = u[2] - 0.5f;
    real_t pm = 0.001f - 0.1f * sca - 0.003f * pitch_r - 0.04f * (la + ra);
    real_t rm = 0.03f * sin_b - 0.015f * roll_r + 0.1f * (la - ra);
    real_t ym = -0.02f * sin_b - 0.05f * yaw_r - 0.01f * (FABSM(la) + FABSM(ra));
    pm *= qbar; rm *= qbar; ym *= qbar;

    out[4] = 5.8823528f * pm;
    out[3] = (3.364222f * rm + 0.27744448f * ym);
    out[5] = (0.27744448f * rm + 2.4920163f * ym);
}

void port_state_model(real_t *out, const real_t *s, const real_t *u,
                      const real_t *wind) {
    /* cnmpc.c:320-363 */
    real_t acc[6], a[4], w[4];
    port_x8_dynamics(acc, s, u, wind);
    out[0] = s[3]; out[1] = s[4]; out[2] = s[5];
    a[0] = s[6]; a[1] = s[7]; a[2] = s[8]; a[3] = -s[9];
    quat_rotate(&out[3], a, acc);
    a[3] = -a[3];
    w[0] = -s[10]; w[1] = -s[11]; w[2] = -s[12]; w[3] = 0.0;
    quat_mul(&out[6], w, a);
    out[6] *= 0.5; out[7] *= 0.5; out[8] *= 0.5; out[9] *= 0.5;
    out[10] = acc[3]; out[11] = acc[4]; out[12] = acc[5];
}

__attribute__((unused)) static void axpy13(real_t *r, const real_t *d, real_t a, const real_t *s) {
    for (int i = 0; i < NS; i++) r[i] = s[i] + d[i] * a;
}

void port_rk4(real_t *out, const real_t *s, const real_t *u, const real_t *wind,
              real_t delta) {
#if defined(PORT_INTEGRATOR_HEUN)
    /* include/integrator.h:66-79 (C++ variant only; the C path has RK4 only) */
    real_t k1[NS], k2[NS], t[NS];
    port_state_model(k1, s, u, wind);
    for (int i = 0; i < NS; i++) t[i] = s[i] + delta * k1[i];
    port_state_model(k2, t, u, wind);
    real_t half = delta * (real_t)0.5;
    for (int i = 0; i < NS; i++) out[i] = s[i] + half * (k1[i] + k2[i]);
#elif defined(PORT_INTEGRATOR_EULER)
    /* include/integrator.h:81-91 */
    real_t k1[NS];
    port_state_model(k1, s, u, wind);
    for (int i = 0; i < NS; i++) out[i] = s[i] + delta * k1[i];
#else
    /* cnmpc.c:365-400 */
    real_t a[NS], b[NS], c[NS], d[NS], t[NS];
    port_state_model(a, s, u, wind);
    axpy13(t, a, delta * 0.5f, s);
    port_state_model(b, t, u, wind);
    axpy13(t, b, delta * 0.5f, s);
    port_state_model(c, t, u, wind);
    axpy13(t, c, delta, s);
    port_state_model(d, t, u, wind);
    real_t d3 = delta * (1.0f / 3.0f), d6 = delta * (1.0f / 6.0f);
    for (int i = 0; i < NS; i++)
        out[i] = s[i] + d3 * (b[i] + c[i]) + d6 * (a[i] + d[i]);
#endif
}

void port_state_to_delta(real_t *dl, const real_t *s1, const real_t *s2) {
    /* cnmpc.c:541-579 */
    for (int i = 0; i < 3; i++) {
        dl[i] = s2[i] - s1[i];
        dl[i + 3] = s2[i + 3] - s1[i + 3];
        dl[i + 9] = s2[i + 10] - s1[i + 10];
    }
    real_t cj[4], e[4];
    cj[0] = s1[6]; cj[1] = s1[7]; cj[2] = s1[8]; cj[3] = -s1[9];
    quat_mul(e, &s2[6], cj);
    if (e[3] < 0) { e[0] = -e[0]; e[1] = -e[1]; e[2] = -e[2]; e[3] = -e[3]; }
    real_t d = (real_t)4.0 / ((real_t)1.0 + e[3]);     /* NMPC_MRP_F / (NMPC_MRP_A + w) */
    dl[6] = e[0] * d; dl[7] = e[1] * d; dl[8] = e[2] * d;
}

void port_linearise(const real_t *x, const real_t *u, const real_t *xnext,
                    const real_t *wind, const real_t *lower,
                    const real_t *upper, real_t *jac, real_t *res) {
    /* cnmpc.c:583-667 */
    real_t nominal[NS], moved[NS], col[NX];
    port_rk4(nominal, x, u, wind, STEP_LENGTH);
    port_state_to_delta(res, xnext, nominal);
    for (int j = 0; j < NZ; j++) {
        real_t pr[NREF];
        real_t h = FD_STEP, hinv = (real_t)(1.0 / FD_STEP);
        for (int i = 0; i < NS; i++) pr[i] = x[i];
        pr[13] = u[0]; pr[14] = u[1]; pr[15] = u[2];
        if (j < 6) {
            pr[j] += FD_STEP;
        } else if (j <= 8) {
            real_t dp[3] = {0.0, 0.0, 0.0}, dq[4], t[4];
            dp[j - 6] = FD_STEP;
            dq[3] = -(FD_STEP * FD_STEP) + (real_t)16.0 / (real_t)(16.0 + FD_STEP * FD_STEP);
            dq[0] = ((real_t)1.0 / (real_t)4.0) * ((real_t)1.0 + dq[3]) * dp[0];
            dq[1] = ((real_t)1.0 / (real_t)4.0) * ((real_t)1.0 + dq[3]) * dp[1];
            dq[2] = ((real_t)1.0 / (real_t)4.0) * ((real_t)1.0 + dq[3]) * dp[2];
            quat_mul(t, dq, &pr[6]);
            pr[6] = t[0]; pr[7] = t[1]; pr[8] = t[2]; pr[9] = t[3];
        } else if (j < NX) {
            pr[j + 1] += FD_STEP;
        } else {
            h *= (upper[j - NX] - lower[j - NX]);
            hinv = (real_t)1.0 / h;
            pr[j + 1] += h;
        }
        port_rk4(moved, pr, &pr[13], wind, STEP_LENGTH);
        port_state_to_delta(col, nominal, moved);
        for (int r = 0; r < NX; r++) jac[NZ * r + j] = col[r] * hinv;
    }
}

/* ------------------------------------------------------------- lifecycle */

unsigned port_real_bytes(void) { return (unsigned)sizeof(real_t); }

port_t *port_create(int horizon) {
    port_t *p = calloc(1, sizeof(*p));
    int N = horizon;
    p->N = N;
    p->xref = calloc((size_t)(N + 1) * NS, sizeof(real_t));
    p->uref = calloc((size_t)N * NU, sizeof(real_t));
    p->rec = calloc((size_t)N + 1, sizeof(stage_t));
    p->slot = calloc((size_t)N + 1, sizeof(int));
    p->lambda = calloc((size_t)N * NX, sizeof(real_t));
    p->dlambda = calloc((size_t)N * NX, sizeof(real_t));
    p->grad = calloc((size_t)N * NX, sizeof(real_t));
    p->gradTry = calloc((size_t)N * NX, sizeof(real_t));
    p->hess = calloc((size_t)N * 2 * NX * NX, sizeof(real_t));
    p->chol = calloc((size_t)N * 2 * NX * NX, sizeof(real_t));
    p->ieq = calloc((size_t)(N + 1) * NZ, sizeof(int));
    p->prev = calloc((size_t)(N + 1) * NZ, sizeof(int));
    for (int k = 0; k <= N; k++) p->slot[k] = k;
    return p;
}

void port_destroy(port_t *p) {
    if (!p) return;
    free(p->xref); free(p->uref); free(p->rec); free(p->slot);
    free(p->lambda); free(p->dlambda); free(p->grad); free(p->gradTry);
    free(p->hess); free(p->chol); free(p->ieq); free(p->prev);
    free(p);
}

void port_set_state_weights(port_t *p, const real_t *w) { memcpy(p->wq, w, sizeof p->wq); }
void port_set_control_weights(port_t *p, const real_t *w) { memcpy(p->wr, w, sizeof p->wr); }
void port_set_terminal_weights(port_t *p, const real_t *w) { memcpy(p->wt, w, sizeof p->wt); }
void port_set_lower_control_bound(port_t *p, const real_t *b) { memcpy(p->lb, b, sizeof p->lb); }
void port_set_upper_control_bound(port_t *p, const real_t *b) { memcpy(p->ub, b, sizeof p->ub); }
void port_set_wind_velocity(port_t *p, real_t x, real_t y, real_t z) {
    p->wind[0] = x; p->wind[1] = y; p->wind[2] = z;
}

static stage_t *stage_at(const port_t *p, int k) { return &p->rec[p->slot[k]]; }
static const real_t *stage_H(const port_t *p, int k) {
    return (k < p->N) ? p->Hreg : p->Hterm;
}

static void mark_data_changed(port_t *p) {
    /* setup_qp.c:39-50 */
    for (int i = 0; i < (p->N + 1) * NZ; i++) p->prev[i] = -42;
}

/* A diagonal weight block that is the identity to within 1e-15 carries no
 * data in the solver and behaves as exact ones (utils.c:106-117,
 * setup_qp.c:229-258). */
static int is_identity(const real_t *w, int n) {
    for (int i = 0; i < n; i++)
        if (fabs(w[i] - 1.0) > 1e-15) return 0;
    return 1;
}

/* setup_qp.c:547-585 for the clipping solver with a diagonal Hessian.
 * The stage's linear term g is identically zero in this controller. */
static void restage(port_t *p, int k) {
    stage_t *s = stage_at(p, k);
    const real_t *H = stage_H(p, k);
    int nV = s->nV, N = p->N;
    for (int i = 0; i < nV; i++) s->q[i] = 0.0;
    if (k < N) {
        const real_t *l1 = &p->lambda[k * NX];
        for (int j = 0; j < NZ; j++) s->qStep[j] = 0.;
        for (int i = 0; i < NX; i++)
            for (int j = 0; j < NZ; j++) s->qStep[j] += s->C[i * NZ + j] * l1[i];
        real_t ps = 0.;
        for (int i = 0; i < NX; i++) ps += l1[i] * s->c[i];
        s->pStep = ps;
    } else {
        for (int j = 0; j < nV; j++) s->qStep[j] = 0.0;
        s->pStep = 0.;
    }
    if (k > 0) {
        const real_t *l0 = &p->lambda[(k - 1) * NX];
        for (int i = 0; i < NX; i++) s->qStep[i] -= l0[i];
    }
    for (int i = 0; i < nV; i++) s->qStep[i] += 0.0;       /* + g */
    for (int i = 0; i < nV; i++) s->dz[i] = s->qStep[i] / H[i];
    for (int i = 0; i < nV; i++) s->dz[i] = -s->dz[i];
    for (int i = 0; i < nV; i++) s->zUnc[i] = 0.0;
}

void port_init(port_t *p) {
    /* cnmpc.c:872-948 */
    int N = p->N;
    for (int k = 0; k <= N; k++) {
        p->slot[k] = k;
        p->rec[k].nV = (k < N) ? NZ : NX;
        p->rec[k].fresh = 1;        /* no solution to linearise about yet */
    }
    mark_data_changed(p);

    int idq = is_identity(p->wq, NX), idr = is_identity(p->wr, NU);
    for (int i = 0; i < NX; i++) p->Hreg[i] = idq ? 1. : p->wq[i];
    for (int i = 0; i < NU; i++) p->Hreg[NX + i] = idr ? 1. : p->wr[i];
    int idt = is_identity(p->wt, NX);
    for (int i = 0; i < NX; i++) p->Hterm[i] = idt ? 1. : p->wt[i];

    for (int k = 0; k < N; k++) {
        stage_t *s = stage_at(p, k);
        memset(s->C, 0, sizeof s->C);
        memset(s->c, 0, sizeof s->c);
        for (int i = 0; i < NX; i++) { s->zLow[i] = -OPT_INFTY; s->zUpp[i] = OPT_INFTY; }
        for (int i = 0; i < NU; i++) { s->zLow[NX + i] = p->lb[i]; s->zUpp[NX + i] = p->ub[i]; }
    }
    stage_t *t = stage_at(p, N);
    for (int i = 0; i < NX; i++) { t->zLow[i] = -OPT_INFTY; t->zUpp[i] = OPT_INFTY; }

    for (int k = 0; k <= N; k++) restage(p, k);
    mark_data_changed(p);
}

void port_set_reference_point(port_t *p, const real_t *ref, unsigned i) {
    /* cnmpc.c:1022-1080 */
    memcpy(&p->xref[i * NS], ref, NS * sizeof(real_t));
    if (i > 0 && i <= (unsigned)p->N) {
        int k = (int)i - 1;
        real_t *ur = &p->uref[k * NU];
        stage_t *s = stage_at(p, k);
        memcpy(ur, &ref[NS], NU * sizeof(real_t));
        for (int j = 0; j < NX; j++) { s->zLow[j] = -OPT_INFTY; s->zUpp[j] = OPT_INFTY; }
        for (int j = 0; j < NU; j++) {
            s->zLow[NX + j] = p->lb[j] - ur[j];
            s->zUpp[NX + j] = p->ub[j] - ur[j];
        }
        port_linearise(&p->xref[k * NS], ur, ref, p->wind, p->lb, p->ub, s->C, s->c);
        restage(p, k);
        s->fresh = 1;
    }
}

/* ------------------------------------------------------- stage QP (clipping) */

static void clip_to_bounds(real_t *v, real_t *y, const real_t *lo,
                           const real_t *up, int n) {
    /* stage_qp_solver_clipping.c:174-198 */
    for (int i = 0; i < n; i++) {
        y[2 * i] = lo[i] - v[i];
        y[2 * i + 1] = v[i] - up[i];
        if (y[2 * i] >= -OPT_ACT_TOL) v[i] = lo[i];
        else if (y[2 * i + 1] >= -OPT_ACT_TOL) v[i] = up[i];
    }
}

/* Commit a step of length a (stage_qp_solver_clipping.c:134-165). */
static void stage_step(stage_t *s, real_t a) {
    int n = s->nV;
    for (int i = 0; i < n; i++) s->zUnc[i] = s->zUnc[i] + a * s->dz[i];
    for (int i = 0; i < n; i++) s->z[i] = s->zUnc[i];
    clip_to_bounds(s->z, s->y, s->zLow, s->zUpp, n);
    for (int i = 0; i < n; i++) s->q[i] = s->q[i] + a * s->qStep[i];
    s->p = s->p + a * s->pStep;
}

static real_t stage_objective(const real_t *H, const real_t *z, const real_t *q,
                              real_t p, int n) {
    real_t zHz = 0., qz = 0.;
    for (int j = 0; j < n; j++) zHz += H[j] * z[j] * z[j];
    real_t v = 0.5 * zHz;
    for (int j = 0; j < n; j++) qz += q[j] * z[j];
    v += qz;
    v += p;
    return v;
}

static real_t objective(port_t *p) {
    /* dual_qp.c:1542-1566 */
    real_t obj = 0.;
    for (int k = 0; k <= p->N; k++) {
        stage_t *s = stage_at(p, k);
        obj += stage_objective(stage_H(p, k), s->z, s->q, s->p, s->nV);
    }
    return obj;
}

/* dual_qp.c:1574-1617: trial step; overwrites z and y, leaves zUnc, q, p. */
static real_t trial_objective(port_t *p, real_t a) {
    real_t obj = 0.;
    for (int k = 0; k <= p->N; k++) {
        stage_t *s = stage_at(p, k);
        int n = s->nV;
        for (int i = 0; i < n; i++) s->z[i] = s->zUnc[i] + a * s->dz[i];
        clip_to_bounds(s->z, s->y, s->zLow, s->zUpp, n);
        for (int i = 0; i < n; i++) s->tmp[i] = s->q[i] + a * s->qStep[i];
        real_t pt = s->p + a * s->pStep;
        obj += stage_objective(stage_H(p, k), s->z, s->tmp, pt, n);
    }
    return obj;
}

static void active_set(port_t *p) {
    /* dual_qp.c:1627-1668 */
    for (int k = 0; k <= p->N; k++) {
        stage_t *s = stage_at(p, k);
        for (int i = 0; i < s->nV; i++) {
            int v;
            if (s->y[2 * i] > OPT_EQ_TOL) v = -1;
            else if (s->y[2 * i + 1] > OPT_EQ_TOL) v = 1;
            else v = 0;
            p->ieq[k * NZ + i] = v;
        }
    }
}

static int compare_active_sets(port_t *p) {
    /* dual_qp.c:1677-1699; returns the last stage whose set changed or -1 */
    int last = -1;
    for (int k = 0; k <= p->N; k++) {
        stage_t *s = stage_at(p, k);
        s->changed = 0;
        for (int i = 0; i < s->nV; i++)
            if (p->ieq[k * NZ + i] != p->prev[k * NZ + i]) { s->changed = 1; last = k; }
    }
    return last;
}

/* ---------------------------------------------------------- Newton system */

#define HS(M, k, l, i, j) (M)[(k) * 2 * NX * NX + (i) * 2 * NX + (1 + (l)) * NX + (j)]

static void stage_gradient(real_t *g, const stage_t *s, const real_t *zk,
                           const real_t *znext) {
    for (int i = 0; i < NX; i++) {
        real_t a = 0.;
        for (int j = 0; j < NZ; j++) a += s->C[i * NZ + j] * zk[j];
        g[i] = a;
    }
    for (int i = 0; i < NX; i++) g[i] += s->c[i];
    for (int i = 0; i < NX; i++) g[i] -= znext[i];
}

static real_t norm2(const real_t *v, int n) {
    real_t r = 0.;
    for (int i = 0; i < n; i++) r += v[i] * v[i];
    return sqrt(r);
}

static int setup_newton(port_t *p) {
    /* dual_qp.c:423-566 */
    int N = p->N;
    for (int k = 0; k < N; k++)
        stage_gradient(&p->grad[k * NX], stage_at(p, k), stage_at(p, k)->z,
                       stage_at(p, k + 1)->z);
    if (norm2(p->grad, NX * N) < OPT_STAT_TOL) return ST_OPTIMAL;

    real_t M[NX * NX], Zt[NZ * NX];
    for (int k = 0; k < N; k++) {
        stage_t *s = stage_at(p, k), *sn = stage_at(p, k + 1);
        if (!(s->changed || sn->changed)) continue;
        const real_t *Hn = stage_H(p, k + 1), *H = stage_H(p, k);
        for (int i = 0; i < NX * NX; i++) M[i] = 0.;
        for (int i = 0; i < NX; i++) {
            M[i * NX + i] = 1. / Hn[i];
            if (sn->y[2 * i] >= OPT_EQ_TOL || sn->y[2 * i + 1] >= OPT_EQ_TOL)
                M[i * NX + i] = 0.;
        }
        for (int l = 0; l < NZ; l++)
            for (int j = 0; j < NX; j++) Zt[l * NX + j] = s->C[j * NZ + l] / H[l];
        for (int l = 0; l < NZ; l++) {
            if (s->y[2 * l] <= OPT_EQ_TOL && s->y[2 * l + 1] <= OPT_EQ_TOL)
                for (int i = 0; i < NX; i++)
                    for (int j = 0; j < NX; j++)
                        M[i * NX + j] += s->C[i * NZ + l] * Zt[l * NX + j];
        }
        for (int i = 0; i < NX; i++)
            for (int j = 0; j < NX; j++) HS(p->hess, k, 0, i, j) = M[i * NX + j];
    }
    for (int k = 1; k < N; k++) {
        stage_t *s = stage_at(p, k);
        if (!s->changed) continue;
        const real_t *H = stage_H(p, k);
        for (int i = 0; i < NX; i++)
            for (int j = 0; j < NX; j++) {
                real_t v = s->C[i * NZ + j] / H[j];
                if (s->y[2 * j] <= OPT_EQ_TOL && s->y[2 * j + 1] <= OPT_EQ_TOL)
                    HS(p->hess, k, -1, i, j) = -v;
                else
                    HS(p->hess, k, -1, i, j) = 0.;
            }
    }
    return ST_OK;
}

static int factor_reverse(port_t *p, int last) {
    /* dual_qp.c:817-908 */
    int N = p->N;
    real_t *L = p->chol;
    const real_t *A = p->hess;
    int start = (last >= 0) ? ((last < N - 1) ? last : N - 1) : -1;
    for (int k = start; k >= 0; --k) {
        for (int j = NX - 1; j >= 0; --j) {
            real_t sum = HS(A, k, 0, j, j);
            for (int l = j + 1; l < NX; ++l) sum -= HS(L, k, 0, l, j) * HS(L, k, 0, l, j);
            if (k < N - 1)
                for (int l = 0; l < NX; ++l)
                    sum -= HS(L, k + 1, -1, l, j) * HS(L, k + 1, -1, l, j);
            if (sum < 1.e2 * OPT_EQ_TOL) return ST_DIV_ZERO;
            HS(L, k, 0, j, j) = sqrt(sum);
            for (int i = j - 1; i >= 0; --i) {
                sum = HS(A, k, 0, j, i);
                for (int l = j + 1; l < NX; ++l) sum -= HS(L, k, 0, l, i) * HS(L, k, 0, l, j);
                if (k < N - 1)
                    for (int l = 0; l < NX; ++l)
                        sum -= HS(L, k + 1, -1, l, i) * HS(L, k + 1, -1, l, j);
                HS(L, k, 0, j, i) = sum / HS(L, k, 0, j, j);
            }
            if (k > 0)
                for (int i = NX - 1; i >= 0; --i) {
                    sum = HS(A, k, -1, j, i);
                    for (int l = j + 1; l < NX; ++l)
                        sum -= HS(L, k, -1, l, i) * HS(L, k, 0, l, j);
                    HS(L, k, -1, j, i) = sum / HS(L, k, 0, j, j);
                }
        }
    }
    return ST_OK;
}

static int factor_newton(port_t *p, int last) {
    /* dual_qp.c:601-698 */
    int N = p->N;
    real_t minDiag = OPT_INFTY;
    int st = factor_reverse(p, last);
    if (st == ST_OK)
        for (int k = 0; k < N; k++)
            for (int i = 0; i < NX; i++)
                if (minDiag > HS(p->chol, k, 0, i, i)) minDiag = HS(p->chol, k, 0, i, i);
    if (st == ST_DIV_ZERO || minDiag < OPT_DIAG_REG_TOL) {
        for (int k = 0; k < N; k++)
            for (int j = 0; j < NX; j++) HS(p->hess, k, 0, j, j) += OPT_REG_PARAM;
        p->regularized = 1;
        st = factor_reverse(p, N + 1);
        if (st != ST_OK) return st;
    } else if (st != ST_OK) {
        return st;
    }
    return ST_OK;
}

static void solve_newton(port_t *p) {
    /* dual_qp.c:993-1055 */
    int N = p->N;
    const real_t *L = p->chol, *g = p->grad;
    real_t *x = p->dlambda;
    for (int k = N - 1; k >= 0; --k)
        for (int i = NX - 1; i >= 0; --i) {
            real_t sum = g[k * NX + i];
            for (int j = i + 1; j < NX; ++j) sum -= HS(L, k, 0, j, i) * x[k * NX + j];
            if (k < N - 1)
                for (int j = 0; j < NX; ++j) sum -= HS(L, k + 1, -1, j, i) * x[(k + 1) * NX + j];
            x[k * NX + i] = sum / HS(L, k, 0, i, i);
        }
    for (int k = 0; k < N; ++k)
        for (int i = 0; i < NX; ++i) {
            real_t sum = x[k * NX + i];
            if (k > 0)
                for (int j = 0; j < NX; ++j) sum -= HS(L, k, -1, i, j) * x[(k - 1) * NX + j];
            for (int j = 0; j < i; ++j) sum -= HS(L, k, 0, i, j) * x[k * NX + j];
            x[k * NX + i] = sum / HS(L, k, 0, i, i);
        }
}

/* dual_qp.c:309-379 with stage_qp_solver_clipping.c:42-98 */
static void stage_directions(port_t *p, const real_t *dl) {
    int N = p->N;
    for (int k = 0; k <= N; k++) {
        stage_t *s = stage_at(p, k);
        int n = s->nV;
        if (k < N) {
            const real_t *l1 = &dl[k * NX];
            for (int j = 0; j < NZ; j++) s->qStep[j] = 0.;
            for (int i = 0; i < NX; i++)
                for (int j = 0; j < NZ; j++) s->qStep[j] += s->C[i * NZ + j] * l1[i];
            real_t ps = 0.;
            for (int i = 0; i < NX; i++) ps += l1[i] * s->c[i];
            s->pStep = ps;
        } else {
            for (int j = 0; j < n; j++) s->qStep[j] = 0.0;
            s->pStep = 0.;
        }
        if (k > 0) {
            const real_t *l0 = &dl[(k - 1) * NX];
            for (int i = 0; i < NX; i++) s->qStep[i] -= l0[i];
        }
    }
    for (int k = 0; k <= N; k++) {
        stage_t *s = stage_at(p, k);
        const real_t *H = stage_H(p, k);
        for (int i = 0; i < s->nV; i++) s->dz[i] = s->qStep[i] / H[i];
        for (int i = 0; i < s->nV; i++) s->dz[i] = -s->dz[i];
    }
}

/* -------------------------------------------------------------- line search */

static int backtracking(port_t *p, real_t *alpha, int *it, real_t alphaMax,
                        real_t objInc) {
    /* dual_qp.c:1294-1334 with alphaMin = 0 */
    real_t minProg = OPT_LS_MIN_REL * fabs(objInc) + OPT_LS_MIN_ABS;
    real_t nrm = norm2(p->dlambda, NX * p->N);
    *alpha = alphaMax;
    for (; *it < OPT_MAX_LS_ITER; ++(*it)) {
        real_t obj = trial_objective(p, *alpha);
        p->nBacktrack++;
        if (obj > objInc + minProg) return ST_OK;
        *alpha = (*alpha) * OPT_LS_REDUCTION;
        if (nrm * (*alpha - 0.) < OPT_EQ_TOL) { *alpha = 0.; return ST_LS_MIN_STEP; }
    }
    return ST_LS_MAXITER;
}

static real_t trial_slope(port_t *p, real_t a) {
    int N = p->N;
    for (int k = 0; k <= N; k++) {
        stage_t *s = stage_at(p, k);
        for (int i = 0; i < s->nV; i++) s->tmp[i] = s->zUnc[i] + a * s->dz[i];
        clip_to_bounds(s->tmp, s->y, s->zLow, s->zUpp, s->nV);
    }
    for (int k = 0; k < N; k++)
        stage_gradient(&p->gradTry[k * NX], stage_at(p, k), stage_at(p, k)->tmp,
                       stage_at(p, k + 1)->tmp);
    real_t r = 0.;
    for (int i = 0; i < NX * N; i++) r += p->gradTry[i] * p->dlambda[i];
    return r;
}

static int bisection(port_t *p, real_t *alpha, int *it, real_t alphaMin,
                     real_t alphaMax) {
    /* dual_qp.c:1341-1470 */
    real_t alphaC = 1.0, slope;
    real_t nd = norm2(p->dlambda, NX * p->N);
    real_t normal = (1. < nd) ? 1. : nd;
    for (; *it < OPT_MAX_LS_REFINE; ++(*it)) {
        slope = trial_slope(p, alphaMax);
        p->nBisect++;
        if (fabs(slope / normal) <= OPT_LS_STAT_TOL) { *alpha = alphaMax; return ST_OK; }
        if (slope / normal < 0.) break;
        alphaMin = alphaMax;
        alphaMax *= OPT_LS_INCREASE;
        if (alphaMax > OPT_LS_MAX_STEP) { *alpha = alphaMin; return ST_LS_MAX_STEP; }
    }
    for (; *it < OPT_MAX_LS_REFINE; ++(*it)) {
        alphaC = 0.5 * (alphaMin + alphaMax);
        slope = trial_slope(p, alphaC);
        p->nBisect++;
        if (fabs(slope / normal) <= OPT_LS_STAT_TOL) { *alpha = alphaC; return ST_OK; }
        if (slope > 0) alphaMin = alphaC; else alphaMax = alphaC;
    }
    *alpha = alphaC;
    return ST_LS_MAXITER;
}

static int step_length(port_t *p, real_t *objInc) {
    /* dual_qp.c:1119-1286 */
    int N = p->N, n = NX * N, st;
    real_t alphaMin = OPT_INFTY, alphaMax = 1.;
    p->lsIter = 0;
    for (int k = 0; k <= N; k++) {
        stage_t *s = stage_at(p, k);
        for (int i = 0; i < s->nV; i++) {
            real_t r0 = s->dz[i] / s->y[2 * i], r1 = s->dz[i] / -s->y[2 * i + 1];
            real_t a = 1. / ((r0 > r1) ? r0 : r1);
            if (a > 0. && a < alphaMin) alphaMin = a;
        }
    }
    if (alphaMin > 1. - OPT_EQ_TOL && !p->regularized) {
        p->alpha = 1.;
        p->nFullStep++;
        for (int i = 0; i < n; i++) p->lambda[i] = p->lambda[i] + p->alpha * p->dlambda[i];
        for (int k = 0; k <= N; k++) stage_step(stage_at(p, k), p->alpha);
        *objInc = objective(p);
        return ST_OK;
    }
    st = backtracking(p, &p->alpha, &p->lsIter, alphaMax, *objInc);
    if (st == ST_LS_MIN_STEP) return st;
    {
        real_t cand = p->alpha / OPT_LS_REDUCTION;
        alphaMax = (alphaMax < cand) ? alphaMax : cand;
    }
    st = bisection(p, &p->alpha, &p->lsIter, alphaMin, alphaMax);
    for (int i = 0; i < n; i++) p->lambda[i] += p->alpha * p->dlambda[i];
    for (int k = 0; k <= N; k++) stage_step(stage_at(p, k), p->alpha);
    *objInc = objective(p);
    return st;
}

/* ------------------------------------------------------------------ solve */

static int dual_newton_solve(port_t *p) {
    /* dual_qp.c:39-223 */
    int N = p->N, last, st;
    real_t objInc;
    p->numIter = 0;
    p->nBacktrack = p->nBisect = p->nFullStep = 0;
    for (int k = 0; k <= N; k++) stage_step(stage_at(p, k), 1);
    objInc = objective(p);
    active_set(p);
    last = compare_active_sets(p);

    for (p->numIter = 1; p->numIter <= OPT_MAX_ITER; ++p->numIter) {
        p->regularized = 0;
        st = setup_newton(p);
        if (st != ST_OK) return st;
        st = factor_newton(p, last);
        if (st != ST_OK) return st;
        solve_newton(p);
        stage_directions(p, p->dlambda);
        st = step_length(p, &objInc);
        if (st == ST_LS_MIN_STEP) return ST_NO_ASCENT;
        memcpy(p->prev, p->ieq, sizeof(int) * (size_t)(N + 1) * NZ);
        active_set(p);
        last = compare_active_sets(p);
    }
    p->numIter = OPT_MAX_ITER;
    return ST_ITER_LIMIT;
}

void port_feedback_step(port_t *p, const real_t *meas) {
    /* cnmpc.c:675-726, 953-956 */
    stage_t *s0 = stage_at(p, 0);
    real_t d[NX];
    port_state_to_delta(d, p->xref, meas);
    for (int i = 0; i < NX; i++) { s0->zLow[i] = d[i]; s0->zUpp[i] = d[i]; }
    for (int i = 0; i < NU; i++) {
        s0->zLow[NX + i] = p->lb[i] - p->uref[i];
        s0->zUpp[NX + i] = p->ub[i] - p->uref[i];
    }
    mark_data_changed(p);
    for (int k = 0; k < p->N; k++) stage_at(p, k)->fresh = 0;
    p->lastStatus = dual_newton_solve(p);
    if (p->lastStatus == ST_OPTIMAL) {
        for (int i = 0; i < NU; i++) p->u[i] = p->uref[i] + s0->z[NX + i];
        p->lastResult = 1;
    } else {
        p->lastResult = 0;
    }
}

/*
 * Real-time-iteration preparation step.  The reference leaves its preparation
 * step empty (cnmpc.c:950-951, src/ocp.cpp:245-250) although its comments give
 * it the re-linearisation of the horizon (src/ocp.cpp:315-322); this is that
 * work, defined here because the reference holds no code for it: every stage
 * that carries a solution of the last feedback step (all but those a reference
 * point has re-linearised since) is linearised about that solution instead of
 * about the reference trajectory,
 *     x = x_ref[k] (+) z_k[0:12],  u = u_ref[k] + z_k[12:15],
 * where (+) adds position, velocity and body rates and applies the attitude
 * deviation as the MRP (a = 1, f = 4) quaternion multiplied from the left, the
 * composition the finite differences use (cnmpc.c:617-634) with the scalar part
 * in its intended form (16 - p.p) / (16 + p.p) (src/ocp.cpp:122-126).  The stage
 * keeps its meaning as a deviation from the reference:
 *     C_k = d f / d (x, u) at (x, u),   c_k = delta(x_ref[k+1], f(x, u)) - C_k z_k.
 * Stages without a solution are linearised about the reference again.  The
 * multipliers restart from zero, every stage is re-staged against them
 * (setup_qp.c:547-585) and the constant terms of the stage objectives return to
 * zero: multipliers of the old linearisation put the stage solutions into the
 * control bounds all along the horizon, which costs the dual Newton iteration
 * more than its five steps, while from zero it needs two to four.  When the
 * last feedback step did not end with NMPC_OK its stage variables are no
 * solution, and the stages are linearised at zero deviation instead.
 */
void port_state_plus_delta(real_t *out, const real_t *x, const real_t *d) {
    for (int i = 0; i < 3; i++) {
        out[i] = x[i] + d[i];
        out[3 + i] = x[3 + i] + d[3 + i];
        out[10 + i] = x[10 + i] + d[9 + i];
    }
    real_t pp = d[6] * d[6] + d[7] * d[7] + d[8] * d[8];
    real_t dq[4];
    dq[3] = ((real_t)16.0 - pp) / ((real_t)16.0 + pp);
    real_t sc = (real_t)0.25 * ((real_t)1.0 + dq[3]);
    dq[0] = sc * d[6]; dq[1] = sc * d[7]; dq[2] = sc * d[8];
    quat_mul(&out[6], dq, &x[6]);
}

void port_preparation_step_rti(port_t *p) {
    int N = p->N, any = 0;
    for (int k = 0; k < N; k++) any |= !stage_at(p, k)->fresh;
    if (!any) return;
    /* An unconverged solve left no solution to linearise about, and its
     * multipliers are the progress made so far: nothing changes, the next
     * feedback step carries on from there as it does without this step. */
    if (!p->lastResult) return;
    for (int k = 0; k < N; k++) {
        stage_t *s = stage_at(p, k);
        if (s->fresh) {     /* entered the horizon since: about the reference, as it was */
            port_linearise(&p->xref[k * NS], &p->uref[k * NU], &p->xref[(k + 1) * NS], p->wind, p->lb,
                           p->ub, s->C, s->c);
        } else {
            real_t x[NS], u[NU];
            port_state_plus_delta(x, &p->xref[k * NS], s->z);
            for (int i = 0; i < NU; i++) u[i] = p->uref[k * NU + i] + s->z[NX + i];
            port_linearise(x, u, &p->xref[(k + 1) * NS], p->wind, p->lb, p->ub, s->C, s->c);
            for (int r = 0; r < NX; r++) {
                real_t acc = 0.;
                for (int j = 0; j < NZ; j++) acc += s->C[r * NZ + j] * s->z[j];
                s->c[r] = s->c[r] - acc;
            }
        }
    }
    /* the duals, and the constant terms that follow them, start afresh */
    memset(p->lambda, 0, sizeof(real_t) * (size_t)N * NX);
    for (int k = 0; k <= N; k++) { stage_at(p, k)->p = 0.; restage(p, k); }
    for (int k = 0; k < N; k++) stage_at(p, k)->fresh = 1;
    mark_data_changed(p);
}

int port_get_controls(port_t *p, real_t *u) {
    memcpy(u, p->u, sizeof p->u);
    return p->lastResult ? 0 : 1;
}

void port_update_horizon(port_t *p, const real_t *ref) {
    /* cnmpc.c:971-986, setup_qp.c:619-662 */
    int N = p->N;
    memmove(p->xref, p->xref + NS, sizeof(real_t) * NS * (size_t)N);
    memmove(p->uref, p->uref + NU, sizeof(real_t) * NU * (size_t)(N - 1));
    for (int k = 0; k < N - 1; k++)
        for (int i = 0; i < NX; i++) p->lambda[k * NX + i] = p->lambda[(k + 1) * NX + i];
    int freed = p->slot[0];
    for (int k = 0; k < N - 1; k++) p->slot[k] = p->slot[k + 1];
    p->slot[N - 1] = freed;
    port_set_reference_point(p, ref, (unsigned)N);
}

/* ------------------------------------------------------------ introspection */

int port_horizon(const port_t *p) { return p->N; }
int port_num_iter(const port_t *p) { return p->numIter; }
int port_last_result(const port_t *p) { return p->lastResult; }
int port_last_status(const port_t *p) { return p->lastStatus; }
real_t port_alpha(const port_t *p) { return p->alpha; }
int port_line_search_iter(const port_t *p) { return p->lsIter; }
int port_hessian_regularized(const port_t *p) { return p->regularized; }
void port_trial_counts(const port_t *p, int *out3) {
    out3[0] = p->nBacktrack; out3[1] = p->nBisect; out3[2] = p->nFullStep;
}

void port_get_stage(const port_t *p, unsigned k, real_t *o) {
    const stage_t *s = stage_at(p, (int)k);
    size_t nv = (size_t)s->nV;
    memset(o, 0, 332 * sizeof(real_t));
    if ((int)k < p->N) {
        memcpy(o, s->C, sizeof s->C);
        memcpy(o + 180, s->c, sizeof s->c);
    }
    memcpy(o + 192, s->zLow, nv * sizeof(real_t));
    memcpy(o + 207, s->zUpp, nv * sizeof(real_t));
    memcpy(o + 222, s->q, nv * sizeof(real_t));
    o[237] = s->p;
    memcpy(o + 238, s->z, nv * sizeof(real_t));
    memcpy(o + 253, s->y, 2 * nv * sizeof(real_t));
    memcpy(o + 283, s->qStep, nv * sizeof(real_t));
    o[298] = s->pStep;
    memcpy(o + 299, s->zUnc, nv * sizeof(real_t));
    memcpy(o + 314, s->dz, nv * sizeof(real_t));
}

unsigned port_global_size(const port_t *p) {
    unsigned N = (unsigned)p->N;
    return 3 * 12 * N + 2 * 288 * N + 2 * 15 * (N + 1) + 13 * (N + 1) + 3 * N + 3;
}

void port_get_global(const port_t *p, real_t *o) {
    size_t N = (size_t)p->N;
    memcpy(o, p->lambda, 12 * N * sizeof(real_t)); o += 12 * N;
    memcpy(o, p->dlambda, 12 * N * sizeof(real_t)); o += 12 * N;
    memcpy(o, p->grad, 12 * N * sizeof(real_t)); o += 12 * N;
    memcpy(o, p->hess, 288 * N * sizeof(real_t)); o += 288 * N;
    memcpy(o, p->chol, 288 * N * sizeof(real_t)); o += 288 * N;
    for (size_t k = 0; k <= N; k++)
        for (int i = 0; i < 15; i++)
            *o++ = (i < stage_at(p, (int)k)->nV) ? (real_t)p->ieq[k * 15 + i] : 0.0;
    for (size_t k = 0; k <= N; k++)
        for (int i = 0; i < 15; i++)
            *o++ = (i < stage_at(p, (int)k)->nV) ? (real_t)p->prev[k * 15 + i] : 0.0;
    memcpy(o, p->xref, 13 * (N + 1) * sizeof(real_t)); o += 13 * (N + 1);
    memcpy(o, p->uref, 3 * N * sizeof(real_t)); o += 3 * N;
    memcpy(o, p->u, 3 * sizeof(real_t));
}

/* ---------------------------------------------------------- batched driver */

typedef struct {
    int horizon, first, last;
    const real_t *wq, *wr, *wt, *lo, *up, *refs, *meas, *wind;
    real_t *u;
    int *status, *iters;
} job_t;

static void *cold_worker(void *arg) {
    job_t *j = arg;
    int N = j->horizon;
    for (int b = j->first; b < j->last; b++) {
        port_t *p = port_create(N);
        port_set_state_weights(p, j->wq);
        port_set_control_weights(p, j->wr);
        port_set_terminal_weights(p, j->wt);
        port_set_lower_control_bound(p, j->lo);
        port_set_upper_control_bound(p, j->up);
        port_init(p);
        const real_t *w = &j->wind[(size_t)b * 3];
        port_set_wind_velocity(p, w[0], w[1], w[2]);
        const real_t *r = &j->refs[(size_t)b * (size_t)(N + 1) * NREF];
        for (int i = 0; i <= N; i++) port_set_reference_point(p, &r[i * NREF], (unsigned)i);
        port_feedback_step(p, &j->meas[(size_t)b * NS]);
        j->status[b] = port_get_controls(p, &j->u[(size_t)b * 3]);
        j->iters[b] = p->numIter;
        port_destroy(p);
    }
    return 0;
}

double port_batch_cold_solve(int horizon, int count, int threads,
                             const real_t *wq, const real_t *wr, const real_t *wt,
                             const real_t *lo, const real_t *up, const real_t *refs,
                             const real_t *meas, const real_t *wind, real_t *u,
                             int *status, int *iters) {
    if (threads < 1) threads = 1;
    if (threads > count) threads = count;
    pthread_t *tid = calloc((size_t)threads, sizeof(*tid));
    job_t *jobs = calloc((size_t)threads, sizeof(*jobs));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; t++) {
        job_t j = {horizon, (int)((long)count * t / threads),
                   (int)((long)count * (t + 1) / threads),
                   wq, wr, wt, lo, up, refs, meas, wind, u, status, iters};
        jobs[t] = j;
        pthread_create(&tid[t], 0, cold_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(tid[t], 0);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(tid); free(jobs);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ---- batched stage linearisation (CPU baseline of the sample workload) ---- */
typedef struct {
    long first, last;
    const real_t *x, *u, *xn, *w, *lo, *up;
    real_t *jac, *res;
} lin_job_t;

static void *lin_worker(void *arg) {
    const lin_job_t *j = (const lin_job_t *)arg;
    for (long i = j->first; i < j->last; i++)
        port_linearise(j->x + 13 * i, j->u + 3 * i, j->xn + 13 * i, j->w + 3 * i, j->lo, j->up,
                       j->jac + 180 * i, j->res + 12 * i);
    return 0;
}

double port_batch_linearise(long count, int threads, const real_t *x, const real_t *u,
                            const real_t *xnext, const real_t *wind, const real_t *lower3,
                            const real_t *upper3, real_t *jac, real_t *res) {
    if (threads < 1) threads = 1;
    if (threads > count) threads = (int)count;
    pthread_t *tid = calloc((size_t)threads, sizeof(*tid));
    lin_job_t *jobs = calloc((size_t)threads, sizeof(*jobs));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; t++) {
        lin_job_t j = {count * t / threads, count * (t + 1) / threads,
                       x, u, xnext, wind, lower3, upper3, jac, res};
        jobs[t] = j;
        pthread_create(&tid[t], 0, lin_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(tid[t], 0);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(tid); free(jobs);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
