// Device-side aircraft model, integrator and state difference for gfx950.
//
// Every expression keeps the operand order, association and float-rounded
// literals of the reference's pure-C controller so that, compiled without FMA
// contraction, the results are bit-identical to the reference fp64 CPU path:
//   quaternion helpers   ccs-c66x/cnmpc.c:168-240
//   fatan2               ccs-c66x/cnmpc.c:257-280
//   kinematics           ccs-c66x/cnmpc.c:320-363   [src/state.cpp:39-63]
//   integrators          ccs-c66x/cnmpc.c:365-400   [include/integrator.h:46-91]
//   X8 aerodynamics      ccs-c66x/cnmpc.c:402-539   [src/dynamics.cpp:33-118]
//   state difference     ccs-c66x/cnmpc.c:541-579   [src/ocp.cpp:64-88]
#pragma once
#include <hip/hip_runtime.h>

#include "layout.hpp"

#pragma clang fp contract(off)

namespace nmpc {

constexpr int NS = 13;   // state: pos3 vel3 quat(xyzw)4 omega3
constexpr int NU = 3;    // control: throttle, left elevon, right elevon
constexpr int NX = 12;   // delta state (quaternion -> 3 MRP)
constexpr int NZ = 15;   // stage variable: delta state + control

// Literals keep the types they have in the reference: float-suffixed ones are
// float in both precisions, real_t-cast ones follow the build, and the few
// plain double literals (with the double-valued sqrt the reference calls on
// real_t operands) make their expression double in both.
#ifdef NMPC_SINGLE_PRECISION
constexpr real_t kFdStep = (real_t)1.857e-2;        // cnmpc.c:152
#else
constexpr real_t kFdStep = (real_t)1.221e-4;        // cnmpc.c:154
#endif
constexpr real_t kStepLength = (real_t)(1.0 / 50.0);    // include/config.h:60
constexpr real_t kGravity = (real_t)9.80665;
constexpr real_t kRho = (real_t)1.225;
constexpr bool kDoublePrecision = sizeof(real_t) == 8;

#define NMPC_DEV __device__ __forceinline__

// Hamilton product, xyzw storage.
NMPC_DEV void quat_mul(real_t* r, const real_t* a, const real_t* b) {
    r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    r[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    r[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
}

// v' = q v q*, in the t = 2 cross(q.xyz, v) form with the reference's
// statement order.
NMPC_DEV void quat_rotate(real_t* r, const real_t* q, const real_t* v) {
    real_t tx, ty, tz, rx, ry, rz;
    tx = q[1] * v[2];
    ty = q[2] * v[0];
    tx -= q[2] * v[1];
    ty -= q[0] * v[2];
    tz = q[0] * v[1];
    ty *= (real_t)2.0;      // (2.0 is a double literal there; doubling is exact either way)
    tz -= q[1] * v[0];
    tx *= (real_t)2.0;
    tz *= (real_t)2.0;
    rx = v[0]; rx += q[3] * tx; rx += q[1] * tz; rx -= q[2] * ty;
    ry = v[1]; ry += q[3] * ty; ry += q[2] * tx; ry -= q[0] * tz;
    rz = v[2]; rz += q[3] * tz; rz -= q[1] * tx; rz += q[0] * ty;
    r[0] = rx; r[1] = ry; r[2] = rz;
}

// Rational atan2 approximation: float arguments and result; the constants are
// real_t, so the intermediates are double in the fp64 configuration and float
// in the single-precision one.
NMPC_DEV float approx_atan2(float y, float x) {
    const real_t pi = (real_t)3.14159265, pi_2 = (real_t)1.5707963;
    if (x == (real_t)0.0) {
        if (y > (real_t)0.0) return (float)pi_2;
        if (y == (real_t)0.0) return 0.0f;
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

// d = sqrt(x) and y = RN(1 / d) from one iteration.  The root is the compiler's
// own expansion of sqrt (v_rsq_f64, one Goldschmidt step, two corrections)
// without the rescaling it applies to x < 2^-767.  That iteration carries
// h ~ 1 / (2 d) along; refined once more beside the corrections of the root,
// it starts one Newton step and Markstein's correction, which give the
// correctly rounded reciprocal of every d whose significand is not all ones.
// Checked against sqrt, 1 / d and n / d on 2.6e10 random and adversarial
// operands with exponents within +-400 on gfx950 (tools/micro_sqrt.hip): no
// difference.  Callers keep x inside that range and handle the all-ones case.
struct Pivot { double d, y; };
NMPC_DEV Pivot root_and_reciprocal(double x) {
    const double r = __builtin_amdgcn_rsq(x);
    double g = x * r, h = r * 0.5;
    const double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    h = __builtin_fma(h, e, h);
    const double e2 = __builtin_fma(-h, g, 0.5);
    const double h2 = __builtin_fma(h, e2, h);
    const double d0 = __builtin_fma(-g, g, x);
    g = __builtin_fma(d0, h, g);
    const double d1 = __builtin_fma(-g, g, x);
    g = __builtin_fma(d1, h, g);
    const double y0 = h2 + h2;
    const double e0 = __builtin_fma(-g, y0, 1.0);
    const double y1 = __builtin_fma(e0, y0, y0);
    const double e1 = __builtin_fma(-g, y1, 1.0);
    return {g, __builtin_fma(e1, y1, y1)};
}
NMPC_DEV bool significand_all_ones(double d) {
    return (__double_as_longlong(d) & 0xfffffffffffffLL) == 0xfffffffffffffLL;
}

NMPC_DEV real_t sel_max(real_t a, real_t b) { return (a > b) ? a : b; }
NMPC_DEV real_t sel_abs(real_t x) { return x < 0 ? -x : x; }

// Body-frame linear (0..2) and angular (3..5) acceleration.
NMPC_DEV void x8_dynamics(real_t* out, const real_t* s, const real_t* u,
                          const real_t* wind) {
    real_t ned[3], af[3];
    ned[0] = wind[0] - s[3];
    ned[1] = wind[1] - s[4];
    ned[2] = wind[2] - s[5];
    quat_rotate(af, &s[6], ned);

    real_t rx, ry, rz, tx, ty;
    tx = s[7] * (kGravity * 2.0f);
    ty = -s[6] * (kGravity * 2.0f);
    rx = s[9] * tx;
    ry = s[9] * ty;
    ry += s[8] * tx;
    rx -= s[8] * ty;
    rz = kGravity;
    rz -= s[7] * tx;
    rz += s[6] * ty;
    out[0] = rx; out[1] = ry; out[2] = rz;

    real_t ax2 = af[0] * af[0], ay2 = af[1] * af[1], az2 = af[2] * af[2];
    real_t rpm = u[0] * 25000.0f;
    real_t ve2 = (0.0025f * 0.0025f) * rpm * rpm;
    real_t thrust = sel_max(0.0f, ve2 - ax2) *
                    (0.26315789473684f * 0.5f * kRho * 0.025f);

    real_t hv2 = ax2 + ay2;
    real_t qbar = (kRho * 0.5f) * hv2;
    // The reference takes these in double whatever real_t is:
    //   v_inv = (real_t)(1.0 / sqrt(max(1, |airflow|^2))), vv = (real_t)sqrt(ax^2 + az^2),
    //   vv_inv = (real_t)(1.0 / max(1, vv)).
    const double s1 = sel_max(1.0f, hv2 + az2), s2 = ax2 + az2;
    real_t v_inv, vv, vv_inv;
#ifndef NMPC_PLAIN_ROOTS
    // two roots and two divisions taken as two root-and-reciprocal iterations
    // (2 % of the stage linearisation); operands outside the range that form is
    // checked for (zero airflow among them) and roots with an all-ones
    // significand take sqrt and / instead
    const Pivot p1 = root_and_reciprocal(s1), p2 = root_and_reciprocal(s2);
    v_inv = (real_t)p1.y;
    vv = (real_t)p2.d;
    if constexpr (kDoublePrecision) vv_inv = (1.0f > vv) ? (real_t)1.0 : (real_t)p2.y;
    else vv_inv = (real_t)(1.0 / (double)sel_max(1.0f, vv));
    const bool checked = s1 <= 0x1p400 && s2 >= 0x1p-400 && s2 <= 0x1p400 &&
                         !significand_all_ones(p1.d) && !significand_all_ones(p2.d);
    if (__builtin_expect(!checked, 0)) {
        v_inv = (real_t)(1.0 / sqrt(s1));
        vv = (real_t)sqrt(s2);
        vv_inv = (real_t)(1.0 / (double)sel_max(1.0f, vv));
    }
#else
    v_inv = (real_t)(1.0 / sqrt(s1));
    vv = (real_t)sqrt(s2);
    vv_inv = (real_t)(1.0 / (double)sel_max(1.0f, vv));
#endif

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

    real_t la = u[1] - 0.5f, ra = u[2] - 0.5f;
    real_t pm = 0.001f - 0.1f * sca - 0.003f * s[11] - 0.04f * (la + ra);
    real_t rm = 0.03f * sin_b - 0.015f * s[10] + 0.1f * (la - ra);
    real_t ym = -0.02f * sin_b - 0.05f * s[12] -
                0.01f * (sel_abs(la) + sel_abs(ra));
    pm *= qbar; rm *= qbar; ym *= qbar;

    out[4] = 5.8823528f * pm;
    out[3] = (3.364222f * rm + 0.27744448f * ym);
    out[5] = (0.27744448f * rm + 2.4920163f * ym);
}

// State derivative.
NMPC_DEV void state_model(real_t* out, const real_t* s, const real_t* u,
                          const real_t* wind) {
    real_t acc[6], a[4], w[4];
    x8_dynamics(acc, s, u, wind);
    out[0] = s[3]; out[1] = s[4]; out[2] = s[5];
    a[0] = s[6]; a[1] = s[7]; a[2] = s[8]; a[3] = -s[9];
    quat_rotate(&out[3], a, acc);
    a[3] = -a[3];
    w[0] = -s[10]; w[1] = -s[11]; w[2] = -s[12]; w[3] = (real_t)0.0;
    quat_mul(&out[6], w, a);
    out[6] *= (real_t)0.5; out[7] *= (real_t)0.5; out[8] *= (real_t)0.5; out[9] *= (real_t)0.5;
    out[10] = acc[3]; out[11] = acc[4]; out[12] = acc[5];
}

// One integration step.  ta, tb (optional): the increments of the position
// entries, out[i] = (s[i] + ta[i]) + tb[i] for i < 3 (tb only when
// kIntegratorTerms == 2).
//
// Default: classical RK4 with the reference's float-rounded 1/3 and 1/6
// weights, x+ = x + d3 (b + c) + d6 (a + d)  (ccs-c66x/cnmpc.c:365-400).
// -DNMPC_INTEGRATOR_HEUN / -DNMPC_INTEGRATOR_EULER select the two cheaper
// integrators of the reference's compile-time matrix (include/config.h:30-39).
// They exist only in its C++ variant (include/integrator.h:66-91), whose
// formulas these follow in real_t arithmetic; the C path has RK4 only.
#if defined(NMPC_INTEGRATOR_HEUN)
constexpr int kIntegratorTerms = 1;
NMPC_DEV void integrate_step(real_t* out, const real_t* s, const real_t* u,
                             const real_t* wind, real_t delta, real_t* ta = nullptr,
                             real_t* tb = nullptr) {
    real_t k1[NS], k2[NS], t[NS];
    state_model(k1, s, u, wind);
#pragma unroll
    for (int i = 0; i < NS; i++) t[i] = s[i] + delta * k1[i];          // predictor
    state_model(k2, t, u, wind);
    const real_t half = delta * (real_t)0.5;
#pragma unroll
    for (int i = 0; i < NS; i++) {
        const real_t a = half * (k1[i] + k2[i]);
        out[i] = s[i] + a;
        if (i < 3 && ta) ta[i] = a;
    }
}
#elif defined(NMPC_INTEGRATOR_EULER)
constexpr int kIntegratorTerms = 1;
NMPC_DEV void integrate_step(real_t* out, const real_t* s, const real_t* u,
                             const real_t* wind, real_t delta, real_t* ta = nullptr,
                             real_t* tb = nullptr) {
    real_t k1[NS];
    state_model(k1, s, u, wind);
#pragma unroll
    for (int i = 0; i < NS; i++) {
        const real_t a = delta * k1[i];
        out[i] = s[i] + a;
        if (i < 3 && ta) ta[i] = a;
    }
}
#else
constexpr int kIntegratorTerms = 2;
NMPC_DEV void integrate_step(real_t* out, const real_t* s, const real_t* u,
                             const real_t* wind, real_t delta, real_t* ta = nullptr,
                             real_t* tb = nullptr) {
    real_t k[NS], t[NS], bc[NS], ad[NS];
    state_model(k, s, u, wind);
#pragma unroll
    for (int i = 0; i < NS; i++) { ad[i] = k[i]; t[i] = s[i] + k[i] * (delta * 0.5f); }
    state_model(k, t, u, wind);
#pragma unroll
    for (int i = 0; i < NS; i++) { bc[i] = k[i]; t[i] = s[i] + k[i] * (delta * 0.5f); }
    state_model(k, t, u, wind);
#pragma unroll
    for (int i = 0; i < NS; i++) { bc[i] = bc[i] + k[i]; t[i] = s[i] + k[i] * delta; }
    state_model(k, t, u, wind);
    const real_t d3 = delta * (1.0f / 3.0f), d6 = delta * (1.0f / 6.0f);
#pragma unroll
    for (int i = 0; i < NS; i++) {
        const real_t a = d3 * bc[i], b = d6 * (ad[i] + k[i]);
        out[i] = s[i] + a + b;
        if (i < 3 && ta) { ta[i] = a; tb[i] = b; }
    }
}
#endif

// 12-vector difference s2 (-) s1; the attitude part is the MRP (a=1, f=4) of
// q2 (x) q1* with the real part forced non-negative.
NMPC_DEV void state_to_delta(real_t* dl, const real_t* s1, const real_t* s2) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
        dl[i] = s2[i] - s1[i];
        dl[i + 3] = s2[i + 3] - s1[i + 3];
        dl[i + 9] = s2[i + 10] - s1[i + 10];
    }
    real_t cj[4], e[4];
    cj[0] = s1[6]; cj[1] = s1[7]; cj[2] = s1[8]; cj[3] = -s1[9];
    quat_mul(e, &s2[6], cj);
    if (e[3] < 0) { e[0] = -e[0]; e[1] = -e[1]; e[2] = -e[2]; e[3] = -e[3]; }
    real_t d = (real_t)4.0 / ((real_t)1.0 + e[3]);
    dl[6] = e[0] * d; dl[7] = e[1] * d; dl[8] = e[2] * d;
}

// x (+) d: the state a 12-vector deviation d leads to from x.  Position,
// velocity and body rates add; the attitude deviation is applied as the MRP
// (a = 1, f = 4) quaternion multiplied from the left, the composition the
// finite differences use (ccs-c66x/cnmpc.c:617-634), with the scalar part in
// its intended form (src/ocp.cpp:122-126).  x and out may not alias.
NMPC_DEV void state_plus_delta(real_t* out, const real_t* x, const real_t* d) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
        out[i] = x[i] + d[i];
        out[3 + i] = x[3 + i] + d[3 + i];
        out[10 + i] = x[10 + i] + d[9 + i];
    }
    const real_t pp = d[6] * d[6] + d[7] * d[7] + d[8] * d[8];
    real_t dq[4];
    dq[3] = ((real_t)16.0 - pp) / ((real_t)16.0 + pp);
    const real_t sc = (real_t)0.25 * ((real_t)1.0 + dq[3]);
    dq[0] = sc * d[6]; dq[1] = sc * d[7]; dq[2] = sc * d[8];
    quat_mul(&out[6], dq, &x[6]);
}

// Perturbed copy of (x, u) for finite-difference column j of the stage
// Jacobian (ccs-c66x/cnmpc.c:600-648).  j == 15 leaves the point unperturbed.
// Returns 1/h for the column.
// All element selections are compile-time-indexed selects so that x and u stay
// in registers (a run-time subscript would push them to scratch memory).
NMPC_DEV real_t perturb_point(real_t* x, real_t* u, int j, const real_t* lower,
                              const real_t* upper) {
    real_t hinv = (real_t)(1.0 / kFdStep);
    if (j < 6) {
#pragma unroll
        for (int i = 0; i < 6; i++) x[i] = (i == j) ? x[i] + kFdStep : x[i];
    } else if (j <= 8) {
        real_t dp[3], dq[4], t[4];
#pragma unroll
        for (int i = 0; i < 3; i++) dp[i] = (i == j - 6) ? kFdStep : (real_t)0.0;
        dq[3] = -(kFdStep * kFdStep) + (real_t)16.0 / (real_t)(16.0 + kFdStep * kFdStep);
        dq[0] = ((real_t)1.0 / (real_t)4.0) * ((real_t)1.0 + dq[3]) * dp[0];
        dq[1] = ((real_t)1.0 / (real_t)4.0) * ((real_t)1.0 + dq[3]) * dp[1];
        dq[2] = ((real_t)1.0 / (real_t)4.0) * ((real_t)1.0 + dq[3]) * dp[2];
        quat_mul(t, dq, &x[6]);
        x[6] = t[0]; x[7] = t[1]; x[8] = t[2]; x[9] = t[3];
    } else if (j < NX) {
#pragma unroll
        for (int i = 10; i < 13; i++) x[i] = (i == j + 1) ? x[i] + kFdStep : x[i];
    } else if (j < NZ) {
        real_t range = (real_t)0.0;
#pragma unroll
        for (int i = 0; i < 3; i++) range = (i == j - NX) ? (upper[i] - lower[i]) : range;
        real_t h = kFdStep;
        h *= range;
        hinv = (real_t)1.0 / h;
#pragma unroll
        for (int i = 0; i < 3; i++) u[i] = (i == j - NX) ? u[i] + h : u[i];
    }
    return hinv;
}

}  // namespace nmpc
