// HIP kernels (gfx950) for the batched NMPC hot path.
//
//   k_store_refs      reference points -> xref / uref rings
//   k_init_stages     nmpc_init for every instance      (cnmpc.c:872-948)
//   k_linearise       RK4 + finite-difference stage linearisation, one lane per
//                     (instance, stage, column), fused with the stage-QP
//                     re-staging                         (cnmpc.c:583-667, 1022-1080;
//                                                        qpDUNES/setup_qp.c:547-585)
//   k_linearise_free  the same stage body on caller-supplied samples
//   k_rk4_free        one RK4 step on caller-supplied samples
//   k_qp_begin        initial-value embedding + opening step of the dual Newton
//                     solve                              (cnmpc.c:675-702, dual_qp.c:56-100)
//   k_newton_setup    dual gradient, convergence test, Newton Hessian blocks
//                                                        (dual_qp.c:423-597)
//   k_newton_solve    reverse block-tridiagonal Cholesky, regularisation,
//                     two substitutions                  (dual_qp.c:601-698, 817-908, 993-1055)
//   k_newton_step     stage directions, step-length selection, commit, active
//                     set bookkeeping                    (dual_qp.c:309-379, 1119-1470,
//                                                        1542-1699; stage_qp_solver_clipping.c)
//   k_qp_finish       iteration-limit status, controls   (dual_qp.c:219-222, cnmpc.c:705-726)
//
// The QP kernels run one 64-lane wavefront (= one workgroup) per instance: all
// solver control flow is per instance, hence wave-uniform.  Sums whose order
// the reference fixes (objective, norms, slopes) are accumulated in that order.
// Built with -ffp-contract=off: no FMA contraction anywhere.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"
#include "model.hpp"

#pragma clang fp contract(off)

namespace nmpc {

namespace {

constexpr double kInfty = 1.e12;
constexpr double kEqTol = 2.221e-16;
constexpr double kActTol = 1e4 * 2.221e-16;
constexpr double kDiagRegTol = 1.e-10;
constexpr double kRegParam = 1.e-6;
constexpr double kStatTol = (double)1e-3f;
constexpr int kMaxIter = 5;
constexpr int kMaxLsIter = 19;
constexpr int kMaxLsRefine = 40;
constexpr double kLsReduction = 0.1;
constexpr double kLsIncrease = 1.5;
constexpr double kLsMinAbs = 2.221e-16;
constexpr double kLsMinRel = 1.e-14;
constexpr double kLsStatTol = 1.e-3;
constexpr double kLsMaxStep = 1.;

constexpr int kWave = 64;
constexpr int kChunk = 512;          // doubles of LDS used by ordered sums

NMPC_DEV int stage_slot(const BatchView& v, int k) {
    return k < v.N ? (v.head + k) % v.N : v.N;
}
NMPC_DEV int lam_slot(const BatchView& v, int k) { return (v.lhead + k) % v.N; }
NMPC_DEV int xref_slot(const BatchView& v, int k) { return (v.xhead + k) % (v.N + 1); }

// Per-instance base pointers.
struct Inst {
    const double* xref; const double* uref;
    double *C, *c, *zLow, *zUpp, *q, *z, *qStep, *zUnc, *dz, *tmp, *y, *p, *pStep;
    double *lam, *dl, *grad, *hess, *chol;
    int8_t *ieq, *prev, *chg;
    Scalars* sc;
};

NMPC_DEV Inst instance(const BatchView& v, int b) {
    Inst I;
    const size_t N = (size_t)v.N, N1 = N + 1, sb = (size_t)b;
    I.xref = v.xref + sb * N1 * NS;
    I.uref = v.uref + sb * N * NU;
    I.C = v.C + sb * N * (NX * NZ);
    I.c = v.c + sb * N * NX;
    I.zLow = v.zLow + sb * N1 * NZ;
    I.zUpp = v.zUpp + sb * N1 * NZ;
    I.q = v.q + sb * N1 * NZ;
    I.z = v.z + sb * N1 * NZ;
    I.qStep = v.qStep + sb * N1 * NZ;
    I.zUnc = v.zUnc + sb * N1 * NZ;
    I.dz = v.dz + sb * N1 * NZ;
    I.tmp = v.tmp + sb * N1 * NZ;
    I.y = v.y + sb * N1 * 2 * NZ;
    I.p = v.p + sb * N1;
    I.pStep = v.pStep + sb * N1;
    I.lam = v.lam + sb * N * NX;
    I.dl = v.dl + sb * N * NX;
    I.grad = v.grad + sb * N * NX;
    I.hess = v.hess + sb * N * 2 * NX * NX;
    I.chol = v.chol + sb * N * 2 * NX * NX;
    I.ieq = v.ieq + sb * N1 * NZ;
    I.prev = v.prev + sb * N1 * NZ;
    I.chg = v.chg + sb * N1;
    I.sc = v.sc + sb;
    return I;
}

NMPC_DEV const double* stage_hessian(const BatchView& v, int k) {
    return k < v.N ? v.H : v.Ht;
}

// ------------------------------------------------------------------------
// Stage-QP re-staging for the clipping solver with a diagonal Hessian and a
// zero linear term (qpDUNES/setup_qp.c:547-585, stage_qp_solver_clipping.c:42-98):
// q := 0, qStep := C' lambda_{k+1} - lambda_k (+ g), pStep := c . lambda_{k+1},
// dz := -(qStep / H), zUnc := 0.  One thread does a whole stage.
NMPC_DEV void restage_serial(const BatchView& v, const Inst& I, int k) {
    const int s = stage_slot(v, k), nV = k < v.N ? NZ : NX;
    const double* H = stage_hessian(v, k);
    double qs[NZ];
    double ps = 0.;
    if (k < v.N) {
        const double* l1 = I.lam + (size_t)lam_slot(v, k) * NX;
        const double* Ck = I.C + (size_t)s * (NX * NZ);
        const double* ck = I.c + (size_t)s * NX;
#pragma unroll
        for (int j = 0; j < NZ; j++) qs[j] = 0.;
        for (int i = 0; i < NX; i++) {
            const double li = l1[i];
#pragma unroll
            for (int j = 0; j < NZ; j++) qs[j] += Ck[i * NZ + j] * li;
        }
        for (int i = 0; i < NX; i++) ps += l1[i] * ck[i];
    } else {
#pragma unroll
        for (int j = 0; j < NZ; j++) qs[j] = 0.0;
    }
    if (k > 0) {
        const double* l0 = I.lam + (size_t)lam_slot(v, k - 1) * NX;
#pragma unroll
        for (int i = 0; i < NX; i++) qs[i] -= l0[i];
    }
    I.pStep[s] = ps;
#pragma unroll
    for (int i = 0; i < NZ; i++) {
        if (i < nV) {
            double t = qs[i] + 0.0;
            I.qStep[s * NZ + i] = t;
            double d = t / H[i];
            I.dz[s * NZ + i] = -d;
            I.zUnc[s * NZ + i] = 0.0;
            I.q[s * NZ + i] = 0.0;
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------
// refs: [B][count][16] staging buffer; point i of the range goes to horizon
// position first + i.  Control part of position p >= 1 belongs to stage p - 1.
__global__ void k_store_refs(BatchView v, const double* __restrict__ refs,
                             int first, int count) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = v.B * count;
    if (t >= total) return;
    const int b = t / count, i = t % count, pos = first + i;
    const double* r = refs + (size_t)t * 16;
    double* x = v.xref + ((size_t)b * (v.N + 1) + xref_slot(v, pos)) * NS;
#pragma unroll
    for (int e = 0; e < NS; e++) x[e] = r[e];
    if (pos >= 1) {
        double* u = v.uref + ((size_t)b * v.N + stage_slot(v, pos - 1)) * NU;
        u[0] = r[13]; u[1] = r[14]; u[2] = r[15];
    }
}

// ------------------------------------------------------------------------
// nmpc_init: zero Jacobians and residuals, global bounds, re-stage every
// stage QP against the current multipliers, invalidate the active-set record.
// One thread per (instance, stage).
__global__ void k_init_stages(BatchView v) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int N1 = v.N + 1;
    if (t >= v.B * N1) return;
    const int b = t / N1, k = t % N1;
    Inst I = instance(v, b);
    const int s = stage_slot(v, k);
    if (k < v.N) {
        for (int e = 0; e < NX * NZ; e++) I.C[(size_t)s * (NX * NZ) + e] = 0.0;
        for (int e = 0; e < NX; e++) I.c[s * NX + e] = 0.0;
    }
    for (int e = 0; e < NX; e++) {
        I.zLow[s * NZ + e] = -kInfty;
        I.zUpp[s * NZ + e] = kInfty;
    }
    if (k < v.N)
        for (int e = 0; e < NU; e++) {
            I.zLow[s * NZ + NX + e] = v.lb[e];
            I.zUpp[s * NZ + NX + e] = v.ub[e];
        }
    restage_serial(v, I, k);
    for (int e = 0; e < NZ; e++) I.prev[k * NZ + e] = -42;
}

// ------------------------------------------------------------------------
// Stage linearisation.  Lane j of each 16-lane group integrates the reference
// point perturbed along column j (j == 15: unperturbed).  The nominal end state
// is broadcast inside the group; lane j < 15 then owns column j of the 12x15
// Jacobian, lane 15 owns the residual.
NMPC_DEV void linearise_lane(const double* x0, const double* u0, const double* xnext,
                             const double* wind, const double* lower,
                             const double* upper, int j, double* out12) {
    double x[NS], u[NU], moved[NS], nominal[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) x[i] = x0[i];
#pragma unroll
    for (int i = 0; i < NU; i++) u[i] = u0[i];
    const double hinv = perturb_point(x, u, j, lower, upper);
    rk4_step(moved, x, u, wind, kStepLength);
#pragma unroll
    for (int i = 0; i < NS; i++) nominal[i] = __shfl(moved[i], 15, 16);
    double base[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) base[i] = (j == 15) ? xnext[i] : nominal[i];
    state_to_delta(out12, base, moved);
    if (j < 15) {
#pragma unroll
        for (int r = 0; r < NX; r++) out12[r] = out12[r] * hinv;
    }
}

__global__ __launch_bounds__(256) void k_linearise(BatchView v, int first, int count) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = (int)(gid & 15);
    long st = gid >> 4;
    const long total = (long)v.B * count;
    const bool live = st < total;
    if (!live) st = total - 1;              // keep the whole group in the shuffles
    const int b = (int)(st / count), k = first + (int)(st % count);
    Inst I = instance(v, b);
    const int s = stage_slot(v, k);

    double x0[NS], xn[NS], u0[NU], w[3];
    const double* xr = I.xref + (size_t)xref_slot(v, k) * NS;
    const double* xq = I.xref + (size_t)xref_slot(v, k + 1) * NS;
#pragma unroll
    for (int i = 0; i < NS; i++) { x0[i] = xr[i]; xn[i] = xq[i]; }
#pragma unroll
    for (int i = 0; i < NU; i++) { u0[i] = I.uref[s * NU + i]; w[i] = v.wind[(size_t)b * 3 + i]; }

    double col[NX];
    linearise_lane(x0, u0, xn, w, v.lb, v.ub, j, col);
    if (!live) return;

    // multipliers of the two coupling constraints this stage takes part in
    const double* l1 = I.lam + (size_t)lam_slot(v, k) * NX;
    if (j < 15) {
        double* Ck = I.C + (size_t)s * (NX * NZ);
#pragma unroll
        for (int r = 0; r < NX; r++) Ck[r * NZ + j] = col[r];
        double qs = 0.;
#pragma unroll
        for (int i = 0; i < NX; i++) qs += col[i] * l1[i];
        if (k > 0 && j < NX) qs -= I.lam[(size_t)lam_slot(v, k - 1) * NX + j];
        qs = qs + 0.0;
        const double d = qs / v.H[j];
        I.qStep[s * NZ + j] = qs;
        I.dz[s * NZ + j] = -d;
        I.zUnc[s * NZ + j] = 0.0;
        I.q[s * NZ + j] = 0.0;
        double lo = -kInfty, up = kInfty;
        if (j >= NX) {
            double uj = 0.0, lbj = 0.0, ubj = 0.0;
#pragma unroll
            for (int i = 0; i < NU; i++) {
                uj = (i == j - NX) ? u0[i] : uj;
                lbj = (i == j - NX) ? v.lb[i] : lbj;
                ubj = (i == j - NX) ? v.ub[i] : ubj;
            }
            lo = lbj - uj;
            up = ubj - uj;
        }
        I.zLow[s * NZ + j] = lo;
        I.zUpp[s * NZ + j] = up;
    } else {
        double ps = 0.;
#pragma unroll
        for (int i = 0; i < NX; i++) {
            I.c[s * NX + i] = col[i];
            ps += l1[i] * col[i];
        }
        I.pStep[s] = ps;
    }
}

// Stateless variant: n samples, inputs [n][13] [n][3] [n][13] [n][3],
// outputs jac [n][180] (row-major 12x15) and res [n][12].
__global__ __launch_bounds__(256) void k_linearise_free(
    long n, const double* __restrict__ xs, const double* __restrict__ us,
    const double* __restrict__ xnexts, const double* __restrict__ winds,
    Bounds bnd, double* __restrict__ jac, double* __restrict__ res) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = (int)(gid & 15);
    long st = gid >> 4;
    const bool live = st < n;
    if (!live) st = n - 1;
    double x0[NS], xn[NS], u0[NU], w[3], col[NX];
#pragma unroll
    for (int i = 0; i < NS; i++) { x0[i] = xs[st * NS + i]; xn[i] = xnexts[st * NS + i]; }
#pragma unroll
    for (int i = 0; i < NU; i++) { u0[i] = us[st * NU + i]; w[i] = winds[st * 3 + i]; }
    linearise_lane(x0, u0, xn, w, bnd.lb, bnd.ub, j, col);
    if (!live) return;
    if (j < 15) {
#pragma unroll
        for (int r = 0; r < NX; r++) jac[st * (NX * NZ) + r * NZ + j] = col[r];
    } else {
#pragma unroll
        for (int r = 0; r < NX; r++) res[st * NX + r] = col[r];
    }
}

__global__ __launch_bounds__(256) void k_rk4_free(long n, const double* __restrict__ xs,
                           const double* __restrict__ us,
                           const double* __restrict__ winds, double delta,
                           double* __restrict__ out) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    double x[NS], u[NU], w[3], r[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) x[i] = xs[t * NS + i];
#pragma unroll
    for (int i = 0; i < NU; i++) { u[i] = us[t * NU + i]; w[i] = winds[t * 3 + i]; }
    rk4_step(r, x, u, w, delta);
#pragma unroll
    for (int i = 0; i < NS; i++) out[t * NS + i] = r[i];
}

// ========================================================================
// Dual Newton solve, one wavefront per instance.
// ========================================================================
namespace {

// Sum term(0) + term(1) + ... + term(n-1) in index order, starting from 0.
// term(r) is evaluated by lane r % 64 (any side effects included); the
// accumulation is done redundantly by every lane, so the result is uniform.
template <class F>
NMPC_DEV double ordered_sum(int n, double* lds, F term) {
    const int lane = threadIdx.x;
    double acc = 0.;
    for (int base = 0; base < n; base += kChunk) {
        const int len = (n - base < kChunk) ? n - base : kChunk;
        for (int r = lane; r < len; r += kWave) lds[r] = term(base + r);
        __syncthreads();
        for (int t = 0; t < len; t++) acc += lds[t];
        __syncthreads();
    }
    return acc;
}

NMPC_DEV int wave_max(int x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        int y = __shfl_xor(x, o);
        x = y > x ? y : x;
    }
    return x;
}

NMPC_DEV double wave_min(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double y = __shfl_xor(x, o);
        x = y < x ? y : x;
    }
    return x;
}

NMPC_DEV double bcast_lane(double x, int src) {   // src must be wave-uniform
    int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
}

// Clip v to [lo, up] recording the gaps first
// (stage_qp_solver_clipping.c:174-198).
NMPC_DEV double clip_gap(double val, double lo, double up, double* y0, double* y1) {
    *y0 = lo - val;
    *y1 = val - up;
    if (*y0 >= -kActTol) val = lo;
    else if (*y1 >= -kActTol) val = up;
    return val;
}

// 0.5 z'Hz + q'z + p with the reference's accumulation order
// (dual_qp.c:1552-1559, matrix_vector.c:1887-1900, 2100-2112).
NMPC_DEV double stage_value(const double* H, const double* z, const double* q,
                            double p, int nV) {
    double zHz = 0., qz = 0.;
#pragma unroll
    for (int i = 0; i < NZ; i++) if (i < nV) zHz += H[i] * z[i] * z[i];
    double val = 0.5 * zHz;
#pragma unroll
    for (int i = 0; i < NZ; i++) if (i < nV) qz += q[i] * z[i];
    val += qz;
    val += p;
    return val;
}

// Commit a step of length a on stage k and return its objective value
// (stage_qp_solver_clipping.c:134-165 followed by dual_qp.c:1548-1563).
NMPC_DEV double stage_commit(const BatchView& v, const Inst& I, int k, double a) {
    const int s = stage_slot(v, k), nV = k < v.N ? NZ : NX, o = s * NZ;
    const double* H = stage_hessian(v, k);
    double z[NZ], q[NZ];
#pragma unroll
    for (int i = 0; i < NZ; i++) {
        if (i < nV) {
            const double zu = I.zUnc[o + i] + a * I.dz[o + i];
            I.zUnc[o + i] = zu;
            double y0, y1;
            z[i] = clip_gap(zu, I.zLow[o + i], I.zUpp[o + i], &y0, &y1);
            I.z[o + i] = z[i];
            I.y[2 * o + 2 * i] = y0;
            I.y[2 * o + 2 * i + 1] = y1;
            q[i] = I.q[o + i] + a * I.qStep[o + i];
            I.q[o + i] = q[i];
        } else {
            z[i] = 0.; q[i] = 0.;
        }
    }
    const double p = I.p[s] + a * I.pStep[s];
    I.p[s] = p;
    return stage_value(H, z, q, p, nV);
}

// Objective at trial step a: overwrites z and y only (dual_qp.c:1574-1617).
NMPC_DEV double stage_trial_value(const BatchView& v, const Inst& I, int k, double a) {
    const int s = stage_slot(v, k), nV = k < v.N ? NZ : NX, o = s * NZ;
    const double* H = stage_hessian(v, k);
    double z[NZ], q[NZ];
#pragma unroll
    for (int i = 0; i < NZ; i++) {
        if (i < nV) {
            const double zu = I.zUnc[o + i] + a * I.dz[o + i];
            double y0, y1;
            z[i] = clip_gap(zu, I.zLow[o + i], I.zUpp[o + i], &y0, &y1);
            I.z[o + i] = z[i];
            I.y[2 * o + 2 * i] = y0;
            I.y[2 * o + 2 * i + 1] = y1;
            q[i] = I.q[o + i] + a * I.qStep[o + i];
        } else {
            z[i] = 0.; q[i] = 0.;
        }
    }
    const double p = I.p[s] + a * I.pStep[s];
    return stage_value(H, z, q, p, nV);
}

// Trial primal vector at step a into tmp, gaps into y (dual_qp.c:1370-1376).
NMPC_DEV void stage_trial_point(const BatchView& v, const Inst& I, int k, double a) {
    const int s = stage_slot(v, k), nV = k < v.N ? NZ : NX, o = s * NZ;
#pragma unroll
    for (int i = 0; i < NZ; i++) {
        if (i < nV) {
            const double zu = I.zUnc[o + i] + a * I.dz[o + i];
            double y0, y1;
            I.tmp[o + i] = clip_gap(zu, I.zLow[o + i], I.zUpp[o + i], &y0, &y1);
            I.y[2 * o + 2 * i] = y0;
            I.y[2 * o + 2 * i + 1] = y1;
        }
    }
}

// Row r = 12 k + i of the dual gradient: C_k z_k + c_k - z_{k+1}
// (dual_qp.c:574-597) evaluated on the primal array `zz`.
NMPC_DEV double gradient_row(const BatchView& v, const Inst& I, const double* zz, int r) {
    const int k = r / NX, i = r % NX;
    const int s = stage_slot(v, k), sn = stage_slot(v, k + 1);
    const double* Crow = I.C + (size_t)s * (NX * NZ) + i * NZ;
    const double* zk = zz + s * NZ;
    double a = 0.;
#pragma unroll
    for (int j = 0; j < NZ; j++) a += Crow[j] * zk[j];
    a += I.c[s * NX + i];
    a -= zz[sn * NZ + i];
    return a;
}

// Active-set code of every variable of stage k (dual_qp.c:1627-1668) and
// whether it differs from the previous record (dual_qp.c:1677-1699).
NMPC_DEV int stage_active_set(const BatchView& v, const Inst& I, int k) {
    const int s = stage_slot(v, k), nV = k < v.N ? NZ : NX;
    int changed = 0;
    for (int i = 0; i < nV; i++) {
        const double y0 = I.y[2 * s * NZ + 2 * i], y1 = I.y[2 * s * NZ + 2 * i + 1];
        int8_t code = (y0 > kEqTol) ? -1 : ((y1 > kEqTol) ? 1 : 0);
        I.ieq[k * NZ + i] = code;
        if (code != I.prev[k * NZ + i]) changed = 1;
    }
    I.chg[k] = (int8_t)changed;
    return changed;
}

// Refresh the active-set record of every stage; returns the highest stage
// index whose set changed, or -1.
NMPC_DEV int refresh_active_sets(const BatchView& v, const Inst& I) {
    int last = -1;
    for (int k = threadIdx.x; k <= v.N; k += kWave)
        if (stage_active_set(v, I, k)) last = k;
    return wave_max(last);
}

NMPC_DEV double commit_all(const BatchView& v, const Inst& I, double a, double* lds) {
    return ordered_sum(v.N + 1, lds,
                       [&](int k) { return stage_commit(v, I, k, a); });
}

}  // namespace

// ------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void k_qp_begin(BatchView v,
                                                    const double* __restrict__ meas) {
    __shared__ double lds[kChunk];
    const int b = blockIdx.x, lane = threadIdx.x;
    Inst I = instance(v, b);

    // initial-value embedding: pin the stage-0 state to the measured deviation
    if (lane == 0) {
        double x0[NS], m[NS], d[NX];
        const double* xr = I.xref + (size_t)xref_slot(v, 0) * NS;
#pragma unroll
        for (int i = 0; i < NS; i++) { x0[i] = xr[i]; m[i] = meas[(size_t)b * NS + i]; }
        state_to_delta(d, x0, m);
        const int s0 = stage_slot(v, 0);
#pragma unroll
        for (int i = 0; i < NX; i++) { I.zLow[s0 * NZ + i] = d[i]; I.zUpp[s0 * NZ + i] = d[i]; }
#pragma unroll
        for (int i = 0; i < NU; i++) {
            const double ur = I.uref[s0 * NU + i];
            I.zLow[s0 * NZ + NX + i] = v.lb[i] - ur;
            I.zUpp[s0 * NZ + NX + i] = v.ub[i] - ur;
        }
    }
    for (int e = lane; e < (v.N + 1) * NZ; e += kWave) I.prev[e] = -42;
    __syncthreads();

    const double obj = commit_all(v, I, 1, lds);
    const int last = refresh_active_sets(v, I);
    if (lane == 0) {
        I.sc->objInc = obj;
        I.sc->last = last;
        I.sc->status = ST_RUNNING;
        I.sc->numIter = 0;
        I.sc->regularized = 0;
        I.sc->lsIter = 0;
    }
}

// ------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void k_newton_setup(BatchView v, int it) {
    __shared__ double lds[kChunk];
    __shared__ double Cs[NX * NZ];
    __shared__ double Zs[NZ * NX];
    const int b = blockIdx.x, lane = threadIdx.x;
    Inst I = instance(v, b);
    if (I.sc->status != ST_RUNNING) return;
    const int N = v.N;

    // gradient and its norm
    const double nn = ordered_sum(NX * N, lds, [&](int r) {
        const double g = gradient_row(v, I, I.z, r);
        I.grad[r] = g;
        return g * g;
    });
    if (lane == 0) { I.sc->numIter = it; I.sc->regularized = 0; }
    if (sqrt(nn) < kStatTol) {
        if (lane == 0) I.sc->status = ST_OPTIMAL;
        return;
    }

    // Newton Hessian blocks of stages whose active set changed
    for (int k = 0; k < N; k++) {
        const bool ck = I.chg[k] != 0, cn = I.chg[k + 1] != 0;
        if (!(ck || cn)) continue;
        const int s = stage_slot(v, k), sn = stage_slot(v, k + 1);
        const double* Hn = stage_hessian(v, k + 1);
        const double* Cg = I.C + (size_t)s * (NX * NZ);
        // free columns of stage k; fixed states of stage k+1
        bool fr = false, fx = false;
        if (lane < NZ)
            fr = (I.y[2 * s * NZ + 2 * lane] <= kEqTol) &&
                 (I.y[2 * s * NZ + 2 * lane + 1] <= kEqTol);
        if (lane < NX)
            fx = (I.y[2 * sn * NZ + 2 * lane] >= kEqTol) ||
                 (I.y[2 * sn * NZ + 2 * lane + 1] >= kEqTol);
        const unsigned long long freeMask = __ballot(fr);
        const unsigned long long fixMask = __ballot(fx);
        for (int e = lane; e < NX * NZ; e += kWave) {
            Cs[e] = Cg[e];
            const int l = e / NX, j = e % NX;
            Zs[e] = Cg[j * NZ + l] / v.H[l];
        }
        __syncthreads();
        // diagonal block: E P_{k+1} E' + C_k P_k C_k'
        if (lane < 48) {
            const int i = lane >> 2, j0 = (lane & 3) * 3;
            double m[3];
#pragma unroll
            for (int t = 0; t < 3; t++) {
                m[t] = 0.;
                if (i == j0 + t) {
                    m[t] = 1. / Hn[i];
                    if ((fixMask >> i) & 1ull) m[t] = 0.;
                }
            }
            for (int l = 0; l < NZ; l++) {
                if (!((freeMask >> l) & 1ull)) continue;
                const double ci = Cs[i * NZ + l];
#pragma unroll
                for (int t = 0; t < 3; t++) m[t] += ci * Zs[l * NX + j0 + t];
            }
            double* Hd = I.hess + (size_t)k * 2 * NX * NX + i * 2 * NX + NX + j0;
#pragma unroll
            for (int t = 0; t < 3; t++) Hd[t] = m[t];
        }
        // sub-diagonal block: -C_k[:, :12] P_k
        if (k >= 1 && ck) {
            for (int e = lane; e < NX * NX; e += kWave) {
                const int i = e / NX, j = e % NX;
                const double val = Cs[i * NZ + j] / v.H[j];
                I.hess[(size_t)k * 2 * NX * NX + i * 2 * NX + j] =
                    ((freeMask >> j) & 1ull) ? -val : 0.;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------
namespace {

#define HS(M, k, l, i, j) (M)[(size_t)(k) * 2 * NX * NX + (i) * 2 * NX + (1 + (l)) * NX + (j)]

// Reverse block-tridiagonal Cholesky from block `start` down to 0
// (dual_qp.c:817-908).  Lanes 0-11 own one column of the diagonal block each,
// lanes 16-27 one column of the sub-diagonal block; pivots and finished column
// entries are broadcast from the owning lane.  Returns ST_OK or ST_DIV_ZERO;
// *minDiag is lowered to the smallest pivot root written.
NMPC_DEV int factor_reverse(const BatchView& v, const Inst& I, int start) {
    const int N = v.N, lane = threadIdx.x, t = lane & 15;
    const bool diagLane = lane < 16, subLane = (lane >= 16 && lane < 32);
    const bool colLane = t < NX;
    const int tc = colLane ? t : 0;           // clamp so idle lanes read valid memory
    double Ln[NX];                            // column tc of chol(k+1, -1) on diag lanes
    int status = ST_OK;

    for (int k = start; k >= 0; --k) {
        const bool below = k < N - 1;         // a block row k+1 exists
        double A[NX], L[NX];
        if (k == start) {
#pragma unroll
            for (int l = 0; l < NX; l++) Ln[l] = below ? HS(I.chol, k + 1, -1, l, tc) : 0.;
        }
        // column tc of this block row's diagonal (diag lanes) or sub (sub lanes) part
        {
            const int part = diagLane ? 0 : -1;
#pragma unroll
            for (int l = 0; l < NX; l++) A[l] = HS(I.hess, k, part, l, tc);
        }
#pragma unroll
        for (int l = 0; l < NX; l++) L[l] = 0.;

#pragma unroll
        for (int jj = NX - 1; jj >= 0; --jj) {
            if (status == ST_OK) {
                double sum = A[jj];
#pragma unroll
                for (int l = jj + 1; l < NX; ++l) {
                    const double ljj = bcast_lane(L[l], jj);      // chol(k,0,l,jj)
                    sum -= L[l] * ljj;
                }
                if (below) {
#pragma unroll
                    for (int l = 0; l < NX; ++l) {
                        const double njj = bcast_lane(Ln[l], jj); // chol(k+1,-1,l,jj)
                        const double nsum = sum - Ln[l] * njj;
                        sum = diagLane ? nsum : sum;
                    }
                }
                const double piv = bcast_lane(sum, jj);
                if (piv < 1.e2 * kEqTol) {
                    status = ST_DIV_ZERO;
                } else {
                    const double d = sqrt(piv);
                    const double quot = sum / d;
                    L[jj] = (diagLane && t == jj) ? d : quot;
                }
            }
        }
        if (status != ST_OK) break;

        // write back: lower triangle of the diagonal block, full sub block
        if (diagLane && colLane) {
#pragma unroll
            for (int l = 0; l < NX; l++)
                if (l >= t) HS(I.chol, k, 0, l, t) = L[l];
        }
        if (subLane && colLane && k > 0) {
#pragma unroll
            for (int l = 0; l < NX; l++) HS(I.chol, k, -1, l, t) = L[l];
        }
        // hand this block row's sub part to the diagonal lanes for block k-1
#pragma unroll
        for (int l = 0; l < NX; l++) Ln[l] = __shfl(L[l], (lane + 16) & 63);
    }
    return status;
}

// L' x = g bottom-up, then L x = x top-down (dual_qp.c:993-1055).
NMPC_DEV void solve_reverse(const BatchView& v, const Inst& I) {
    const int N = v.N, lane = threadIdx.x;
    const int i = lane < NX ? lane : 0;
    double xk[NX], xo[NX];        // solution of this / the neighbouring block (uniform)
#pragma unroll
    for (int j = 0; j < NX; j++) { xk[j] = 0.; xo[j] = 0.; }

    for (int k = N - 1; k >= 0; --k) {
        const bool below = k < N - 1;
        double L[NX], pn[NX];
#pragma unroll
        for (int j = 0; j < NX; j++) L[j] = (j >= i) ? HS(I.chol, k, 0, j, i) : 0.;
        if (below) {
#pragma unroll
            for (int j = 0; j < NX; j++) pn[j] = HS(I.chol, k + 1, -1, j, i) * xo[j];
        }
        const double g = I.grad[k * NX + i];
        double mine = 0.;
#pragma unroll
        for (int r = NX - 1; r >= 0; --r) {
            double sum = g;
#pragma unroll
            for (int j = r + 1; j < NX; ++j) sum -= L[j] * xk[j];
            if (below) {
#pragma unroll
                for (int j = 0; j < NX; ++j) sum -= pn[j];
            }
            double dd = L[0];
#pragma unroll
            for (int j = 1; j < NX; j++) dd = (j == i) ? L[j] : dd;
            const double x = sum / dd;
            xk[r] = bcast_lane(x, r);
            mine = (lane == r) ? x : mine;
        }
        if (lane < NX) I.dl[k * NX + lane] = mine;
#pragma unroll
        for (int j = 0; j < NX; j++) xo[j] = xk[j];
    }
    __syncthreads();

    for (int k = 0; k < N; ++k) {
        double Ld[NX], Ls[NX];
#pragma unroll
        for (int j = 0; j < NX; j++) {
            Ld[j] = (j <= i) ? HS(I.chol, k, 0, i, j) : 0.;
            Ls[j] = (k > 0) ? HS(I.chol, k, -1, i, j) : 0.;
        }
        double sum = I.dl[k * NX + i];
        if (k > 0) {
#pragma unroll
            for (int j = 0; j < NX; ++j) sum -= Ls[j] * xo[j];
        }
        double dd = Ld[0];
#pragma unroll
        for (int j = 1; j < NX; j++) dd = (j == i) ? Ld[j] : dd;
        double mine = 0.;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const double x = sum / dd;          // final for lane j at this point
            const double xj = bcast_lane(x, j);
            xk[j] = xj;
            mine = (lane == j) ? x : mine;
            sum -= Ld[j] * xj;                  // only lanes i > j use this later
        }
        __syncthreads();
        if (lane < NX) I.dl[k * NX + lane] = mine;
#pragma unroll
        for (int j = 0; j < NX; j++) xo[j] = xk[j];
    }
}

}  // namespace

__global__ __launch_bounds__(kWave) void k_newton_solve(BatchView v) {
    const int b = blockIdx.x, lane = threadIdx.x;
    Inst I = instance(v, b);
    if (I.sc->status != ST_RUNNING) return;
    const int N = v.N;
    const int last = I.sc->last;

    // dual_qp.c:601-698
    int st = ST_OK;
    if (last >= 0) st = factor_reverse(v, I, last < N - 1 ? last : N - 1);
    __syncthreads();
    double minDiag = kInfty;
    if (st == ST_OK) {
        for (int e = lane; e < N * NX; e += kWave) {
            const double d = HS(I.chol, e / NX, 0, e % NX, e % NX);
            if (minDiag > d) minDiag = d;
        }
        minDiag = wave_min(minDiag);
    }
    if (st == ST_DIV_ZERO || minDiag < kDiagRegTol) {
        for (int e = lane; e < N * NX; e += kWave)
            HS(I.hess, e / NX, 0, e % NX, e % NX) += kRegParam;
        if (lane == 0) I.sc->regularized = 1;
        __syncthreads();
        st = factor_reverse(v, I, N - 1);
        __syncthreads();
        if (st != ST_OK) {
            if (lane == 0) I.sc->status = st;
            return;
        }
    }
    solve_reverse(v, I);
}

// ------------------------------------------------------------------------
namespace {

// Stage directions for the multiplier step dl (dual_qp.c:309-379):
// qStep = C' dl_k - dl_{k-1}, pStep = c . dl_k, dz = -(qStep / H).  Returns
// this stage's smallest positive step to an active-set change
// (stage_qp_solver_clipping.c:110-126), or `amin` if none is smaller.
NMPC_DEV double stage_direction(const BatchView& v, const Inst& I, int k, double amin) {
    const int N = v.N, s = stage_slot(v, k), nV = k < N ? NZ : NX, o = s * NZ;
    const double* H = stage_hessian(v, k);
    double qs[NZ];
    double ps = 0.;
    if (k < N) {
        const double* l1 = I.dl + k * NX;
        const double* Ck = I.C + (size_t)s * (NX * NZ);
#pragma unroll
        for (int j = 0; j < NZ; j++) qs[j] = 0.;
        for (int i = 0; i < NX; i++) {
            const double li = l1[i];
#pragma unroll
            for (int j = 0; j < NZ; j++) qs[j] += Ck[i * NZ + j] * li;
        }
        for (int i = 0; i < NX; i++) ps += l1[i] * I.c[s * NX + i];
    } else {
#pragma unroll
        for (int j = 0; j < NZ; j++) qs[j] = 0.0;
    }
    if (k > 0) {
        const double* l0 = I.dl + (k - 1) * NX;
#pragma unroll
        for (int i = 0; i < NX; i++) qs[i] -= l0[i];
    }
    I.pStep[s] = ps;
#pragma unroll
    for (int i = 0; i < NZ; i++) {
        if (i < nV) {
            I.qStep[o + i] = qs[i];
            const double d = qs[i] / H[i];
            const double dz = -d;
            I.dz[o + i] = dz;
            const double r0 = dz / I.y[2 * o + 2 * i];
            const double r1 = dz / -I.y[2 * o + 2 * i + 1];
            const double a = 1. / ((r0 > r1) ? r0 : r1);
            if (a > 0. && a < amin) amin = a;
        }
    }
    return amin;
}

NMPC_DEV double trial_slope(const BatchView& v, const Inst& I, double a, double* lds) {
    for (int k = threadIdx.x; k <= v.N; k += kWave) stage_trial_point(v, I, k, a);
    __syncthreads();
    return ordered_sum(NX * v.N, lds, [&](int r) {
        return gradient_row(v, I, I.tmp, r) * I.dl[r];
    });
}

}  // namespace

__global__ __launch_bounds__(kWave) void k_newton_step(BatchView v) {
    __shared__ double lds[kChunk];
    const int b = blockIdx.x, lane = threadIdx.x;
    Inst I = instance(v, b);
    if (I.sc->status != ST_RUNNING) return;
    const int N = v.N, n = NX * N;
    const bool regularized = I.sc->regularized != 0;
    double objInc = I.sc->objInc;

    double amin = kInfty;
    for (int k = lane; k <= N; k += kWave) amin = stage_direction(v, I, k, amin);
    const double alphaMinAS = wave_min(amin);
    __syncthreads();

    double alpha;
    int lsIter = 0;
    int st = ST_OK;
    if (alphaMinAS > 1. - kEqTol && !regularized) {
        alpha = 1.;
    } else {
        // backtracking on the dual objective (dual_qp.c:1294-1334)
        double alphaMax = 1.;
        const double minProg = kLsMinRel * fabs(objInc) + kLsMinAbs;
        const double nd = sqrt(ordered_sum(n, lds, [&](int r) { return I.dl[r] * I.dl[r]; }));
        alpha = alphaMax;
        st = ST_LS_MAXITER;
        for (; lsIter < kMaxLsIter; ++lsIter) {
            const double obj = ordered_sum(N + 1, lds, [&](int k) {
                return stage_trial_value(v, I, k, alpha);
            });
            if (obj > objInc + minProg) { st = ST_OK; break; }
            alpha = alpha * kLsReduction;
            if (nd * (alpha - 0.) < kEqTol) { alpha = 0.; st = ST_LS_MIN_STEP; break; }
        }
        if (st == ST_LS_MIN_STEP) {
            if (lane == 0) {
                I.sc->status = ST_NO_ASCENT;
                I.sc->alpha = alpha;
                I.sc->lsIter = lsIter;
            }
            return;
        }
        {
            const double cand = alpha / kLsReduction;
            alphaMax = (alphaMax < cand) ? alphaMax : cand;
        }
        // gradient bisection (dual_qp.c:1341-1470)
        double alphaMin = alphaMinAS, alphaC = 1.0, slope;
        const double normal = (1. < nd) ? 1. : nd;
        bool done = false;
        st = ST_LS_MAXITER;
        for (; lsIter < kMaxLsRefine; ++lsIter) {
            slope = trial_slope(v, I, alphaMax, lds);
            if (fabs(slope / normal) <= kLsStatTol) { alpha = alphaMax; st = ST_OK; done = true; break; }
            if (slope / normal < 0.) break;
            alphaMin = alphaMax;
            alphaMax *= kLsIncrease;
            if (alphaMax > kLsMaxStep) { alpha = alphaMin; st = ST_LS_MAX_STEP; done = true; break; }
        }
        if (!done) {
            for (; lsIter < kMaxLsRefine; ++lsIter) {
                alphaC = 0.5 * (alphaMin + alphaMax);
                slope = trial_slope(v, I, alphaC, lds);
                if (fabs(slope / normal) <= kLsStatTol) { alpha = alphaC; st = ST_OK; done = true; break; }
                if (slope > 0) alphaMin = alphaC; else alphaMax = alphaC;
            }
            if (!done) alpha = alphaC;
        }
    }

    // commit: multipliers, stage variables, objective
    for (int r = lane; r < n; r += kWave) {
        double* lp = I.lam + (size_t)lam_slot(v, r / NX) * NX + r % NX;
        *lp = *lp + alpha * I.dl[r];
    }
    objInc = commit_all(v, I, alpha, lds);

    // active-set bookkeeping (dual_qp.c:197-214)
    for (int e = lane; e < (N + 1) * NZ; e += kWave) I.prev[e] = I.ieq[e];
    __syncthreads();
    const int last = refresh_active_sets(v, I);
    if (lane == 0) {
        I.sc->objInc = objInc;
        I.sc->alpha = alpha;
        I.sc->lsIter = lsIter;
        I.sc->last = last;
    }
}

// ------------------------------------------------------------------------
// One thread per instance.
__global__ void k_qp_finish(BatchView v) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= v.B) return;
    Inst I = instance(v, b);
    Scalars* sc = I.sc;
    if (sc->status == ST_RUNNING) {
        sc->status = ST_ITER_LIMIT;
        sc->numIter = kMaxIter;
    }
    if (sc->status == ST_OPTIMAL) {
        const int s0 = stage_slot(v, 0);
        for (int i = 0; i < NU; i++)
            sc->u[i] = I.uref[s0 * NU + i] + I.z[s0 * NZ + NX + i];
        sc->lastResult = 1;
    } else {
        sc->lastResult = 0;
    }
}

// After the ring offsets advanced by one: the new last multiplier block is a
// copy of the old last one (qpDUNES_shiftLambda leaves lambda[N-1] in place).
__global__ void k_shift_lambda_tail(BatchView v) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= v.B * NX) return;
    const int b = t / NX, i = t % NX;
    double* lam = v.lam + (size_t)b * v.N * NX;
    lam[(size_t)lam_slot(v, v.N - 1) * NX + i] = lam[(size_t)lam_slot(v, v.N - 2) * NX + i];
}

// controls / status / iterations gather: out_u [B][3], out_status [B], out_iters [B]
__global__ void k_gather_results(BatchView v, double* __restrict__ out_u,
                                 int* __restrict__ out_status, int* __restrict__ out_iters) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= v.B) return;
    const Scalars* sc = v.sc + b;
    out_u[b * 3 + 0] = sc->u[0];
    out_u[b * 3 + 1] = sc->u[1];
    out_u[b * 3 + 2] = sc->u[2];
    out_status[b] = sc->lastResult ? 0 : 1;
    out_iters[b] = sc->numIter;
}

// ========================================================================
// launchers
// ========================================================================
static inline unsigned blocks_for(long n, int bs) { return (unsigned)((n + bs - 1) / bs); }

void launch_store_refs(const BatchView& v, const double* refs, int first, int count, hipStream_t s) {
    const long n = (long)v.B * count;
    hipLaunchKernelGGL(k_store_refs, dim3(blocks_for(n, 256)), dim3(256), 0, s, v, refs, first, count);
}
void launch_init_stages(const BatchView& v, hipStream_t s) {
    const long n = (long)v.B * (v.N + 1);
    hipLaunchKernelGGL(k_init_stages, dim3(blocks_for(n, 64)), dim3(64), 0, s, v);
}
void launch_linearise(const BatchView& v, int first, int count, hipStream_t s) {
    const long n = (long)v.B * count * 16;
    hipLaunchKernelGGL(k_linearise, dim3(blocks_for(n, 256)), dim3(256), 0, s, v, first, count);
}
void launch_linearise_free(long n, const double* x, const double* u, const double* xn,
                           const double* w, const Bounds& bnd, double* jac, double* res,
                           hipStream_t s) {
    hipLaunchKernelGGL(k_linearise_free, dim3(blocks_for(n * 16, 256)), dim3(256), 0, s, n, x, u, xn,
                       w, bnd, jac, res);
}
void launch_rk4_free(long n, const double* x, const double* u, const double* w, double delta,
                     double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_rk4_free, dim3(blocks_for(n, 256)), dim3(256), 0, s, n, x, u, w, delta, out);
}
void launch_qp_begin(const BatchView& v, const double* meas, hipStream_t s) {
    hipLaunchKernelGGL(k_qp_begin, dim3(v.B), dim3(kWave), 0, s, v, meas);
}
void launch_newton_setup(const BatchView& v, int it, hipStream_t s) {
    hipLaunchKernelGGL(k_newton_setup, dim3(v.B), dim3(kWave), 0, s, v, it);
}
void launch_newton_solve(const BatchView& v, hipStream_t s) {
    hipLaunchKernelGGL(k_newton_solve, dim3(v.B), dim3(kWave), 0, s, v);
}
void launch_newton_step(const BatchView& v, hipStream_t s) {
    hipLaunchKernelGGL(k_newton_step, dim3(v.B), dim3(kWave), 0, s, v);
}
void launch_qp_finish(const BatchView& v, hipStream_t s) {
    hipLaunchKernelGGL(k_qp_finish, dim3(blocks_for(v.B, 256)), dim3(256), 0, s, v);
}
void launch_shift_lambda_tail(const BatchView& v, hipStream_t s) {
    hipLaunchKernelGGL(k_shift_lambda_tail, dim3(blocks_for((long)v.B * NX, 256)), dim3(256), 0, s, v);
}
void launch_gather_results(const BatchView& v, double* u, int* status, int* iters, hipStream_t s) {
    hipLaunchKernelGGL(k_gather_results, dim3(blocks_for(v.B, 256)), dim3(256), 0, s, v, u, status, iters);
}

int max_newton_iterations() { return kMaxIter; }

}  // namespace nmpc
