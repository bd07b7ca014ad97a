// Shared device helpers of the QP kernels: solver constants, per-instance
// addressing, ring-offset maps and the scalar building blocks whose operation
// order is fixed by the reference's qpDUNES 1.0beta
// (ccs-c66x/qpDUNES/stage_qp_solver_clipping.c, dual_qp.c, matrix_vector.c).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "layout.hpp"
#include "model.hpp"

#pragma clang fp contract(off)

namespace nmpc {

// Effective options: qpDUNES defaults (setup_qp.c:667-725) with the overrides
// of ccs-c66x/cnmpc.c:891-894.
// The option fields are real_t (qpDUNES/types.h:487-543): every value is rounded
// to real_t where it is stored, and the plain double literals next to them in
// the solver's expressions make those expressions double in both precisions.
constexpr real_t kInfty = (real_t)1.e12;
constexpr real_t kEqTol = (real_t)2.221e-16;
constexpr real_t kActTol = (real_t)(1e4 * kEqTol);
constexpr real_t kDiagRegTol = (real_t)1.e-10;
constexpr real_t kRegParam = (real_t)1.e-6;
constexpr real_t kStatTol = (real_t)1e-3f;
constexpr int kMaxIter = 5;
constexpr int kMaxLsIter = 19;
constexpr int kMaxLsRefine = 40;
constexpr real_t kLsReduction = (real_t)0.1;
constexpr real_t kLsIncrease = (real_t)1.5;
constexpr real_t kLsMinAbs = kEqTol;
constexpr real_t kLsMinRel = (real_t)1.e-14;
constexpr real_t kLsStatTol = (real_t)1.e-3;
constexpr real_t kLsMaxStep = (real_t)1.;
// unit roundoff of real_t
constexpr double kUnitRoundoff = sizeof(real_t) == 8 ? 1.1102230246251565e-16 : 5.9604644775390625e-8;

// The stage-QP kernels exist twice: for batches whose instances all have the
// batch's weights and bounds (kernel arguments), and, compiled with
// -DNMPC_OWN_CFG=1, for batches with a record per instance (BatchView::cfg); the
// choice is made at compile time so that the first form carries no trace of the
// second.
#ifndef NMPC_OWN_CFG
#define NMPC_OWN_CFG 0
#endif
constexpr bool kOwnCfg = NMPC_OWN_CFG != 0;

constexpr int kWave = 64;
constexpr int kWabsStride = 16;      // BatchView::wabs: 15 stage variables, then the stage's pStep

// Ring maps.  Offsets and positions are both below the ring length, so one
// conditional subtraction takes the place of a remainder (an integer division
// is some thirty instructions on this hardware).
NMPC_DEV int ring_wrap(int x, int len) { return x >= len ? x - len : x; }
NMPC_DEV int stage_slot(const BatchView& v, int k) {        // 0 <= k <= N
    return k < v.N ? ring_wrap(v.head + k, v.N) : v.N;
}
// inverse of stage_slot
NMPC_DEV int slot_position(const BatchView& v, int s) {     // 0 <= s <= N
    return s < v.N ? ring_wrap(s - v.head + v.N, v.N) : v.N;
}
NMPC_DEV int lam_slot(const BatchView& v, int k) { return ring_wrap(v.lhead + k, v.N); }       // 0 <= k < N
NMPC_DEV int xref_slot(const BatchView& v, int k) { return ring_wrap(v.xhead + k, v.N + 1); }  // 0 <= k <= N

// Per-instance base pointers.
struct Inst {
    const real_t* xref; const real_t* uref;
    real_t *C, *c, *zLow, *zUpp, *q, *z, *qStep, *zUnc, *dz, *tmp, *y, *p, *pStep;
    real_t *lam, *dl, *grad, *hess, *chol;
    real_t* wabs;
    int8_t *ieq, *prev, *chg;
    Scalars* sc;
    const real_t* cfg;      // kOwnCfg: the instance's own weights and bounds (BatchView::cfg)
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
    I.wabs = v.wabs ? v.wabs + sb * N1 * kWabsStride : nullptr;
    I.y = v.y + sb * N1 * 2 * NZ;
    I.p = v.p + sb * N1;
    I.pStep = v.pStep + sb * N1;
    I.lam = v.lam + sb * N * NX;
    I.dl = v.dl + sb * N * NX;
    I.grad = v.grad + sb * N * NX;
    I.hess = v.hess + sb * N * kCholRow;
    I.chol = v.chol + sb * N * kCholRow;
    I.ieq = v.ieq + sb * N1 * NZ;
    I.prev = v.prev + sb * N1 * NZ;
    I.chg = v.chg + sb * N1;
    I.sc = v.sc + sb;
    I.cfg = kOwnCfg ? v.cfg + sb * kCfgStride : nullptr;
    return I;
}

// Instance that item i (a workgroup, or a 16-lane group) of a launch works on, -1: none.
NMPC_DEV int launch_instance(const BatchView& v, int i) {
    if (v.list) {
        const int at = v.listRange[0] + i;
        return at < v.listRange[1] ? v.list[at] : -1;
    }
    return i < v.nb ? v.b0 + i : -1;
}

// Where row l of a diagonal block starts in a block row of the Newton Hessian or of
// its factor (kCholDiagRow), l known only at run time.
NMPC_DEV int chol_diag_row(int l) {
    int at = kCholDiagRow[0];
#pragma unroll
    for (int i = 1; i < NX; i++) at = (l == i) ? kCholDiagRow[i] : at;
    return at;
}

NMPC_DEV const real_t* stage_hessian(const BatchView& v, int k) {
    return k < v.N ? v.H : v.Ht;
}
// Entry j of the diagonal of stage k's Hessian.
NMPC_DEV real_t hessian_entry(const BatchView& v, const Inst& I, int k, int j) {
    if constexpr (kOwnCfg) return I.cfg[(k < v.N ? kCfgH : kCfgHt) + j];
    else return k < v.N ? v.H[j] : v.Ht[j];
}
NMPC_DEV real_t lower_control_bound(const BatchView& v, const Inst& I, int i) {
    if constexpr (kOwnCfg) return I.cfg[kCfgLb + i];
    else return v.lb[i];
}
NMPC_DEV real_t upper_control_bound(const BatchView& v, const Inst& I, int i) {
    if constexpr (kOwnCfg) return I.cfg[kCfgUb + i];
    else return v.ub[i];
}

NMPC_DEV int wave_max(int x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        int y = __shfl_xor(x, o);
        x = y > x ? y : x;
    }
    return x;
}

NMPC_DEV real_t wave_min(real_t x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        real_t y = __shfl_xor(x, o);
        x = y < x ? y : x;
    }
    return x;
}

// Value held by lane SRC of this lane's 16-lane DPP row (every lane of the row
// must be active).
template <int SRC>
NMPC_DEV real_t row_bcast(real_t x) {
    return __builtin_amdgcn_update_dpp((real_t)0.0, x, 0x150 + SRC, 0xf, 0xf, true);
}

NMPC_DEV double bcast_lane(double x, int src) {   // src must be wave-uniform
    int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
}
NMPC_DEV float bcast_lane(float x, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), src));
}

// Clip val to [lo, up], recording the gaps first
// (stage_qp_solver_clipping.c:174-198).
NMPC_DEV real_t clip_gap(real_t val, real_t lo, real_t up, real_t* y0, real_t* y1) {
    *y0 = lo - val;
    *y1 = val - up;
    if (*y0 >= -kActTol) val = lo;
    else if (*y1 >= -kActTol) val = up;
    return val;
}

// 0.5 z'Hz + q'z + p with the reference's accumulation order
// (dual_qp.c:1552-1559, matrix_vector.c:1887-1900, 2100-2112).  z and q may
// point to LDS.
NMPC_DEV real_t stage_value(const real_t* H, const real_t* z, const real_t* q,
                            real_t p, int nV) {
    real_t zHz = 0., qz = 0.;
#pragma unroll
    for (int i = 0; i < NZ; i++) if (i < nV) zHz += H[i] * z[i] * z[i];
    real_t val = 0.5 * zHz;
#pragma unroll
    for (int i = 0; i < NZ; i++) if (i < nV) qz += q[i] * z[i];
    val += qz;
    val += p;
    return val;
}

// Row r = 12 k + i of the dual gradient: C_k z_k + c_k - z_{k+1}
// (dual_qp.c:574-597) evaluated on the primal array `zz` (slot-indexed).
// A row of C is 15 contiguous doubles, 8-byte aligned: fetch it as seven
// 16-byte loads and one 8-byte load, which halves the number of cache-line
// look-ups of a wavefront whose lanes each read their own row.
typedef real_t pair_u __attribute__((ext_vector_type(2), aligned(sizeof(real_t))));
// Naturally aligned pair: the two gaps of a stage variable (y is [variable][2]
// from a 256-byte aligned base), two consecutive entries of a Hessian block row.
typedef real_t pair_a __attribute__((ext_vector_type(2)));
NMPC_DEV void store_pair(real_t* p, real_t a, real_t b) {
    pair_a v;
    v.x = a;
    v.y = b;
    *(pair_a*)p = v;
}
NMPC_DEV void load_row15(const real_t* row, real_t (&out)[NZ]) {
#pragma unroll
    for (int j = 0; j + 1 < NZ; j += 2) {
        const pair_u p = *(const pair_u*)(row + j);
        out[j] = p.x;
        out[j + 1] = p.y;
    }
    out[NZ - 1] = row[NZ - 1];
}

NMPC_DEV real_t gradient_row(const BatchView& v, const Inst& I, const real_t* zz, int r) {
    const int k = r / NX, i = r % NX;
    const int s = stage_slot(v, k), sn = stage_slot(v, k + 1);
    const real_t* Crow = I.C + (size_t)s * (NX * NZ) + i * NZ;
    const real_t* zk = zz + s * NZ;
    real_t a = 0.;
#pragma unroll
    for (int j = 0; j < NZ; j++) a += Crow[j] * zk[j];
    a += I.c[s * NX + i];
    a -= zz[sn * NZ + i];
    return a;
}

}  // namespace nmpc
