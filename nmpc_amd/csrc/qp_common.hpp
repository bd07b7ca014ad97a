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

// Value held by lane SRC of this lane's 16-lane DPP row (every lane of the row
// must be active).
template <int SRC>
NMPC_DEV double row_bcast(double x) {
    return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + SRC, 0xf, 0xf, true);
}

NMPC_DEV double bcast_lane(double x, int src) {   // src must be wave-uniform
    int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
}

// Clip val to [lo, up], recording the gaps first
// (stage_qp_solver_clipping.c:174-198).
NMPC_DEV double clip_gap(double val, double lo, double up, double* y0, double* y1) {
    *y0 = lo - val;
    *y1 = val - up;
    if (*y0 >= -kActTol) val = lo;
    else if (*y1 >= -kActTol) val = up;
    return val;
}

// 0.5 z'Hz + q'z + p with the reference's accumulation order
// (dual_qp.c:1552-1559, matrix_vector.c:1887-1900, 2100-2112).  z and q may
// point to LDS.
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

// Row r = 12 k + i of the dual gradient: C_k z_k + c_k - z_{k+1}
// (dual_qp.c:574-597) evaluated on the primal array `zz` (slot-indexed).
// A row of C is 15 contiguous doubles, 8-byte aligned: fetch it as seven
// 16-byte loads and one 8-byte load, which halves the number of cache-line
// look-ups of a wavefront whose lanes each read their own row.
typedef double pair_u __attribute__((ext_vector_type(2), aligned(8)));
NMPC_DEV void load_row15(const double* row, double (&out)[NZ]) {
#pragma unroll
    for (int j = 0; j + 1 < NZ; j += 2) {
        const pair_u p = *(const pair_u*)(row + j);
        out[j] = p.x;
        out[j + 1] = p.y;
    }
    out[NZ - 1] = row[NZ - 1];
}

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

}  // namespace nmpc
