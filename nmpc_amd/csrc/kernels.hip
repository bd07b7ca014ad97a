// HIP kernels (gfx950) for the batched NMPC hot path, part 1: reference
// storage, controller initialisation and the stage linearisation.
//
//   k_store_refs      reference points -> xref / uref rings
//   k_init_stages     nmpc_init for every instance      (cnmpc.c:872-948)
//   k_linearise       RK4 + finite-difference stage linearisation, one lane per
//                     (instance, stage, column), fused with the stage-QP
//                     re-staging                         (cnmpc.c:583-667, 1022-1080;
//                                                        qpDUNES/setup_qp.c:547-585)
//   k_linearise_free  the same stage body on caller-supplied samples
//   k_rk4_free        one RK4 step on caller-supplied samples
//   k_qp_finish       iteration-limit status, controls   (dual_qp.c:219-222, cnmpc.c:705-726)
//   k_shift_lambda_tail, k_gather_results
//
// The dual Newton QP solve lives in qp_stage.hip and qp_newton.hip.
// Built with -ffp-contract=off: no FMA contraction anywhere.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

#include "kernels.hpp"
#include "qp_common.hpp"

// Minimum resident wavefronts per SIMD the stage linearisation is compiled for.
// The stateless kernel fits three (165 registers).  The batch kernel also holds
// the values its re-staging needs across the integration (178 registers): held
// to three it spills them and is 9 % slower than at two.
#ifndef NMPC_LINEARISE_WAVES
#define NMPC_LINEARISE_WAVES 2
#endif
#ifndef NMPC_LINEARISE_FREE_WAVES
#define NMPC_LINEARISE_FREE_WAVES 3
#endif

#pragma clang fp contract(off)

namespace nmpc {

// ------------------------------------------------------------------------
// Reference points from waypoints (interpolate_reference of the reference's
// harness, scripts/nmpc-xplane-synthetic.py:62-78): waypoints [B][W][17] rows
// (t, 16-vector), sorted by t per instance; point i of the output is the
// piecewise-linear interpolation at t0 + (first + i) dt with the attitude
// re-normalised (scaled by 1 / |q|, scripts/vectors.py:299-310).  Past the last
// waypoint the last waypoint is returned as it is; before the first one the
// first (the harness never asks for that).
__global__ void k_refs_from_waypoints(int B, const real_t* __restrict__ wp, int W, real_t t0,
                                      real_t dt, int first, int count, real_t* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * count) return;
    const int b = t / count, i = t % count;
    const real_t* P = wp + (size_t)b * W * 17;
    const real_t at = t0 + (real_t)(first + i) * dt;
    int index = 0;                              // bisect.bisect: first waypoint later than `at`
    while (index < W && !(at < P[index * 17])) ++index;
    real_t* o = out + (size_t)t * 16;
    if (index >= W || index == 0) {
        const real_t* src = P + (index >= W ? W - 1 : 0) * 17;
        for (int j = 0; j < 16; j++) o[j] = src[1 + j];
        return;
    }
    const real_t* a = P + (index - 1) * 17;
    const real_t* c = P + index * 17;
    const real_t r = (at - a[0]) / (c[0] - a[0]);
    real_t pt[16];
#pragma unroll
    for (int j = 0; j < 16; j++) pt[j] = a[1 + j] + (c[1 + j] - a[1 + j]) * r;
    real_t m = 0.;
#pragma unroll
    for (int j = 6; j < 10; j++) m += pt[j] * pt[j];
    const real_t scale = 1.0 / sqrt(m);
#pragma unroll
    for (int j = 6; j < 10; j++) pt[j] = pt[j] * scale;
#pragma unroll
    for (int j = 0; j < 16; j++) o[j] = pt[j];
}

// ------------------------------------------------------------------------
// refs: [B][count][16] staging buffer; point i of the range goes to horizon
// position first + i.  Control part of position p >= 1 belongs to stage p - 1.
__global__ void k_store_refs(BatchView v, const real_t* __restrict__ refs,
                             int first, int count) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = v.B * count;
    if (t >= total) return;
    const int b = t / count, i = t % count, pos = first + i;
    const real_t* r = refs + (size_t)t * 16;
    real_t* x = v.xref + ((size_t)b * (v.N + 1) + xref_slot(v, pos)) * NS;
#pragma unroll
    for (int e = 0; e < NS; e++) x[e] = r[e];
    if (pos >= 1) {
        real_t* u = v.uref + ((size_t)b * v.N + stage_slot(v, pos - 1)) * NU;
        u[0] = r[13]; u[1] = r[14]; u[2] = r[15];
    }
}

// ------------------------------------------------------------------------
// nmpc_init: global bounds, every stage QP re-staged against the current
// multipliers, active-set record invalidated.  The Jacobians C and residuals c
// are zero at this point (the caller clears them), so the re-staging
// (qpDUNES/setup_qp.c:547-585, stage_qp_solver_clipping.c:42-98: q := 0,
// qStep := C' lambda_{k+1} - lambda_k + g, pStep := c . lambda_{k+1},
// dz := -(qStep / H), zUnc := 0) multiplies the multipliers by literal zeros.
// One thread per (instance, position, variable): contiguous stores.
// Linearising a stage rewrites every vector this kernel writes for it, so the
// host runs the kernel in two parts: the active-set record and the terminal
// stage at once (part 1), the stages 0..N-1 (part 2) only if something turns up
// that reads them before every stage has been linearised.
__global__ void k_init_stages(BatchView v, int part) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = (v.N + 1) * NZ;
    if (t >= (long)v.B * per) return;
    const int b = (int)(t / per), r = (int)(t % per);
    const int k = r / NZ, i = r - k * NZ;
    Inst I = instance(v, b);
    if (part == 1) I.prev[r] = -42;
    if ((part == 1) != (k == v.N)) return;
    const int nV = k < v.N ? NZ : NX;
    if (i >= nV) return;
    const int s = stage_slot(v, k);
    const real_t* H = stage_hessian(v, k);
    real_t qs = 0., ps = 0.;
    if (k < v.N) {
        const real_t* l1 = I.lam + (size_t)lam_slot(v, k) * NX;
        for (int m = 0; m < NX; m++) {
            const real_t li = l1[m];
            qs += 0.0 * li;
            ps += li * 0.0;
        }
    }
    if (k > 0 && i < NX) qs -= I.lam[(size_t)lam_slot(v, k - 1) * NX + i];
    if (i == 0) I.pStep[s] = ps;
    const real_t qv = qs + 0.0;
    const int e = s * NZ + i;
    I.qStep[e] = qv;
    const real_t d = qv / H[i];
    I.dz[e] = -d;
    I.zUnc[e] = 0.0;
    I.q[e] = 0.0;
    real_t lo = -kInfty, up = kInfty;
    if (i >= NX) { lo = v.lb[i - NX]; up = v.ub[i - NX]; }
    I.zLow[e] = lo;
    I.zUpp[e] = up;
}

// ------------------------------------------------------------------------
// Stage linearisation.  Lane j of each 16-lane group integrates the reference
// point perturbed along column j (j == 15: unperturbed).  The nominal end state
// is broadcast inside the group; lane j < 15 then owns column j of the 12x15
// Jacobian, lane 15 owns the residual.
//
// The kernel is bound by instruction issue, not by memory, so it is written to
// keep registers low enough for three wavefronts per SIMD: lane i < 13 of a
// group fetches entry i of the stage's state and of the next state (one
// register pair each instead of 26 per lane; the 13 entries are one contiguous
// 104-byte read per group), every lane collects the state through DPP row
// broadcasts, and the next state is only unpacked once the integration is over.
using Seq12 = std::make_integer_sequence<int, NX>;
template <int... I>
NMPC_DEV void gather_row(real_t (&out)[NS], real_t mine, std::integer_sequence<int, I...>) {
    ((out[I] = row_bcast<I>(mine)), ...);
}
template <int... I>
NMPC_DEV void bcast_from_last(real_t (&out)[NS], const real_t (&in)[NS],
                              std::integer_sequence<int, I...>) {
    ((out[I] = row_bcast<15>(in[I])), ...);
}

// zdev: null, or the stage variable z (15 entries) the expansion point is moved
// by: x (+) z[0:12], u + z[12:15]; this lane's entry of it comes back in *zj.
template <int... I>
NMPC_DEV void gather_row15(real_t (&out)[NZ], real_t mine, std::integer_sequence<int, I...>) {
    ((out[I] = row_bcast<I>(mine)), ...);
}

NMPC_DEV void linearise_lane(const real_t* xr, const real_t* xq, const real_t* ur,
                             const real_t* wr, const real_t* lower, const real_t* upper,
                             int j, real_t* out12, const real_t* zdev = nullptr,
                             real_t* zj = nullptr) {
    using Seq13 = std::make_integer_sequence<int, NS>;
    const int li = j < NS ? j : 0;
    const real_t mine = xr[li], mine_next = xq[li];
    real_t x[NS], u[NU], wind[3], moved[NS], base[NS], next[NS];
    gather_row(x, mine, Seq13{});
#pragma unroll
    for (int i = 0; i < NU; i++) { u[i] = ur[i]; wind[i] = wr[i]; }
    if (zdev) {
        const real_t zmine = zdev[j < NZ ? j : 0];
        real_t z[NZ], x0[NS];
        gather_row15(z, zmine, std::make_integer_sequence<int, NZ>{});
#pragma unroll
        for (int i = 0; i < NS; i++) x0[i] = x[i];
        state_plus_delta(x, x0, z);
#pragma unroll
        for (int i = 0; i < NU; i++) u[i] = u[i] + z[NX + i];
        *zj = zmine;
    }
    const real_t hinv = perturb_point(x, u, j, lower, upper);
    rk4_step(moved, x, u, wind, kStepLength);
    bcast_from_last(base, moved, Seq13{});              // nominal end state
    gather_row(next, mine_next, Seq13{});               // (every lane of the row takes part)
#pragma unroll
    for (int i = 0; i < NS; i++) base[i] = (j == 15) ? next[i] : base[i];
    state_to_delta(out12, base, moved);
    if (j < 15) {
#pragma unroll
        for (int r = 0; r < NX; r++) out12[r] = out12[r] * hinv;
    }
}

template <int... I>
NMPC_DEV real_t dot_with_row(const real_t (&col)[NX], real_t mine, std::integer_sequence<int, I...>) {
    real_t acc = 0.;
    ((acc += col[I] * row_bcast<I>(mine)), ...);
    return acc;
}

// The multipliers, bounds and Hessian entry the re-staging needs are fetched
// before the integration, one value per lane (a wavefront that asked for them
// afterwards would sit out the memory latency with nothing left to do), and
// the multiplier block is shared out by row broadcasts at the end.
// ABOUT_SOLUTION: the expansion point of every stage is moved by the stage's
// current solution z and the residual is taken back by C z, so that the stage
// stays a deviation from the reference: c = delta(x_ref[k+1], f(x, u)) - C z
// (the real-time-iteration preparation step; the reference has no code for it).
template <int... J>
NMPC_DEV real_t sum_over_columns(real_t mine, std::integer_sequence<int, J...>) {
    real_t acc = 0.;
    ((acc += row_bcast<J>(mine)), ...);
    return acc;
}

template <bool ABOUT_SOLUTION>
__global__ __launch_bounds__(256, NMPC_LINEARISE_WAVES) void k_linearise(BatchView v, int first, int count) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = (int)(gid & 15);
    long st = gid >> 4;
    const long total = (long)v.B * count;
    const bool live = st < total;
    if (!live) st = total - 1;              // keep the whole group in the broadcasts
    const int b = (int)(st / count), k = first + (int)(st % count);
    const int s = stage_slot(v, k);
    const real_t* ur = v.uref + ((size_t)b * v.N + s) * NU;
    // multipliers of the two coupling constraints this stage takes part in
    const real_t* lam = v.lam + (size_t)b * v.N * NX;
    const real_t lam_next = lam[(size_t)lam_slot(v, k) * NX + (j < NX ? j : 0)];
    const bool has_prev = k > 0 && j < NX;
    const real_t lam_prev = has_prev ? lam[(size_t)lam_slot(v, k - 1) * NX + j] : 0.;
    real_t hj = v.H[0], lo = -kInfty, up = kInfty;
#pragma unroll
    for (int i = 1; i < NZ; i++) hj = (i == j) ? v.H[i] : hj;
    if (j >= NX && j < NZ) {
        const real_t uj = ur[j - NX];
        real_t lbj = 0.0, ubj = 0.0;
#pragma unroll
        for (int i = 0; i < NU; i++) {
            lbj = (i == j - NX) ? v.lb[i] : lbj;
            ubj = (i == j - NX) ? v.ub[i] : ubj;
        }
        lo = lbj - uj;
        up = ubj - uj;
    }
    real_t col[NX];
    {
        const real_t* xref = v.xref + (size_t)b * (v.N + 1) * NS;
        real_t zj = 0.;
        const real_t* zdev = ABOUT_SOLUTION ? v.z + ((size_t)b * (v.N + 1) + s) * NZ : nullptr;
        linearise_lane(xref + (size_t)xref_slot(v, k) * NS, xref + (size_t)xref_slot(v, k + 1) * NS,
                       ur, v.wind + (size_t)b * 3, v.lb, v.ub, j, col, zdev, &zj);
        if (ABOUT_SOLUTION) {
            // row r of C z: the columns' products added in column order; lane 15 keeps them
#pragma unroll
            for (int r = 0; r < NX; r++) {
                const real_t prod = (j < NZ) ? col[r] * zj : 0.;
                const real_t cz = sum_over_columns(prod, std::make_integer_sequence<int, NZ>{});
                if (j == 15) col[r] = col[r] - cz;
            }
        }
    }
    // C' lambda_{k+1} (lanes j < 15) and c . lambda_{k+1} (lane 15): the same sum
    const real_t dot = dot_with_row(col, lam_next, Seq12{});
    if (!live) return;

    const size_t sb = (size_t)b, N = (size_t)v.N, N1 = N + 1;
    if (j < 15) {
        real_t* Ck = v.C + (sb * N + s) * (NX * NZ);
#pragma unroll
        for (int r = 0; r < NX; r++) Ck[r * NZ + j] = col[r];
        real_t qs = dot;
        if (has_prev) qs -= lam_prev;
        qs = qs + 0.0;
        const real_t d = qs / hj;
        const size_t e = (sb * N1 + s) * NZ + j;
        v.qStep[e] = qs;
        v.dz[e] = -d;
        v.zUnc[e] = 0.0;
        v.q[e] = 0.0;
        v.zLow[e] = lo;
        v.zUpp[e] = up;
    } else {
        real_t* cs = v.c + (sb * N + s) * NX;
#pragma unroll
        for (int i = 0; i < NX; i++) cs[i] = col[i];
        v.pStep[sb * N1 + s] = dot;
    }
}

// Stateless variant: n samples, inputs [n][13] [n][3] [n][13] [n][3],
// outputs jac [n][180] (row-major 12x15) and res [n][12].
__global__ __launch_bounds__(256, NMPC_LINEARISE_FREE_WAVES) void k_linearise_free(
    long n, const real_t* __restrict__ xs, const real_t* __restrict__ us,
    const real_t* __restrict__ xnexts, const real_t* __restrict__ winds,
    Bounds bnd, real_t* __restrict__ jac, real_t* __restrict__ res) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = (int)(gid & 15);
    long st = gid >> 4;
    const bool live = st < n;
    if (!live) st = n - 1;
    real_t col[NX];
    linearise_lane(xs + st * NS, xnexts + st * NS, us + st * NU, winds + st * 3, bnd.lb, bnd.ub, j, col);
    if (!live) return;
    if (j < 15) {
#pragma unroll
        for (int r = 0; r < NX; r++) jac[st * (NX * NZ) + r * NZ + j] = col[r];
    } else {
#pragma unroll
        for (int r = 0; r < NX; r++) res[st * NX + r] = col[r];
    }
}

__global__ __launch_bounds__(256) void k_rk4_free(long n, const real_t* __restrict__ xs,
                           const real_t* __restrict__ us,
                           const real_t* __restrict__ winds, real_t delta,
                           real_t* __restrict__ out) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    real_t x[NS], u[NU], w[3], r[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) x[i] = xs[t * NS + i];
#pragma unroll
    for (int i = 0; i < NU; i++) { u[i] = us[t * NU + i]; w[i] = winds[t * 3 + i]; }
    rk4_step(r, x, u, w, delta);
#pragma unroll
    for (int i = 0; i < NS; i++) out[t * NS + i] = r[i];
}

// ------------------------------------------------------------------------
// End of a solve (dual_qp.c:219-222, cnmpc.c:705-726).  One thread per instance.
__global__ void k_qp_finish(BatchView v) {
    const int b = v.b0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= v.b0 + v.nb) return;
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
    real_t* lam = v.lam + (size_t)b * v.N * NX;
    lam[(size_t)lam_slot(v, v.N - 1) * NX + i] = lam[(size_t)lam_slot(v, v.N - 2) * NX + i];
}

// controls / status / iterations gather: out_u [B][3], out_status [B], out_iters [B]
__global__ void k_gather_results(BatchView v, real_t* __restrict__ out_u,
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

void launch_store_refs(const BatchView& v, const real_t* refs, int first, int count, hipStream_t s) {
    const long n = (long)v.B * count;
    hipLaunchKernelGGL(k_store_refs, dim3(blocks_for(n, 256)), dim3(256), 0, s, v, refs, first, count);
}
void launch_refs_from_waypoints(int B, const real_t* wp, int W, real_t t0, real_t dt, int first,
                                int count, real_t* out, hipStream_t s) {
    const long n = (long)B * count;
    hipLaunchKernelGGL(k_refs_from_waypoints, dim3(blocks_for(n, 256)), dim3(256), 0, s, B, wp, W, t0,
                       dt, first, count, out);
}
void launch_init_stages(const BatchView& v, int part, hipStream_t s) {
    const long n = (long)v.B * (v.N + 1) * NZ;
    hipLaunchKernelGGL(k_init_stages, dim3(blocks_for(n, 256)), dim3(256), 0, s, v, part);
}
void launch_linearise(const BatchView& v, int first, int count, hipStream_t s, bool about_solution) {
    const long n = (long)v.B * count * 16;
    if (about_solution)
        hipLaunchKernelGGL(k_linearise<true>, dim3(blocks_for(n, 256)), dim3(256), 0, s, v, first, count);
    else
        hipLaunchKernelGGL(k_linearise<false>, dim3(blocks_for(n, 256)), dim3(256), 0, s, v, first, count);
}
void launch_linearise_free(long n, const real_t* x, const real_t* u, const real_t* xn,
                           const real_t* w, const Bounds& bnd, real_t* jac, real_t* res,
                           hipStream_t s) {
    hipLaunchKernelGGL(k_linearise_free, dim3(blocks_for(n * 16, 256)), dim3(256), 0, s, n, x, u, xn,
                       w, bnd, jac, res);
}
void launch_rk4_free(long n, const real_t* x, const real_t* u, const real_t* w, real_t delta,
                     real_t* out, hipStream_t s) {
    hipLaunchKernelGGL(k_rk4_free, dim3(blocks_for(n, 256)), dim3(256), 0, s, n, x, u, w, delta, out);
}
void launch_qp_finish(const BatchView& v, hipStream_t s) {
    hipLaunchKernelGGL(k_qp_finish, dim3(blocks_for(v.nb, 256)), dim3(256), 0, s, v);
}
void launch_shift_lambda_tail(const BatchView& v, hipStream_t s) {
    hipLaunchKernelGGL(k_shift_lambda_tail, dim3(blocks_for((long)v.B * NX, 256)), dim3(256), 0, s, v);
}
void launch_gather_results(const BatchView& v, real_t* u, int* status, int* iters, hipStream_t s) {
    hipLaunchKernelGGL(k_gather_results, dim3(blocks_for(v.B, 256)), dim3(256), 0, s, v, u, status, iters);
}

int max_newton_iterations() { return kMaxIter; }

}  // namespace nmpc
