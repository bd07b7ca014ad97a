// HIP kernels (gfx950) for the batched NMPC hot path, part 1: reference
// storage, controller initialisation and the stage linearisation.
//
//   k_store_refs      reference points -> xref / uref rings
//   k_init_stages     nmpc_init for every instance      (cnmpc.c:872-948)
//   k_linearise       RK4 + finite-difference stage linearisation, one lane per
//                     (instance, stage, integrated column), fused with the stage-QP
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
#ifndef NMPC_LINEARISE_WAVES
#define NMPC_LINEARISE_WAVES 3
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
// One thread per entry: consecutive lanes read consecutive words.
__global__ void k_store_refs(BatchView v, const real_t* __restrict__ refs,
                             int first, int count) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)v.B * count * 16) return;
    const int e = (int)(t & 15);
    const long p = t >> 4;
    const int b = (int)(p / count), i = (int)(p % count), pos = first + i;
    const real_t val = refs[t];
    if (e < NS)
        v.xref[((size_t)b * (v.N + 1) + xref_slot(v, pos)) * NS + e] = val;
    else if (pos >= 1)
        v.uref[((size_t)b * v.N + stage_slot(v, pos - 1)) * NU + (e - NS)] = val;
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
// that reads them before every stage has been linearised.  Part 3 is part 1 for
// the instances whose last solve converged (real-time-iteration preparation).
// OWN: every instance has weights and control bounds of its own (BatchView::cfg).
template <bool OWN>
__global__ void k_init_stages(BatchView v, int part) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = (v.N + 1) * NZ;
    if (t >= (long)v.B * per) return;
    const int b = (int)(t / per), r = (int)(t % per);
    const int k = r / NZ, i = r - k * NZ;
    Inst I = instance(v, b);
    if (part == 3) {
        if (I.sc->lastResult == 0) return;
        part = 1;
    }
    if (part == 1) I.prev[r] = -42;
    if ((part == 1) != (k == v.N)) return;
    const int nV = k < v.N ? NZ : NX;
    if (i >= nV) return;
    const int s = stage_slot(v, k);
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
    const real_t* own = v.cfg + (size_t)b * kCfgStride;         // OWN only
    real_t h;
    if constexpr (OWN) h = own[(k < v.N ? kCfgH : kCfgHt) + i];
    else h = stage_hessian(v, k)[i];
    const real_t d = qv / h;
    I.dz[e] = -d;
    I.zUnc[e] = 0.0;
    I.q[e] = 0.0;
    real_t lo = -kInfty, up = kInfty;
    if (i >= NX) {      // the control bounds nmpc_init saw
        if constexpr (OWN) { lo = own[kCfgLb0 + i - NX]; up = own[kCfgUb0 + i - NX]; }
        else { lo = v.lb[i - NX]; up = v.ub[i - NX]; }
    }
    I.zLow[e] = lo;
    I.zUpp[e] = up;
}

// ------------------------------------------------------------------------
// Stage linearisation.  Thirteen lanes per stage: lane 0 of a group integrates
// the unperturbed point, lane c > 0 the point perturbed along column c + 2 of
// the 12x15 Jacobian (velocity, attitude, body rates, controls).  The three
// position columns take no integration of their own: the model never reads the
// position, so their end state is the nominal one with one position entry
// re-added from the nominal lane's Runge-Kutta sums, and their Jacobian entries
// follow from the nominal lane alone (the differences of equal entries are
// evaluated, not assumed zero, so that non-finite states propagate as they do
// in the reference).
//
// A 256-lane workgroup holds 19 stages (247 lanes).  The groups do not line up
// with wavefronts or DPP rows, so states travel through LDS: 13 lanes fetch the
// 13 entries of the stage's state (one contiguous 104-byte read), every lane
// reads them back; the nominal end state comes the same way.  Jacobian and
// residual are assembled in an LDS tile and leave in 16-byte row-contiguous
// stores.  The kernel is bound by instruction issue, not memory; everything
// that is only needed after the integration (multipliers, Hessian, bounds)
// waits in LDS rather than in registers, which keeps three wavefronts per SIMD.
constexpr int kLanesPerStage = 13;
constexpr int kStagesPerBlock = 19;
constexpr int kTileRow = NX * NZ + NX;      // 180 Jacobian entries (row-major 12x15), 12 residuals
constexpr int kPerVec = 16 / (int)sizeof(real_t);
typedef real_t Vec16 __attribute__((ext_vector_type(kPerVec)));

struct LineariseLds {
    alignas(16) real_t tile[kStagesPerBlock][kTileRow];
    real_t x[kStagesPerBlock][NS];          // expansion point (reference state)
    real_t next[kStagesPerBlock][NS];       // reference state of the next stage
    real_t moved[kStagesPerBlock][NS];      // nominal end state
    real_t u[kStagesPerBlock][NU];          // reference control
    real_t wind[kStagesPerBlock][3];
    // batch kernel only
    real_t lam[kStagesPerBlock][NX];        // multipliers of the coupling to stage k + 1
    real_t lam_prev[kStagesPerBlock][NX];   // ... to stage k - 1
    real_t z[kStagesPerBlock][NZ];          // stage solution (real-time iteration)
    real_t H[NZ], lb[NU], ub[NU];
    // the same per stage for batches whose instances have weights and bounds of their own
    real_t own[kStagesPerBlock][NZ + 2 * NU];
    int row[kStagesPerBlock];               // (instance, slot) row of the stage in C and c
    int inst[kStagesPerBlock], pos[kStagesPerBlock], slot[kStagesPerBlock];
};

// Lanes of one group, after L.x[g], L.next[g], L.u[g], L.wind[g] (and L.z[g])
// are in place: leaves
// this lane's column (c > 0, scaled by 1 / h) or the residual (c == 0) in col
// and in the tile; the nominal lane also writes the position columns.  The
// caller synchronises before the tile is read.
// MOVE: the expansion point is moved by the stage solution, x (+) z[0:12],
// u + z[12:15].
template <bool MOVE>
NMPC_DEV void linearise_lanes(LineariseLds& L, bool active, int g, int c, const real_t* lower,
                              const real_t* upper, real_t (&col)[NX]) {
    const int j = c == 0 ? 15 : c + 2;
    real_t moved[NS];
    real_t hinv = (real_t)0.0;
    if (active) {
        real_t x[NS], u[NU], wind[3], ta[3], tb[3];
#pragma unroll
        for (int i = 0; i < NS; i++) x[i] = L.x[g][i];
#pragma unroll
        for (int i = 0; i < NU; i++) { u[i] = L.u[g][i]; wind[i] = L.wind[g][i]; }
        if (MOVE) {
            real_t z[NZ], x0[NS];
#pragma unroll
            for (int i = 0; i < NZ; i++) z[i] = L.z[g][i];
#pragma unroll
            for (int i = 0; i < NS; i++) x0[i] = x[i];
            state_plus_delta(x, x0, z);
#pragma unroll
            for (int i = 0; i < NU; i++) u[i] = u[i] + z[NX + i];
        }
        const real_t p0[3] = {x[0], x[1], x[2]};
        __builtin_assume(j >= 3);                   // no lane integrates a position column
        hinv = perturb_point(x, u, j, lower, upper);
        integrate_step(moved, x, u, wind, kStepLength, ta, tb);
        if (c == 0) {
#pragma unroll
            for (int i = 0; i < NS; i++) L.moved[g][i] = moved[i];
            // Columns 0..2 (cnmpc.c:600-606 with the perturbed entry carried
            // through the last line of the integrator, cnmpc.c:393-397 for RK4).
            real_t diag[3], same[NX];
#pragma unroll
            for (int i = 0; i < 3; i++) {
                real_t pm = (p0[i] + kFdStep) + ta[i];
                if constexpr (kIntegratorTerms == 2) pm = pm + tb[i];
                diag[i] = (pm - moved[i]) * hinv;
                same[i] = (moved[i] - moved[i]) * hinv;
                same[3 + i] = (moved[3 + i] - moved[3 + i]) * hinv;
                same[9 + i] = (moved[10 + i] - moved[10 + i]) * hinv;
            }
            real_t cj[4], e[4];
            cj[0] = moved[6]; cj[1] = moved[7]; cj[2] = moved[8]; cj[3] = -moved[9];
            quat_mul(e, &moved[6], cj);
            if (e[3] < 0) { e[0] = -e[0]; e[1] = -e[1]; e[2] = -e[2]; e[3] = -e[3]; }
            const real_t d = (real_t)4.0 / ((real_t)1.0 + e[3]);
#pragma unroll
            for (int i = 0; i < 3; i++) same[6 + i] = (e[i] * d) * hinv;
#pragma unroll
            for (int r = 0; r < NX; r++)
#pragma unroll
                for (int jp = 0; jp < 3; jp++) L.tile[g][r * NZ + jp] = (r == jp) ? diag[jp] : same[r];
        }
    }
    __syncthreads();
    if (active) {
        const real_t* bp = c == 0 ? L.next[g] : L.moved[g];
        real_t base[NS];
#pragma unroll
        for (int i = 0; i < NS; i++) base[i] = bp[i];
        state_to_delta(col, base, moved);
        if (c > 0) {
            real_t* t = &L.tile[g][j];
#pragma unroll
            for (int r = 0; r < NX; r++) { col[r] = col[r] * hinv; t[r * NZ] = col[r]; }
        } else {
            real_t* t = &L.tile[g][NX * NZ];
#pragma unroll
            for (int r = 0; r < NX; r++) t[r] = col[r];
        }
    }
}

// Jacobian rows and residual of the workgroup's stages from the tile to global
// memory; stage gg goes to row row_of(gg) of jac ([.][180]) and res ([.][12]).
template <bool WIDE, typename RowOf>
NMPC_DEV void store_tiles(const LineariseLds& L, int nvalid, real_t* jac, real_t* res, RowOf row_of) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    for (int gg = wave; gg < nvalid; gg += 4) {
        const long row = row_of(gg);                // the same for the whole wavefront
        if (row < 0) continue;
        real_t* jr = jac + row * (NX * NZ);
        real_t* rr = res + row * NX;
        const real_t* tl = L.tile[gg];
        if (WIDE) {
            constexpr int chunks = kTileRow / kPerVec, jchunks = NX * NZ / kPerVec;
#pragma unroll
            for (int i = 0; i < (chunks + 63) / 64; i++) {
                const int e = i * 64 + lane;
                if (e < chunks) {
                    real_t* dst = e < jchunks ? jr + e * kPerVec : rr + (e - jchunks) * kPerVec;
                    // Jacobians are next read a kernel later, after more data than the
                    // caches hold: stored non-temporally
                    __builtin_nontemporal_store(*reinterpret_cast<const Vec16*>(tl + e * kPerVec),
                                                reinterpret_cast<Vec16*>(dst));
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < kTileRow / 64; i++) {
                const int e = i * 64 + lane;
                real_t* dst = e < NX * NZ ? jr + e : rr + (e - NX * NZ);
                *dst = tl[e];
            }
        }
    }
}

// The stage-QP re-staging is fused in (qpDUNES/setup_qp.c:547-585,
// stage_qp_solver_clipping.c:42-98): qStep := C' lambda_{k+1} - lambda_k + g,
// pStep := c . lambda_{k+1}, dz := -(qStep / H), zUnc := 0, q := 0, bounds.
// ABOUT_SOLUTION: the expansion point of every stage is moved by the stage's
// current solution z and the residual is taken back by C z, so that the stage
// stays a deviation from the reference: c = delta(x_ref[k+1], f(x, u)) - C z
// (the real-time-iteration preparation step; the reference has no code for it).
// only_solved: the stages of instances whose last solve did not converge are
// left as they are.
// OWN: every instance has weights and control bounds of its own (BatchView::cfg).
template <bool ABOUT_SOLUTION, bool OWN>
__global__ __launch_bounds__(256, NMPC_LINEARISE_WAVES) void k_linearise(BatchView v, int first, int count,
                                                                         int only_solved) {
    __shared__ LineariseLds L;
    const int t = threadIdx.x, g = t / kLanesPerStage, c = t - g * kLanesPerStage;
    const long total = (long)v.B * count;
    const long st = (long)blockIdx.x * kStagesPerBlock + g;
    bool active = g < kStagesPerBlock && st < total;
    if (active && only_solved) active = v.sc[(unsigned)st / (unsigned)count].lastResult != 0;
    if (!OWN && t == 0) {           // kernel arguments: static indices only
#pragma unroll
        for (int i = 0; i < NZ; i++) L.H[i] = v.H[i];
#pragma unroll
        for (int i = 0; i < NU; i++) { L.lb[i] = v.lb[i]; L.ub[i] = v.ub[i]; }
    }
    if (active) {
        // (the launcher keeps B * count below 2^31: 32-bit division)
        const unsigned sq = (unsigned)st / (unsigned)count;
        const int b = (int)sq, k = first + (int)((unsigned)st - sq * (unsigned)count);
        const int s = stage_slot(v, k);
        // every load is issued before the first one is waited for
        const real_t* xref = v.xref + (size_t)b * (v.N + 1) * NS;
        const real_t* lam = v.lam + (size_t)b * v.N * NX;   // multipliers of the two couplings of the stage
        const int cl = c < NX ? c : 0, cu = c < NU + 3 ? c : 0;
        const real_t xv = xref[(size_t)xref_slot(v, k) * NS + c];
        const real_t nv = xref[(size_t)xref_slot(v, k + 1) * NS + c];
        const real_t lv = lam[(size_t)lam_slot(v, k) * NX + cl];
        const real_t pv = lam[(size_t)lam_slot(v, k > 0 ? k - 1 : 0) * NX + cl];
        const real_t uw = cu < NU ? v.uref[((size_t)b * v.N + s) * NU + cu] : v.wind[(size_t)b * 3 + (cu - NU)];
        real_t zv = 0., zw = 0.;
        if (ABOUT_SOLUTION) {
            const real_t* zp = v.z + ((size_t)b * (v.N + 1) + s) * NZ;
            zv = zp[c];
            zw = zp[c < NZ - kLanesPerStage ? kLanesPerStage + c : 0];
        }
        L.x[g][c] = xv;
        L.next[g][c] = nv;
        if (c < NX) { L.lam[g][c] = lv; L.lam_prev[g][c] = k > 0 ? pv : (real_t)0.0; }
        if (c < NU) L.u[g][c] = uw;
        else if (c < NU + 3) L.wind[g][c - NU] = uw;
        if (ABOUT_SOLUTION) {
            L.z[g][c] = zv;
            if (c < NZ - kLanesPerStage) L.z[g][kLanesPerStage + c] = zw;
        }
        if (c == 0) { L.row[g] = b * v.N + s; L.inst[g] = b; L.pos[g] = k; L.slot[g] = s; }
        if (OWN) {      // latched stage Hessian, live bounds
            const real_t* cfg = v.cfg + (size_t)b * kCfgStride;
            L.own[g][c] = cfg[kCfgH + c];
            if (c < NZ - kLanesPerStage) L.own[g][kLanesPerStage + c] = cfg[kCfgH + kLanesPerStage + c];
            if (c < 2 * NU) L.own[g][NZ + c] = cfg[kCfgLb + c];
        }
    } else if (c == 0 && g < kStagesPerBlock) {
        L.row[g] = -1;                              // no tile to store for this group
    }
    __syncthreads();
    real_t col[NX];
    if constexpr (OWN) {
        const int gg = active ? g : 0;
        linearise_lanes<ABOUT_SOLUTION>(L, active, g, c, &L.own[gg][NZ], &L.own[gg][NZ + NU], col);
    } else {
        linearise_lanes<ABOUT_SOLUTION>(L, active, g, c, v.lb, v.ub, col);
    }
    real_t dot_own = 0.;                            // own column of C . lambda_{k+1}
    if (active && c > 0) {
#pragma unroll
        for (int r = 0; r < NX; r++) dot_own += col[r] * L.lam[g][r];
    }
    __syncthreads();                                // the tile is complete
    if (ABOUT_SOLUTION) {
        if (active && c > 0) {                      // row c - 1 of C z, the columns in order
            const int r = c - 1;
            real_t cz = 0.;
#pragma unroll
            for (int jj = 0; jj < NZ; jj++) cz += L.tile[g][r * NZ + jj] * L.z[g][jj];
            L.tile[g][NX * NZ + r] = L.tile[g][NX * NZ + r] - cz;
        }
        __syncthreads();
    }
    // Global stores only from here on: a workgroup barrier behind a store waits
    // for the store to complete.  (Instance, position and slot come back from
    // LDS rather than staying in registers across the integration.)
    const int gs = active ? g : 0;
    const int b = L.inst[gs], k = L.pos[gs], s = L.slot[gs];
    const size_t zrow = ((size_t)b * (v.N + 1) + s) * NZ;
    auto restage = [&](int jv, real_t dot) {        // stage variable jv, dot = column jv of C . lambda_{k+1}
        real_t qs = dot;
        if (k > 0 && jv < NX) qs -= L.lam_prev[g][jv];
        qs = qs + 0.0;
        const real_t d = qs / (OWN ? L.own[g][jv] : L.H[jv]);
        real_t lo = -kInfty, up = kInfty;
        if (jv >= NX) {
            const real_t uj = L.u[g][jv - NX];
            lo = (OWN ? L.own[g][NZ + jv - NX] : L.lb[jv - NX]) - uj;
            up = (OWN ? L.own[g][NZ + NU + jv - NX] : L.ub[jv - NX]) - uj;
        }
        const size_t e = zrow + jv;
// Plain stores: fifteen lanes write a stage's 120 bytes, the lanes of the next
        // stage the 120 behind them, and the L2 puts the 32-byte sectors the two share
        // together before they leave.  Streaming stores wrote each such sector twice
        // (1.21 GB per launch of 409 600 stages against 0.95 GB now and 0.925 GB
        // algorithmic, rocprofv3 --pmc WRITE_SIZE; the kernel's time is the same).
        v.qStep[e] = qs;
        v.dz[e] = -d;
        v.zUnc[e] = (real_t)0.0;
        v.q[e] = (real_t)0.0;
        v.zLow[e] = lo;
        v.zUpp[e] = up;
    };
    if (active && c > 0) restage(c + 2, dot_own);
    if (active && c < 4) {          // the residual (c == 0) and the position columns against lambda_{k+1}
        const real_t* src = &L.tile[g][c == 0 ? NX * NZ : c - 1];
        const int stride = c == 0 ? 1 : NZ;
        real_t dot = 0.;
#pragma unroll
        for (int r = 0; r < NX; r++) dot += src[r * stride] * L.lam[g][r];
        if (c == 0) v.pStep[(size_t)b * (v.N + 1) + s] = dot;
        else restage(c - 1, dot);
    }
    const long left = total - (long)blockIdx.x * kStagesPerBlock;
    store_tiles<true>(L, left < kStagesPerBlock ? (int)left : kStagesPerBlock, v.C, v.c,
                      [&](int gg) { return (long)__builtin_amdgcn_readfirstlane(L.row[gg]); });
}

// Stateless variant: n samples, inputs [n][13] [n][3] [n][13] [n][3],
// outputs jac [n][180] (row-major 12x15) and res [n][12].
// WIDE: jac and res are 16-byte aligned.
template <bool WIDE>
__global__ __launch_bounds__(256, NMPC_LINEARISE_FREE_WAVES) void k_linearise_free(
    long n, const real_t* __restrict__ xs, const real_t* __restrict__ us,
    const real_t* __restrict__ xnexts, const real_t* __restrict__ winds,
    Bounds bnd, real_t* __restrict__ jac, real_t* __restrict__ res) {
    __shared__ LineariseLds L;
    const int t = threadIdx.x, g = t / kLanesPerStage, c = t - g * kLanesPerStage;
    const long first = (long)blockIdx.x * kStagesPerBlock;
    long st = first + g;
    const bool active = g < kStagesPerBlock && st < n;
    if (!active) st = 0;
    if (active) {
        const int cu = c < NU + 3 ? c : 0;          // (the loads are issued together)
        const real_t xv = xs[st * NS + c], nv = xnexts[st * NS + c];
        const real_t uw = cu < NU ? us[st * NU + cu] : winds[st * 3 + (cu - NU)];
        L.x[g][c] = xv;
        L.next[g][c] = nv;
        if (c < NU) L.u[g][c] = uw;
        else if (c < NU + 3) L.wind[g][c - NU] = uw;
    }
    __syncthreads();
    real_t col[NX];
    linearise_lanes<false>(L, active, g, c, bnd.lb, bnd.ub, col);
    __syncthreads();
    const long left = n - first;
    store_tiles<WIDE>(L, left < kStagesPerBlock ? (int)left : kStagesPerBlock, jac, res,
                      [&](int gg) { return first + gg; });
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
    integrate_step(r, x, u, w, delta);
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

// Real-time-iteration preparation: multipliers and the constant terms of the stage
// objectives of every instance whose last solve converged return to zero.
__global__ void k_rti_reset(BatchView v) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = v.N * NX + v.N + 1;
    if (t >= (long)v.B * per) return;
    const int b = (int)(t / per), r = (int)(t - (long)b * per);
    if (v.sc[b].lastResult == 0) return;
    if (r < v.N * NX) v.lam[(size_t)b * v.N * NX + r] = 0.;
    else v.p[(size_t)b * (v.N + 1) + (r - v.N * NX)] = 0.;
}

// ------------------------------------------------------------------------
// Per-instance weights and control bounds (BatchView::cfg, layout.hpp).
// Entries [at, at + n) of the records of instances [first, first + count): from
// values[count][n], or the same n values for every instance (one small vector in
// the kernel argument).
struct CfgVector { real_t v[NZ]; };
__global__ void k_cfg_set(real_t* cfg, int first, int count, int at, int n, const real_t* values,
                          CfgVector same) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count * n) return;
    const int b = t / n, i = t - b * n;
    real_t x = same.v[0];
#pragma unroll
    for (int j = 1; j < NZ; j++) x = (i == j) ? same.v[j] : x;
    if (values) x = values[t];
    cfg[(size_t)(first + b) * kCfgStride + at + i] = x;
}

// nmpc_init: the stage Hessians as the solver sees them and the bounds of this
// moment (ccs-c66x/cnmpc.c:899-931).  A weight block within 1e-15 of the identity
// carries no data in the reference's solver and behaves as exact ones
// (qpDUNES/utils.c:106-117, setup_qp.c:229-258); state and control weights are
// separate blocks there, as are the terminal weights.
__global__ void k_cfg_latch(real_t* cfg, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    real_t* r = cfg + (size_t)b * kCfgStride;
    auto identity = [&](int at, int n) {
        bool id = true;
        for (int i = 0; i < n; i++) id = id && !(fabs((double)r[at + i] - 1.0) > 1e-15);
        return id;
    };
    const bool idq = identity(kCfgWq, NX), idr = identity(kCfgWr, NU), idt = identity(kCfgWt, NX);
    for (int i = 0; i < NX; i++) r[kCfgH + i] = idq ? (real_t)1. : r[kCfgWq + i];
    for (int i = 0; i < NU; i++) r[kCfgH + NX + i] = idr ? (real_t)1. : r[kCfgWr + i];
    for (int i = 0; i < NX; i++) r[kCfgHt + i] = idt ? (real_t)1. : r[kCfgWt + i];
    for (int i = 0; i < NU; i++) { r[kCfgLb0 + i] = r[kCfgLb + i]; r[kCfgUb0 + i] = r[kCfgUb + i]; }
}

// ========================================================================
// launchers
// ========================================================================
static inline unsigned blocks_for(long n, int bs) { return (unsigned)((n + bs - 1) / bs); }

void launch_store_refs(const BatchView& v, const real_t* refs, int first, int count, hipStream_t s) {
    const long n = (long)v.B * count * 16;
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
    if (v.cfg) hipLaunchKernelGGL(k_init_stages<true>, dim3(blocks_for(n, 256)), dim3(256), 0, s, v, part);
    else hipLaunchKernelGGL(k_init_stages<false>, dim3(blocks_for(n, 256)), dim3(256), 0, s, v, part);
}
void launch_linearise(const BatchView& v, int first, int count, hipStream_t s, bool about_solution,
                      bool only_solved) {
    const unsigned blocks = blocks_for((long)v.B * count, kStagesPerBlock);
    const int only = about_solution || only_solved ? 1 : 0;
    if (about_solution && v.cfg)
        hipLaunchKernelGGL((k_linearise<true, true>), dim3(blocks), dim3(256), 0, s, v, first, count, only);
    else if (about_solution)
        hipLaunchKernelGGL((k_linearise<true, false>), dim3(blocks), dim3(256), 0, s, v, first, count, only);
    else if (v.cfg)
        hipLaunchKernelGGL((k_linearise<false, true>), dim3(blocks), dim3(256), 0, s, v, first, count, only);
    else
        hipLaunchKernelGGL((k_linearise<false, false>), dim3(blocks), dim3(256), 0, s, v, first, count, only);
}
void launch_cfg_set(real_t* cfg, int first, int count, int at, int n, const real_t* values,
                    const real_t* same, hipStream_t s) {
    CfgVector sv{};
    if (same) for (int i = 0; i < n; i++) sv.v[i] = same[i];
    hipLaunchKernelGGL(k_cfg_set, dim3(blocks_for((long)count * n, 256)), dim3(256), 0, s, cfg, first, count,
                       at, n, values, sv);
}
void launch_cfg_latch(real_t* cfg, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_cfg_latch, dim3(blocks_for(B, 256)), dim3(256), 0, s, cfg, B);
}
void launch_rti_reset(const BatchView& v, hipStream_t s) {
    const long n = (long)v.B * (v.N * NX + v.N + 1);
    hipLaunchKernelGGL(k_rti_reset, dim3(blocks_for(n, 256)), dim3(256), 0, s, v);
}
void launch_linearise_free(long n, const real_t* x, const real_t* u, const real_t* xn,
                           const real_t* w, const Bounds& bnd, real_t* jac, real_t* res,
                           hipStream_t s) {
    const unsigned blocks = blocks_for(n, kStagesPerBlock);
    if ((((uintptr_t)jac | (uintptr_t)res) & 15) == 0)
        hipLaunchKernelGGL(k_linearise_free<true>, dim3(blocks), dim3(256), 0, s, n, x, u, xn, w, bnd, jac, res);
    else
        hipLaunchKernelGGL(k_linearise_free<false>, dim3(blocks), dim3(256), 0, s, n, x, u, xn, w, bnd, jac, res);
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
