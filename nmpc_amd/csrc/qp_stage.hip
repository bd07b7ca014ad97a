// Stage-parallel half of the dual Newton solve: one workgroup per controller
// instance, of 256 threads, or of 1 024 in the form for long horizons.
//
//   k_qp_begin     initial-value embedding and the opening step
//                  (ccs-c66x/cnmpc.c:675-702, qpDUNES/dual_qp.c:56-100),
//                  then the Newton set-up of iteration 1
//   k_newton_step  stage directions for the multiplier step, step-length
//                  selection (full step / backtracking / gradient bisection),
//                  commit, active-set bookkeeping
//                  (dual_qp.c:309-379, 1119-1470, 1542-1699;
//                   stage_qp_solver_clipping.c:42-198),
//                  then the Newton set-up of the next iteration
//   Newton set-up = dual gradient, convergence test, Newton Hessian block rows
//                  (dual_qp.c:423-597).  It is fused behind the kernels above so
//                  that instances that are done stepping go on while others are
//                  still in their line search.
//   k_line_search_fast / _tail / _exact / _commit / _hessian
//                  step-length searches that go on after their first bisection
//                  trial (dual_qp.c:1341-1470), handed over by k_newton_step: a
//                  first attempt on a regrouped slope with a rounding bound, the
//                  row-ordered continuation of what that leaves open, then the
//                  commit and the Newton set-up of those instances
//
// Element-wise stage work (step, clip, gaps, linear term) runs over the flat
// slot-ordered index e = 15 slot + i, so global accesses are contiguous across
// lanes; the clipped primal and linear term of 64 stages at a time are kept in
// LDS for the per-stage objective values.  Every sum whose order the reference
// fixes (objective over stages, norms and slopes over multiplier rows) is
// produced in parallel and accumulated in index order by one wavefront.
// All control flow depends only on per-instance values, so it is uniform over
// the workgroup.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <utility>

#include "kernels.hpp"
#include "qp_common.hpp"

#pragma clang fp contract(off)

// This file is compiled four times: with the weights and bounds of the batch or
// of each instance (qp_common.hpp, kOwnCfg), and with workgroups of 256 or of
// 1 024 threads (NMPC_STAGE_THREADS).  The four sets of kernels and launchers live
// in four inline namespaces, and the launchers the library calls (kernels.hpp)
// pick between them at the end of the first compilation.
#ifndef NMPC_STAGE_THREADS
#define NMPC_STAGE_THREADS 256
#endif
#define NMPC_WIDE (NMPC_STAGE_THREADS != 256)
#if NMPC_OWN_CFG && NMPC_WIDE
#define NMPC_CFG_NS perinstance_wide
#elif NMPC_OWN_CFG
#define NMPC_CFG_NS perinstance
#elif NMPC_WIDE
#define NMPC_CFG_NS batchwide_wide
#else
#define NMPC_CFG_NS batchwide
#endif

namespace nmpc {
inline namespace NMPC_CFG_NS {

namespace {

// The wide form is for launches of few instances with long horizons: at 500 stages
// an instance's passes are 30 elements per thread of a 256-thread workgroup, and a
// few hundred such workgroups leave most of the chip idle.
constexpr int kStageThreads = NMPC_STAGE_THREADS;
static_assert(kStageThreads == 256 || kStageThreads == 1024, "workgroups of 4 or 16 wavefronts");
constexpr int kStageWaves = kStageThreads / kWave;
// k_qp_begin waits on memory most of the time: six resident wavefronts per
// SIMD (80 registers per lane, eight of them spilled) measured 7 % faster than
// seven and 9 % faster than five (96 registers, no spills).  k_newton_step needs about
// 120 registers; capping it lower spills and is slower.
#ifndef NMPC_BEGIN_WAVES
#define NMPC_BEGIN_WAVES (NMPC_WIDE ? 4 : 6)
#endif
#define NMPC_BEGIN_OCC __attribute__((amdgpu_waves_per_eu(NMPC_BEGIN_WAVES, NMPC_BEGIN_WAVES)))
constexpr int kStageChunk = kStageThreads / 4;      // stage slots whose z, q sit in LDS at once
constexpr int kBufDoubles = 2 * kStageChunk * NZ;   // 1920 (7680): z + q, reused as sum buffer
constexpr int kRedDoubles = 2 + 2 * kStageWaves;    // a result, then two values per wavefront

// dynamic LDS: buf[kBufDoubles] | vals[N+1] | red[kRedDoubles] | chg[N+1] (int)
struct StageLds {
    real_t* z;      // [64*15]
    real_t* q;      // [64*15]
    real_t* buf;    // = z, 1920 doubles for ordered sums
    real_t* vals;   // [N+1] stage objective values by slot
    real_t* red;    // [kRedDoubles]
    int* chg;       // [N+1] active set of slot changed
    real_t* terms;  // [12 N] slope terms of the fused sweep (k_newton_step), or null
    real_t* sqr;    // [12 N] squared gradient rows of the same sweep
    real_t* prevz;  // [15] clipped point of the stage before the current chunk
};

NMPC_DEV StageLds carve(real_t* base, int N) {
    StageLds L;
    L.z = base;
    L.q = base + kStageChunk * NZ;
    L.buf = base;
    L.vals = base + kBufDoubles;
    L.red = L.vals + (N + 1);
    L.chg = (int*)(L.red + kRedDoubles);
    L.terms = nullptr;
    L.sqr = nullptr;
    L.prevz = nullptr;
    return L;
}

// k_newton_step: room for the slope terms its first sweep produces on the way,
// when the gradient fits one accumulation buffer.
NMPC_DEV bool sweep_has_slope(int N) { return NX * N <= 1920; }
NMPC_DEV StageLds carve_step(real_t* base, int N) {
    StageLds L = carve(base, N);
    if (sweep_has_slope(N)) {
        L.prevz = (real_t*)L.chg + ((size_t)(N + 1) * sizeof(int) + 7) / 8 * (8 / sizeof(real_t));
        L.terms = L.prevz + 16;
        L.sqr = L.terms + NX * N;
    }
    return L;
}

inline size_t stage_lds_bytes(int N) {
    return (size_t)(kBufDoubles + (N + 1) + kRedDoubles) * sizeof(real_t) + (size_t)(N + 1) * sizeof(int);
}
inline size_t step_lds_bytes(int N) {
    const size_t base = (stage_lds_bytes(N) + 7) / 8 * 8;
    return base + (NX * N <= 1920 ? (size_t)(16 + 2 * NX * N) * sizeof(real_t) : 0);
}

// acc + buf[0] + buf[1] + ... + buf[len-1], added strictly in that order, by
// the sixteen lanes of a DPP row (all of them active, with the same acc, buf and
// len).  A lane that reads the values itself pays an LDS instruction for every
// two adds, and a wavefront on its own issues one instruction in about seven
// cycles (7.2 ns per value measured, tools/micro_acc.hip).  Here each lane reads
// four consecutive values of a block of 64,
// and every lane then takes the 64 values in order through row broadcasts folded
// into the add (v_fmac_f64 has a VOP2 form, which accepts a DPP row broadcast on
// its first source, and acc + x * 1.0 rounds exactly as acc + x does).  One
// instruction per value and no LDS access in the dependent chain: 5.3 ns per
// value on MI355X, 6.4 microseconds for 1200; the rows of a wavefront run
// different sums side by side.  Every lane of the row returns the sum.
template <int SRC>
NMPC_DEV void add_row_value(double& acc, double x, double one) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc)
                 : "v"(x), "v"(one), "n"(SRC));
}
template <int SRC>
NMPC_DEV void add_row_value(float& acc, float x, float one) {
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc)
                 : "v"(x), "v"(one), "n"(SRC));
}
template <int... I>
NMPC_DEV void add_row_block(real_t& acc, const real_t (&c)[4], real_t one,
                            std::integer_sequence<int, I...>) {
    (add_row_value<I / 4>(acc, c[I % 4], one), ...);
}
template <int... I>
NMPC_DEV void add_row_block_partial(real_t& acc, const real_t (&c)[4], real_t one, int count,
                                    std::integer_sequence<int, I...>) {
    ((I < count ? add_row_value<I / 4>(acc, c[I % 4], one) : (void)0), ...);
}
NMPC_DEV real_t accumulate_by_row(real_t acc, const real_t* buf, int len) {
    using Seq64 = std::make_integer_sequence<int, 64>;
    const int l = threadIdx.x & 15;
    real_t one = 1.0;
    asm volatile("" : "+v"(one));               // keeps it in a vector register pair
    real_t c[4], nx[4];
    auto fetch = [&](real_t (&d)[4], int t) {
#pragma unroll
        for (int i = 0; i < 4; i++) d[i] = (t + 4 * l + i < len) ? buf[t + 4 * l + i] : 0.;
    };
    fetch(c, 0);
    int t = 0;
    for (; t + 64 <= len; t += 64) {
        fetch(nx, t + 64);
        add_row_block(acc, c, one, Seq64{});
#pragma unroll
        for (int i = 0; i < 4; i++) c[i] = nx[i];
    }
    if (t < len) add_row_block_partial(acc, c, one, len - t, Seq64{});
    return acc;
}

// Sum term(0) + ... + term(n-1) in index order starting from 0.  Terms are
// produced by all threads; wavefront 0 accumulates.  Uniform result.
template <class F>
NMPC_DEV real_t block_ordered_sum(int n, const StageLds& L, F term) {
    const int tid = threadIdx.x;
    real_t acc = 0.;
    for (int base = 0; base < n; base += kBufDoubles) {
        const int len = (n - base < kBufDoubles) ? n - base : kBufDoubles;
        int r = tid;
        for (; r + kStageThreads < len; r += 2 * kStageThreads) {   // two terms in flight
            const real_t t0 = term(base + r);
            const real_t t1 = term(base + r + kStageThreads);
            L.buf[r] = t0;
            L.buf[r + kStageThreads] = t1;
        }
        if (r < len) L.buf[r] = term(base + r);
        __syncthreads();
        if (tid < kWave) acc = accumulate_by_row(acc, L.buf, len);
        __syncthreads();
    }
    if (tid == 0) L.red[0] = acc;
    __syncthreads();
    const real_t out = L.red[0];
    __syncthreads();
    return out;
}

template <int... I>
NMPC_DEV void wave_sums(real_t (&x)[sizeof...(I)], std::integer_sequence<int, I...>) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ((x[I] += __shfl_xor(x[I], o)), ...);
}

// Sums over the workgroup of each thread's x[0] and x[1], in no particular
// order.  Uniform result.
NMPC_DEV void block_sums(real_t (&x)[2], const StageLds& L) {
    wave_sums(x, std::make_integer_sequence<int, 2>{});
    const int tid = threadIdx.x;
    if ((tid & (kWave - 1)) == 0) {
        L.red[2 * (tid / kWave)] = x[0];
        L.red[2 * (tid / kWave) + 1] = x[1];
    }
    __syncthreads();
    real_t a = 0., b = 0.;
#pragma unroll
    for (int w = 0; w < kStageWaves; w++) { a += L.red[2 * w]; b += L.red[2 * w + 1]; }
    __syncthreads();
    x[0] = a;
    x[1] = b;
}

NMPC_DEV real_t block_min(real_t x, const StageLds& L) {
    x = wave_min(x);
    const int tid = threadIdx.x;
    if ((tid & 63) == 0) L.red[1 + (tid >> 6)] = x;
    __syncthreads();
    real_t m = L.red[1];
#pragma unroll
    for (int w = 1; w < kStageWaves; w++) m = (L.red[1 + w] < m) ? L.red[1 + w] : m;
    __syncthreads();
    return m;
}

NMPC_DEV int block_max(int x, const StageLds& L) {
    x = wave_max(x);
    const int tid = threadIdx.x;
    int* r = (int*)(L.red + 1 + kStageWaves);
    if ((tid & 63) == 0) r[tid >> 6] = x;
    __syncthreads();
    int m = r[0];
#pragma unroll
    for (int w = 1; w < kStageWaves; w++) m = (r[w] > m) ? r[w] : m;
    __syncthreads();
    return m;
}

enum PassMode { kBegin, kCommit, kTrialValue, kTrialPoint };

// One pass over every stage at step length a.
//   kBegin / kCommit  apply the step (stage_qp_solver_clipping.c:134-165), refresh the
//                     active-set record (dual_qp.c:1627-1699); returns the objective
//                     (dual_qp.c:1542-1566)
//   kTrialValue       objective at a trial step; only z and y are overwritten
//                     (dual_qp.c:1574-1617)
//   kTrialPoint       clipped trial point into tmp, gaps into y (dual_qp.c:1370-1376);
//                     returns 0
template <PassMode MODE>
NMPC_DEV real_t stage_pass(const BatchView& v, const Inst& I, real_t a, const StageLds& L) {
    const int tid = threadIdx.x, N = v.N;
    const int eEnd = NZ * N + NX;            // terminal slot holds 12 variables
    for (int c0 = 0; c0 <= N; c0 += kStageChunk) {
        const int c1 = (c0 + kStageChunk <= N + 1) ? c0 + kStageChunk : N + 1;
        const int e0 = c0 * NZ;
        const int e1 = (c1 * NZ < eEnd) ? c1 * NZ : eEnd;
        for (int e = e0 + tid; e < e1; e += kStageThreads) {
            const real_t zu = I.zUnc[e] + a * I.dz[e];
            if (MODE == kBegin || MODE == kCommit) I.zUnc[e] = zu;
            real_t y0, y1;
            const real_t zc = clip_gap(zu, I.zLow[e], I.zUpp[e], &y0, &y1);
            store_pair(I.y + 2 * e, y0, y1);
            if (MODE == kTrialPoint) {
                I.tmp[e] = zc;
            } else {
                I.z[e] = zc;
                const real_t qv = I.q[e] + a * I.qStep[e];
                if (MODE != kTrialValue) I.q[e] = qv;
                L.z[e - e0] = zc;
                L.q[e - e0] = qv;
            }
            if (MODE == kBegin || MODE == kCommit) {
                const int s = e / NZ, i = e - s * NZ;
                const int idx = slot_position(v, s) * NZ + i;
                const int8_t code = (y0 > kEqTol) ? -1 : ((y1 > kEqTol) ? 1 : 0);
                if (MODE == kBegin) {
                    I.prev[idx] = -42;
                } else {
                    const int8_t old = I.ieq[idx];
                    I.prev[idx] = old;
                    if (code != old) atomicOr(&L.chg[s], 1);
                }
                I.ieq[idx] = code;
            }
        }
        __syncthreads();
        if (MODE != kTrialPoint) {
            for (int s = c0 + tid; s < c1; s += kStageThreads) {
                const int nV = s < N ? NZ : NX;
                const real_t pt = I.p[s] + a * I.pStep[s];
                if (MODE != kTrialValue) I.p[s] = pt;
                if constexpr (kOwnCfg)
                    L.vals[s] = stage_value(I.cfg + (s < N ? kCfgH : kCfgHt), L.z + (s - c0) * NZ, L.q + (s - c0) * NZ, pt, nV);
                else
                    L.vals[s] = stage_value(s < N ? v.H : v.Ht, L.z + (s - c0) * NZ, L.q + (s - c0) * NZ, pt, nV);
            }
        }
        __syncthreads();
    }
    if (MODE == kTrialPoint) return 0.;
    // objective: stage values summed in horizon order
    if (tid < kWave) {
        // vals is slot-ordered: positions 0..N-1 are slots head..N-1, 0..head-1, then slot N
        real_t acc = accumulate_by_row(0., L.vals + v.head, N - v.head);
        acc = accumulate_by_row(acc, L.vals, v.head);
        acc += L.vals[N];
        if (tid == 0) L.red[0] = acc;
    }
    __syncthreads();
    const real_t obj = L.red[0];
    __syncthreads();
    return obj;
}

// Publish actSetHasChanged per horizon position and return the last changed
// position (-1 if none).
template <bool ALL>
NMPC_DEV int publish_changes(const BatchView& v, const Inst& I, const StageLds& L) {
    int last = -1;
    for (int k = threadIdx.x; k <= v.N; k += kStageThreads) {
        const int c = ALL ? 1 : L.chg[stage_slot(v, k)];
        I.chg[k] = (int8_t)c;
        if (c) last = k;
    }
    return block_max(last, L);
}

// ------------------------------------------------------------------------
// Dual gradient and its norm (convergence test), then the Newton Hessian block
// rows of every stage whose active set changed (dual_qp.c:423-597).  Each of
// the four wavefronts assembles every fourth block row: it stages C_k and
// C_k P_k in its own slice of LDS, 42 lanes then own two entries each of the
// lower triangle of the 12x12 diagonal block and add the free columns' dyads in
// column order.
// gradKept: grad already holds the gradient at z (a slope trial at the committed
// step stored it).
// Returns false when the gradient norm passes the convergence test.
NMPC_DEV bool setup_gradient(const BatchView& v, const Inst& I, const StageLds& L, int it,
                             bool gradKept, const real_t* normKept = nullptr) {
    const int tid = threadIdx.x, N = v.N, n = NX * N;
    bool converged;
    if (gradKept && normKept) {
        converged = sqrt(*normKept) < kStatTol;
    } else {
        // The test asks one thing of the sum of the squared rows: is its root below the
        // tolerance.  The terms are not negative, so a sum in any order is within
        // (n - 1) u / (1 - (n - 1) u) of the exact one, relatively, as the row-ordered
        // sum is; unless the unordered sum sits within twice that (taken generously,
        // with room for the roundings of the threshold's square and of the root) of the
        // squared tolerance it answers for the row-ordered sum, whose 5.3 ns per term
        // (32 microseconds at 500 stages) are then not spent.
        real_t sq[2] = {0., 0.};
        int r = tid;
        for (; r + kStageThreads < n; r += 2 * kStageThreads) {     // two rows in flight
            const real_t g0 = gradKept ? I.grad[r] : gradient_row(v, I, I.z, r);
            const real_t g1 = gradKept ? I.grad[r + kStageThreads] : gradient_row(v, I, I.z, r + kStageThreads);
            if (!gradKept) { I.grad[r] = g0; I.grad[r + kStageThreads] = g1; }
            sq[0] += g0 * g0;
            sq[0] += g1 * g1;
        }
        if (r < n) {
            const real_t g = gradKept ? I.grad[r] : gradient_row(v, I, I.z, r);
            if (!gradKept) I.grad[r] = g;
            sq[0] += g * g;
        }
        block_sums(sq, L);
        const real_t tol2 = kStatTol * kStatTol;
        const real_t rel = (real_t)(4.0 * (double)n * kUnitRoundoff);
        const real_t edge = (real_t)(64.0 * kUnitRoundoff);
        bool below = sq[0] * (1. + rel) < tol2 * (1. - edge);
        bool above = sq[0] * (1. - rel) > tol2 * (1. + edge);       // both false for NaN
#ifdef NMPC_TEST_ALWAYS_REPEAT_BLOCKS         // test builds: the row-ordered sum on every odd iteration
        if (it & 1) below = above = false;
#endif
        if (below || above) {
            converged = below;
        } else {
            // (every thread reads back the rows it wrote itself)
            const real_t nn = block_ordered_sum(n, L, [&](int r) {
                const real_t g = I.grad[r];
                return g * g;
            });
            converged = sqrt(nn) < kStatTol;
        }
    }
    if (tid == 0) { I.sc->numIter = it; I.sc->regularized = 0; }
    if (converged) {
        if (tid == 0) I.sc->status = ST_OPTIMAL;
        return false;
    }
    return true;
}

// Newton Hessian block rows of the stages whose active set changed, every
// WAVES-th block row per wavefront; slices: kSliceDoubles of LDS per wavefront:
// C_k P_k, then C_k, whose place (and 108 entries more) the finished block row
// takes before it leaves.
constexpr int kSliceDoubles = NX * NZ + 2 * NX * NX;
constexpr int kDiagPairs = 42;      // pairs (i, j), (i, j + 1), j even, j <= i < 12
static_assert((kStageThreads / kWave) * kSliceDoubles <= kBufDoubles, "slices live in buf");
// WITH_GRADIENT: every block row is assembled (the caller has marked every
// stage changed), and the twelve rows of the dual gradient that belong to a
// block row (gradient_row's expression, term for term) are made from the
// Jacobian while it is in LDS, which saves reading it a second time.
// A workgroup that shares an instance with others takes the block rows first + w,
// first + w + stride, ... (stride: the wavefronts of all of them).
template <int WAVES, bool WITH_GRADIENT = false>
NMPC_DEV void assemble_hessian(const BatchView& v, const Inst& I, real_t* slices, int first = 0,
                               int stride = WAVES) {
    const int tid = threadIdx.x, N = v.N;
    const int w = tid >> 6, lane = tid & 63;
    real_t* Zs = slices + w * kSliceDoubles;
    real_t* Cs = Zs + NX * NZ;
    // The Jacobian entries and gaps of the next block row to assemble are fetched
    // while the current one is worked on.
    struct Fetched { real_t c[3], y[4]; };
    auto needed = [&](int k) {
        return k < N && (WITH_GRADIENT || I.chg[k] != 0 || I.chg[k + 1] != 0);
    };
    auto fetch = [&](int k, Fetched& f) {
        const int s = stage_slot(v, k), sn = stage_slot(v, k + 1);
        const real_t* Cg = I.C + (size_t)s * (NX * NZ);
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int e = lane + t * kWave;
            // (read once here, and too far from its next use to be kept on chip)
            f.c[t] = (e < NX * NZ) ? __builtin_nontemporal_load(&Cg[e]) : 0.;
        }
        const int ly = lane < NZ ? lane : 0, lx = lane < NX ? lane : 0;
        const pair_a yk = *(const pair_a*)(I.y + 2 * s * NZ + 2 * ly);
        const pair_a yn = *(const pair_a*)(I.y + 2 * sn * NZ + 2 * lx);
        f.y[0] = yk.x; f.y[1] = yk.y;
        f.y[2] = yn.x; f.y[3] = yn.y;
    };
    // lane p < kDiagPairs: entries (di, dj) and (di, dj + 1) of the diagonal block,
    // dj even, the rows in order; dAt: where the pair sits in the block row
    int di = 0, dj = lane;
#pragma unroll
    for (int r = 0; r < NX - 1; r++) {
        const int pairs = di / 2 + 1;           // of row di
        if (dj >= pairs) { dj -= pairs; di++; }
    }
    dj *= 2;
    if (lane >= kDiagPairs) { di = 0; dj = 0; }
    const int dAt = chol_diag_row(di) + dj;
    int k = first + w;
    while (k < N && !needed(k)) k += stride;
    Fetched cur{}, nxt{};
    if (k < N) fetch(k, cur);
    while (k < N) {
        int kn = k + stride;
        while (kn < N && !needed(kn)) kn += stride;
        if (kn < N) fetch(kn, nxt);
        const bool ck = I.chg[k] != 0;
        // free columns of stage k; fixed states of stage k+1
        const bool fr = lane < NZ && cur.y[0] <= kEqTol && cur.y[1] <= kEqTol;
        const bool fx = lane < NX && (cur.y[2] >= kEqTol || cur.y[3] >= kEqTol);
        const unsigned long long freeMask = __ballot(fr);
        const unsigned long long fixMask = __ballot(fx);
        // C_k and C_k P_k, both laid out as C_k is: consecutive lanes write consecutive words
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int e = lane + t * kWave;
            if (e < NX * NZ) {
                const int l = e % NZ;
                Cs[e] = cur.c[t];
                Zs[e] = cur.c[t] / hessian_entry(v, I, 0, l);
            }
        }
        real_t* zs = Cs + NX * NZ;                      // free until the finished row is put together
        if (WITH_GRADIENT && lane < NZ) zs[lane] = I.z[stage_slot(v, k) * NZ + lane];
        // this wavefront's LDS writes land before its reads below
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (WITH_GRADIENT && lane < NX) {
            const real_t ck = I.c[stage_slot(v, k) * NX + lane];
            const real_t zn = I.z[stage_slot(v, k + 1) * NZ + lane];
            real_t a = 0.;
#pragma unroll 5
            for (int j = 0; j < NZ; j++) a += Cs[lane * NZ + j] * zs[j];
            a += ck;
            a -= zn;
            I.grad[k * NX + lane] = a;
        }
        // diagonal block: E P_{k+1} E' + C_k P_k C_k', the entries (i, j <= i) the
        // factorisation reads (dual_qp.c:835, 871), two of a row per lane
        const bool wholeRow = k >= 1 && ck;             // the sub-diagonal block is written too
        real_t m[2] = {0., 0.};
        if (lane < kDiagPairs) {
#pragma unroll
            for (int t = 0; t < 2; t++) {
                if (di == dj + t) {
                    m[t] = 1. / hessian_entry(v, I, k + 1, di);
                    if ((fixMask >> di) & 1ull) m[t] = 0.;
                }
            }
            for (int l = 0; l < NZ; l++) {
                if (!((freeMask >> l) & 1ull)) continue;
                const real_t ci = Cs[di * NZ + l];
#pragma unroll
                for (int t = 0; t < 2; t++) m[t] += ci * Zs[(dj + t) * NZ + l];
            }
        }
        // (entry dj + 1 of a row of odd length is the padding behind it)
        if (!wholeRow) {
            if (lane < kDiagPairs) store_pair(I.hess + (size_t)k * kCholRow + dAt, m[0], m[1]);
        } else {
            // The block row leaves as whole 16-byte pieces of consecutive addresses:
            // it is put together where C_k was (a wavefront's LDS accesses are
            // served in program order, so the reads above come first).
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            real_t* row = Cs;
            if (lane < kDiagPairs) store_pair(row + dAt, m[0], m[1]);
            // sub-diagonal block: -C_k[:, :12] P_k, the quotients staged above
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const int e = lane + t * kWave;
                const int si = e / NX, sj = e % NX;
                if (e < NX * NX)
                    row[e] = ((freeMask >> sj) & 1ull) ? -Zs[si * NZ + sj] : 0.;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            pair_a* out = (pair_a*)(I.hess + (size_t)k * kCholRow);
            const pair_a* in = (const pair_a*)row;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int e = lane + t * kWave;
                // (0.75 GB per batch of 4096 that the Newton solve reads once: streamed past
                // the caches, which the stage vectors need more)
                if (e < kCholRow / 2) __builtin_nontemporal_store(in[e], &out[e]);
            }
        }
        // reads done before the next block row overwrites the slice
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        k = kn;
        cur = nxt;
    }
}

NMPC_DEV void newton_setup(const BatchView& v, const Inst& I, const StageLds& L, int it,
                           bool gradKept = false, const real_t* normKept = nullptr) {
    if (setup_gradient(v, I, L, it, gradKept, normKept))
        assemble_hessian<kStageThreads / kWave>(v, I, L.buf);
}

// ------------------------------------------------------------------------
// The gradient-bisection step-length search (dual_qp.c:1341-1470) as a state
// machine: trial_alpha() names the next trial step, advance() consumes its
// slope.  Phase 0 grows the bracket from its upper end, phase 1 bisects it.
struct Bisection {
    real_t aMin, aMax, aMid;
    real_t alpha;       // result, valid when done
    int it;             // line-search trial counter (shared with the backtracking phase)
    int phase;
    bool done;
};

NMPC_DEV void settle(Bisection& s) {
    if (s.done) return;
    if (s.it >= kMaxLsRefine) {         // trial budget spent: keep the last midpoint
        s.alpha = s.aMid;
        s.done = true;
    }
}

NMPC_DEV real_t trial_alpha(const Bisection& s) {
    return s.phase == 0 ? s.aMax : 0.5 * (s.aMin + s.aMax);
}

NMPC_DEV void advance(Bisection& s, real_t slope, real_t normal) {
    if (s.phase == 0) {
        if (fabs(slope / normal) <= kLsStatTol) { s.alpha = s.aMax; s.done = true; return; }
        if (slope / normal < 0.) { s.phase = 1; settle(s); return; }
        s.aMin = s.aMax;
        s.aMax *= kLsIncrease;
        if (s.aMax > kLsMaxStep) { s.alpha = s.aMin; s.done = true; return; }
        ++s.it;
        if (s.it >= kMaxLsRefine) s.phase = 1;
    } else {
        s.aMid = 0.5 * (s.aMin + s.aMax);
        if (fabs(slope / normal) <= kLsStatTol) { s.alpha = s.aMid; s.done = true; return; }
        if (slope > 0) s.aMin = s.aMid; else s.aMax = s.aMid;
        ++s.it;
    }
    settle(s);
}

// advance() for a trial whose two decisions are already known: `flat` =
// |slope / normal| <= tolerance (the search ends there), `negative` = the slope
// is below zero.  The slope's value itself never enters the search state.
NMPC_DEV void advance_decided(Bisection& s, bool flat, bool negative) {
    if (s.phase == 0) {
        if (flat) { s.alpha = s.aMax; s.done = true; return; }
        if (negative) { s.phase = 1; settle(s); return; }
        s.aMin = s.aMax;
        s.aMax *= kLsIncrease;
        if (s.aMax > kLsMaxStep) { s.alpha = s.aMin; s.done = true; return; }
        ++s.it;
        if (s.it >= kMaxLsRefine) s.phase = 1;
    } else {
        s.aMid = 0.5 * (s.aMin + s.aMax);
        if (flat) { s.alpha = s.aMid; s.done = true; return; }
        if (negative) s.aMax = s.aMid; else s.aMin = s.aMid;
        ++s.it;
    }
    settle(s);
}

// A slope is the sum of n terms added in row order, and the search only asks
// two things of it: is |slope / normal| within the tolerance, and is it
// negative.  Any order of summation lands within (n - 1) u sum|t| / (1 - (n - 1) u)
// of the true sum, so a tree sum s and the magnitude sum m settle both
// questions for the row-order sum too whenever |s| is further from the
// threshold than twice that bound (taken generously, with room for the rounding
// of m, of the threshold and of the quotient: 64 units roundoff of the threshold).  Otherwise the caller adds the
// terms in order.
NMPC_DEV bool slope_decided(real_t s, real_t m, int n, real_t normal, bool* flat, bool* negative) {
    const real_t thr = kLsStatTol * normal;
    const real_t slack = (real_t)(4.0 * (double)n * kUnitRoundoff * (double)m + 64.0 * kUnitRoundoff * (double)thr);
    *flat = fabs(s) < thr;
    *negative = s < 0.;
    return fabs(fabs(s) - thr) > slack;         // false for NaN operands as well
}

// The state after a trial that did not end the search and whose slope was
// negative / positive: advance() without the tests a real slope needs.
NMPC_DEV void advance_assuming(Bisection& s, bool negative) {
    if (s.phase == 0) {
        if (negative) { s.phase = 1; settle(s); return; }
        s.aMin = s.aMax;
        s.aMax *= kLsIncrease;
        if (s.aMax > kLsMaxStep) { s.alpha = s.aMin; s.done = true; return; }
        ++s.it;
        if (s.it >= kMaxLsRefine) s.phase = 1;
    } else {
        s.aMid = 0.5 * (s.aMin + s.aMax);
        if (negative) s.aMax = s.aMid; else s.aMin = s.aMid;
        ++s.it;
    }
    settle(s);
}

// Commit of a step whose clipped point and gaps are already in z and y (the last
// value trial ran at the same step length and nothing has touched them since):
// what is left is the unclipped point, the linear term, the offsets and the
// active-set record.  Every value is produced by the expression the full pass
// uses, so the state is the one stage_pass<kCommit> would leave.
NMPC_DEV void commit_kept_trial(const BatchView& v, const Inst& I, real_t a, const StageLds& L) {
    const int tid = threadIdx.x, N = v.N;
    const int eEnd = NZ * N + NX;
    for (int e = tid; e < eEnd; e += kStageThreads) {
        I.zUnc[e] = I.zUnc[e] + a * I.dz[e];
        I.q[e] = I.q[e] + a * I.qStep[e];
        const pair_a yy = *(const pair_a*)(I.y + 2 * e);
        const real_t y0 = yy.x, y1 = yy.y;
        const int s = e / NZ, i = e - s * NZ;
        const int idx = slot_position(v, s) * NZ + i;
        const int8_t code = (y0 > kEqTol) ? -1 : ((y1 > kEqTol) ? 1 : 0);
        const int8_t old = I.ieq[idx];
        I.prev[idx] = old;
        if (code != old) atomicOr(&L.chg[s], 1);
        I.ieq[idx] = code;
    }
    for (int s = tid; s <= N; s += kStageThreads) I.p[s] = I.p[s] + a * I.pStep[s];
    __syncthreads();
}

// Commit of the chosen step: multipliers, stage variables, objective, active
// sets; then the Newton set-up of iteration nextIt.  keptObjective points to the
// objective of the value trial that z and y still hold, or is null.
NMPC_DEV void commit_step(const BatchView& v, const Inst& I, const StageLds& L, real_t alpha,
                          int lsIter, int nextIt, const real_t* keptObjective = nullptr,
                          bool gradKept = false, bool withHessian = true,
                          const real_t* normKept = nullptr) {
    const int tid = threadIdx.x, n = NX * v.N;
    for (int r = tid; r < n; r += kStageThreads) {
        real_t* lp = I.lam + (size_t)lam_slot(v, r / NX) * NX + r % NX;
        *lp = *lp + alpha * I.dl[r];
    }
    real_t objInc;
    if (keptObjective) {
        commit_kept_trial(v, I, alpha, L);
        objInc = *keptObjective;
    } else {
        objInc = stage_pass<kCommit>(v, I, alpha, L);
    }
    const int last = publish_changes<false>(v, I, L);
    if (tid == 0) {
        I.sc->objInc = objInc;
        I.sc->alpha = alpha;
        I.sc->lsIter = lsIter;
        I.sc->last = last;
    }
    if (nextIt <= kMaxIter) {
        __syncthreads();
        if (withHessian) newton_setup(v, I, L, nextIt, gradKept && keptObjective, normKept);
        else setup_gradient(v, I, L, nextIt, gradKept && keptObjective);
    }
}

// Stage directions for the multiplier step dl (dual_qp.c:309-379): qStep =
// C' dl_k - dl_{k-1}, pStep = c . dl_k, dz = -(qStep / H); the smallest positive
// step to an active-set change (stage_qp_solver_clipping.c:110-126), taken from
// the gaps of the current point; and, in the same sweep, the value trial at the
// full step (dual_qp.c:1574-1617), which every branch of the step-length
// selection starts with or ends at, together with the rows of the dual gradient
// at that trial point (dual_qp.c:574-597): they give the slope of the first
// bisection trial and, if the full step is committed, the gradient of the next
// Newton set-up, and they read each Jacobian block while the direction pass
// has just fetched it.  Stages are visited in horizon order, kStageChunk at a
// time; the rows of a stage are done once its successor's point is known.
// Returns the trial objective; *aminOut the smallest step; with L.terms:
// *slopeOut the slope at the full step, and grad is written when keepGrad.
NMPC_DEV real_t directions_and_full_step_trial(const BatchView& v, const Inst& I, const StageLds& L,
                                               bool keepGrad, real_t* aminOut, real_t* slopeOut,
                                               real_t* normOut) {
    const int tid = threadIdx.x, N = v.N;
    const real_t a = 1.;
    const bool rows = L.terms != nullptr;
    real_t amin = kInfty;
    for (int c0 = 0; c0 <= N; c0 += kStageChunk) {
        const int c1 = (c0 + kStageChunk <= N + 1) ? c0 + kStageChunk : N + 1;
        for (int idx = tid; idx < (c1 - c0) * NZ; idx += kStageThreads) {
            const int kk = idx / NZ, j = idx - kk * NZ, k = c0 + kk;
            if (k == N && j >= NX) continue;
            const int s = stage_slot(v, k), e = s * NZ + j;
            real_t qs, qa = 0.;     // qa: the same sum over magnitudes (k_line_search_fast's error bound)
            if (k < N) {
                const real_t* Cc = I.C + (size_t)s * (NX * NZ) + j;
                const real_t* l1 = I.dl + k * NX;
                qs = 0.;
#pragma unroll
                for (int i = 0; i < NX; i++) {
                    const real_t cv = Cc[i * NZ], lv = l1[i];
                    qs += cv * lv;
                    qa += fabs(cv) * fabs(lv);
                }
            } else {
                qs = 0.0;
            }
            if (k > 0 && j < NX) {
                const real_t lp = I.dl[(k - 1) * NX + j];
                qs -= lp;
                qa += fabs(lp);
            }
            I.qStep[e] = qs;
            if (I.wabs) I.wabs[s * kWabsStride + j] = qa;
            const real_t d = qs / hessian_entry(v, I, k, j);
            const real_t dzv = -d;
            I.dz[e] = dzv;
            const pair_a yy = *(const pair_a*)(I.y + 2 * e);
            const real_t r0 = dzv / yy.x;
            const real_t r1 = dzv / -yy.y;
            const real_t aa = 1. / ((r0 > r1) ? r0 : r1);
            if (aa > 0. && aa < amin) amin = aa;
            // value trial at the full step
            const real_t zu = I.zUnc[e] + a * dzv;
            real_t y0, y1;
            const real_t zc = clip_gap(zu, I.zLow[e], I.zUpp[e], &y0, &y1);
            store_pair(I.y + 2 * e, y0, y1);
            I.z[e] = zc;
            L.z[idx] = zc;
            L.q[idx] = I.q[e] + a * qs;
        }
        __syncthreads();
        for (int k = c0 + tid; k < c1; k += kStageThreads) {
            const int s = stage_slot(v, k);
            real_t ps = 0., pa = 0.;
            if (k < N) {
                const real_t* l1 = I.dl + k * NX;
#pragma unroll
                for (int i = 0; i < NX; i++) {
                    const real_t lv = l1[i], cv = I.c[s * NX + i];
                    ps += lv * cv;
                    pa += fabs(lv) * fabs(cv);
                }
            }
            I.pStep[s] = ps;
            if (I.wabs) I.wabs[s * kWabsStride + NZ] = pa;
            const int nV = k < N ? NZ : NX;
            const real_t pt = I.p[s] + a * ps;
            if constexpr (kOwnCfg)
                L.vals[k] = stage_value(I.cfg + (k < N ? kCfgH : kCfgHt), L.z + (k - c0) * NZ, L.q + (k - c0) * NZ, pt, nV);
            else
                L.vals[k] = stage_value(k < N ? v.H : v.Ht, L.z + (k - c0) * NZ, L.q + (k - c0) * NZ, pt, nV);
        }
        if (rows) {
            // gradient rows of the stages whose successor is in LDS now
            const int kFirst = c0 > 0 ? c0 - 1 : 0;
            for (int rr = tid; rr < (c1 - 1 - kFirst) * NX; rr += kStageThreads) {
                const int kk = rr / NX, i = rr - kk * NX, k = kFirst + kk;
                const int s = stage_slot(v, k);
                const real_t* Crow = I.C + (size_t)s * (NX * NZ) + i * NZ;
                const real_t* zk = (k >= c0) ? L.z + (k - c0) * NZ : L.prevz;
                real_t g = 0.;
#pragma unroll
                for (int j = 0; j < NZ; j++) g += Crow[j] * zk[j];
                g += I.c[s * NX + i];
                g -= L.z[(k + 1 - c0) * NZ + i];
                if (keepGrad) I.grad[k * NX + i] = g;
                L.terms[k * NX + i] = g * I.dl[k * NX + i];
                L.sqr[k * NX + i] = g * g;
            }
        }
        __syncthreads();
        if (rows && c1 <= N) {                  // the last stage of this chunk, for the next one
            if (tid < NZ) L.prevz[tid] = L.z[(c1 - 1 - c0) * NZ + tid];
            __syncthreads();
        }
    }
    // objective (stage values in horizon order), slope and squared gradient norm,
    // side by side
    if (tid < 3 * 16) {             // one DPP row each
        const int w = tid >> 4;
        const real_t* src = (w == 0 || !rows) ? L.vals : (w == 1 ? L.terms : L.sqr);
        const real_t sv = accumulate_by_row(0., src, (w == 0 || !rows) ? N + 1 : NX * N);
        if ((tid & 15) == 0) L.red[w] = sv;
    }
    __syncthreads();
    const real_t obj = L.red[0];
    *slopeOut = L.red[1];
    *normOut = L.red[2];
    __syncthreads();                            // block_min reuses red[1..]
    *aminOut = block_min(amin, L);
    return obj;
}

// Largest horizon whose stage variables fit the per-thread registers of
// k_line_search_tail (kTailElems values of each vector per thread).
constexpr int kTailThreads = 1024;
constexpr int kTailElems = 2;
constexpr int kTailDepth = 3;       // bisection levels settled per round
constexpr int kTailPoints = (1 << kTailDepth) - 1;  // trial steps per round: a binary tree of outcomes
constexpr int kTailMaxHorizon = 106;    // the tree's trial points and slope terms fill the 160 KB of LDS
// Slope-term lists are accumulated side by side, one DPP row each; a stride that is
// 5 mod 32 doubles keeps the rows on different LDS banks.
NMPC_DEV int tail_term_stride(int n) { return n + ((5 - n % 32) + 32) % 32; }
static_assert(kTailPoints * (NZ * (kTailMaxHorizon + 1) + NX * kTailMaxHorizon + 32) * 8 + 256 <= 160 * 1024,
              "trial points and slope terms must fit LDS");
static_assert(NZ * (kTailMaxHorizon + 1) <= kTailElems * kTailThreads, "stage vectors must fit");
static_assert(NX / 2 * kTailMaxHorizon <= kTailThreads, "two gradient rows per thread");
NMPC_DEV bool tail_capable(const BatchView& v) {
    return v.tailList != nullptr && v.wabs != nullptr && v.N <= 508;     // kFastHorizonLarge
}
inline size_t tail_lds_bytes(int N) {
    return (size_t)kTailPoints * (NZ * (N + 1) + NX * N + 32) * sizeof(real_t) + 256;
}
}  // namespace

// ------------------------------------------------------------------------
__global__ __launch_bounds__(kStageThreads) NMPC_BEGIN_OCC void k_qp_begin(BatchView v,
                                                            const real_t* __restrict__ meas) {
    extern __shared__ real_t lds_raw[];
    const StageLds L = carve(lds_raw, v.N);
    const int b = v.b0 + blockIdx.x, tid = threadIdx.x;
    Inst I = instance(v, b);
    if (blockIdx.x == 0 && tid < 32 && v.tailCount) v.tailCount[tid] = 0;     // both sets

    // initial-value embedding: pin the stage-0 state to the measured deviation
    if (tid == 0) {
        real_t x0[NS], m[NS], d[NX];
        const real_t* xr = I.xref + (size_t)xref_slot(v, 0) * NS;
#pragma unroll
        for (int i = 0; i < NS; i++) { x0[i] = xr[i]; m[i] = meas[(size_t)b * NS + i]; }
        state_to_delta(d, x0, m);
        const int s0 = stage_slot(v, 0);
#pragma unroll
        for (int i = 0; i < NX; i++) { I.zLow[s0 * NZ + i] = d[i]; I.zUpp[s0 * NZ + i] = d[i]; }
#pragma unroll
        for (int i = 0; i < NU; i++) {
            const real_t ur = I.uref[s0 * NU + i];
            I.zLow[s0 * NZ + NX + i] = lower_control_bound(v, I, i) - ur;
            I.zUpp[s0 * NZ + NX + i] = upper_control_bound(v, I, i) - ur;
        }
    }
    __syncthreads();

    const real_t obj = stage_pass<kBegin>(v, I, 1, L);
    const int last = publish_changes<true>(v, I, L);
    if (tid == 0) {
        I.sc->objInc = obj;
        I.sc->last = last;
        I.sc->status = ST_RUNNING;
        I.sc->numIter = 0;
        I.sc->regularized = 0;
        I.sc->lsIter = 0;
        I.sc->lsDeferred = 0;
    }
    __syncthreads();
    // The first Newton set-up assembles every block row.  The convergence test
    // (dual_qp.c:423-446) asks whether the root of the sum of the squared
    // gradient rows is below the tolerance; the sum is at least each of its
    // terms, so one row beyond twice the tolerance settles it without the sum,
    // and gradient and Hessian then come out of one pass over the Jacobians.
    // Otherwise: gradient, ordered sum, test, Hessian, as in later iterations.
    const int probe = NX * v.N < 4 * NX ? NX * v.N : 4 * NX;
    bool beyond = false;
    if (tid < probe) beyond = fabs(gradient_row(v, I, I.z, tid)) > 2 * kStatTol;
    if (__syncthreads_or(beyond)) {
        if (tid == 0) I.sc->numIter = 1;
        assemble_hessian<kStageThreads / kWave, true>(v, I, L.buf);
    } else {
        newton_setup(v, I, L, 1);
    }
}

// ------------------------------------------------------------------------
// hessianApart: the Hessian block rows of the next set-up are left to k_step_hessian.
__global__ __launch_bounds__(kStageThreads) void k_newton_step(BatchView v, int nextIt, int hessianApart) {
    extern __shared__ real_t lds_raw[];
    const StageLds L = carve_step(lds_raw, v.N);
    const int b = launch_instance(v, blockIdx.x), tid = threadIdx.x;
    if (b < 0) return;
    Inst I = instance(v, b);
    if (I.sc->status != ST_RUNNING) return;
    const int N = v.N, n = NX * N;
    const bool regularized = I.sc->regularized != 0;
    real_t objInc = I.sc->objInc;

    for (int s = tid; s <= N; s += kStageThreads) L.chg[s] = 0;

    real_t alphaMinAS, fullStepSlope, fullStepNorm2;
    const bool keepGrad = nextIt <= kMaxIter;   // the last iteration leaves the gradient record alone
    const bool haveSlope = L.terms != nullptr;
    // z and y hold the value trial at step length zAlpha, whose objective is objKept
    real_t objKept = directions_and_full_step_trial(v, I, L, keepGrad, &alphaMinAS, &fullStepSlope,
                                                    &fullStepNorm2);
    real_t zAlpha = 1.;
    // step length whose gradient grad holds, if any
    real_t gradAlpha = (haveSlope && keepGrad) ? 1. : __builtin_nan("");
    bool gradFromSweep = haveSlope && keepGrad;     // false once a slope trial rewrites grad

    real_t alpha;
    int lsIter = 0;
    if (alphaMinAS > 1. - kEqTol && !regularized) {
        alpha = 1.;
    } else {
        // backtracking on the dual objective (dual_qp.c:1294-1334)
        real_t alphaMax = 1.;
        // fabs is double-valued in the reference, which makes the whole expression double
        const real_t minProg = (real_t)((double)kLsMinRel * fabs((double)objInc) + (double)kLsMinAbs);
        const real_t nd = sqrt(I.sc->dlNorm2);
        alpha = alphaMax;
        int st = ST_LS_MAXITER;
        for (; lsIter < kMaxLsIter; ++lsIter) {
            if (!(alpha == zAlpha)) {
                objKept = stage_pass<kTrialValue>(v, I, alpha, L);
                zAlpha = alpha;
            }
            if (objKept > objInc + minProg) { st = ST_OK; break; }
            alpha = alpha * kLsReduction;
            if (nd * (alpha - (real_t)0.) < kEqTol) { alpha = 0.; st = ST_LS_MIN_STEP; break; }
        }
        if (st == ST_LS_MIN_STEP) {
            if (haveSlope && keepGrad) {
                // The solve ends here, and the reference's gradient record is still the
                // one of this iteration's set-up, which the sweep above has replaced with
                // the full step's: make it again from the committed point, clip(zUnc)
                // (z and y stay the last value trial's, as in the reference).
                const int eEnd = NZ * N + NX;
                for (int e = tid; e < eEnd; e += kStageThreads) {
                    real_t y0, y1;
                    I.tmp[e] = clip_gap(I.zUnc[e], I.zLow[e], I.zUpp[e], &y0, &y1);
                }
                __syncthreads();
                for (int r = tid; r < n; r += kStageThreads) I.grad[r] = gradient_row(v, I, I.tmp, r);
            }
            if (tid == 0) {
                I.sc->status = ST_NO_ASCENT;
                I.sc->alpha = alpha;
                I.sc->lsIter = lsIter;
            }
            return;
        }
        {
            const real_t cand = alpha / kLsReduction;
            alphaMax = (alphaMax < cand) ? alphaMax : cand;
        }
        // gradient bisection (dual_qp.c:1341-1470).  Nearly every search ends with
        // its first trial; the few that go on are handed to k_line_search_tail,
        // which evaluates them faster than this kernel's resources allow.  A trial
        // at the step length z already holds needs no new trial point.
        const real_t normal = (1. < nd) ? 1. : nd;
        Bisection bs{alphaMinAS, alphaMax, 1.0, 0., lsIter, 0, false};
        // One trial: the slope at step length `at`, and the search state moved on by it.
        // The slope terms summed in any order, with their magnitudes, settle the
        // search's two questions unless the sum sits within rounding of the tolerance
        // (slope_decided); only then are they added in row order.
        auto trial = [&](real_t at) {
            if (haveSlope && at == 1.) {            // the first sweep made it
                advance(bs, fullStepSlope, normal);
                return;
            }
            const real_t* zs = I.z;
            bool keep = false;
            if (at == zAlpha) {
                // the gradient at z is what the next Newton set-up starts from if this
                // step length is committed: keep it (the last iteration has no set-up
                // and leaves the gradient record alone)
                keep = nextIt <= kMaxIter;
                if (keep) { gradAlpha = at; gradFromSweep = false; }
            } else {
                stage_pass<kTrialPoint>(v, I, at, L);
                zAlpha = __builtin_nan("");         // y no longer belongs to z
                zs = I.tmp;
            }
            real_t sm[2] = {0., 0.};
            for (int r = tid; r < n; r += kStageThreads) {
                const real_t g = gradient_row(v, I, zs, r);
                if (keep) I.grad[r] = g;
                const real_t t = g * I.dl[r];
                sm[0] += t;
                sm[1] += fabs(t);
            }
            block_sums(sm, L);
            bool flat, neg;
            bool settled = slope_decided(sm[0], sm[1], n, normal, &flat, &neg);
#ifdef NMPC_TEST_ALWAYS_REPEAT_BLOCKS         // test builds: both ways
            if (nextIt & 1) settled = false;
#endif
            if (settled) {
                advance_decided(bs, flat, neg);
            } else {
                const real_t slope = block_ordered_sum(n, L, [&](int r) {
                    return gradient_row(v, I, zs, r) * I.dl[r];
                });
                advance(bs, slope, normal);
            }
        };
        settle(bs);
        const bool handOver = tail_capable(v);
        int trials = 0;
        while (!bs.done) {
            if (handOver && trials >= 1) {
                if (tid == 0) {
                    Scalars* sc = I.sc;
                    sc->lsMin = bs.aMin; sc->lsMax = bs.aMax; sc->lsMid = bs.aMid;
                    sc->lsNormal = normal;
                    sc->lsIter = bs.it;
                    sc->lsPhase = bs.phase;
                    sc->lsDeferred = 1;
                    const int at = atomicAdd(&v.tailCount[nextIt - 2], 1);
                    v.tailList[at] = b;
                }
                return;
            }
            trial(trial_alpha(bs));
            ++trials;
        }
        alpha = bs.alpha;
        lsIter = bs.it;
    }
    // grad can only be the first sweep's here when it belongs to the full step
    commit_step(v, I, L, alpha, lsIter, nextIt, (alpha == zAlpha) ? &objKept : nullptr,
                alpha == gradAlpha, !hessianApart,
                (haveSlope && gradFromSweep && alpha == 1.) ? &fullStepNorm2 : nullptr);
}

// Newton Hessian block rows after a k_newton_step that left them out, `parts`
// workgroups per instance: in a launch of few instances, or of long horizons, an
// instance's block rows are a latency chain for the four wavefronts of its step,
// while most of the chip has nothing to do.  Instances whose search was handed
// over get theirs from k_line_search_hessian, after their commit.
__global__ __launch_bounds__(kStageThreads) void k_step_hessian(BatchView v, int parts) {
    extern __shared__ real_t lds_raw[];
    const int b = launch_instance(v, blockIdx.x / parts);
    if (b < 0) return;
    Inst I = instance(v, b);
    if (I.sc->status != ST_RUNNING || I.sc->lsDeferred) return;
    constexpr int kWaves = kStageThreads / kWave;
    assemble_hessian<kWaves>(v, I, lds_raw, (blockIdx.x % parts) * kWaves, parts * kWaves);
}

// ------------------------------------------------------------------------
// First attempt at the step-length searches k_newton_step handed over: one
// wavefront per search.
//
// The slope at a trial point z is  sum_k dl_k . (C_k z_k + c_k - z_{k+1}[0:12]),
// which the reference accumulates row by row.  The search only asks whether it
// is within the tolerance of zero and whether it is negative, and regrouped by
// stage variable the same number is  sum_e w_e z_e + sum_k pStep_k  with
//     w_e = (C_k' dl_k)_j - dl_{k-1,j} [j < 12, k >= 1]      for e = (stage k, variable j),
// which is the stage-QP direction qStep the Newton step has already computed
// (dual_qp.c:309-344): no pass over C at all.  Only the control variables and
// the pinned state of stage 0 have finite bounds; as long as every other
// variable stays clear of its bounds over the whole range of trial steps,
// z_e(a) = zUnc_e + a dz_e there and their part of the sum is A + a B with two
// numbers per search.  What is left per trial is a dot product over the 3 N + 12
// bounded variables.
// Rounding makes this differ from the row-ordered sum, but both stay within
//     (rows + variables + 64) u M,   M = sum_e a_e max|z_e| + sum_k |c_k| . |dl_k|,
//     a_e = (|C_k|' |dl_k|)_j + |dl_{k-1,j}|          (wabs, from k_newton_step)
// of the exact sum (u = unit roundoff; at most 19 products and additions per
// row, the rows in any order; likewise per variable, plus the rounding of the
// trial point itself), so whenever the regrouped sum is further than twice that
// from the tolerance threshold it answers both questions as the row-ordered sum
// does.  A search whose next trial is not settled that way, or with an
// unbounded variable near a bound, is left in the state it has reached to
// k_line_search_tail, which adds in the reference's order.
// Each round evaluates the next trial step together with every step that can
// follow it within kTailDepth levels, like k_line_search_tail.
NMPC_DEV bool clips(real_t val, real_t lo, real_t up) {
    return (lo - val >= -kActTol) || (val - up >= -kActTol);
}

// In single precision the regrouped sum settles nothing: the row-ordered sum of
// 24-bit terms is rounding noise itself near the root of the slope (its error
// bound, some 1e-4 of the terms' magnitudes, exceeds the slopes a search ends
// with), so only the same additions in the same order reproduce its sign.  That
// build skips the first attempt, and the row-ordered kernels take the searches
// from the list k_newton_step filled (counters 0.., entries 0..) instead of the
// list of those left open (counters 8.., entries B..).
constexpr bool kFirstAttempt = kDoublePrecision;
constexpr int kOpenList = kFirstAttempt ? 8 : 0;

// kFastSlots bounded variables per lane in registers: 6 for horizons up to 124 (four
// searches per 256-lane workgroup); 0 = in LDS instead, for horizons up to 508 (one
// search per 64-lane workgroup, 60 KB).
constexpr int kFastHorizonSmall = 124, kFastHorizonLarge = 508;
constexpr int kFastBoxed = 3 * kFastHorizonLarge + NX;
static_assert(3 * kFastHorizonSmall + NX <= 6 * kWave, "bounded variables must fit the wavefront");

// Few searches in a launch (at most kFewSearches) are a matter of latency, not of
// throughput: k_line_search_few takes them, one 1 024-thread workgroup each, and the
// kernels for many leave them alone (fewTaken: that kernel runs in front of this one).
constexpr int kFewSearches = 128;

template <int kFastSlots>
__global__ __launch_bounds__(kFastSlots ? 256 : kWave) void k_line_search_fast(BatchView v, int nextIt,
                                                                                int fewTaken) {
    __shared__ real_t box[kFastSlots ? 1 : 5][kFastSlots ? 1 : kFastBoxed];
    constexpr int kWavesPerBlock = kFastSlots ? 4 : 1;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * kWavesPerBlock + threadIdx.x / kWave));
    const int nwaves = gridDim.x * kWavesPerBlock;
    const int N = v.N, n = NX * N, eEnd = NZ * N + NX, nBounded = 3 * N + NX;
    const int count = __builtin_amdgcn_readfirstlane(v.tailCount[nextIt - 2]);
    if (fewTaken && count <= kFewSearches) return;
    const int s0 = stage_slot(v, 0);
    for (int item = wave; item < count; item += nwaves) {
        const int b = __builtin_amdgcn_readfirstlane(v.tailList[item]);    // uniform: addresses stay scalar
        Inst I = instance(v, b);
        Bisection bs{I.sc->lsMin, I.sc->lsMax, I.sc->lsMid, 0., I.sc->lsIter, I.sc->lsPhase, false};
        const real_t normal = I.sc->lsNormal;
#ifdef NMPC_TAIL_STATS
        const long long t0 = clock64();
        int rounds = 0;
        const int it0 = bs.it;
#endif
        settle(bs);
        // no trial step lies beyond this one
        real_t ahi = 1.;
        if (bs.aMax > ahi) ahi = bs.aMax;
        if (bs.aMin > ahi) ahi = bs.aMin;

        // the variables without finite bounds: A, B and their share of M
        real_t fixed[3] = {0., 0., 0.};
        bool nearBound = false;
        for (int e0 = lane; e0 < eEnd; e0 += 5 * kWave) {     // five values of each vector in flight
            real_t fzu[5], fdz[5], fw[5], flo[5], fup[5], fa[5];
            bool on[5];
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const int e = e0 + i * kWave, s = e / NZ, j = e - s * NZ;
                on[i] = e < eEnd && j < NX && s != s0;
                const int ec = on[i] ? e : 0, ea = on[i] ? s * kWabsStride + j : 0;
                fzu[i] = I.zUnc[ec]; fdz[i] = I.dz[ec]; fw[i] = I.qStep[ec];
                flo[i] = I.zLow[ec]; fup[i] = I.zUpp[ec]; fa[i] = I.wabs[ea];
            }
#pragma unroll
            for (int i = 0; i < 5; i++) {
                if (!on[i]) continue;
                nearBound = nearBound || clips(fzu[i], flo[i], fup[i]) || clips(fzu[i] + ahi * fdz[i], flo[i], fup[i]);
                fixed[0] += fw[i] * fzu[i];
                fixed[1] += fw[i] * fdz[i];
                fixed[2] += fa[i] * (fabs(fzu[i]) + ahi * fabs(fdz[i]));
            }
        }
        // pStep and its magnitudes, by slot
        for (int s = lane; s <= N; s += kWave) {
            fixed[0] += I.pStep[s];
            fixed[2] += I.wabs[s * kWabsStride + NZ];
        }
        // the bounded ones (the state of stage 0, then the controls) stay in registers,
        // or for long horizons in the wavefront's LDS
        constexpr bool kInLds = kFastSlots == 0;
        constexpr int kRegs = kInLds ? 1 : kFastSlots;
        real_t zu[kRegs], dzv[kRegs], lo[kRegs], up[kRegs], wt[kRegs];
        auto fetch_bounded = [&](int c, real_t (&val)[5]) {     // false: c is past the last one
            const bool in = c < nBounded;
            const int k = c < NX ? 0 : (c - NX) / NU;
            const int j = c < NX ? c : NX + (c - NX) - k * NU;
            const int s = stage_slot(v, in ? k : 0), e = s * NZ + j;
            val[0] = in ? I.zUnc[e] : 0.;
            val[1] = in ? I.dz[e] : 0.;
            val[2] = in ? I.zLow[e] : -kInfty;
            val[3] = in ? I.zUpp[e] : kInfty;
            val[4] = in ? I.qStep[e] : 0.;
            if (in) {
                real_t y0, y1;
                const real_t z0 = fabs(clip_gap(val[0], val[2], val[3], &y0, &y1));
                const real_t z1 = fabs(clip_gap(val[0] + ahi * val[1], val[2], val[3], &y0, &y1));
                fixed[2] += I.wabs[s * kWabsStride + j] * (z0 > z1 ? z0 : z1);
            }
        };
        if constexpr (kInLds) {
            for (int c = lane; c < nBounded; c += kWave) {
                real_t val[5];
                fetch_bounded(c, val);
#pragma unroll
                for (int q = 0; q < 5; q++) box[q][c] = val[q];
            }
            // (each lane reads back only what it wrote itself)
        } else {
#pragma unroll
            for (int m = 0; m < kRegs; m++) {
                real_t val[5];
                fetch_bounded(lane + m * kWave, val);
                zu[m] = val[0]; dzv[m] = val[1]; lo[m] = val[2]; up[m] = val[3]; wt[m] = val[4];
            }
        }
        wave_sums(fixed, std::make_integer_sequence<int, 3>{});
        const real_t partA = fixed[0], partB = fixed[1], bound = fixed[2];
        bool open = __builtin_amdgcn_ballot_w64(nearBound) != 0;

        // trial steps in heap order (node 1 is the next trial, nodes 2i and 2i + 1 the
        // trials after node i when its slope is negative / positive), each reached by
        // walking down from the root so that one search state is live at a time
        real_t at[kTailPoints];
        auto plan_round = [&]() {
            const real_t at1 = bs.done ? (real_t)0. : trial_alpha(bs);
#pragma unroll
            for (int i = 1; i <= kTailPoints; i++) {
                Bisection st = bs;
                const int depth = 31 - __builtin_clz(i);
#pragma unroll
                for (int d = depth - 1; d >= 0; d--)
                    if (!st.done) advance_assuming(st, ((i >> d) & 1) == 0);
                at[i - 1] = st.done ? at1 : trial_alpha(st);
            }
        };
#ifdef NMPC_TAIL_STATS
        const long long t1 = clock64();
#endif
        while (!bs.done && !open) {
#ifdef NMPC_TAIL_STATS
            ++rounds;
#endif
            plan_round();
            real_t sums[kTailPoints];
#pragma unroll
            for (int p = 0; p < kTailPoints; p++) sums[p] = 0.;
            if constexpr (kInLds) {
#pragma unroll 2
                for (int c = lane; c < nBounded; c += kWave) {
                    const real_t a0 = box[0][c], a1 = box[1][c], a2 = box[2][c], a3 = box[3][c], a4 = box[4][c];
#pragma unroll
                    for (int p = 0; p < kTailPoints; p++) {
                        real_t y0, y1;
                        sums[p] += a4 * clip_gap(a0 + at[p] * a1, a2, a3, &y0, &y1);
                    }
                }
            } else {
#pragma unroll
                for (int p = 0; p < kTailPoints; p++) {
#pragma unroll
                    for (int m = 0; m < kRegs; m++) {
                        real_t y0, y1;
                        sums[p] += wt[m] * clip_gap(zu[m] + at[p] * dzv[m], lo[m], up[m], &y0, &y1);
                    }
                }
            }
            wave_sums(sums, std::make_integer_sequence<int, kTailPoints>{});
#pragma unroll
            for (int p = 0; p < kTailPoints; p++) sums[p] += partA + at[p] * partB;
            int node = 1;
#pragma unroll
            for (int level = 0; level < kTailDepth; level++) {
                if (!bs.done && !open) {
                    real_t sv = sums[0];
#pragma unroll
                    for (int q = 1; q < kTailPoints; q++) sv = (q == node - 1) ? sums[q] : sv;
                    bool flat, neg;
                    bool settled = slope_decided(sv, bound, n + eEnd + 64, normal, &flat, &neg);
#ifdef NMPC_TEST_ALWAYS_REPEAT_BLOCKS
                    if (bs.it & 1) settled = false;     // test build: exercise the hand-over
#endif
                    if (settled) {
                        advance_decided(bs, flat, neg);
                        node = 2 * node + (neg ? 0 : 1);
                    } else {
                        open = true;
                    }
                }
            }
        }
        if (lane == 0) {
            Scalars* sc = I.sc;
            if (open) {
                sc->lsMin = bs.aMin; sc->lsMax = bs.aMax; sc->lsMid = bs.aMid;
                sc->lsIter = bs.it;
                sc->lsPhase = bs.phase;
                const int to = atomicAdd(&v.tailCount[8 + nextIt - 2], 1);
                v.tailList[v.B + to] = b;
            } else {
                sc->alpha = bs.alpha;
                sc->lsIter = bs.it;
            }
#ifdef NMPC_TAIL_STATS
            // [5] searches, [6] rounds, [7] trials, [13] set-up, [14] rounds: cycles / 256
            atomicAdd(&v.tailCount[5], 1);
            atomicAdd(&v.tailCount[6], rounds);
            atomicAdd(&v.tailCount[7], bs.it - it0);
            atomicAdd(&v.tailCount[13], (int)((t1 - t0) >> 8));
            atomicAdd(&v.tailCount[14], (int)((clock64() - t1) >> 8));
#endif
        }
    }
}

// ------------------------------------------------------------------------
// Continues the step-length searches k_newton_step handed over.  A 1024-thread
// workgroup owns one such instance at a time: the stage vectors the trial
// points are made of stay in registers, trial points and slope terms live in
// LDS only (the reference's scratch copies of them, zVecTmp and y, are
// rewritten by the commit anyway), and each round evaluates the next trial step
// together with every step that can follow it within kTailDepth levels (a
// binary tree: each trial is either left or right of the root of the slope),
// so one round of ordered accumulations, which costs 6 microseconds however
// many run side by side, settles kTailDepth levels.  Leaves the step length
// and trial count in the instance's scalars for k_line_search_commit.
__global__ __launch_bounds__(kTailThreads) void k_line_search_tail(BatchView v, int nextIt) {
    extern __shared__ real_t lds_raw[];
    const int tid = threadIdx.x;
    const int N = v.N, n = NX * N, eEnd = NZ * N + NX, E = NZ * (N + 1);
    real_t* tp = lds_raw;                                   // [points][E] clipped trial points
    real_t* terms = lds_raw + kTailPoints * E;              // [points][stride >= n] slope terms
    const int ts = tail_term_stride(n);
    real_t* slopes = terms + kTailPoints * ts;              // [points]
    real_t* plan = slopes + 8;                              // [0] search over, [1..points] trial steps
    real_t* mags = plan + 8;                                // [points] sums of the terms' magnitudes
    real_t* redo = mags + 8;                                // [0] != 0: this round needs the row-order sums
    // gradient rows: thread t < 6 N owns rows h and h + 6 of block t / 6
    const int rk = tid / (NX / 2), rh = tid % (NX / 2);
    const bool rowOn = rk < N;
    const int count = v.tailCount[kOpenList + nextIt - 2];  // the searches k_line_search_fast left open
    for (int item = blockIdx.x; item < count; item += gridDim.x) {
        const int b = v.tailList[(kOpenList ? v.B : 0) + item];
        Inst I = instance(v, b);
        Bisection bs{I.sc->lsMin, I.sc->lsMax, I.sc->lsMid, 0., I.sc->lsIter, I.sc->lsPhase, false};
        const real_t normal = I.sc->lsNormal;

        real_t zu[kTailElems], dzv[kTailElems], lo[kTailElems], up[kTailElems];
#pragma unroll
        for (int m = 0; m < kTailElems; m++) {
            const int e = tid + m * kTailThreads;
            const bool in = e < eEnd;
            zu[m] = in ? I.zUnc[e] : 0.;
            dzv[m] = in ? I.dz[e] : 0.;
            lo[m] = in ? I.zLow[e] : 0.;
            up[m] = in ? I.zUpp[e] : 0.;
        }
        // this thread's two Jacobian rows, residuals and multiplier-step entries
        real_t ca[NZ], cb[NZ], resA = 0., resB = 0., dlA = 0., dlB = 0.;
#pragma unroll
        for (int j = 0; j < NZ; j++) { ca[j] = 0.; cb[j] = 0.; }
        if (rowOn) {
            const int s = stage_slot(v, rk);
            const real_t* blk = I.C + (size_t)s * (NX * NZ);
            load_row15(blk + rh * NZ, ca);
            load_row15(blk + (rh + NX / 2) * NZ, cb);
            resA = I.c[s * NX + rh];
            resB = I.c[s * NX + rh + NX / 2];
            dlA = I.dl[rk * NX + rh];
            dlB = I.dl[rk * NX + rh + NX / 2];
        }
        settle(bs);
        // The search state lives in wavefront 0; the others only need each round's
        // trial steps and the end of the search, which it leaves in LDS.
        // Trial steps in heap order: node 1 is the next trial, nodes 2i and 2i+1
        // the trials after node i when its slope is negative / positive.
        auto plan_round = [&]() {
            real_t at[kTailPoints + 1];
            Bisection st[kTailPoints + 1];
            st[1] = bs;
#pragma unroll
            for (int i = 1; i <= kTailPoints; i++) {
                at[i] = st[i].done ? at[1] : trial_alpha(st[i]);
                if (2 * i + 1 <= kTailPoints) {
                    st[2 * i] = st[i];
                    st[2 * i + 1] = st[i];
                    if (!st[i].done) {
                        advance_assuming(st[2 * i], true);
                        advance_assuming(st[2 * i + 1], false);
                    }
                }
            }
            if (tid == 0) {
#pragma unroll
                for (int i = 1; i <= kTailPoints; i++) plan[i] = at[i];
                plan[0] = bs.done ? 1. : 0.;
            }
        };
        if (tid < kWave) plan_round();
        __syncthreads();
        while (plan[0] == 0.) {
            real_t at[kTailPoints + 1];
#pragma unroll
            for (int i = 1; i <= kTailPoints; i++) at[i] = plan[i];
#pragma unroll
            for (int m = 0; m < kTailElems; m++) {
                const int e = tid + m * kTailThreads;
                if (e < eEnd) {
#pragma unroll
                    for (int p = 0; p < kTailPoints; p++) {
                        real_t y0, y1;
                        tp[p * E + e] = clip_gap(zu[m] + at[p + 1] * dzv[m], lo[m], up[m], &y0, &y1);
                    }
                }
            }
            __syncthreads();
            if (rowOn) {
                const int s = stage_slot(v, rk), sn = stage_slot(v, rk + 1);
#pragma unroll 1
                for (int p = 0; p < kTailPoints; p++) {
                    const real_t* zk = tp + p * E + s * NZ;
                    real_t a = 0., b2 = 0.;
#pragma unroll
                    for (int j = 0; j < NZ; j++) {
                        const real_t zj = zk[j];
                        a += ca[j] * zj;
                        b2 += cb[j] * zj;
                    }
                    a += resA;
                    b2 += resB;
                    a -= tp[p * E + sn * NZ + rh];
                    b2 -= tp[p * E + sn * NZ + rh + NX / 2];
                    terms[p * ts + rk * NX + rh] = a * dlA;
                    terms[p * ts + rk * NX + rh + NX / 2] = b2 * dlB;
                }
            }
            __syncthreads();
            // Tree sums first: wavefront p < 7 adds up the slope terms of trial point
            // p, wavefront 8 + p their magnitudes.  They settle the search's questions
            // unless a slope sits within rounding of the tolerance; only then are the
            // terms added in row order (one DPP row per trial point, 6 microseconds).
            {
                const int w = tid >> 6, lane = tid & 63, p = w & 7;
                if (p < kTailPoints) {
                    const bool magnitude = w >= 8;
                    const real_t* tv = terms + p * ts;
                    real_t acc = 0.;
                    for (int r = lane; r < n; r += kWave) {
                        const real_t x = tv[r];
                        acc += magnitude ? fabs(x) : x;
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
                    if (lane == 0) (magnitude ? mags : slopes)[p] = acc;
                }
            }
            __syncthreads();
            // wavefront 0 walks down the tree along the slopes found and plans the
            // next round; everyone has read this round's plan before the first
            // barrier above
            if (tid < kWave) {
                const Bisection before = bs;
                bool settled = true;
                int node = 1;
#pragma unroll
                for (int level = 0; level < kTailDepth; level++) {
                    if (!bs.done && settled) {
                        bool flat, neg;
                        settled = slope_decided(slopes[node - 1], mags[node - 1], n, normal, &flat, &neg);
                        if (settled) {
                            advance_decided(bs, flat, neg);
                            node = 2 * node + (neg ? 0 : 1);
                        }
                    }
                }
                if (!settled) bs = before;
                if (tid == 0) redo[0] = settled ? 0. : 1.;
                if (settled) plan_round();
            }
            __syncthreads();
            if (redo[0] != 0.) {
                if (tid < 8 * 16) {     // one DPP row per trial point (the eighth row idles along)
                    const int p = (tid >> 4) < kTailPoints ? (tid >> 4) : kTailPoints - 1;
                    const real_t sv = accumulate_by_row(0., terms + p * ts, n);
                    if ((tid & 15) == 0 && (tid >> 4) < kTailPoints) slopes[p] = sv;
                }
                __syncthreads();
                if (tid < kWave) {
                    int node = 1;
#pragma unroll
                    for (int level = 0; level < kTailDepth; level++) {
                        if (!bs.done) {
                            const real_t sv = slopes[node - 1];
                            const bool neg = bs.phase == 0 ? (sv / normal < 0.) : !(sv > 0);
                            advance(bs, sv, normal);
                            node = 2 * node + (neg ? 0 : 1);
                        }
                    }
                    plan_round();
                }
                __syncthreads();
            }
        }
        if (tid == 0) {
            I.sc->alpha = bs.alpha;
            I.sc->lsIter = bs.it;
        }
        // every wavefront has read plan[0] of this item before wavefront 0 plans the next one
        __syncthreads();
    }
}

// The same continuation for horizons whose trial points do not fit the LDS of
// k_line_search_tail: the trial loop of k_newton_step (one 256-thread workgroup
// per search, trial points through global memory, row-ordered slope sums).
// One search from the state in the instance's scalars to its end, by the whole workgroup.
NMPC_DEV void exact_search(const BatchView& v, const Inst& I, const StageLds& L) {
    const int n = NX * v.N;
    {
        Bisection bs{I.sc->lsMin, I.sc->lsMax, I.sc->lsMid, 0., I.sc->lsIter, I.sc->lsPhase, false};
        const real_t normal = I.sc->lsNormal;
        settle(bs);
        while (!bs.done) {
            stage_pass<kTrialPoint>(v, I, trial_alpha(bs), L);
            // the slope terms summed in any order, with their magnitudes, settle the
            // search's two questions unless the sum sits within rounding of the
            // tolerance (slope_decided); only then are they added in row order
            real_t sm[2] = {0., 0.};
            for (int r = threadIdx.x; r < n; r += kStageThreads) {
                const real_t t = gradient_row(v, I, I.tmp, r) * I.dl[r];
                sm[0] += t;
                sm[1] += fabs(t);
            }
            block_sums(sm, L);
            bool flat, neg;
            bool settled = slope_decided(sm[0], sm[1], n, normal, &flat, &neg);
#ifdef NMPC_TEST_ALWAYS_REPEAT_BLOCKS
            if (bs.it & 2) settled = false;         // test build: both ways
#endif
            if (settled) {
                advance_decided(bs, flat, neg);
            } else {
                const real_t slope = block_ordered_sum(n, L, [&](int r) {
                    return gradient_row(v, I, I.tmp, r) * I.dl[r];
                });
                advance(bs, slope, normal);
            }
        }
        if (threadIdx.x == 0) {
            I.sc->alpha = bs.alpha;
            I.sc->lsIter = bs.it;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(kStageThreads) void k_line_search_exact(BatchView v, int nextIt) {
    extern __shared__ real_t lds_raw[];
    const StageLds L = carve(lds_raw, v.N);
    const int count = v.tailCount[kOpenList + nextIt - 2];
    for (int item = blockIdx.x; item < count; item += gridDim.x)
        exact_search(v, instance(v, v.tailList[(kOpenList ? v.B : 0) + item]), L);
}

// Commit and next Newton set-up of the instances k_line_search_tail finished.
__global__ __launch_bounds__(kStageThreads) void k_line_search_commit(BatchView v, int nextIt) {
    extern __shared__ real_t lds_raw[];
    const StageLds L = carve(lds_raw, v.N);
    const int tid = threadIdx.x;
    const int count = v.tailCount[nextIt - 2];
    for (int item = blockIdx.x; item < count; item += gridDim.x) {
        Inst I = instance(v, v.tailList[item]);
        if (I.sc->lsDeferred != 1) continue;        // k_line_search_few has done it all (uniform)
        const real_t alpha = I.sc->alpha;
        const int lsIter = I.sc->lsIter;
        for (int s = tid; s <= v.N; s += kStageThreads) L.chg[s] = 0;
        __syncthreads();
        if (tid == 0) I.sc->lsDeferred = 2;         // committed, Hessian block rows to come
        commit_step(v, I, L, alpha, lsIter, nextIt, nullptr, false, false);
        __syncthreads();
    }
}

// Newton Hessian block rows of the same instances, sixteen wavefronts each: a
// lone instance's block rows are a latency chain per wavefront.
__global__ __launch_bounds__(kTailThreads) void k_line_search_hessian(BatchView v, int nextIt) {
    extern __shared__ real_t lds_raw[];
    const int count = v.tailCount[nextIt - 2];
    for (int item = blockIdx.x; item < count; item += gridDim.x) {
        Inst I = instance(v, v.tailList[item]);
        if (I.sc->lsDeferred != 2) continue;        // not k_line_search_commit's (uniform)
        if (I.sc->status == ST_RUNNING) assemble_hessian<kTailThreads / kWave>(v, I, lds_raw);
        __syncthreads();
        if (threadIdx.x == 0) I.sc->lsDeferred = 0;
    }
}

#if NMPC_WIDE
// ------------------------------------------------------------------------
// k_line_search_fast's attempt at one search by a workgroup of sixteen wavefronts.
// All of them share the pass over the stage vectors; the bounded variables go to LDS;
// in a round, wavefront w < 14 adds up half of the bounded variables' terms for trial
// point w / 2, and every wavefront then takes the fourteen parts in the same order
// and goes through the same decisions, so the search state stays identical on all
// of them without being exchanged (one barrier per round).  The sums are grouped
// differently from k_line_search_fast's, which the rounding bound covers like any
// other order.  Every thread returns whether the search is left open (its state is
// then in the instance's scalars).
// lds: 5 (3 N + 12) + 4 * 16 + 32 values.
NMPC_DEV bool team_search(const BatchView& v, const Inst& I, real_t* lds) {
    const int tid = threadIdx.x, lane = tid & (kWave - 1), w = tid / kWave;
    const int N = v.N, n = NX * N, eEnd = NZ * N + NX, nBounded = 3 * N + NX;
    real_t* box = lds;                              // [5][nBounded]
    real_t* share = lds + 5 * nBounded;             // [4 kStageWaves]
    real_t* psum = share + 4 * kStageWaves;         // [2][16], by round parity
    const int s0 = stage_slot(v, 0);
    Bisection bs{I.sc->lsMin, I.sc->lsMax, I.sc->lsMid, 0., I.sc->lsIter, I.sc->lsPhase, false};
    const real_t normal = I.sc->lsNormal;
    settle(bs);
    real_t ahi = 1.;
    if (bs.aMax > ahi) ahi = bs.aMax;
    if (bs.aMin > ahi) ahi = bs.aMin;

    // the variables without finite bounds: A, B and their share of M
    real_t fixed[3] = {0., 0., 0.};
    bool nearBound = false;
    for (int e0 = tid; e0 < eEnd; e0 += 2 * kStageThreads) {
        real_t fzu[2], fdz[2], fw[2], flo[2], fup[2], fa[2];
        bool on[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int e = e0 + i * kStageThreads, s = e / NZ, j = e - s * NZ;
            on[i] = e < eEnd && j < NX && s != s0;
            const int ec = on[i] ? e : 0, ea = on[i] ? s * kWabsStride + j : 0;
            fzu[i] = I.zUnc[ec]; fdz[i] = I.dz[ec]; fw[i] = I.qStep[ec];
            flo[i] = I.zLow[ec]; fup[i] = I.zUpp[ec]; fa[i] = I.wabs[ea];
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            if (!on[i]) continue;
            nearBound = nearBound || clips(fzu[i], flo[i], fup[i]) || clips(fzu[i] + ahi * fdz[i], flo[i], fup[i]);
            fixed[0] += fw[i] * fzu[i];
            fixed[1] += fw[i] * fdz[i];
            fixed[2] += fa[i] * (fabs(fzu[i]) + ahi * fabs(fdz[i]));
        }
    }
    for (int s = tid; s <= N; s += kStageThreads) {
        fixed[0] += I.pStep[s];
        fixed[2] += I.wabs[s * kWabsStride + NZ];
    }
    // the bounded ones (the state of stage 0, then the controls) into LDS
    for (int c = tid; c < nBounded; c += kStageThreads) {
        const int k = c < NX ? 0 : (c - NX) / NU;
        const int j = c < NX ? c : NX + (c - NX) - k * NU;
        const int s = stage_slot(v, k), e = s * NZ + j;
        const real_t zu = I.zUnc[e], dz = I.dz[e], lo = I.zLow[e], up = I.zUpp[e];
        box[c] = zu; box[nBounded + c] = dz; box[2 * nBounded + c] = lo; box[3 * nBounded + c] = up;
        box[4 * nBounded + c] = I.qStep[e];
        real_t y0, y1;
        const real_t z0 = fabs(clip_gap(zu, lo, up, &y0, &y1));
        const real_t z1 = fabs(clip_gap(zu + ahi * dz, lo, up, &y0, &y1));
        fixed[2] += I.wabs[s * kWabsStride + j] * (z0 > z1 ? z0 : z1);
    }
    wave_sums(fixed, std::make_integer_sequence<int, 3>{});
    const bool nearAny = __builtin_amdgcn_ballot_w64(nearBound) != 0;
    if (lane == 0) {
        share[4 * w] = fixed[0]; share[4 * w + 1] = fixed[1]; share[4 * w + 2] = fixed[2];
        share[4 * w + 3] = nearAny ? 1. : 0.;
    }
    __syncthreads();
    fixed[0] = fixed[1] = fixed[2] = 0.;
    nearBound = false;
    if (lane < kStageWaves) {
        fixed[0] = share[4 * lane]; fixed[1] = share[4 * lane + 1]; fixed[2] = share[4 * lane + 2];
        nearBound = share[4 * lane + 3] != 0.;
    }
    wave_sums(fixed, std::make_integer_sequence<int, 3>{});
    const real_t partA = fixed[0], partB = fixed[1], bound = fixed[2];
    bool open = __builtin_amdgcn_ballot_w64(nearBound) != 0;

    real_t at[kTailPoints];
    auto plan_round = [&]() {       // as in k_line_search_fast
        const real_t at1 = bs.done ? (real_t)0. : trial_alpha(bs);
#pragma unroll
        for (int i = 1; i <= kTailPoints; i++) {
            Bisection st = bs;
            const int depth = 31 - __builtin_clz(i);
#pragma unroll
            for (int d = depth - 1; d >= 0; d--)
                if (!st.done) advance_assuming(st, ((i >> d) & 1) == 0);
            at[i - 1] = st.done ? at1 : trial_alpha(st);
        }
    };
    const int point = w >> 1;
    const int mid = (nBounded + 1) / 2;
    const int cFrom = (w & 1) ? mid : 0, cTo = (w & 1) ? nBounded : mid;
    for (int round = 0; !bs.done && !open; ++round) {
        plan_round();
        real_t* part = psum + (round & 1) * 16;
        if (w < 2 * kTailPoints) {
            real_t atp = at[0];
#pragma unroll
            for (int q = 1; q < kTailPoints; q++) atp = (q == point) ? at[q] : atp;
            real_t mine[1] = {0.};
            for (int c = cFrom + lane; c < cTo; c += kWave) {
                real_t y0, y1;
                mine[0] += box[4 * nBounded + c] *
                           clip_gap(box[c] + atp * box[nBounded + c], box[2 * nBounded + c], box[3 * nBounded + c], &y0, &y1);
            }
            wave_sums(mine, std::make_integer_sequence<int, 1>{});
            if (lane == 0) part[w] = mine[0];
        }
        __syncthreads();
        real_t sums[kTailPoints];
#pragma unroll
        for (int p = 0; p < kTailPoints; p++) {
            sums[p] = part[2 * p] + part[2 * p + 1];
            sums[p] += partA + at[p] * partB;
        }
        int node = 1;
#pragma unroll
        for (int level = 0; level < kTailDepth; level++) {
            if (!bs.done && !open) {
                real_t sv = sums[0];
#pragma unroll
                for (int q = 1; q < kTailPoints; q++) sv = (q == node - 1) ? sums[q] : sv;
                bool flat, neg;
                bool settled = slope_decided(sv, bound, n + eEnd + 64, normal, &flat, &neg);
#ifdef NMPC_TEST_ALWAYS_REPEAT_BLOCKS
                if (bs.it & 1) settled = false;     // test build: exercise the hand-over
#endif
                if (settled) {
                    advance_decided(bs, flat, neg);
                    node = 2 * node + (neg ? 0 : 1);
                } else {
                    open = true;
                }
            }
        }
    }
    if (tid == 0) {
        Scalars* sc = I.sc;
        if (open) {             // the state the row-ordered continuation goes on from
            sc->lsMin = bs.aMin; sc->lsMax = bs.aMax; sc->lsMid = bs.aMid;
            sc->lsIter = bs.it;
            sc->lsPhase = bs.phase;
        } else {
            sc->alpha = bs.alpha;
            sc->lsIter = bs.it;
        }
    }
    __syncthreads();
    return open;
}

// At most kFewSearches searches in the launch: each gets a workgroup of sixteen
// wavefronts for the first attempt, the commit of the step and the next Newton
// set-up with its Hessian block rows (what k_line_search_fast, _commit and _hessian
// do in three launches).  A search the first attempt leaves open goes on in the same
// workgroup with trial points through global memory and row-ordered sums
// (exact_search, what k_line_search_exact does), so none of a launch's few searches
// is left for the kernels for many.
__global__ __launch_bounds__(kStageThreads) void k_line_search_few(BatchView v, int nextIt) {
    extern __shared__ real_t lds_raw[];
    const int tid = threadIdx.x;
    const int count = v.tailCount[nextIt - 2];
    if (count > kFewSearches) return;
    const StageLds L = carve(lds_raw, v.N);         // (the search is done with the LDS when the commit begins)
    for (int item = blockIdx.x; item < count; item += gridDim.x) {
        const int b = v.tailList[item];
        Inst I = instance(v, b);
        if (team_search(v, I, lds_raw)) exact_search(v, I, L);  // uniform
        const real_t alpha = I.sc->alpha;
        const int lsIter = I.sc->lsIter;
        for (int s = tid; s <= v.N; s += kStageThreads) L.chg[s] = 0;
        __syncthreads();
        if (tid == 0) I.sc->lsDeferred = 0;
        commit_step(v, I, L, alpha, lsIter, nextIt, nullptr, false, true);
        __syncthreads();
    }
}
#endif

// Dynamic LDS beyond the default cap of 64 KB has to be asked for.
static void allow_lds(const void* kernel, size_t bytes) {
    if (bytes > 64 * 1024)
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
void qp_begin(const BatchView& v, const real_t* meas, hipStream_t s) {
    allow_lds((const void*)k_qp_begin, stage_lds_bytes(v.N));
    hipLaunchKernelGGL(k_qp_begin, dim3(v.nb), dim3(kStageThreads), stage_lds_bytes(v.N), s, v, meas);
}
void newton_step(const BatchView& v, int next_it, hipStream_t s) {
    // Long horizons: 1 024 x 500 cold 16.38 -> 16.06 ms; at horizon 100 the second launch
    // costs what it saves (4 096 x 100: 5.15 against 5.17 ms).  NMPC_HESSIAN_APART=0 / 1: never / always.
    static const int forced = getenv("NMPC_HESSIAN_APART") ? atoi(getenv("NMPC_HESSIAN_APART")) : -1;
    const bool apart = next_it <= kMaxIter && (forced >= 0 ? forced != 0 : v.N >= 200);
    allow_lds((const void*)k_newton_step, step_lds_bytes(v.N));
    hipLaunchKernelGGL(k_newton_step, dim3(v.nb), dim3(kStageThreads), step_lds_bytes(v.N), s, v, next_it,
                       apart ? 1 : 0);
    if (apart) {
        int parts = (v.N + 31) / 32;
        parts = parts > 16 ? 16 : parts;
        allow_lds((const void*)k_step_hessian, kStageWaves * kSliceDoubles * sizeof(real_t));
        hipLaunchKernelGGL(k_step_hessian, dim3(v.nb * parts), dim3(kStageThreads),
                           kStageWaves * kSliceDoubles * sizeof(real_t), s, v, parts);
    }
}
bool tail_enabled(const BatchView& v) {
    return v.tailList != nullptr && v.wabs != nullptr && v.N <= kFastHorizonLarge;
}
#if NMPC_WIDE
void line_search_few(const BatchView& v, int next_it, hipStream_t s) {
    const int grid = v.nb < kFewSearches ? v.nb : kFewSearches;
    const size_t search = (size_t)(5 * (3 * v.N + NX) + 4 * kStageWaves + 32) * sizeof(real_t);
    const size_t lds = stage_lds_bytes(v.N) > search ? stage_lds_bytes(v.N) : search;
    allow_lds((const void*)k_line_search_few, lds);
    hipLaunchKernelGGL(k_line_search_few, dim3(grid), dim3(kStageThreads), lds, s, v, next_it);
}
#endif
void line_search_tail(const BatchView& v, int next_it, hipStream_t s, bool few_taken) {
    const dim3 grid_fast((v.nb + 3) / 4);
    // (a batch of at most kFewSearches instances never has more searches than that)
    if (kFirstAttempt && !(few_taken && v.nb <= kFewSearches)) {
        if (v.N <= kFastHorizonSmall)
            hipLaunchKernelGGL(k_line_search_fast<6>, grid_fast, dim3(256), 0, s, v, next_it, few_taken ? 1 : 0);
        else
            hipLaunchKernelGGL(k_line_search_fast<0>, dim3(v.nb), dim3(kWave), 0, s, v, next_it, few_taken ? 1 : 0);
    }
    // what the first attempt left open
    if (v.N <= kTailMaxHorizon) {
        // Each workgroup takes most of a CU's LDS, so placing one waits for the CU to
        // drain, kernels of other streams included; after the first attempt few
        // searches are left (about one in 5 000), and a small grid loops over them.
        const int cap = kFirstAttempt ? 32 : 256;
        const int grid = v.nb < cap ? v.nb : cap;
        const size_t lds = tail_lds_bytes(v.N);
        if (lds > 64 * 1024)    // beyond the default cap on dynamic LDS
            (void)hipFuncSetAttribute((const void*)k_line_search_tail,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_line_search_tail, dim3(grid), dim3(kTailThreads), lds, s, v, next_it);
    } else {
        const int grid = v.nb < 2048 ? v.nb : 2048;
        allow_lds((const void*)k_line_search_exact, stage_lds_bytes(v.N));
        hipLaunchKernelGGL(k_line_search_exact, dim3(grid), dim3(kStageThreads), stage_lds_bytes(v.N), s, v,
                           next_it);
    }
    // the search kernel fills a CU with one workgroup; the two below are lighter,
    // and warm-started batches hand over half of their instances
    const int grid_commit = v.nb < 2048 ? v.nb : 2048;
    const int grid_hessian = v.nb < 512 ? v.nb : 512;
    allow_lds((const void*)k_line_search_commit, stage_lds_bytes(v.N));
    hipLaunchKernelGGL(k_line_search_commit, dim3(grid_commit), dim3(kStageThreads),
                       stage_lds_bytes(v.N), s, v, next_it);
    if (next_it <= kMaxIter)
        hipLaunchKernelGGL(k_line_search_hessian, dim3(grid_hessian), dim3(kTailThreads),
                           (kTailThreads / kWave) * kSliceDoubles * sizeof(real_t), s, v, next_it);
}

}  // inline namespace

#if !NMPC_OWN_CFG && !NMPC_WIDE
#define NMPC_OTHER_COMPILATION(ns)                                                  \
    namespace ns {                                                                  \
    void qp_begin(const BatchView& v, const real_t* meas, hipStream_t s);           \
    void newton_step(const BatchView& v, int next_it, hipStream_t s);               \
    void line_search_tail(const BatchView& v, int next_it, hipStream_t s, bool few_taken); \
    void line_search_few(const BatchView& v, int next_it, hipStream_t s);           \
    }
NMPC_OTHER_COMPILATION(perinstance)
NMPC_OTHER_COMPILATION(batchwide_wide)
NMPC_OTHER_COMPILATION(perinstance_wide)
#undef NMPC_OTHER_COMPILATION

namespace {
// A batch too small to fill the chip with 256-thread workgroups (it has no parts either).
bool small_batch(const BatchView& v) { return !v.list && v.nb <= 256; }
// Workgroups of 1 024 threads where a launch of few instances leaves most of the chip
// idle: every stage kernel of a small batch, the opening one included, and from 200
// stages on the kernels of the long part of a two-part iteration (the opening kernel
// of a larger batch has every instance to work on and stays as it is).
// NMPC_WIDE_STAGE=0 / 1: never / always.
bool wide_stage(const BatchView& v) {
    static const int forced = getenv("NMPC_WIDE_STAGE") ? atoi(getenv("NMPC_WIDE_STAGE")) : -1;
    if (!kDoublePrecision) return false;
    // the long part of a two-part iteration, or a batch too small to have parts; a launch
    // over every instance of a larger batch is bound by throughput and slower this way
    return forced >= 0 ? forced != 0 : ((v.N >= 200 && v.listPart == 1) || small_batch(v));
}
}  // namespace

void launch_qp_begin(const BatchView& v, const real_t* meas, hipStream_t s) {
    static const int forced = getenv("NMPC_WIDE_STAGE") ? atoi(getenv("NMPC_WIDE_STAGE")) : -1;
    if (kDoublePrecision && (forced >= 0 ? forced != 0 : small_batch(v))) {
        if (v.cfg) perinstance_wide::qp_begin(v, meas, s);
        else batchwide_wide::qp_begin(v, meas, s);
    } else {
        if (v.cfg) perinstance::qp_begin(v, meas, s);
        else batchwide::qp_begin(v, meas, s);
    }
}
void launch_newton_step(const BatchView& v, int next_it, hipStream_t s) {
    if (wide_stage(v)) {
        if (v.cfg) perinstance_wide::newton_step(v, next_it, s);
        else batchwide_wide::newton_step(v, next_it, s);
    } else {
        if (v.cfg) perinstance::newton_step(v, next_it, s);
        else batchwide::newton_step(v, next_it, s);
    }
}
bool line_search_tail_enabled(const BatchView& v) { return batchwide::tail_enabled(v); }
void launch_line_search_tail(const BatchView& v, int next_it, hipStream_t s) {
    // k_line_search_few in front: the device decides by the number of searches which
    // kernels take them (NMPC_FEW_SEARCHES=0: always the kernels for many)
    static const bool few = !(getenv("NMPC_FEW_SEARCHES") && atoi(getenv("NMPC_FEW_SEARCHES")) == 0);
    const bool taken = few && kDoublePrecision && v.N <= 508;       // kFastHorizonLarge
    if (taken) {
        if (v.cfg) perinstance_wide::line_search_few(v, next_it, s);
        else batchwide_wide::line_search_few(v, next_it, s);
        if (!v.list && v.nb <= 128) return;         // kFewSearches: a batch of that size never has more
    }
    if (wide_stage(v)) {
        if (v.cfg) perinstance_wide::line_search_tail(v, next_it, s, taken);
        else batchwide_wide::line_search_tail(v, next_it, s, taken);
    } else {
        if (v.cfg) perinstance::line_search_tail(v, next_it, s, taken);
        else batchwide::line_search_tail(v, next_it, s, taken);
    }
}
#endif

}  // namespace nmpc
