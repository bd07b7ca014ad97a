// Newton-system half of the dual Newton solve: four controller instances per
// 64-lane wavefront.
//
//   k_newton_solve    reverse block-tridiagonal Cholesky, regularisation,
//                     two substitutions (dual_qp.c:601-698, 817-908, 993-1055)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <utility>

#include "kernels.hpp"
#include "qp_common.hpp"

#pragma clang fp contract(off)

namespace nmpc {

namespace {


// ------------------------------------------------------------------------
// k_newton_solve works in 16-lane groups, four controller instances per
// wavefront.  Lane t < 12 of a group owns column t of the current diagonal
// block, of the current sub-diagonal block and of the sub-diagonal block below.
// Values of another column of the same instance come from a DPP row broadcast
// (the 16 lanes of a group are one DPP row); nothing goes through LDS.
// Every lane of the wavefront stays active throughout, results of groups that
// have nothing to do are computed and dropped.
constexpr int kGroupsPerWave = 4;
[[maybe_unused]] constexpr unsigned kOneWavePerSimd = 1024;   // 256 CUs x 4 SIMDs

using Seq12 = std::make_integer_sequence<int, NX>;

NMPC_DEV real_t group_min(real_t x) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        const real_t y = __shfl_xor(x, o, 16);
        x = y < x ? y : x;
    }
    return x;
}

// The substitutions divide by pivots that are known long before the numerator
// is.  With y = RN(1 / d) prepared ahead, the correctly rounded quotient takes
// three dependent instructions instead of about ten (Markstein's sequence:
// q0 = RN(n y) is a faithful quotient, the residual is exact in an FMA, and the
// corrected value rounds to RN(n / d)).  The faithfulness argument excludes
// divisors whose significand is all ones; blocks that hold one use plain
// division.  Checked against n / d on 3.4e10 random, adversarial and exact
// (zero numerators of either sign among them) operand pairs on gfx950
// (tools/micro_div.hip): no difference.  Divisors are positive here (square
// roots); over a negative one a zero quotient would keep the numerator's sign.
NMPC_DEV bool plain_division_needed(real_t d) {
#ifdef NMPC_TEST_ALWAYS_REPEAT_BLOCKS     // test builds: the substitutions divide plainly too
    return true;
#else
    if constexpr (!kDoublePrecision) return true;   // the fast forms are checked for double only
    return significand_all_ones((double)d);
#endif
}
template <bool FAST>
NMPC_DEV real_t divide(real_t n, const Pivot& p) {
    if (!FAST) return n / p.d;
    // The residual is taken with its sign turned: a residual of exactly zero comes
    // out of the FMA as +0 and enters the correction as -0, which leaves q0 and
    // the sign of a zero quotient as they are ((-0) / d is -0).
    const real_t q0 = n * p.y;
    const real_t rn = __builtin_fma(p.d, q0, -n);
    return __builtin_fma(-rn, p.y, q0);
}

// Column JJ of block row k of the reverse Cholesky (dual_qp.c:832-903): pivot,
// the entries left of it in the diagonal block and the whole row of the
// sub-diagonal block.  L / Ls / Ln hold this lane's column of chol(k,0),
// chol(k,-1) and chol(k+1,-1); on entry L[JJ] and Ls[JJ] still hold the Newton
// Hessian entries they are computed from.
//
// The reference takes the products away from the pivot one after the other,
// the one with the row just finished first, so a column is one chain of up to
// 23 dependent subtractions.  Only that chain has to be serial: all products
// of column JJ but the first are known while column JJ+1 is still summing, and
// are made there, between its subtractions (Products q), each pinned in
// place so that the compiler keeps the interleaving.
struct Products {
    real_t a[NX];   // L[l]  * chol(k,0,l,JJ),     l >= JJ+2
    real_t b[NX];   // Ls[l] * chol(k,0,l,JJ),     l >= JJ+2
    real_t c[NX];   // Ln[l] * chol(k+1,-1,l,JJ),  all l
};

#define NMPC_PIN(x) asm volatile("" : "+v"(x))

// Item I of the products of column JJ: the twelve c first, then the (a, b) pairs.
template <int JJ, int I>
NMPC_DEV void make_product(Products& q, const real_t (&L)[NX], const real_t (&Ls)[NX],
                           const real_t (&Ln)[NX]) {
    if constexpr (JJ >= 0 && I < NX) {
        q.c[I] = Ln[I] * row_bcast<JJ>(Ln[I]);
        NMPC_PIN(q.c[I]);
    } else if constexpr (JJ >= 0 && JJ + 2 + (I - NX) < NX) {
        constexpr int l = JJ + 2 + (I - NX);
        const real_t ljj = row_bcast<JJ>(L[l]);
        q.a[l] = L[l] * ljj;
        q.b[l] = Ls[l] * ljj;
        NMPC_PIN(q.a[l]);
        NMPC_PIN(q.b[l]);
    }
}

template <int JJ, int... I>
NMPC_DEV void make_products(Products& q, const real_t (&L)[NX], const real_t (&Ls)[NX],
                            const real_t (&Ln)[NX], std::integer_sequence<int, I...>) {
    (make_product<JJ, I>(q, L, Ls, Ln), ...);
}

// Step S of column JJ's chain, followed by item S of the next column's products.
template <int JJ, int S>
NMPC_DEV void chain_step(real_t& sum, real_t& sub, const Products& p, Products& q,
                         const real_t (&L)[NX], const real_t (&Ls)[NX], const real_t (&Ln)[NX]) {
    constexpr int nAB = JJ + 2 < NX ? NX - 2 - JJ : 0;
    if constexpr (S < nAB) {
        sum -= p.a[JJ + 2 + S];
        sub -= p.b[JJ + 2 + S];
        NMPC_PIN(sub);
    } else {
        sum -= p.c[S - nAB];
    }
    NMPC_PIN(sum);
    make_product<JJ - 1, S>(q, L, Ls, Ln);
}

template <int JJ, int... S>
NMPC_DEV void chain_steps(real_t& sum, real_t& sub, const Products& p, Products& q,
                          const real_t (&L)[NX], const real_t (&Ls)[NX], const real_t (&Ln)[NX],
                          std::integer_sequence<int, S...>) {
    (chain_step<JJ, S>(sum, sub, p, q, L, Ls, Ln), ...);
}

// FAST takes root and reciprocal together and leaves in `ones` the largest low
// word of a root: a root whose significand is all ones (the one case the
// reciprocal is not good for) shows there as 0xffffffff, and the caller then
// repeats the block with FAST off.
template <bool FAST, int JJ>
NMPC_DEV void factor_column(real_t (&L)[NX], real_t (&Ls)[NX], const real_t (&Ln)[NX],
                            Products& p, int t, int& failCol, real_t& diag, uint32_t& ones) {
    constexpr int nAB = JJ + 2 < NX ? NX - 2 - JJ : 0;
    real_t sum = L[JJ], sub = Ls[JJ];
    if constexpr (JJ + 1 < NX) {
        const real_t ljj = row_bcast<JJ>(L[JJ + 1]);    // chol(k,0,JJ+1,JJ), just finished
        sum -= L[JJ + 1] * ljj;
        sub -= Ls[JJ + 1] * ljj;
    }
    Products q;
    chain_steps<JJ>(sum, sub, p, q, L, Ls, Ln, std::make_integer_sequence<int, nAB + NX>{});
    // the next column has one product more than this chain has steps
    make_product<JJ - 1, nAB + NX>(q, L, Ls, Ln);
    const real_t piv = row_bcast<JJ>(sum);
    if (failCol < 0 && piv < 1.e2 * kEqTol) failCol = JJ;
    real_t d, ld, ls;
    if constexpr (FAST) {
        const Pivot pv = root_and_reciprocal(piv);
        d = pv.d;
        ld = divide<true>(sum, pv);
        ls = divide<true>(sub, pv);
        const uint32_t lo = (uint32_t)__double_as_longlong(pv.d);
        ones = lo > ones ? lo : ones;
    } else {
        d = sqrt(piv);
        ld = sum / d;
        ls = sub / d;
    }
    diag = (t == JJ) ? d : diag;
    L[JJ] = (t == JJ) ? d : ld;
    Ls[JJ] = ls;
    if constexpr (JJ > 0) p = q;
}

template <bool FAST, int... JJ>
NMPC_DEV void factor_columns(real_t (&L)[NX], real_t (&Ls)[NX], const real_t (&Ln)[NX],
                             int t, int& failCol, real_t& diag, uint32_t& ones,
                             std::integer_sequence<int, JJ...>) {
    // The last block row has nothing below it: Ln is zero there, and taking
    // +0 away leaves a sum as it is, so its c terms need no branch around them.
    Products p;
    make_products<NX - 1>(p, L, Ls, Ln, Seq12{});
    (factor_column<FAST, NX - 1 - JJ>(L, Ls, Ln, p, t, failCol, diag, ones), ...);
}

// Row R of the bottom-up substitution L' x = g for block k (dual_qp.c:1003-1026).
template <bool FAST, int R>
NMPC_DEV void backsolve_row(real_t (&xk)[NX], real_t& mine, const real_t (&L)[NX],
                            const real_t (&pn)[NX], real_t g, const Pivot& dd, int t) {
    real_t sum = g;
#pragma unroll
    for (int j = R + 1; j < NX; ++j) sum -= L[j] * xk[j];
#pragma unroll
    for (int j = 0; j < NX; ++j) sum -= pn[j];          // +0 in the last block row
    const real_t x = divide<FAST>(sum, dd);
    xk[R] = row_bcast<R>(x);
    mine = (t == R) ? x : mine;
}

template <bool FAST, int... R>
NMPC_DEV void backsolve_rows(real_t (&xk)[NX], real_t& mine, const real_t (&L)[NX],
                             const real_t (&pn)[NX], real_t g, const Pivot& dd, int t,
                             std::integer_sequence<int, R...>) {
    (backsolve_row<FAST, NX - 1 - R>(xk, mine, L, pn, g, dd, t), ...);
}

// Column J of the top-down substitution L x = y for block k
// (dual_qp.c:1029-1052), column-oriented: once x_J is final every later row
// subtracts its L(i,J) x_J, which keeps each row's terms in ascending J order.
template <bool FAST, int J>
NMPC_DEV void forward_column(real_t (&xk)[NX], real_t& mine, real_t& sum,
                             const real_t (&Ld)[NX], const Pivot& dd, int t) {
    const real_t x = divide<FAST>(sum, dd);
    const real_t xj = row_bcast<J>(x);
    xk[J] = xj;
    mine = (t == J) ? x : mine;
    sum -= Ld[J] * xj;
}

template <bool FAST, int... J>
NMPC_DEV void forward_columns(real_t (&xk)[NX], real_t& mine, real_t& sum,
                              const real_t (&Ld)[NX], const Pivot& dd, int t,
                              std::integer_sequence<int, J...>) {
    (forward_column<FAST, J>(xk, mine, sum, Ld, dd, t), ...);
}

// Row l of a block row (24 doubles, 192 bytes) into LDS at image + l * 1024 +
// lane * 16.  The instruction offset moves both the global and the LDS
// address, so the LDS base handed over makes up the difference.
typedef __attribute__((address_space(3))) char* lds_bytes;
template <int... Lrow>
NMPC_DEV void fetch_rows_to_lds(const real_t* g0, lds_bytes image, std::integer_sequence<int, Lrow...>) {
    (__builtin_amdgcn_global_load_lds(
         (const __attribute__((address_space(1))) void*)g0,
         (__attribute__((address_space(3))) void*)(image + Lrow * (1024 - 192)), 16, Lrow * 192, 0),
     ...);
}

constexpr int kCholPieces = (kCholRow + 2 * NX - 1) / (2 * NX);     // LDS-DMA pieces of 24 values

// Entry t of a 12-vector held in registers, t known only at run time.
NMPC_DEV real_t L_diag(const real_t (&L)[NX], int t) {
    real_t d = L[0];
#pragma unroll
    for (int l = 1; l < NX; l++) d = (t == l) ? L[l] : d;
    return d;
}

// One sweep from block N-1 down to 0: blocks k <= start of an `active` group
// are factorised (dual_qp.c:817-908), the others read their factor back; each
// block then takes its step of the bottom-up substitution, which only needs
// the factor rows just produced.  Returns ST_OK or ST_DIV_ZERO per group and
// lowers minDiag to the smallest diagonal factor entry seen.
// Block rows are fetched one iteration ahead, either into registers (STAGED
// false) or straight into LDS with global_load_lds_dwordx4 (STAGED true): no
// registers are held while the data is on its way, which brings the kernel
// under 256 registers so that two wavefronts can share a SIMD; the price is a
// wait for the LDS image at the top of every block, so batches that do not
// fill the chip twice over keep the register form.
// STAGED: lane t < 12 of a group brings in entries 2t and 2t+1 of each of the
// twelve 24-entry rows, so the LDS image of row l is rows[l][group][24 of 32];
// lane t then reads its column of the sub-diagonal block (entry t) and of the
// diagonal block (entry 12 + t).
template <bool STAGED>
NMPC_DEV int backward_pass(const BatchView& v, const Inst& I, int start, bool active, int t,
                           real_t& minDiag) {
    const int N = v.N;
    const bool col = t < NX;
    const int tc = col ? t : 0;
    int status = ST_OK;
    real_t Ln[NX], pn[NX], xk[NX];      // pn[j] = chol(k+1,-1,j,t) * x_{k+1}[j]
#pragma unroll
    for (int j = 0; j < NX; j++) { Ln[j] = 0.; pn[j] = 0.; xk[j] = 0.; }

    // When any group of the wavefront factorises a block, every group does,
    // starting from the Newton Hessian: a block above a group's `start` has the
    // same Hessian and the same factor below it as when it was last factorised
    // (that is why the reference skips it), so the result is the factor it would
    // have read back, and the columns need no per-lane choice between a computed
    // and a fetched value.  Groups whose results are dropped fetch nothing and
    // work on zeros.
    __shared__ real_t rows[STAGED ? NX * kWave * 2 : 1];
    const lds_bytes image = (lds_bytes)rows;
    const int slot = (threadIdx.x >> 4) * 32;           // this group's part of a row image
    real_t nL[STAGED ? 1 : NX], nLs[STAGED ? 1 : NX], ng = 0., nd = 1.;
    if constexpr (STAGED) {
        if (!active) {
#pragma unroll
            for (int l = 0; l < NX; l++) {
                rows[l * 128 + 2 * threadIdx.x] = 0.;
                rows[l * 128 + 2 * threadIdx.x + 1] = 0.;
            }
        }
    } else {
#pragma unroll
        for (int l = 0; l < NX; l++) { nL[l] = 0.; nLs[l] = 0.; }
    }
    // Lanes right of the diagonal (t > l) receive whatever follows row l in a factor
    // block row: they only ever feed their own, unused sums.
    const int diagAt = chol_diag_row(tc) + tc;
    auto fetch_block_row = [&](bool fromH, int kk) {
        if constexpr (STAGED) {
            const real_t* g0 = (fromH ? I.hess : I.chol) + (size_t)kk * kCholRow + 2 * tc;
            if (active && col) fetch_rows_to_lds(g0, image, std::make_integer_sequence<int, kCholPieces>{});
            if (active) ng = I.grad[kk * NX + tc];
        } else if (active) {
            // the Newton Hessian and the factor have one layout: one load per value
            const real_t* row = (fromH ? I.hess : I.chol) + (size_t)kk * kCholRow + tc;
#pragma unroll
            for (int l = 0; l < NX; l++) {
                nL[l] = row[kCholDiagRow[l]];
                nLs[l] = row[NX * l];
            }
            if (!fromH) nd = I.chol[(size_t)kk * kCholRow + diagAt];
            ng = I.grad[kk * NX + tc];
        }
    };
    bool nextFromHess = __any(active && N - 1 <= start);
    fetch_block_row(nextFromHess, N - 1);

    for (int k = N - 1; k >= 0; --k) {
        const bool fromHess = nextFromHess;
        const bool factorThis = active && status == ST_OK && k <= start;
        real_t L[NX], Ls[NX];
        real_t dd;                                      // chol(k,0,t,t) unless refactorised below
        if constexpr (STAGED) {
            // the block row fetched during the last iteration has landed ...
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // value o of a block row sits in piece o / 24 at o % 24
            const real_t* mine = rows + slot + tc;
#pragma unroll
            for (int l = 0; l < NX; l++) {
                Ls[l] = mine[(l / 2) * 128 + (l % 2) * NX];
                L[l] = mine[(kCholDiagRow[l] / (2 * NX)) * 128 + kCholDiagRow[l] % (2 * NX)];
            }
            // ... and is in registers before the next one overwrites the image
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dd = (fromHess || !active) ? 1. : L_diag(L, tc);
        } else {
#pragma unroll
            for (int l = 0; l < NX; l++) { L[l] = nL[l]; Ls[l] = nLs[l]; }
            dd = nd;
        }
        const real_t g = ng;
        if (k > 0) {
            // a failure in this block is not known yet; a group that fails stops
            // using its results, so the guess "still factorising" is harmless
            nextFromHess = __any(active && status == ST_OK && k - 1 <= start);
            fetch_block_row(nextFromHess, k - 1);
        }
        if (fromHess) {
            int failCol = -1;
            uint32_t ones = 0;
            // (single precision takes roots and quotients plainly: the fast forms are
            // checked for double only)
            factor_columns<kDoublePrecision>(L, Ls, Ln, t, failCol, dd, ones, Seq12{});
#ifdef NMPC_TEST_ALWAYS_REPEAT_BLOCKS     // test builds: exercise the repeat path on every block
            ones = 0xffffffffu;
#endif
            if (kDoublePrecision && __any(ones == 0xffffffffu)) {
                if (active) {
#pragma unroll
                    for (int l = 0; l < NX; l++) {
                        L[l] = I.hess[(size_t)k * kCholRow + kCholDiagRow[l] + tc];
                        Ls[l] = I.hess[(size_t)k * kCholRow + NX * l + tc];
                    }
                }
                failCol = -1;
                factor_columns<false>(L, Ls, Ln, t, failCol, dd, ones, Seq12{});
            }
            if (factorThis && col) {
                // A failed pivot in column failCol ends the reference's factorisation
                // before it writes row failCol (dual_qp.c:855-866): the rows below it
                // stay as they were.
                real_t* row = I.chol + (size_t)k * kCholRow;
#pragma unroll
                for (int l = 0; l < NX; l++)
                    if (l > failCol && l >= t) row[kCholDiagRow[l] + t] = L[l];
                if (k > 0) {
#pragma unroll
                    for (int l = 0; l < NX; l++)
                        if (l > failCol) row[NX * l + t] = Ls[l];
                }
            }
            if (factorThis && failCol >= 0) status = ST_DIV_ZERO;
        }
        if (col && minDiag > dd) minDiag = dd;

        // bottom-up substitution, block k
        real_t mine = 0.;
        const Pivot pv{dd, 1.0 / dd};
        if (__any(plain_division_needed(dd)))
            backsolve_rows<false>(xk, mine, L, pn, g, pv, t, Seq12{});
        else
            backsolve_rows<true>(xk, mine, L, pn, g, pv, t, Seq12{});
        // into scratch: the step itself is only written once the factor is known to be good
        if (active && status == ST_OK && col) I.tmp[k * NX + t] = mine;
#pragma unroll
        for (int j = 0; j < NX; j++) { pn[j] = Ls[j] * xk[j]; Ln[j] = Ls[j]; }
    }
    return status;
}

// Top-down substitution L x = y over all blocks (dual_qp.c:1029-1052); lane t
// owns row t of each block.
// Also leaves the squared norm of the step, summed in row order, in the
// instance's scalars (the line search normalises with it, dual_qp.c:1305,1352).
struct ForwardRow {     // row tc of a block row: diagonal block, sub-diagonal block, right-hand side, pivot
    real_t Ld[NX], Lr[NX], sum, d;
};

NMPC_DEV void forward_pass(const BatchView& v, const Inst& I, bool live, int t) {
    const int N = v.N;
    const bool col = t < NX;
    const int tc = col ? t : 0;
    real_t norm2 = 0.;
    // Row tc of both blocks, twelve values each: past the diagonal the values are
    // those of the rows stored behind it, and they only reach sums whose result
    // has been taken already (a row is finished at its own column).
    const int diagRow = chol_diag_row(tc);
    auto fetch_row = [&](int kk, ForwardRow& r) {
        const real_t* row = I.chol + (size_t)kk * kCholRow;
        const pair_a* pd = (const pair_a*)(row + diagRow);
        const pair_a* pr = (const pair_a*)(row + NX * tc);
#pragma unroll
        for (int j = 0; j < NX / 2; j++) {
            const pair_a d = pd[j], s = pr[j];
            r.Ld[2 * j] = d.x; r.Ld[2 * j + 1] = d.y;
            r.Lr[2 * j] = s.x; r.Lr[2 * j + 1] = s.y;
        }
        r.sum = I.tmp[kk * NX + tc];
        r.d = row[diagRow + tc];
    };
    // Block k from `cur` while block k + 1 is on its way into `nxt`; x of the block
    // above in xin, x of this block into xout.  Two copies of the body take turns on
    // the two sets, so nothing is moved from one to the other.
    auto block = [&](int k, ForwardRow& cur, ForwardRow& nxt, const real_t (&xin)[NX], real_t (&xout)[NX]) {
        if (k + 1 < N && live) fetch_row(k + 1, nxt);
        real_t sum = cur.sum;
        if (k > 0) {
#pragma unroll
            for (int j = 0; j < NX; ++j) sum -= cur.Lr[j] * xin[j];
        }
        real_t mine = 0.;
        const Pivot pv{cur.d, 1.0 / cur.d};
        if (__any(plain_division_needed(cur.d)))
            forward_columns<false>(xout, mine, sum, cur.Ld, pv, t, Seq12{});
        else
            forward_columns<true>(xout, mine, sum, cur.Ld, pv, t, Seq12{});
        if (live && col) I.dl[k * NX + t] = mine;
#pragma unroll
        for (int j = 0; j < NX; j++) norm2 += xout[j] * xout[j];
    };
    ForwardRow a, b;
    real_t xa[NX], xb[NX];
#pragma unroll
    for (int j = 0; j < NX; j++) { a.Ld[j] = 0.; a.Lr[j] = 0.; b.Ld[j] = 0.; b.Lr[j] = 0.; xa[j] = 0.; xb[j] = 0.; }
    a.sum = b.sum = 0.;
    a.d = b.d = 1.;
    if (live) fetch_row(0, a);
    for (int k = 0; k < N; k += 2) {
        block(k, a, b, xb, xa);
        if (k + 1 < N) block(k + 1, b, a, xa, xb);
    }
    if (live && t == 0) I.sc->dlNorm2 = norm2;
}


#ifdef NMPC_SOLVE_STATS
__device__ unsigned long long g_team_stats[16];   // per wave of a team: cycles, cycles waited, looks, sweeps
#endif
// ------------------------------------------------------------------------
// Team form of the bottom-up sweep: two wavefronts on two SIMDs of a CU work on
// the same four instances.  Wave 0 factorises block after block exactly as
// backward_pass does (the same column routine) and hands the finished block row
// over through LDS; wave 1, one block behind, does the bottom-up substitution
// (dual_qp.c:1003-1026), which needs nothing but the finished rows, and stores
// the factor.  That takes the substitution's 330 instructions and the 24 stores
// out of the 1 800 a lone wavefront issues per factorised block, one every five
// or six cycles; where SIMDs are idle anyway (few instances with long sweeps, a
// small batch) the second wavefront is free.
//
// Lane i of both wavefronts works on the same (instance, column) pair, so a value
// passes through one LDS word per lane.  No workgroup barrier inside the sweep (one
// stops the wavefront for 40 to 130 cycles, tools/micro_barrier.hip): wave 0 writes
// a block row and then a counter, the LDS carries out a wavefront's accesses in
// order, and wave 1, finding the counter far enough, reads the rows behind it and
// says so in a counter of its own, which wave 0 looks at before it reuses the
// buffer (two block rows, by block parity).
//
// What the team does not handle itself -- a failed pivot, a pivot root whose
// significand is all ones -- makes it stop (`bail`) before the block is handed
// over; the caller then repeats the sweep with backward_pass, which is possible
// because everything the team has stored by then is what that sweep stores again.
namespace team {

constexpr int kWaves = 2;
constexpr int kMaxInstances = 4 * 512;      // two teams per CU; beyond, the one-wavefront form has the higher throughput

typedef int flags4 __attribute__((ext_vector_type(4)));

// Top-down sweep: wave 1 fetches the rows of the blocks ahead and hands them over
// through a ring of kRing slots, thirteen 16-byte words per lane: six pairs of the
// lane's row of the diagonal block, six of the sub-diagonal block, {right-hand side, pivot}.
constexpr int kRing = 4;
constexpr int kRowWords = NX + 1;

struct Shared {
    pair_a row[2][NX][kWave];       // by block parity: row l of the block row, {chol(k,0,l,t), chol(k,-1,l,t)} of the lane's column t
    pair_a ring[kRing][kRowWords][kWave];
    real_t minDiag[kWave];
    // [0] blocks handed over by wave 0, [1] blocks taken by wave 1, [3] the sweep is given up
    __attribute__((aligned(16))) int count[4];
    // top-down sweep: [0] blocks handed over by wave 1, [1] blocks taken by wave 0,
    // [2] wave 0's word to wave 1 after the bottom-up sweeps: 1 feed the top-down sweep, 2 nothing to do
    __attribute__((aligned(16))) int fed[4];
#ifdef NMPC_SOLVE_STATS
    long long waited[kWaves], looked[kWaves];     // per wavefront: cycles and looks spent in waits
#endif
};
#ifdef NMPC_SOLVE_STATS
struct WaitClock {
    Shared& sh;
    long long t0;
    int n = 0;
    __device__ explicit WaitClock(Shared& s) : sh(s), t0(clock64()) {}
    __device__ ~WaitClock() {
        if ((threadIdx.x & (kWave - 1)) == 0) {
            sh.waited[threadIdx.x >> 6] += clock64() - t0;
            sh.looked[threadIdx.x >> 6] += n;
        }
    }
};
#define NMPC_WAIT_CLOCK(sh) WaitClock wait_clock_(sh)
#define NMPC_WAIT_LOOK() (++wait_clock_.n)
#else
#define NMPC_WAIT_CLOCK(sh)
#define NMPC_WAIT_LOOK()
#endif

// Counters are read and written as LDS accesses proper (a volatile access through
// a generic pointer would go the flat way and wait for every load in flight).
#define NMPC_LDS __attribute__((address_space(3)))
NMPC_DEV flags4 peek(const int* group) { return *(const volatile NMPC_LDS flags4*)group; }
// After the data it stands for (the compiler keeps the order, the LDS does the rest).
NMPC_DEV void post(int* counter, int value) {
    asm volatile("" ::: "memory");
    *(volatile NMPC_LDS int*)counter = value;
    asm volatile("" ::: "memory");
}
// Until count[WHICH] >= value; false when the sweep is given up.  No wait goes on
// for ever: after kPatience looks (a tenth of a second, where the counter normally
// is there at the first) the sweep is given up from here.
constexpr int kPatience = 1 << 21;
template <int WHICH>
NMPC_DEV int await_fed(Shared& sh, int value) {     // -> the counter as seen (>= value), or -1 after kPatience looks
    NMPC_WAIT_CLOCK(sh);
    for (int looks = 0;;) {
        NMPC_WAIT_LOOK();
        const int f = peek(sh.fed)[WHICH];
        if (f >= value) { asm volatile("" ::: "memory"); return f; }
        if (++looks >= kPatience) return -1;
    }
}
template <int WHICH>
NMPC_DEV bool await(Shared& sh, int value) {
    NMPC_WAIT_CLOCK(sh);
    for (int looks = 0;;) {
        NMPC_WAIT_LOOK();
        const flags4 f = peek(sh.count);
        if (f[3] != 0) return false;
        if (f[WHICH] >= value) { asm volatile("" ::: "memory"); return true; }
        if (++looks >= kPatience) {
            post(&sh.count[3], 2);
            return false;
        }
    }
}

// Wave 0: backward_pass without the substitution and the stores; block rows are
// fetched one block ahead into registers.
NMPC_DEV void factor_sweep(Shared& sh, const BatchView& v, const Inst& I, int start, bool active, int lane) {
    const int N = v.N;
    const int t = lane & 15;
    const bool col = t < NX;
    const int tc = col ? t : 0;
    real_t Ln[NX], nL[NX], nLs[NX];
#pragma unroll
    for (int j = 0; j < NX; j++) { Ln[j] = 0.; nL[j] = 0.; nLs[j] = 0.; }
    auto fetch_block_row = [&](bool fromH, int kk) {
        if (!active) return;
        const real_t* row = (fromH ? I.hess : I.chol) + (size_t)kk * kCholRow + tc;
#pragma unroll
        for (int l = 0; l < NX; l++) {
            nL[l] = row[kCholDiagRow[l]];
            nLs[l] = row[NX * l];
        }
    };
    bool nextFromHess = __any(active && N - 1 <= start);
    fetch_block_row(nextFromHess, N - 1);
    for (int k = N - 1; k >= 0; --k) {
        const int i = N - 1 - k;
        const bool fromHess = nextFromHess;
        real_t L[NX], Ls[NX];
#pragma unroll
        for (int l = 0; l < NX; l++) { L[l] = nL[l]; Ls[l] = nLs[l]; }
        if (k > 0) {
            nextFromHess = __any(active && k - 1 <= start);
            fetch_block_row(nextFromHess, k - 1);
        }
        if (fromHess) {
            int failCol = -1;
            uint32_t ones = 0;
            real_t dd = 1.;
            // (single precision takes roots and quotients plainly, as backward_pass does)
            factor_columns<kDoublePrecision>(L, Ls, Ln, t, failCol, dd, ones, Seq12{});
            bool bad = active && ((k <= start && failCol >= 0) || (kDoublePrecision && ones == 0xffffffffu));
#ifdef NMPC_TEST_ALWAYS_REPEAT_BLOCKS     // test builds: give up half-way through every sweep
            bad = bad || k == N / 2;
#endif
            if (__any(bad)) {           // before the block is handed over: wave 1 stores nothing of it
                post(&sh.count[3], 1);
                return;
            }
        }
        // wave 1 has taken the block row before last out of this buffer
        if (i >= 2 && !await<1>(sh, i - 1)) return;
#pragma unroll
        for (int l = 0; l < NX; l++) {
            pair_a out;
            out.x = L[l];
            out.y = Ls[l];
            sh.row[k & 1][l][lane] = out;
        }
        post(&sh.count[0], i + 1);
#pragma unroll
        for (int j = 0; j < NX; j++) Ln[j] = Ls[j];
    }
}

// Wave 1: the bottom-up substitution of block m once its rows are there, and the
// stores of the rows that were factorised.
NMPC_DEV void substitution_sweep(Shared& sh, const BatchView& v, const Inst& I, int start, bool active,
                                 int lane, real_t& minDiag) {
    const int N = v.N;
    const int t = lane & 15;
    const bool col = t < NX;
    const int tc = col ? t : 0;
    real_t L[NX], Ls[NX], xk[NX], pn[NX];
#pragma unroll
    for (int l = 0; l < NX; l++) { xk[l] = 0.; pn[l] = 0.; }
    real_t ng = active ? I.grad[(N - 1) * NX + tc] : 0.;
    for (int i = 0; i < N; ++i) {
        const int m = N - 1 - i;
        const real_t gm = ng;
        if (m > 0 && active) ng = I.grad[(m - 1) * NX + tc];
        if (!await<0>(sh, i + 1)) return;
#pragma unroll
        for (int l = 0; l < NX; l++) {
            const pair_a in = sh.row[m & 1][l][lane];
            L[l] = in.x;
            Ls[l] = in.y;
        }
        const real_t own = sh.row[m & 1][tc][lane].x;       // chol(m,0,t,t)
        post(&sh.count[1], i + 1);
        const real_t dd = active ? own : 1.;
        if (col && minDiag > dd) minDiag = dd;
        const Pivot pv{dd, 1.0 / dd};
        real_t mine = 0.;
        if (__any(plain_division_needed(dd)))
            backsolve_rows<false>(xk, mine, L, pn, gm, pv, t, Seq12{});
        else
            backsolve_rows<true>(xk, mine, L, pn, gm, pv, t, Seq12{});
        if (active && col) {
            // into scratch: the step itself is only written once the factor is known to be good
            I.tmp[m * NX + t] = mine;
            if (m <= start) {
                real_t* row = I.chol + (size_t)m * kCholRow;
#pragma unroll
                for (int l = 0; l < NX; l++)
                    if (l >= t) row[kCholDiagRow[l] + t] = L[l];
                if (m > 0) {
#pragma unroll
                    for (int l = 0; l < NX; l++) row[NX * l + t] = Ls[l];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NX; j++) pn[j] = Ls[j] * xk[j];
    }
}

// Wave 1 during the top-down sweep: the rows of block k (forward_pass's fetch_row)
// from global memory into slot k % kRing, two blocks in flight.
NMPC_DEV void forward_feed(Shared& sh, const BatchView& v, const Inst& I, bool live, int lane) {
    const int N = v.N;
    const int t = lane & 15;
    const int tc = t < NX ? t : 0;
    const int diagRow = chol_diag_row(tc);
    struct Row { pair_a w[kRowWords]; };
    auto fetch = [&](int kk, Row& r) {
        if (!live || kk >= N) return;
        const real_t* row = I.chol + (size_t)kk * kCholRow;
        const pair_a* pd = (const pair_a*)(row + diagRow);
        const pair_a* pr = (const pair_a*)(row + NX * tc);
#pragma unroll
        for (int j = 0; j < NX / 2; j++) { r.w[j] = pd[j]; r.w[NX / 2 + j] = pr[j]; }
        r.w[NX].x = I.tmp[kk * NX + tc];
        r.w[NX].y = row[diagRow + tc];
    };
    auto hand_over = [&](int kk, const Row& r) -> bool {
        if (kk >= N) return true;
        if (kk >= kRing && await_fed<1>(sh, kk - kRing + 1) < 0) return false;     // the slot is free again
#pragma unroll
        for (int j = 0; j < kRowWords; j++) sh.ring[kk % kRing][j][lane] = r.w[j];
        post(&sh.fed[0], kk + 1);
        return true;
    };
    Row a, b;
#pragma unroll
    for (int j = 0; j < kRowWords; j++) { a.w[j].x = 0.; a.w[j].y = 0.; b.w[j] = a.w[j]; }
    a.w[NX].y = 1.; b.w[NX].y = 1.;
    fetch(0, a);
    fetch(1, b);
    for (int k = 0; k < N; k += 2) {
        if (!hand_over(k, a)) return;
        fetch(k + 2, a);
        if (!hand_over(k + 1, b)) return;
        fetch(k + 3, b);
    }
}

// forward_pass with the rows coming out of the ring wave 1 fills.
NMPC_DEV void forward_pass_fed(Shared& sh, const BatchView& v, const Inst& I, bool live, int t, int lane) {
    const int N = v.N;
    const bool col = t < NX;
    real_t norm2 = 0.;
    int seen = 0;                   // blocks known to be in the ring
    bool lost = false;
    auto fetch_row = [&](int kk, ForwardRow& r) {
        if (seen < kk + 1) {
            seen = await_fed<0>(sh, kk + 1);
            if (seen < 0) { lost = true; return; }
        }
        const pair_a (&slot)[kRowWords][kWave] = sh.ring[kk % kRing];
#pragma unroll
        for (int j = 0; j < NX / 2; j++) {
            const pair_a d = slot[j][lane], s = slot[NX / 2 + j][lane];
            r.Ld[2 * j] = d.x; r.Ld[2 * j + 1] = d.y;
            r.Lr[2 * j] = s.x; r.Lr[2 * j + 1] = s.y;
        }
        const pair_a e = slot[NX][lane];
        r.sum = e.x;
        r.d = e.y;
        post(&sh.fed[1], kk + 1);
    };
    auto block = [&](int k, ForwardRow& cur, ForwardRow& nxt, const real_t (&xin)[NX], real_t (&xout)[NX]) {
        if (k + 1 < N) fetch_row(k + 1, nxt);
        real_t sum = cur.sum;
        if (k > 0) {
#pragma unroll
            for (int j = 0; j < NX; ++j) sum -= cur.Lr[j] * xin[j];
        }
        real_t mine = 0.;
        const Pivot pv{cur.d, 1.0 / cur.d};
        if (__any(plain_division_needed(cur.d)))
            forward_columns<false>(xout, mine, sum, cur.Ld, pv, t, Seq12{});
        else
            forward_columns<true>(xout, mine, sum, cur.Ld, pv, t, Seq12{});
        if (live && col) I.dl[k * NX + t] = mine;
#pragma unroll
        for (int j = 0; j < NX; j++) norm2 += xout[j] * xout[j];
    };
    ForwardRow a, b;
    real_t xa[NX], xb[NX];
#pragma unroll
    for (int j = 0; j < NX; j++) { a.Ld[j] = 0.; a.Lr[j] = 0.; b.Ld[j] = 0.; b.Lr[j] = 0.; xa[j] = 0.; xb[j] = 0.; }
    a.sum = b.sum = 0.;
    a.d = b.d = 1.;
    fetch_row(0, a);
    for (int k = 0; k < N && !lost; k += 2) {
        block(k, a, b, xb, xa);
        if (k + 1 < N) block(k + 1, b, a, xa, xb);
    }
    if (lost) {                     // wave 1 did not deliver: the sweep again, on this wave's own
        forward_pass(v, I, live, t);
        return;
    }
    if (live && t == 0) I.sc->dlNorm2 = norm2;
}

// Returns true when the sweep was given up; minDiag as backward_pass leaves it.
NMPC_DEV bool sweep(Shared& sh, const BatchView& v, const Inst& I, int start, bool active, int role,
                    int lane, real_t& minDiag) {
    if (threadIdx.x < 4) { sh.count[threadIdx.x] = 0; sh.fed[threadIdx.x] = 0; }
#ifdef NMPC_SOLVE_STATS
    if (threadIdx.x < kWaves) { sh.waited[threadIdx.x] = 0; sh.looked[threadIdx.x] = 0; }
#endif
    __syncthreads();
#ifdef NMPC_SOLVE_STATS
    const long long sweep0 = clock64();
#endif
    if (role == 0) {
        factor_sweep(sh, v, I, start, active, lane);
    } else {
        substitution_sweep(sh, v, I, start, active, lane, minDiag);
        sh.minDiag[lane] = minDiag;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the factor and the scratch are stored
    }
#ifdef NMPC_SOLVE_STATS
    if (lane == 0) {
        atomicAdd(&g_team_stats[role * 4 + 0], (unsigned long long)(clock64() - sweep0));
        atomicAdd(&g_team_stats[role * 4 + 1], (unsigned long long)sh.waited[role]);
        atomicAdd(&g_team_stats[role * 4 + 2], (unsigned long long)sh.looked[role]);
        atomicAdd(&g_team_stats[role * 4 + 3], 1ull);
    }
#endif
    __syncthreads();
    if (role == 0) minDiag = sh.minDiag[lane];
    return sh.count[3] != 0;
}

}  // namespace team

}  // namespace

// dual_qp.c:601-698 followed by :993-1055
// WAVES = 1: one wavefront per SIMD, block rows prefetched into registers (more
// than 256 of them).  WAVES = 2: two per SIMD, block rows staged through LDS.
// With a list in the view (k_list_fill has just made it) the groups take only
// instances whose solve is still running, those with the longest factorisation
// sweep first, so that the four groups of a wavefront have sweeps of about the
// same length and wavefronts without work end at once.  An instance's results
// do not depend on which instances share its wavefront.
#ifdef NMPC_SOLVE_STATS
// Measurement build (tools/solve_stats.py): cycles of the two sweeps summed over
// wavefronts; [0..7] launches over a whole batch or the long part, [8..15] short parts.
__device__ unsigned long long g_solve_stats[16];
#endif

template <int WAVES>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
void k_newton_solve(BatchView v) {
#ifdef NMPC_SOLVE_STATS
    const long long c0 = clock64();
#endif
    const int lane = threadIdx.x, g = lane >> 4, t = lane & 15;
    const int first = blockIdx.x * kGroupsPerWave;
    const int b0 = launch_instance(v, first);
    if (b0 < 0) return;
    // One dependent chain per wavefront: when wavefronts of other kernels share the
    // SIMD (the other part of a two-part iteration), this one issues first.
    if (WAVES == 1) __builtin_amdgcn_s_setprio(3);
    const int bg = launch_instance(v, first + g);
    const bool have = bg >= 0;
    const int b = have ? bg : b0;
    Inst I = instance(v, b);
    bool live = have && (I.sc->status == ST_RUNNING);
    if (!__any(live)) return;
    const int N = v.N;
    const int last = I.sc->last;
    const int start = (live && last >= 0) ? (last < N - 1 ? last : N - 1) : -1;

    // First sweep from each group's `start`; groups whose factor fails or has a
    // small pivot regularise and take a second sweep over all blocks
    // (dual_qp.c:601-698).  One loop, so that the sweep's code exists once.
    bool sweep = live;
    int from = start;
    for (int pass = 0; pass < 2; ++pass) {
        real_t minDiag = kInfty;
        const int st = backward_pass<WAVES == 2>(v, I, from, sweep, t, minDiag);
        if (pass == 0) {
            minDiag = group_min(minDiag);
            const bool needReg = live && (st == ST_DIV_ZERO || minDiag < kDiagRegTol);
            if (!__any(needReg)) break;
            if (needReg && t < NX)
                for (int k = 0; k < N; k++) I.hess[(size_t)k * kCholRow + chol_diag_row(t) + t] += kRegParam;
            if (needReg && t == 0) I.sc->regularized = 1;
            sweep = needReg;
            from = N - 1;
        } else if (sweep && st != ST_OK) {
            if (t == 0) I.sc->status = st;
            live = false;
        }
    }
    if (!__any(live)) return;
#ifdef NMPC_SOLVE_STATS
    const long long c1 = clock64();
#endif
    forward_pass(v, I, live, t);
#ifdef NMPC_SOLVE_STATS
    const long long c2 = clock64();
    if (threadIdx.x == 0) {
        const int part = (v.list && v.listRange[0] > 0) ? 8 : 0;
        atomicAdd(&g_solve_stats[part + 0], 1ull);
        atomicAdd(&g_solve_stats[part + 1], (unsigned long long)(c1 - c0));
        atomicAdd(&g_solve_stats[part + 2], (unsigned long long)(c2 - c1));
        atomicAdd(&g_solve_stats[part + 3], (unsigned long long)(wave_max(start) + 1));
        atomicMax(&g_solve_stats[part + 4], (unsigned long long)(c1 - c0));
        atomicMax(&g_solve_stats[part + 5], (unsigned long long)(c2 - c1));
    }
#endif
}

// The same solve with the first bottom-up sweep run by a team of two wavefronts
// (namespace team); regularised second sweeps and sweeps the team gives up are
// wave 0's alone, through the code of k_newton_solve; in the top-down
// substitution wave 1 fetches the rows ahead of wave 0.  A workgroup takes
// every gridDim.x-th set of four instances, so a grid of two teams per CU serves
// a list of any length (only the device knows it).
NMPC_DEV void team_solve(team::Shared& sh, const BatchView& v, int first, int role, int lane) {
#ifdef NMPC_SOLVE_STATS
    const long long c0 = clock64();
#endif
    const int g = lane >> 4, t = lane & 15;
    const int b0 = launch_instance(v, first);
    if (b0 < 0) return;
    const int bg = launch_instance(v, first + g);
    const bool have = bg >= 0;
    const int b = have ? bg : b0;
    Inst I = instance(v, b);
    bool live = have && (I.sc->status == ST_RUNNING);
    if (!__any(live)) return;
    const int N = v.N;
    const int last = I.sc->last;
    const int start = (live && last >= 0) ? (last < N - 1 ? last : N - 1) : -1;

    real_t minDiag = kInfty;
    const bool gaveUp = team::sweep(sh, v, I, start, live, role, lane, minDiag);
    if (role != 0) {
        // wave 0 decides what follows the bottom-up sweeps; then this wave feeds its top-down sweep
        if (team::await_fed<2>(sh, 1) == 1) team::forward_feed(sh, v, I, live, lane);
        return;
    }
    int st = ST_OK;
    if (gaveUp) {
        minDiag = kInfty;
        st = backward_pass<false>(v, I, start, live, t, minDiag);
    }
    // regularisation and second sweep as in k_newton_solve (dual_qp.c:601-698)
    minDiag = group_min(minDiag);
    const bool needReg = live && (st == ST_DIV_ZERO || minDiag < kDiagRegTol);
    if (__any(needReg)) {
        if (needReg && t < NX)
            for (int k = 0; k < N; k++) I.hess[(size_t)k * kCholRow + chol_diag_row(t) + t] += kRegParam;
        if (needReg && t == 0) I.sc->regularized = 1;
        real_t unused = kInfty;
        st = backward_pass<false>(v, I, N - 1, needReg, t, unused);
        if (needReg && st != ST_OK) {
            if (t == 0) I.sc->status = st;
            live = false;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // a second sweep's stores are done before wave 1 reads
    if (!__any(live)) {
        team::post(&sh.fed[2], 2);
        return;
    }
    team::post(&sh.fed[2], 1);
#ifdef NMPC_SOLVE_STATS
    const long long c1 = clock64();
#endif
    team::forward_pass_fed(sh, v, I, live, t, lane);
#ifdef NMPC_SOLVE_STATS
    const long long c2 = clock64();
    if (lane == 0) {
        const int part = (v.list && v.listRange[0] > 0) ? 8 : 0;
        atomicAdd(&g_solve_stats[part + 0], 1ull);
        atomicAdd(&g_solve_stats[part + 1], (unsigned long long)(c1 - c0));
        atomicAdd(&g_solve_stats[part + 2], (unsigned long long)(c2 - c1));
        atomicAdd(&g_solve_stats[part + 3], (unsigned long long)(wave_max(start) + 1));
        atomicMax(&g_solve_stats[part + 4], (unsigned long long)(c1 - c0));
        atomicMax(&g_solve_stats[part + 5], (unsigned long long)(c2 - c1));
    }
#endif
}

__global__ __launch_bounds__(team::kWaves * kWave) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_newton_solve_team(BatchView v) {
    __shared__ team::Shared sh;
    const int role = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
    const int count = v.list ? v.listRange[1] - v.listRange[0] : v.nb;
    if (role == 0) __builtin_amdgcn_s_setprio(3);
    for (int first = blockIdx.x * kGroupsPerWave; first < count; first += gridDim.x * kGroupsPerWave) {
        team_solve(sh, v, first, role, lane);
        __syncthreads();            // both wavefronts are done with the counters before the next set resets them
    }
}

// ------------------------------------------------------------------------
// Instance list of a large grid's Newton solve: a counting sort of the running
// instances by the block their factorisation starts from (dual_qp.c:611-623:
// min(lastActSetChangeIdx, N - 1)), in two launches.  After the first Newton
// iteration most instances refactorise a few blocks at the top of the horizon
// and some the whole of it; a wavefront factorises what its longest group needs,
// so four neighbours in instance order cost about three times the blocks that
// four neighbours in this order do.
constexpr int kListThreads = 256;

NMPC_DEV int solve_list_bucket(const BatchView& v, const Scalars& s) {
    if (s.status != ST_RUNNING) return -1;
    const int start = s.last < 0 ? 0 : (s.last < v.N - 1 ? s.last : v.N - 1);
    const int bucket = v.N <= kSolveListBuckets ? start : (int)((long)start * kSolveListBuckets / v.N);
    return kSolveListBuckets - 1 - bucket;          // longest sweeps first
}

__global__ __launch_bounds__(kListThreads) void k_list_count(BatchView v, int* __restrict__ ctr) {
    __shared__ int hist[kSolveListBuckets];
    const int tid = threadIdx.x;
    if (tid < kSolveListBuckets) hist[tid] = 0;
    __syncthreads();
    const int i = blockIdx.x * kListThreads + tid;
    if (i < v.nb) {
        const int bucket = solve_list_bucket(v, v.sc[v.b0 + i]);
        if (bucket >= 0) atomicAdd(&hist[bucket], 1);
    }
    __syncthreads();
    if (tid < kSolveListBuckets && hist[tid] != 0) atomicAdd(&ctr[tid], hist[tid]);
}

// ctr: this launch's counters (histogram complete); spare: the other set, which
// the next list will count into and which nothing reads any more.
// longBuckets: the leading buckets whose instances count as long sweeps.
__global__ __launch_bounds__(kListThreads) void k_list_fill(BatchView v, int* __restrict__ ctr,
                                                            int* __restrict__ spare, int longBuckets) {
    static_assert(kSolveListBuckets == 2 * kWave, "two buckets per lane of the first wavefront");
    __shared__ int base[kSolveListBuckets + 1];
    __shared__ int mine[kSolveListBuckets];         // this workgroup's instances per bucket
    const int tid = threadIdx.x;
    if (tid < kSolveListBuckets) mine[tid] = 0;
    if (tid < kWave) {
        const int a = ctr[2 * tid], c = ctr[2 * tid + 1];
        int sum = a + c;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const int y = __shfl_up(sum, o);
            if (tid >= o) sum += y;
        }
        base[2 * tid] = sum - a - c;
        base[2 * tid + 1] = sum - c;
        if (tid == kWave - 1) base[kSolveListBuckets] = sum;
    }
    if (blockIdx.x == 0)
        for (int j = tid; j < kSolveListStride; j += kListThreads) spare[j] = 0;
    __syncthreads();
    if (blockIdx.x == 0 && tid == 0) {
        const int total = base[kSolveListBuckets], nLong = base[longBuckets];
        ctr[kSolveRangeAll] = 0;       ctr[kSolveRangeAll + 1] = total;
        ctr[kSolveRangeLong] = 0;      ctr[kSolveRangeLong + 1] = nLong;
        ctr[kSolveRangeShort] = nLong; ctr[kSolveRangeShort + 1] = total;
    }
    // one cursor update per workgroup and bucket (a cold solve has every instance in one bucket)
    const int i = blockIdx.x * kListThreads + tid;
    const int bucket = i < v.nb ? solve_list_bucket(v, v.sc[v.b0 + i]) : -1;
    const int rank = bucket >= 0 ? atomicAdd(&mine[bucket], 1) : 0;
    __syncthreads();
    if (tid < kSolveListBuckets && mine[tid] != 0)
        base[tid] += atomicAdd(&ctr[kSolveListBuckets + tid], mine[tid]);
    __syncthreads();
    if (bucket >= 0) v.solveList[base[bucket] + rank] = v.b0 + i;
}

// Two wavefronts per SIMD only pay once there are wavefronts for it: the
// dispatcher packs the workgroups of a grid onto CUs as densely as the
// kernel's resources allow, so a grid of one wavefront per SIMD (1024 on this
// chip) launched with the two-per-SIMD kernel leaves half of the SIMDs idle
// (0.77 ms instead of 0.55 ms per launch at 4096 instances).
bool newton_solve_two_waves(const BatchView& v) {
    // (the staged form's LDS image is laid out for 8-byte entries)
#ifdef NMPC_TEST_ALWAYS_REPEAT_BLOCKS     // test builds take the staged form at every size
    return kDoublePrecision;
#else
    const unsigned blocks = (unsigned)((v.nb + kGroupsPerWave - 1) / kGroupsPerWave);
    return kDoublePrecision && blocks > kOneWavePerSimd * 3 / 2;
#endif
}

// epoch: a count of the lists made for this batch; it picks the counter set.
void launch_list_solves(const BatchView& v, unsigned epoch, hipStream_t s) {
    int* ctr = v.solveCount + (epoch & 1) * kSolveListStride;
    int* spare = v.solveCount + ((epoch + 1) & 1) * kSolveListStride;
    const unsigned blocks = (unsigned)((v.nb + kListThreads - 1) / kListThreads);
    // long sweeps: those that start in the lower 85 % of the horizon's blocks (with the long
    // part's solve run by teams the two parts of an iteration end together there: 5.16 ms per
    // cold solve of 4 096 instances at 0.75, 5.14 at 0.85 and 0.92, 5.26 at 0.6)
    const int nBuckets = v.N < kSolveListBuckets ? v.N : kSolveListBuckets;
    static const double longFrac = getenv("NMPC_LONG_FRAC") ? atof(getenv("NMPC_LONG_FRAC")) : 0.85;
    const int longBuckets = kSolveListBuckets - nBuckets + (int)(longFrac * nBuckets + 0.75);
    hipLaunchKernelGGL(k_list_count, dim3(blocks), dim3(kListThreads), 0, s, v, ctr);
    hipLaunchKernelGGL(k_list_fill, dim3(blocks), dim3(kListThreads), 0, s, v, ctr, spare, longBuckets);
}

BatchView solve_list_view(const BatchView& v, unsigned epoch, int part) {
    BatchView w = v;
    const int* ctr = v.solveCount + (epoch & 1) * kSolveListStride;
    w.list = v.solveList;
    w.listRange = ctr + (part == 1 ? kSolveRangeLong : part == 2 ? kSolveRangeShort : kSolveRangeAll);
    w.listPart = part;
    if (part == 2 && v.tailCount) {
        w.tailCount = v.tailCount + 16;
        w.tailList = v.tailList + 2 * (size_t)v.B;
    }
    return w;
}

// The team form pays where it leaves no SIMD short of work: an unlisted launch of at
// most two teams per CU, or the long-sweep part of a two-part iteration.  The
// length of that part only the device knows; a grid of two teams per CU takes
// whatever it is, in rounds if need be (with the bench workload's sixth of the
// instances one round does up to 12 000 instances).  NMPC_NO_TEAM=1 keeps the
// one-wavefront form.
// In single precision every search that goes on runs in k_line_search_tail, whose
// workgroups of sixteen wavefronts find no room on a CU that holds a team (one
// wavefront per SIMD, most of its registers): on warm control steps, with thousands of
// searches in the other part, the teams cost more than they gain (4 096 x 100: cold
// solve 5.23 -> 4.88 ms, warm step 5.59 -> 6.10 ms), hence few_searches.
static bool team_wanted(const BatchView& v, bool few_searches) {
    static const bool off = getenv("NMPC_NO_TEAM") != nullptr;
    if (off || newton_solve_two_waves(v)) return false;
    if (v.list) return v.listPart == 1 && (kDoublePrecision || few_searches);
    return v.nb <= team::kMaxInstances;
}

void launch_newton_solve(const BatchView& v, hipStream_t s, bool few_searches) {
    const unsigned blocks = (unsigned)((v.nb + kGroupsPerWave - 1) / kGroupsPerWave);
    if (team_wanted(v, few_searches)) {
        const unsigned most = team::kMaxInstances / kGroupsPerWave;
        hipLaunchKernelGGL(k_newton_solve_team, dim3(blocks < most ? blocks : most), dim3(team::kWaves * kWave), 0, s, v);
        return;
    }
    if (newton_solve_two_waves(v)) hipLaunchKernelGGL(k_newton_solve<2>, dim3(blocks), dim3(kWave), 0, s, v);
    else hipLaunchKernelGGL(k_newton_solve<1>, dim3(blocks), dim3(kWave), 0, s, v);
}

#ifdef NMPC_SOLVE_STATS
extern "C" int nmpc_debug_team_stats(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_team_stats), sizeof(unsigned long long) * 16) != hipSuccess)
        return -1;
    if (reset) {
        unsigned long long zero[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_team_stats), zero, sizeof zero) != hipSuccess) return -1;
    }
    return 0;
}
extern "C" int nmpc_debug_solve_stats(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_solve_stats), sizeof(unsigned long long) * 16) != hipSuccess)
        return -1;
    if (reset) {
        unsigned long long zero[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_solve_stats), zero, sizeof zero) != hipSuccess) return -1;
    }
    return 0;
}
#endif

}  // namespace nmpc
