// Newton-system half of the dual Newton solve: one 64-lane wavefront
// (= one workgroup) per controller instance.
//
//   k_newton_setup    dual gradient, convergence test, Newton Hessian blocks
//                     (ccs-c66x/qpDUNES/dual_qp.c:423-597)
//   k_newton_solve    reverse block-tridiagonal Cholesky, regularisation,
//                     two substitutions (dual_qp.c:601-698, 817-908, 993-1055)
//
// All solver control flow is per instance, hence wave-uniform.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

#include "kernels.hpp"
#include "qp_common.hpp"

#pragma clang fp contract(off)

namespace nmpc {

namespace {

constexpr int kChunk = 512;          // doubles of LDS used by ordered sums

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

}  // namespace

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

namespace {

#define HS(M, k, l, i, j) (M)[(size_t)(k) * 2 * NX * NX + (i) * 2 * NX + (1 + (l)) * NX + (j)]

// ------------------------------------------------------------------------
// k_newton_solve works in 16-lane groups, four controller instances per
// wavefront.  Lane t < 12 of a group owns column t of the current diagonal
// block, of the current sub-diagonal block and of the sub-diagonal block below.
// Values of another column of the same instance come from a DPP row broadcast
// (the 16 lanes of a group are one DPP row); nothing goes through LDS.
// Every lane of the wavefront stays active throughout, results of groups that
// have nothing to do are computed and dropped.
constexpr int kGroupsPerWave = 4;

template <int SRC>
NMPC_DEV double row_bcast(double x) {   // value held by lane SRC of this lane's group
    return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + SRC, 0xf, 0xf, false);
}

NMPC_DEV double group_min(double x) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        const double y = __shfl_xor(x, o, 16);
        x = y < x ? y : x;
    }
    return x;
}

// Column JJ of block row k of the reverse Cholesky (dual_qp.c:832-903): pivot,
// the entries left of it in the diagonal block and the whole row of the
// sub-diagonal block.  L / Ls / Ln hold this lane's column of chol(k,0),
// chol(k,-1) and chol(k+1,-1); on entry L[JJ] and Ls[JJ] still hold the Newton
// Hessian entries they are computed from.  Lanes with doit == false keep
// their L[JJ], Ls[JJ].
template <int JJ>
NMPC_DEV void factor_column(double (&L)[NX], double (&Ls)[NX], const double (&Ln)[NX],
                            bool below, bool doit, int t, int& failCol, double& diag) {
    double sum = L[JJ], sub = Ls[JJ];
#pragma unroll
    for (int l = JJ + 1; l < NX; ++l) {
        const double ljj = row_bcast<JJ>(L[l]);         // chol(k,0,l,JJ)
        sum -= L[l] * ljj;
        sub -= Ls[l] * ljj;
    }
    if (below) {
#pragma unroll
        for (int l = 0; l < NX; ++l) {
            const double njj = row_bcast<JJ>(Ln[l]);    // chol(k+1,-1,l,JJ)
            sum -= Ln[l] * njj;
        }
    }
    const double piv = row_bcast<JJ>(sum);
    if (doit && failCol < 0 && piv < 1.e2 * kEqTol) failCol = JJ;
    const double d = sqrt(piv);
    diag = (doit && t == JJ) ? d : diag;
    const double ld = (t == JJ) ? d : sum / d;
    const double ls = sub / d;
    L[JJ] = doit ? ld : L[JJ];
    Ls[JJ] = doit ? ls : Ls[JJ];
}

template <int... JJ>
NMPC_DEV void factor_columns(double (&L)[NX], double (&Ls)[NX], const double (&Ln)[NX],
                             bool below, bool doit, int t, int& failCol, double& diag,
                             std::integer_sequence<int, JJ...>) {
    (factor_column<NX - 1 - JJ>(L, Ls, Ln, below, doit, t, failCol, diag), ...);
}

// Row R of the bottom-up substitution L' x = g for block k (dual_qp.c:1003-1026).
template <int R>
NMPC_DEV void backsolve_row(double (&xk)[NX], double& mine, const double (&L)[NX],
                            const double (&pn)[NX], double g, double dd, bool below, int t) {
    double sum = g;
#pragma unroll
    for (int j = R + 1; j < NX; ++j) sum -= L[j] * xk[j];
    if (below) {
#pragma unroll
        for (int j = 0; j < NX; ++j) sum -= pn[j];
    }
    const double x = sum / dd;
    xk[R] = row_bcast<R>(x);
    mine = (t == R) ? x : mine;
}

template <int... R>
NMPC_DEV void backsolve_rows(double (&xk)[NX], double& mine, const double (&L)[NX],
                             const double (&pn)[NX], double g, double dd, bool below, int t,
                             std::integer_sequence<int, R...>) {
    (backsolve_row<NX - 1 - R>(xk, mine, L, pn, g, dd, below, t), ...);
}

// Column J of the top-down substitution L x = y for block k
// (dual_qp.c:1029-1052), column-oriented: once x_J is final every later row
// subtracts its L(i,J) x_J, which keeps each row's terms in ascending J order.
template <int J>
NMPC_DEV void forward_column(double (&xk)[NX], double& mine, double& sum,
                             const double (&Ld)[NX], double dd, int t) {
    const double x = sum / dd;
    const double xj = row_bcast<J>(x);
    xk[J] = xj;
    mine = (t == J) ? x : mine;
    sum -= Ld[J] * xj;
}

template <int... J>
NMPC_DEV void forward_columns(double (&xk)[NX], double& mine, double& sum,
                              const double (&Ld)[NX], double dd, int t,
                              std::integer_sequence<int, J...>) {
    (forward_column<J>(xk, mine, sum, Ld, dd, t), ...);
}

using Seq12 = std::make_integer_sequence<int, NX>;

// One sweep from block N-1 down to 0: blocks k <= start of an `active` group
// are factorised (dual_qp.c:817-908), the others read their factor back; each
// block then takes its step of the bottom-up substitution, which only needs
// the factor rows just produced.  Returns ST_OK or ST_DIV_ZERO per group and
// lowers minDiag to the smallest diagonal factor entry seen.
NMPC_DEV int backward_pass(const BatchView& v, const Inst& I, int start, bool active, int t,
                           double& minDiag) {
    const int N = v.N;
    const bool col = t < NX;
    const int tc = col ? t : 0;
    int status = ST_OK;
    double Ln[NX], pn[NX], xk[NX];      // pn[j] = chol(k+1,-1,j,t) * x_{k+1}[j]
#pragma unroll
    for (int j = 0; j < NX; j++) { Ln[j] = 0.; pn[j] = 0.; xk[j] = 0.; }

    // Block rows are fetched one iteration ahead.  Lanes that will factorise a
    // block start from the Newton Hessian, the others read their factor back.
    double nL[NX], nLs[NX], ng, nd;
    {
        const bool f = active && (N - 1 <= start);
        const double* src = f ? I.hess : I.chol;
#pragma unroll
        for (int l = 0; l < NX; l++) {
            nL[l] = HS(src, N - 1, 0, l, tc);
            nLs[l] = HS(src, N - 1, -1, l, tc);
        }
        ng = I.grad[(N - 1) * NX + tc];
        nd = HS(I.chol, N - 1, 0, tc, tc);
    }

    for (int k = N - 1; k >= 0; --k) {
        const bool below = k < N - 1;
        const bool factorThis = active && status == ST_OK && k <= start;
        double L[NX], Ls[NX];
#pragma unroll
        for (int l = 0; l < NX; l++) { L[l] = nL[l]; Ls[l] = nLs[l]; }
        const double g = ng;
        double dd = nd;                                // chol(k,0,t,t) unless refactorised below
        if (k > 0) {
            // a failure in this block is not known yet; a group that fails stops
            // using what it fetched, so the guess "still factorising" is harmless
            const bool f = active && status == ST_OK && (k - 1 <= start);
            const double* src = f ? I.hess : I.chol;
#pragma unroll
            for (int l = 0; l < NX; l++) {
                nL[l] = HS(src, k - 1, 0, l, tc);
                nLs[l] = HS(src, k - 1, -1, l, tc);
            }
            ng = I.grad[(k - 1) * NX + tc];
            nd = HS(I.chol, k - 1, 0, tc, tc);
        }
        if (__any(factorThis)) {
            int failCol = -1;
            factor_columns(L, Ls, Ln, below, factorThis, t, failCol, dd, Seq12{});
            if (factorThis && col) {
#pragma unroll
                for (int l = 0; l < NX; l++)
                    if (l >= t) HS(I.chol, k, 0, l, t) = L[l];
                if (k > 0) {
#pragma unroll
                    for (int l = 0; l < NX; l++) HS(I.chol, k, -1, l, t) = Ls[l];
                }
            }
            if (factorThis && failCol >= 0) status = ST_DIV_ZERO;
        }
        if (col && minDiag > dd) minDiag = dd;

        // bottom-up substitution, block k
        double mine = 0.;
        backsolve_rows(xk, mine, L, pn, g, dd, below, t, Seq12{});
        if (active && status == ST_OK && col) I.dl[k * NX + t] = mine;
#pragma unroll
        for (int j = 0; j < NX; j++) { pn[j] = Ls[j] * xk[j]; Ln[j] = Ls[j]; }
    }
    return status;
}

// Top-down substitution L x = y over all blocks (dual_qp.c:1029-1052); lane t
// owns row t of each block.
NMPC_DEV void forward_pass(const BatchView& v, const Inst& I, bool live, int t) {
    const int N = v.N;
    const bool col = t < NX;
    const int tc = col ? t : 0;
    double xk[NX], xo[NX];
#pragma unroll
    for (int j = 0; j < NX; j++) { xk[j] = 0.; xo[j] = 0.; }
    double nLd[NX], nLr[NX], nsum, nd;
#pragma unroll
    for (int j = 0; j < NX; j++) { nLd[j] = HS(I.chol, 0, 0, tc, j); nLr[j] = HS(I.chol, 0, -1, tc, j); }
    nsum = I.dl[tc];
    nd = HS(I.chol, 0, 0, tc, tc);
    for (int k = 0; k < N; ++k) {
        double Ld[NX], Lr[NX];
#pragma unroll
        for (int j = 0; j < NX; j++) { Ld[j] = nLd[j]; Lr[j] = nLr[j]; }
        double sum = nsum;
        const double dd = nd;
        if (k + 1 < N) {
#pragma unroll
            for (int j = 0; j < NX; j++) {
                nLd[j] = HS(I.chol, k + 1, 0, tc, j);
                nLr[j] = HS(I.chol, k + 1, -1, tc, j);
            }
            nsum = I.dl[(k + 1) * NX + tc];
            nd = HS(I.chol, k + 1, 0, tc, tc);
        }
        if (k > 0) {
#pragma unroll
            for (int j = 0; j < NX; ++j) sum -= Lr[j] * xo[j];
        }
        double mine = 0.;
        forward_columns(xk, mine, sum, Ld, dd, t, Seq12{});
        if (live && col) I.dl[k * NX + t] = mine;
#pragma unroll
        for (int j = 0; j < NX; j++) xo[j] = xk[j];
    }
}

}  // namespace

// dual_qp.c:601-698 followed by :993-1055
__global__ __launch_bounds__(kWave) void k_newton_solve(BatchView v) {
    const int lane = threadIdx.x, g = lane >> 4, t = lane & 15;
    const int bRaw = blockIdx.x * kGroupsPerWave + g;
    const int b = bRaw < v.B ? bRaw : v.B - 1;
    Inst I = instance(v, b);
    bool live = (bRaw < v.B) && (I.sc->status == ST_RUNNING);
    if (!__any(live)) return;
    const int N = v.N;
    const int last = I.sc->last;
    const int start = (live && last >= 0) ? (last < N - 1 ? last : N - 1) : -1;

    double minDiag = kInfty;
    int st = backward_pass(v, I, start, live, t, minDiag);
    minDiag = group_min(minDiag);
    const bool needReg = live && (st == ST_DIV_ZERO || minDiag < kDiagRegTol);
    if (__any(needReg)) {
        if (needReg && t < NX)
            for (int k = 0; k < N; k++) HS(I.hess, k, 0, t, t) += kRegParam;
        if (needReg && t == 0) I.sc->regularized = 1;
        double unused = kInfty;
        st = backward_pass(v, I, N - 1, needReg, t, unused);
        if (needReg && st != ST_OK) {
            if (t == 0) I.sc->status = st;
            live = false;
        }
    }
    if (!__any(live)) return;
    forward_pass(v, I, live, t);
}

void launch_newton_setup(const BatchView& v, int it, hipStream_t s) {
    hipLaunchKernelGGL(k_newton_setup, dim3(v.B), dim3(kWave), 0, s, v, it);
}
void launch_newton_solve(const BatchView& v, hipStream_t s) {
    const unsigned blocks = (unsigned)((v.B + kGroupsPerWave - 1) / kGroupsPerWave);
    hipLaunchKernelGGL(k_newton_solve, dim3(blocks), dim3(kWave), 0, s, v);
}

}  // namespace nmpc
