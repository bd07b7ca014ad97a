// Device-memory plan of one batch of NMPC instances (replaces the reference's
// per-process static_qpdata_t / static_interval_t, ccs-c66x/cnmpc.c:52-117, and
// qpDUNES' interval_t / qpData_t, ccs-c66x/qpDUNES/types.h:411-693).
//
// Field-major arrays, instance index slowest: X[b][stage][...].  The Hessians
// of the stage QPs are diagonal and identical for every regular stage, so they
// are two small vectors, not 15x15 matrices per stage.
//
// Horizon shifting moves no data.  Three ring offsets map a horizon position
// to a physical slot:
//   stage ring  (C, c, bounds, z, y, q, p, step data, uref): position k < N is
//               slot (head + k) % N; the terminal stage is always slot N.
//               This is the reference's interval-pointer rotation
//               (qpDUNES/setup_qp.c:619-642).
//   xref ring   N+1 slots, position k is slot (xhead + k) % (N + 1)
//               (the memmove at ccs-c66x/cnmpc.c:976-977).
//   lambda ring N slots, position k is slot (lhead + k) % N
//               (qpDUNES_shiftLambda, qpDUNES/setup_qp.c:650-662).
// The Newton system (hess, chol, grad, dl) and the active-set records are
// indexed by position and never shifted, exactly as in the reference.
#pragma once
#include <stdint.h>

// Arithmetic type of the whole library: double, or float when built with
// -DNMPC_SINGLE_PRECISION (the reference's shipped configuration,
// include/config.h:27-28).  Same typedef as include/cnmpc.h.
#ifdef NMPC_SINGLE_PRECISION
typedef float real_t;
#else
typedef double real_t;
#endif

namespace nmpc {

// Solver status codes: numeric values of qpDUNES' return_t
// (ccs-c66x/qpDUNES/types.h:211-231) plus RUNNING for an unfinished solve.
enum : int {
    ST_RUNNING = -1,
    ST_OK = 0,
    ST_OPTIMAL = 1,
    ST_ITER_LIMIT = 8,
    ST_DIV_ZERO = 9,
    ST_LS_MAXITER = 10,
    ST_LS_MIN_STEP = 11,
    ST_LS_MAX_STEP = 12,
    ST_NO_ASCENT = 13
};

struct Scalars {        // one per instance
    real_t objInc;      // incumbent dual objective
    real_t alpha;       // last step length
    real_t u[3];        // last good controls
    int status;         // ST_*
    int numIter;        // qpData.log.numIter
    int last;           // lastActSetChangeIdx
    int regularized;    // Newton Hessian regularised this iteration
    int lsIter;         // line-search trial count of the last iteration
    int lastResult;     // 1 = nmpc_get_controls would return NMPC_OK
    // step-length search handed from k_newton_step to k_line_search_tail
    real_t dlNorm2;     // squared norm of the Newton step, written by k_newton_solve
    real_t lsMin, lsMax, lsMid, lsNormal;
    int lsPhase;
    int lsDeferred;
};

// Block row of the factor (chol): the sub-diagonal block, rows of twelve, then
// the lower triangle of the diagonal block, row l as l + 1 entries padded to an
// even count.  The entries right of the diagonal, which the reference never
// writes (dual_qp.c:832-903) and which stay zero in its [12][24] rows, are not
// stored: 228 values instead of 288.  The rows of the triangle are placed so
// that none crosses a multiple of 24 values (the piece one LDS-DMA instruction
// of k_newton_solve brings in per instance).
constexpr int kCholRow = 228;
constexpr int kCholDiagRow[12] = {144 + 82, 144 + 70, 144 + 78, 144 + 44, 144 + 72, 144 + 64,
                                  144 + 56, 144 + 48, 144 + 34, 144 + 24, 144 + 12, 144 + 0};
// Reads of a row's first twelve values, whatever its length, stay inside the
// field: its allocation ends with this many spare values.
constexpr int kCholSlack = 16;
// The Newton Hessian's block rows (hess) have the same layout: the factorisation
// reads the sub-diagonal block and the entries (i, j <= i) of the diagonal block
// (dual_qp.c:835, 871, 892), so the entries right of the diagonal, which the
// reference computes and never looks at, are neither computed nor stored.

constexpr int kSolveListBuckets = 128;
constexpr int kSolveListStride = 2 * kSolveListBuckets + 16;
// (begin, end) pairs behind the histogram and the cursors: the whole list, its
// leading part with long factorisation sweeps, the rest
constexpr int kSolveRangeAll = 2 * kSolveListBuckets;
constexpr int kSolveRangeLong = kSolveRangeAll + 2;
constexpr int kSolveRangeShort = kSolveRangeAll + 4;

// Per-instance weights and control bounds (BatchView::cfg), one record of
// kCfgStride values per instance.  The first half is what the setters write; the
// second half is what nmpc_batch_init latches from it, as the reference copies
// weights and bounds into its QP at nmpc_init (ccs-c66x/cnmpc.c:899-931): the
// stage Hessians as the solver sees them (a weight block within 1e-15 of the
// identity counts as exact ones, qpDUNES/utils.c:106-117) and the bounds of that
// moment.  The live bounds are what every later linearisation and feedback step
// read (cnmpc.c:643-645, 1053-1062, 675-702).
constexpr int kCfgStride = 72;
constexpr int kCfgWq = 0, kCfgWr = 12, kCfgWt = 15, kCfgLb = 27, kCfgUb = 30;   // live
constexpr int kCfgH = 36, kCfgHt = 51, kCfgLb0 = 63, kCfgUb0 = 66;              // latched by init

struct BatchView {
    int B, N;
    int b0, nb;         // instance range [b0, b0 + nb) the QP kernels of one launch cover
    int head, xhead, lhead;
    real_t* xref;       // [B][N+1][13]
    real_t* uref;       // [B][N][3]      stage slot
    real_t* wind;       // [B][3]
    real_t* C;          // [B][N][12*15]  stage slot, row-major 12x15
    real_t* c;          // [B][N][12]
    real_t* zLow;       // [B][N+1][15]
    real_t* zUpp;
    real_t* q;
    real_t* z;
    real_t* qStep;
    real_t* zUnc;
    real_t* dz;
    real_t* wabs;       // [B][N+1][16]   magnitudes behind qStep (15) and pStep (1), for k_line_search_fast
    real_t* tmp;        // trial primal vector of the line search; between solves' kernels scratch of k_newton_solve
    real_t* y;          // [B][N+1][30]   (lb - z, z - ub) pairs
    real_t* p;          // [B][N+1]
    real_t* pStep;      // [B][N+1]
    real_t* lam;        // [B][N][12]     lambda ring
    real_t* dl;         // [B][N][12]
    real_t* grad;       // [B][N][12]
    real_t* hess;       // [B][N][kCholRow] (+ kCholSlack): block rows of the Newton Hessian, laid out as the factor's
    real_t* chol;       // [B][N][kCholRow] (+ kCholSlack)
    int8_t* ieq;        // [B][N+1][15]
    int8_t* prev;       // [B][N+1][15]
    int8_t* chg;        // [B][N+1]       actSetHasChanged
    Scalars* sc;        // [B]
    int* tailCount;     // [16] step-length searches handed on, per Newton iteration: [0..7] by
                        //      k_newton_step, [8..15] those k_line_search_fast left open
    int* tailList;      // [2][B] their instances
                        //      (both allocated twice: the second set serves the view of the
                        //      instances with short sweeps when an iteration runs as two parts)
    int* solveCount;    // [2][kSolveListStride] counters of the Newton solve's instance lists, two
                        //      alternating sets: histogram by first block to factorise, fill
                        //      cursors, then the ranges kSolveRange* of the list
    int* solveList;     // [B] instances with a Newton system to solve, longest sweeps first
    // Instances the Newton-iteration kernels of one launch cover: entries
    // [listRange[0], listRange[1]) of list, or all of [b0, b0 + nb) when list is null.
    const int* list;
    const int* listRange;
    int listPart;       // which part of the list the range is: 0 whole, 1 long sweeps, 2 the rest
    real_t H[15];       // regular-stage Hessian diagonal as the solver sees it
    real_t Ht[12];      // terminal-stage Hessian diagonal
    real_t lb[3], ub[3];
    // Null while every instance has the weights and bounds above; otherwise
    // [B][kCfgStride] records that take their place, instance by instance.
    const real_t* cfg;
};

}  // namespace nmpc
