// Host-callable launchers of the HIP kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "layout.hpp"

namespace nmpc {

struct Bounds { real_t lb[3], ub[3]; };

void launch_store_refs(const BatchView& v, const real_t* refs, int first, int count, hipStream_t s);
void launch_refs_from_waypoints(int B, const real_t* wp, int W, real_t t0, real_t dt, int first,
                                int count, real_t* out, hipStream_t s);
void launch_init_stages(const BatchView& v, int part, hipStream_t s);
// about_solution: expansion point moved by each stage's current z (see k_linearise)
void launch_linearise(const BatchView& v, int first, int count, hipStream_t s,
                      bool about_solution = false, bool only_solved = false);
void launch_rti_reset(const BatchView& v, hipStream_t s);
// Per-instance weights and bounds (BatchView::cfg): entries [at, at + n) of the
// records of instances [first, first + count) from values[count][n] (device
// memory) or, values null, the n values of `same` (host memory) for every one of
// them; launch_cfg_latch is nmpc_init's copy of weights and bounds into the QP.
void launch_cfg_set(real_t* cfg, int first, int count, int at, int n, const real_t* values,
                    const real_t* same, hipStream_t s);
void launch_cfg_latch(real_t* cfg, int B, hipStream_t s);
void launch_linearise_free(long n, const real_t* x, const real_t* u, const real_t* xnext,
                           const real_t* wind, const Bounds& bnd, real_t* jac, real_t* res,
                           hipStream_t s);
void launch_rk4_free(long n, const real_t* x, const real_t* u, const real_t* wind, real_t delta,
                     real_t* out, hipStream_t s);
void launch_qp_begin(const BatchView& v, const real_t* meas, hipStream_t s);
// List of the instances whose solve is still running, longest factorisation
// sweeps first: launch_list_solves makes list number `epoch` (the number only has
// to go up by one per list), solve_list_view gives the view of the batch whose
// Newton-iteration kernels (launch_newton_solve, launch_newton_step,
// launch_line_search_tail) cover part 0: the whole list, 1: the long sweeps,
// 2: the rest, with step-length search lists of its own so that parts 1 and 2 can
// run side by side.
bool newton_solve_two_waves(const BatchView& v);
void launch_list_solves(const BatchView& v, unsigned epoch, hipStream_t s);
BatchView solve_list_view(const BatchView& v, unsigned epoch, int part);
// few_searches: the step-length searches of this solve are expected to be few (a solve after a
// linearisation of every stage); the single-precision library only then runs a listed part as teams
void launch_newton_solve(const BatchView& v, hipStream_t s, bool few_searches = true);
void launch_newton_step(const BatchView& v, int next_it, hipStream_t s);
bool line_search_tail_enabled(const BatchView& v);
void launch_line_search_tail(const BatchView& v, int next_it, hipStream_t s);
void launch_qp_finish(const BatchView& v, hipStream_t s);
void launch_shift_lambda_tail(const BatchView& v, hipStream_t s);
void launch_gather_results(const BatchView& v, real_t* u, int* status, int* iters, hipStream_t s);
int max_newton_iterations();

}  // namespace nmpc
