// Host-callable launchers of the HIP kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "layout.hpp"

namespace nmpc {

struct Bounds { double lb[3], ub[3]; };

void launch_store_refs(const BatchView& v, const double* refs, int first, int count, hipStream_t s);
void launch_refs_from_waypoints(int B, const double* wp, int W, double t0, double dt, int first,
                                int count, double* out, hipStream_t s);
void launch_init_stages(const BatchView& v, int part, hipStream_t s);
// about_solution: expansion point moved by each stage's current z (see k_linearise)
void launch_linearise(const BatchView& v, int first, int count, hipStream_t s,
                      bool about_solution = false);
void launch_linearise_free(long n, const double* x, const double* u, const double* xnext,
                           const double* wind, const Bounds& bnd, double* jac, double* res,
                           hipStream_t s);
void launch_rk4_free(long n, const double* x, const double* u, const double* wind, double delta,
                     double* out, hipStream_t s);
void launch_qp_begin(const BatchView& v, const double* meas, hipStream_t s);
void launch_newton_solve(const BatchView& v, hipStream_t s);
void launch_newton_step(const BatchView& v, int next_it, hipStream_t s);
bool line_search_tail_enabled(const BatchView& v);
void launch_line_search_tail(const BatchView& v, int next_it, hipStream_t s);
void launch_qp_finish(const BatchView& v, hipStream_t s);
void launch_shift_lambda_tail(const BatchView& v, hipStream_t s);
void launch_gather_results(const BatchView& v, double* u, int* status, int* iters, hipStream_t s);
int max_newton_iterations();

}  // namespace nmpc
