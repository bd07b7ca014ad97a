// Dependent-issue latency of vector instructions on gfx950: one wavefront runs
// CHAINS independent chains of N dependent operations each; cycles per operation
// of one chain = elapsed / N.  Built and run by hand:
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/micro_latency.hip -o gpurun_out/micro_latency
#include <hip/hip_runtime.h>
#include <stdio.h>

template <typename T, int CHAINS, int OP>
__global__ void k_chain(int n, T seed, T* out, long long* cycles, long long* ticks) {
    T a[CHAINS];
    for (int c = 0; c < CHAINS; c++) a[c] = seed + (T)(threadIdx.x + c);
    const T m = (T)1.0000001, b = (T)1e-9;
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (OP == 0) a[c] = a[c] + b;
                else if (OP == 1) a[c] = a[c] * m;
                else a[c] = __builtin_fma(a[c], m, b);
            }
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    T s = 0;
    for (int c = 0; c < CHAINS; c++) s += a[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { *cycles = t1 - t0; *ticks = w1 - w0; }
}

// One column of the top-down substitution: quotient from a reciprocal (mul, fma, fma), row
// broadcast, mul, sub; STEP selects how much of it runs.
template <int STEP>
__global__ void k_column(int n, double seed, double* out, long long* cycles, long long* ticks) {
    double sum = seed + threadIdx.x, y = 0.5000001, d = 2.0, l = 1.0000001;
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            double x = sum;
            if (STEP >= 1) {
                const double q0 = sum * y;
                const double rn = __builtin_fma(d, q0, -sum);
                x = __builtin_fma(-rn, y, q0);
            }
            if (STEP >= 2) x = __builtin_amdgcn_update_dpp(0.0, x, 0x150 + 3, 0xf, 0xf, true);
            sum -= l * x;
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) { *cycles = t1 - t0; *ticks = w1 - w0; }
}

template <int STEP>
void run_column(const char* name) {
    double* out; long long *cyc, *tk;
    hipMalloc(&out, 8 * 64); hipMalloc(&cyc, 8); hipMalloc(&tk, 8);
    const int n = 20000;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((k_column<STEP>), dim3(1), dim3(64), 0, 0, n, 1.0, out, cyc, tk);
        hipDeviceSynchronize();
    }
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-40s clock64 %.2f per column\n", name, c / ((double)n * 8));
    hipFree(out); hipFree(cyc); hipFree(tk);
}

template <typename T, int CHAINS, int OP>
void run(const char* name, int waves, int lanes = 64) {
    T* out; long long *cyc, *tk;
    hipMalloc(&out, sizeof(T) * 64 * 64);
    hipMalloc(&cyc, 8); hipMalloc(&tk, 8);
    const int n = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_chain<T, CHAINS, OP>), dim3(1), dim3(lanes * waves), 0, 0, n, (T)1.0, out, cyc, tk);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c, t;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&t, tk, 8, hipMemcpyDeviceToHost);
    const double ops = (double)n * 8;
    printf("%-10s lanes %2d chains %d waves/block %2d: clock64 %.2f per op-step, wall ticks(100MHz) %.3f per op-step -> %.1f ns, event %.3f ms (%.2f ns per op-step)\n",
           name, lanes, CHAINS, waves, c / ops, t / ops, t / ops * 10.0, ms, ms * 1e6 / ops);
    hipFree(out); hipFree(cyc); hipFree(tk);
}

int main() {
    run<double, 1, 0>("add_f64", 1); run<double, 2, 0>("add_f64", 1); run<double, 4, 0>("add_f64", 1);
    run<double, 1, 1>("mul_f64", 1); run<double, 2, 1>("mul_f64", 1); run<double, 4, 1>("mul_f64", 1);
    run<double, 1, 2>("fma_f64", 1); run<double, 4, 2>("fma_f64", 1);
    run<float, 1, 0>("add_f32", 1); run<float, 2, 0>("add_f32", 1); run<float, 4, 0>("add_f32", 1);
    run<float, 1, 2>("fma_f32", 1);
    // several wavefronts of one workgroup (one CU): 4 = one per SIMD, 8 = two per SIMD, 12, 16
    run<double, 1, 0>("add_f64", 4); run<double, 1, 0>("add_f64", 8); run<double, 1, 0>("add_f64", 12);
    run<double, 1, 0>("add_f64", 16);
    run<double, 2, 0>("add_f64", 8); run<double, 2, 0>("add_f64", 12);
    // a lone wavefront with 16 or 32 of its 64 lanes enabled: does the SIMD skip the empty quarters?
    run<double, 1, 0>("add_f64", 1, 16); run<double, 4, 0>("add_f64", 1, 16); run<double, 8, 0>("add_f64", 1, 16);
    run<double, 4, 0>("add_f64", 1, 32); run<double, 8, 0>("add_f64", 1, 32); run<double, 8, 0>("add_f64", 1, 64);
    run<double, 8, 1>("mul_f64", 1, 16); run<double, 8, 1>("mul_f64", 1, 64);
    run_column<0>("column: mul, sub");
    run_column<1>("column: quotient (mul, fma, fma), mul, sub");
    run_column<2>("column: quotient, row broadcast, mul, sub");
    return 0;
}
