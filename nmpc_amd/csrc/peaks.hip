// Measured machine peaks next to which bench.py quotes its roofline fractions:
// the fp64 vector rate with and without fused multiply-add (the library is
// built without contraction, so its own ceiling is the FMA-free one) and a
// stream triad for the HBM rate.  Nothing here is on the controller path.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cnmpc.h"

namespace {

constexpr int kChains = 8;      // independent dependency chains per lane
constexpr int kInner = 64;      // chain steps per loop trip (unrolled)

// x <- x * a + b on kChains chains per lane.  FUSED: one v_fma_f64 per step
// (2 flop); otherwise a v_mul_f64 and a v_add_f64 (1 flop each).
template <bool FUSED>
__global__ __launch_bounds__(256) void k_fp64_chains(double* __restrict__ out, double a, double b,
                                                      int trips) {
    double x[kChains];
#pragma unroll
    for (int c = 0; c < kChains; c++) x[c] = 1.0 + 1e-3 * (double)(threadIdx.x + c);
    for (int t = 0; t < trips; t++) {
#pragma unroll
        for (int i = 0; i < kInner; i++) {
#pragma unroll
            for (int c = 0; c < kChains; c++) {
                if (FUSED) {
                    x[c] = __builtin_fma(x[c], a, b);
                } else {
                    double m = x[c] * a;
                    asm volatile("" : "+v"(m));     // keeps the product and the sum apart
                    x[c] = m + b;
                }
            }
        }
    }
    double s = 0.;
#pragma unroll
    for (int c = 0; c < kChains; c++) s += x[c];
    if (s == 123.456) out[blockIdx.x * blockDim.x + threadIdx.x] = s;     // never true: keeps the work
}

typedef double pair_t __attribute__((ext_vector_type(2)));

// a[i] = b[i] + s * c[i], 16 bytes per lane and access.
__global__ __launch_bounds__(256) void k_triad(pair_t* __restrict__ a, const pair_t* __restrict__ b,
                                               const pair_t* __restrict__ c, double s, size_t n2) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        const pair_t bv = b[i], cv = c[i];
        pair_t r;
        r.x = bv.x + s * cv.x;
        r.y = bv.y + s * cv.y;
        a[i] = r;
    }
}

// The same with one contiguous 64 KB piece of each array per workgroup.  Which of
// the two shapes the memory system likes better varies from box to box and with
// the number of resident wavefronts (4.1 to 6.1 TB/s over the shapes
// tools/micro_hbm.hip tries); the measurement reports the best it sees.
__global__ __launch_bounds__(256) void k_triad_pieces(pair_t* __restrict__ a, const pair_t* __restrict__ b,
                                                      const pair_t* __restrict__ c, double s, size_t n2) {
    const size_t per = 4096;            // pairs per workgroup: 64 KB
    const size_t lo = per * blockIdx.x, hi = lo + per < n2 ? lo + per : n2;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
        const pair_t bv = b[i], cv = c[i];
        pair_t r;
        r.x = bv.x + s * cv.x;
        r.y = bv.y + s * cv.y;
        a[i] = r;
    }
}

#define PK_TRY(expr)                                       \
    do {                                                   \
        hipError_t e_ = (expr);                            \
        if (e_ != hipSuccess) { rc = -3; goto done; }      \
    } while (0)

}  // namespace

extern "C" {

int nmpc_measure_fp64_vector_peak(int device, int fused, double* tflops) {
    if (!tflops) return -1;
    int rc = 0;
    double* out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipDeviceProp_t prop;
    float ms = 0.f;
    int blocks = 0;
    const int trips = 400;
    PK_TRY(hipSetDevice(device));
    PK_TRY(hipGetDeviceProperties(&prop, device));
    blocks = prop.multiProcessorCount * 8;      // 8 wavefronts per SIMD
    PK_TRY(hipMalloc((void**)&out, (size_t)blocks * 256 * sizeof(double)));
    PK_TRY(hipEventCreate(&e0));
    PK_TRY(hipEventCreate(&e1));
    for (int pass = 0; pass < 3; pass++) {      // the last pass is the one reported
        PK_TRY(hipEventRecord(e0, nullptr));
        if (fused)
            hipLaunchKernelGGL(k_fp64_chains<true>, dim3(blocks), dim3(256), 0, nullptr, out, 0.999999,
                               1e-6, trips);
        else
            hipLaunchKernelGGL(k_fp64_chains<false>, dim3(blocks), dim3(256), 0, nullptr, out, 0.999999,
                               1e-6, trips);
        PK_TRY(hipGetLastError());
        PK_TRY(hipEventRecord(e1, nullptr));
        PK_TRY(hipEventSynchronize(e1));
        PK_TRY(hipEventElapsedTime(&ms, e0, e1));
    }
    *tflops = 2.0 * kChains * kInner * (double)trips * blocks * 256 / (ms * 1e-3) / 1e12;
done:
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (out) (void)hipFree(out);
    return rc;
}

int nmpc_measure_stream_triad(int device, uint64_t bytes_per_array, double* gbps) {
    if (!gbps || bytes_per_array < 4096) return -1;
    int rc = 0;
    pair_t *a = nullptr, *b = nullptr, *c = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipDeviceProp_t prop;
    float ms = 0.f, best = 1e30f;
    const size_t n2 = bytes_per_array / sizeof(pair_t);
    int blocks = 0;
    PK_TRY(hipSetDevice(device));
    PK_TRY(hipGetDeviceProperties(&prop, device));
    blocks = prop.multiProcessorCount * 8;
    PK_TRY(hipMalloc((void**)&a, n2 * sizeof(pair_t)));
    PK_TRY(hipMalloc((void**)&b, n2 * sizeof(pair_t)));
    PK_TRY(hipMalloc((void**)&c, n2 * sizeof(pair_t)));
    PK_TRY(hipMemset(b, 0, n2 * sizeof(pair_t)));
    PK_TRY(hipMemset(c, 0, n2 * sizeof(pair_t)));
    PK_TRY(hipEventCreate(&e0));
    PK_TRY(hipEventCreate(&e1));
    for (int pass = 0; pass < 12; pass++) {     // four passes of each shape, the first of each untimed
        const int shape = pass / 4;
        PK_TRY(hipEventRecord(e0, nullptr));
        if (shape == 0)
            hipLaunchKernelGGL(k_triad, dim3(blocks), dim3(256), 0, nullptr, a, b, c, 3.0, n2);
        else if (shape == 1)
            hipLaunchKernelGGL(k_triad, dim3(blocks / 4), dim3(256), 0, nullptr, a, b, c, 3.0, n2);
        else
            hipLaunchKernelGGL(k_triad_pieces, dim3((unsigned)((n2 + 4095) / 4096)), dim3(256), 0, nullptr,
                               a, b, c, 3.0, n2);
        PK_TRY(hipGetLastError());
        PK_TRY(hipEventRecord(e1, nullptr));
        PK_TRY(hipEventSynchronize(e1));
        PK_TRY(hipEventElapsedTime(&ms, e0, e1));
        if (pass % 4 > 0 && ms < best) best = ms;
    }
    *gbps = 3.0 * (double)(n2 * sizeof(pair_t)) / (best * 1e-3) / 1e9;
done:
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (c) (void)hipFree(c);
    return rc;
}

}  // extern "C"
