// What the memory system sustains: copy (1 read : 1 write) and triad (2 : 1) of 16-byte lanes,
// grid-stride with U independent accesses per lane in flight, several grid sizes.
//   hipcc --offload-arch=gfx950 -O3 -o micro_hbm micro_hbm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef double pair_t __attribute__((ext_vector_type(2)));

template <int U, bool TRIAD, bool NT>
__global__ __launch_bounds__(256) void k_stream(pair_t* __restrict__ a, const pair_t* __restrict__ b,
                                                const pair_t* __restrict__ c, double s, size_t n2) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n2; i += U * stride) {
        pair_t bv[U], cv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            bv[u] = NT ? __builtin_nontemporal_load(&b[i + u * stride]) : b[i + u * stride];
            if (TRIAD) cv[u] = NT ? __builtin_nontemporal_load(&c[i + u * stride]) : c[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            pair_t r = bv[u];
            if (TRIAD) { r.x = bv[u].x + s * cv[u].x; r.y = bv[u].y + s * cv[u].y; }
            if (NT) __builtin_nontemporal_store(r, &a[i + u * stride]); else a[i + u * stride] = r;
        }
    }
    for (; i < n2; i += stride) {
        pair_t r = b[i];
        if (TRIAD) { r.x += s * c[i].x; r.y += s * c[i].y; }
        a[i] = r;
    }
}

// the same with one contiguous chunk per workgroup
template <bool TRIAD>
__global__ __launch_bounds__(256) void k_chunked(pair_t* __restrict__ a, const pair_t* __restrict__ b,
                                                 const pair_t* __restrict__ c, double s, size_t n2) {
    const size_t per = (n2 + gridDim.x - 1) / gridDim.x;
    const size_t lo = per * blockIdx.x, hi = lo + per < n2 ? lo + per : n2;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
        pair_t r = b[i];
        if (TRIAD) { r.x += s * c[i].x; r.y += s * c[i].y; }
        a[i] = r;
    }
}
template <bool TRIAD>
double run_chunked(pair_t* a, pair_t* b, pair_t* c, size_t n2, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int p = 0; p < 4; p++) {
        hipEventRecord(e0, nullptr);
        k_chunked<TRIAD><<<blocks, 256>>>(a, b, c, 3.0, n2);
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (p && ms < best) best = ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return (TRIAD ? 3.0 : 2.0) * (double)(n2 * sizeof(pair_t)) / (best * 1e-3) / 1e12;
}

template <int U, bool TRIAD, bool NT>
double run(pair_t* a, pair_t* b, pair_t* c, size_t n2, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int p = 0; p < 4; p++) {
        hipEventRecord(e0, nullptr);
        k_stream<U, TRIAD, NT><<<blocks, 256>>>(a, b, c, 3.0, n2);
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (p && ms < best) best = ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return (TRIAD ? 3.0 : 2.0) * (double)(n2 * sizeof(pair_t)) / (best * 1e-3) / 1e12;
}

int main() {
    const size_t bytes = (size_t)1 << 30, n2 = bytes / sizeof(pair_t);
    pair_t *a, *b, *c;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes); hipMemset(c, 0, bytes);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cu = prop.multiProcessorCount;
    for (int per : {1, 2, 3, 4, 6, 8, 16, 64}) {
        const int blocks = cu * per;
        printf("blocks/CU %2d: copy U1 %.2f U2 %.2f U4 %.2f U4nt %.2f | triad U1 %.2f U2 %.2f U4 %.2f U4nt %.2f TB/s\n", per,
               run<1, false, false>(a, b, c, n2, blocks), run<2, false, false>(a, b, c, n2, blocks),
               run<4, false, false>(a, b, c, n2, blocks), run<4, false, true>(a, b, c, n2, blocks),
               run<1, true, false>(a, b, c, n2, blocks), run<2, true, false>(a, b, c, n2, blocks),
               run<4, true, false>(a, b, c, n2, blocks), run<4, true, true>(a, b, c, n2, blocks));
        printf("              chunked: copy %.2f triad %.2f; one workgroup per 64 KB: copy %.2f triad %.2f\n",
               run_chunked<false>(a, b, c, n2, blocks), run_chunked<true>(a, b, c, n2, blocks),
               run_chunked<false>(a, b, c, n2, (int)(bytes >> 16)), run_chunked<true>(a, b, c, n2, (int)(bytes >> 16)));
    }
    return 0;
}
