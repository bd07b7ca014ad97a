// Does the memory system care how the stage vectors of an instance are laid out?
// One workgroup per instance reads six vectors of E doubles and writes six (the traffic
// shape of the stage pass of the dual Newton solve), with the vectors
//   field-major:    twelve arrays [instances][E]            (the library's layout)
//   instance-major: one array [instances][12][E]            (an instance's vectors adjacent)
//   interleaved:    one array [instances][E][12]            (an element's twelve values adjacent)
//   hipcc --offload-arch=gfx950 -O3 -o micro_layout micro_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int E = 1515, F = 12;

template <int LAYOUT>
__global__ __launch_bounds__(256) void k_pass(double* base, int instances) {
    for (int b = blockIdx.x; b < instances; b += gridDim.x) {
        for (int e = threadIdx.x; e < E; e += 256) {
            double v[6];
#pragma unroll
            for (int f = 0; f < 6; f++) {
                size_t at;
                if (LAYOUT == 0) at = ((size_t)f * instances + b) * E + e;
                else if (LAYOUT == 1) at = ((size_t)b * F + f) * E + e;
                else at = ((size_t)b * E + e) * F + f;
                v[f] = base[at];
            }
#pragma unroll
            for (int f = 0; f < 6; f++) {
                size_t at;
                if (LAYOUT == 0) at = ((size_t)(6 + f) * instances + b) * E + e;
                else if (LAYOUT == 1) at = ((size_t)b * F + 6 + f) * E + e;
                else at = ((size_t)b * E + e) * F + 6 + f;
                base[at] = v[f] + 1.0;
            }
        }
    }
}

template <int LAYOUT>
double run(double* base, int instances, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int p = 0; p < 4; p++) {
        hipEventRecord(e0, nullptr);
        k_pass<LAYOUT><<<blocks, 256>>>(base, instances);
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (p && ms < best) best = ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return (double)instances * E * F * 8 / (best * 1e-3) / 1e12;
}

int main() {
    const int instances = 16384;
    double* base;
    const size_t bytes = (size_t)instances * E * F * 8;
    hipMalloc(&base, bytes);
    hipMemset(base, 0, bytes);
    for (int blocks : {256 * 2, 256 * 4, 256 * 6, 256 * 8, instances}) {
        printf("workgroups %5d: field-major %.2f  instance-major %.2f  interleaved %.2f TB/s\n", blocks,
               run<0>(base, instances, blocks), run<1>(base, instances, blocks), run<2>(base, instances, blocks));
    }
    return 0;
}
