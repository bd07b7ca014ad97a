// Cost of a workgroup barrier between wavefronts that sit on different SIMDs of one
// CU, as the team form of the Newton solve uses it (tools/bin/micro_barrier).
//   hipcc --offload-arch=gfx950 -O3 tools/micro_barrier.hip -o tools/bin/micro_barrier
#include <hip/hip_runtime.h>
#include <cstdio>

template <int WAVES, int MODE>
__global__ __launch_bounds__(WAVES * 64) void k(long long* out, int iters, double* sink) {
    __shared__ double buf[2][WAVES * 64];
    const int tid = threadIdx.x;
    double x = tid * 1e-3, acc = 0.;
    buf[0][tid] = x; buf[1][tid] = x;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {            // barrier only
            asm volatile("s_barrier" ::: "memory");
        } else if (MODE == 1) {     // write, wait, barrier
            buf[i & 1][tid] = x;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else if (MODE == 2) {     // write, wait, barrier, read the neighbour wave's word, use it
            buf[i & 1][tid] = x;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            x += buf[i & 1][(tid + 64) % (WAVES * 64)];
        } else if (MODE == 3) {     // no barrier: write, wait, read own word back
            buf[i & 1][tid] = x;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            x += buf[i & 1][tid];
        } else if (MODE == 4) {     // 40 dependent adds between barriers (a chain step)
#pragma unroll
            for (int j = 0; j < 40; j++) { x += 1.0; asm volatile("" : "+v"(x)); }
            asm volatile("s_barrier" ::: "memory");
        } else if (MODE == 5) {     // the same without the barrier
#pragma unroll
            for (int j = 0; j < 40; j++) { x += 1.0; asm volatile("" : "+v"(x)); }
        }
    }
    const long long t1 = clock64();
    acc += x;
    if (tid == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 12345.678) *sink = acc;
}

template <int WAVES, int MODE>
void run(const char* what, int blocks) {
    long long* d; double* s;
    hipMalloc(&d, blocks * sizeof(long long)); hipMalloc(&s, 8);
    const int iters = 20000;
    k<WAVES, MODE><<<blocks, WAVES * 64>>>(d, iters, s);
    k<WAVES, MODE><<<blocks, WAVES * 64>>>(d, iters, s);
    hipDeviceSynchronize();
    long long h[1024];
    hipMemcpy(h, d, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double sum = 0; for (int i = 0; i < blocks; i++) sum += h[i];
    printf("%d waves, %4d workgroups, %-55s %7.1f cycles per iteration\n", WAVES, blocks, what, sum / blocks / iters);
    hipFree(d); hipFree(s);
}

int main() {
    for (int blocks : {1, 256}) {
        run<4, 0>("barrier", blocks);
        run<4, 1>("LDS write, wait, barrier", blocks);
        run<4, 2>("LDS write, wait, barrier, read other wave's word", blocks);
        run<4, 3>("LDS write, wait, read own word (no barrier)", blocks);
        run<4, 4>("40 dependent fp64 adds, barrier", blocks);
        run<4, 5>("40 dependent fp64 adds", blocks);
        run<2, 0>("barrier", blocks);
        run<2, 2>("LDS write, wait, barrier, read other wave's word", blocks);
        run<1, 0>("barrier", blocks);
    }
    return 0;
}
