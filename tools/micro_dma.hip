// Which addresses does the instruction offset of global_load_lds_dwordx4 move?
// Lane i loads 16 bytes at src + i*16 (+ offset?) into LDS at M0 base + i*16 (+ offset?).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro_dma.hip -o variants/micro_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const double* __restrict__ src, double* out) {
    __shared__ double buf[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) buf[i] = -1.0;
    __syncthreads();
    // offset 192 on the instruction, LDS base 1024 - 192 = 832
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 2 * lane),
                                     (__attribute__((address_space(3))) void*)((char*)buf + 832), 16, 192, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 1024; i += 64) out[i] = buf[i];
}
int main() {
    std::vector<double> h(4096);
    for (int i = 0; i < 4096; i++) h[i] = i;
    double *d, *o;
    hipMalloc(&d, 4096 * 8); hipMalloc(&o, 1024 * 8);
    hipMemcpy(d, h.data(), 4096 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    std::vector<double> r(1024);
    hipMemcpy(r.data(), o, 1024 * 8, hipMemcpyDeviceToHost);
    int first = -1;
    for (int i = 0; i < 1024; i++) if (r[i] >= 0) { first = i; break; }
    printf("first written LDS double index %d (128 = base+offset both applied to LDS) holds source double %g (24 = offset applied to global)\n", first, first >= 0 ? r[first] : -1.0);
    printf("next: idx %d val %g ; idx %d val %g\n", first + 1, r[first + 1], first + 2, r[first + 2]);
    return 0;
}
