// Micro-benchmark and exhaustive-style check of a short-latency fp64 division.
//
// n / d (correctly rounded) needs about ten dependent instructions on gfx950.
// When d is known before n -- the pivots of a triangular solve -- its correctly
// rounded reciprocal y can be prepared off the critical path and the quotient
// finished with three dependent instructions (Markstein):
//     q0 = RN(n * y);  r = fma(-d, q0, n) (exact);  q = fma(r, y, q0)
// taken as  rn = fma(d, q0, -n);  q = fma(-rn, y, q0): a residual of exactly zero
// is then +0 where it is made and -0 where it is used, so that an exact quotient,
// a zero one with its sign included, passes through unchanged.
// This program measures both chains and counts results that differ from n / d
// over random and adversarial operands.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o micro_div micro_div.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#pragma clang fp contract(off)

__device__ __forceinline__ double fast_div(double n, double d, double y) {
    const double q0 = n * y;
    const double rn = __builtin_fma(d, q0, -n);
    return __builtin_fma(-rn, y, q0);
}

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

// mode 0: random significands, exponents within +-40 of zero
// mode 1: divisor significand with a run of trailing ones / all ones
// mode 2: numerator = RN(q * d) +- few ulps for a random q (quotients next to a rounding boundary)
// mode 3: exact quotients: numerator +0, -0, or a small integer times a divisor of 26 bits;
//         positive divisors only, as in the solver (its divisors are square roots): a zero
//         over a negative divisor comes out with the sign of the numerator
__global__ void check(uint64_t seed, int mode, int per_thread, unsigned long long* bad,
                      unsigned long long* bad_guarded, double* sample) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long nb = 0, ng = 0;
    for (int it = 0; it < per_thread; it++) {
        const uint64_t a = mix(seed + gid * 0x9e3779b97f4a7c15ULL + (uint64_t)it * 0x632be59bd9b4e019ULL);
        const uint64_t b = mix(a ^ 0xdeadbeefcafef00dULL);
        const uint64_t c = mix(b + 12345);
        uint64_t md = b & 0xfffffffffffffULL, mn = a & 0xfffffffffffffULL;
        if (mode == 1) {
            const int ones = 1 + (int)(c % 52);
            md |= (1ULL << ones) - 1;
            if ((c >> 8) % 4 == 0) md = 0xfffffffffffffULL;
        }
        const int ed = 1023 + (int)((b >> 52) % 81) - 40, en = 1023 + (int)((a >> 52) % 81) - 40;
        double d = __longlong_as_double(((uint64_t)ed << 52) | md);
        double n = __longlong_as_double(((uint64_t)en << 52) | mn);
        if ((a >> 63) & 1) n = -n;
        if ((b >> 63) & 1) d = -d;
        if (mode == 2) {
            const double q = n;
            const double prod = q * d;
            const long long k = (long long)(c % 7) - 3;
            n = __longlong_as_double(__double_as_longlong(prod) + k);
        }
        if (mode == 3) {
            d = __longlong_as_double(__double_as_longlong(d) & 0x7ffffffff8000000LL);
            const long long k = (long long)(c % 65) - 32;
            n = (double)k * d;
            if (c % 5 == 0) n = 0.0;
            if (c % 5 == 1) n = -0.0;
        }
        const double y = 1.0 / d;
        const double ref = n / d;
        const double f = fast_div(n, d, y);
        if (__double_as_longlong(ref) != __double_as_longlong(f)) {
            nb++;
            if (nb == 1) { sample[0] = n; sample[1] = d; }
            if ((__double_as_longlong(d) & 0xfffffffffffffLL) != 0xfffffffffffffLL) ng++;
        }
    }
    if (nb) atomicAdd(bad, nb);
    if (ng) atomicAdd(bad_guarded, ng);
}

template <int FAST>
__global__ void chain(double* out, long long* ticks, double d, int n) {
    double x = out[threadIdx.x];
    const double y = 1.0 / d;
    const long long t0 = wall_clock64();
    for (int i = 0; i < n; i++) {
        x = x + 1.0;
        x = FAST ? fast_div(x, d, y) : x / d;
    }
    const long long t1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) *ticks = t1 - t0;
}

int main() {
    unsigned long long *bad, *badg;
    double* sample;
    hipMalloc(&bad, 8); hipMalloc(&badg, 8); hipMalloc(&sample, 16);
    for (int mode = 0; mode < 4; mode++) {
        hipMemset(bad, 0, 8); hipMemset(badg, 0, 8);
        const int blocks = 1 << 14, threads = 256, per = 1 << 11;   // 2^33 pairs per mode
        check<<<blocks, threads>>>(0x1234567ULL + mode, mode, per, bad, badg, sample);
        hipDeviceSynchronize();
        unsigned long long hb, hg; double hs[2];
        hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
        hipMemcpy(&hg, badg, 8, hipMemcpyDeviceToHost);
        hipMemcpy(hs, sample, 16, hipMemcpyDeviceToHost);
        printf("mode %d: %llu pairs, %llu differ from n/d (%llu with a divisor whose significand is not all ones)",
               mode, (unsigned long long)blocks * threads * per, hb, hg);
        if (hb) printf("  e.g. n=%a d=%a", hs[0], hs[1]);
        printf("\n");
    }
    double* out; long long* ticks;
    hipMalloc(&out, 64 * 8); hipMalloc(&ticks, 8);
    hipMemset(out, 0, 64 * 8);
    const int n = 100000;
    for (int fast = 0; fast < 2; fast++) {
        if (fast) chain<1><<<1, 64>>>(out, ticks, 1.37, n); else chain<0><<<1, 64>>>(out, ticks, 1.37, n);
        hipDeviceSynchronize();
        long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        printf("%s: %.1f ns per (add + divide)\n", fast ? "reciprocal + 3 dependent ops" : "n / d", t * 10.0 / n);
    }
    return 0;
}
