// Check of the pivot square root with reciprocal used by the Newton factorisation.
//
// The factorisation needs d = sqrt(p) and two quotients n / d per column.  The
// compiler's sqrt is v_rsq_f64 followed by one Goldschmidt step and two
// corrections (plus a rescaling that only acts for p < 2^-767); the same
// iteration carries h ~ 1/(2 d) along, so a reciprocal y = RN(1/d) costs four
// more fused operations (one Newton step, then Markstein's correction, which
// gives the correctly rounded reciprocal of any d whose significand is not all
// ones), and each quotient three (see micro_div.hip).
// This program counts, over random and adversarial p and n:
//   - d  != sqrt(p)
//   - y  != 1.0 / d     (split by "significand of d all ones")
//   - q  != n / d
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o micro_sqrt micro_sqrt.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#pragma clang fp contract(off)

struct Pivot { double d, y; };

__device__ __forceinline__ Pivot root_and_reciprocal(double x) {
    const double r = __builtin_amdgcn_rsq(x);
    double g = x * r, h = r * 0.5;
    const double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    h = __builtin_fma(h, e, h);
    const double e2 = __builtin_fma(-h, g, 0.5);
    const double h2 = __builtin_fma(h, e2, h);
    const double d0 = __builtin_fma(-g, g, x);
    g = __builtin_fma(d0, h, g);
    const double d1 = __builtin_fma(-g, g, x);
    g = __builtin_fma(d1, h, g);
    const double y0 = h2 + h2;
    const double e0 = __builtin_fma(-g, y0, 1.0);
    const double y1 = __builtin_fma(e0, y0, y0);
    const double e1 = __builtin_fma(-g, y1, 1.0);
    const double y = __builtin_fma(e1, y1, y1);
    return {g, y};
}

__device__ __forceinline__ double fast_div(double n, double d, double y) {
    const double q0 = n * y;
    const double r = __builtin_fma(-d, q0, n);
    return __builtin_fma(r, y, q0);
}

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

// mode 0: random significands, exponent of p within +-400, of n within +-300
// mode 1: p = RN(s * s) +- few ulps for a random s (roots next to a rounding boundary),
//         every fourth s with a run of trailing ones or all ones
// mode 2: p in [2.2e-14, 1e-12] (just above the smallest pivot the factorisation accepts)
__global__ void check(uint64_t seed, int mode, int per_thread, unsigned long long* counts, double* sample) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bad_d = 0, bad_y = 0, bad_y_ones = 0, bad_q = 0, ones = 0;
    for (int it = 0; it < per_thread; it++) {
        const uint64_t a = mix(seed + gid * 0x9e3779b97f4a7c15ULL + (uint64_t)it * 0x632be59bd9b4e019ULL);
        const uint64_t b = mix(a ^ 0xdeadbeefcafef00dULL);
        const uint64_t c = mix(b + 12345);
        uint64_t mp = b & 0xfffffffffffffULL;
        const uint64_t mn = a & 0xfffffffffffffULL;
        int ep = 1023 + (int)((b >> 52) % 801) - 400;
        const int en = 1023 + (int)((a >> 52) % 601) - 300;
        if (mode == 2) ep = 1023 - 46 + (int)((b >> 52) % 6);
        double p = __longlong_as_double(((uint64_t)ep << 52) | mp);
        if (mode == 1) {
            uint64_t ms = mp;
            if ((c >> 8) % 4 == 0) {
                const int run = 1 + (int)(c % 52);
                ms |= (1ULL << run) - 1;
                if ((c >> 16) % 4 == 0) ms = 0xfffffffffffffULL;
            }
            const double s = __longlong_as_double(((uint64_t)(1023 + (ep - 1023) / 2) << 52) | ms);
            const long long k = (long long)((c >> 24) % 7) - 3;
            p = __longlong_as_double(__double_as_longlong(s * s) + k);
        }
        double n = __longlong_as_double(((uint64_t)en << 52) | mn);
        if ((a >> 63) & 1) n = -n;
        if ((c >> 40) % 64 == 0) n = 0.0;
        const Pivot pv = root_and_reciprocal(p);
        const double d = sqrt(p);
        const bool all_ones = (__double_as_longlong(d) & 0xfffffffffffffLL) == 0xfffffffffffffLL;
        if (all_ones) ones++;
        if (__double_as_longlong(d) != __double_as_longlong(pv.d)) {
            bad_d++;
            if (bad_d == 1) sample[0] = p;
            continue;
        }
        if (__double_as_longlong(1.0 / d) != __double_as_longlong(pv.y)) {
            if (all_ones) bad_y_ones++; else { bad_y++; if (bad_y == 1) sample[1] = p; }
            continue;
        }
        if (__double_as_longlong(n / d) != __double_as_longlong(fast_div(n, d, pv.y))) {
            bad_q++;
            if (bad_q == 1) { sample[2] = p; sample[3] = n; }
        }
    }
    if (bad_d) atomicAdd(counts + 0, bad_d);
    if (bad_y) atomicAdd(counts + 1, bad_y);
    if (bad_y_ones) atomicAdd(counts + 2, bad_y_ones);
    if (bad_q) atomicAdd(counts + 3, bad_q);
    if (ones) atomicAdd(counts + 4, ones);
}

int main() {
    unsigned long long* counts;
    double* sample;
    if (hipMalloc(&counts, 5 * 8) != hipSuccess || hipMalloc(&sample, 4 * 8) != hipSuccess) return 1;
    int rc = 0;
    for (int mode = 0; mode < 3; mode++) {
        (void)hipMemset(counts, 0, 5 * 8);
        (void)hipMemset(sample, 0, 4 * 8);
        const int blocks = 1 << 14, threads = 256, per = 1 << 11;   // 2^33 samples per mode
        check<<<blocks, threads>>>(0x7654321ULL + mode, mode, per, counts, sample);
        if (hipDeviceSynchronize() != hipSuccess) return 1;
        unsigned long long h[5]; double hs[4];
        (void)hipMemcpy(h, counts, 5 * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hs, sample, 4 * 8, hipMemcpyDeviceToHost);
        printf("mode %d: %llu samples: root differs %llu, reciprocal differs %llu (+%llu of %llu roots with an "
               "all-ones significand), quotient differs %llu\n",
               mode, (unsigned long long)blocks * threads * per, h[0], h[1], h[2], h[4], h[3]);
        if (h[0]) printf("   root: p=%a\n", hs[0]);
        if (h[1]) printf("   reciprocal: p=%a\n", hs[1]);
        if (h[3]) printf("   quotient: p=%a n=%a\n", hs[2], hs[3]);
        if (h[0] || h[1] || h[3]) rc = 2;
    }
    return rc;
}
