// Micro-benchmark: cost of an in-order fp64 accumulation of n LDS values by one
// wavefront (the reference fixes the summation order of its objective, norm and
// slope sums, so these cannot be tree reductions).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o micro_acc micro_acc.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <utility>

#pragma clang fp contract(off)

template <int AHEAD>
__device__ double acc_scalar(double acc, const double* buf, int len) {
    int t = 0;
    double c[AHEAD], nx[AHEAD];
    if (len >= AHEAD) {
#pragma unroll
        for (int i = 0; i < AHEAD; i++) c[i] = buf[i];
        for (t = AHEAD; t + AHEAD <= len; t += AHEAD) {
#pragma unroll
            for (int i = 0; i < AHEAD; i++) nx[i] = buf[t + i];
#pragma unroll
            for (int i = 0; i < AHEAD; i++) acc += c[i];
#pragma unroll
            for (int i = 0; i < AHEAD; i++) c[i] = nx[i];
        }
#pragma unroll
        for (int i = 0; i < AHEAD; i++) acc += c[i];
    }
    for (; t < len; t++) acc += buf[t];
    return acc;
}

template <int AHEAD>
__device__ double acc_pair(double acc, const double* buf, int len) {
    int t = 0;
    double2 c[AHEAD / 2], nx[AHEAD / 2];
    if (len >= AHEAD) {
#pragma unroll
        for (int i = 0; i < AHEAD / 2; i++) c[i] = *(const double2*)(buf + 2 * i);
        for (t = AHEAD; t + AHEAD <= len; t += AHEAD) {
#pragma unroll
            for (int i = 0; i < AHEAD / 2; i++) nx[i] = *(const double2*)(buf + t + 2 * i);
#pragma unroll
            for (int i = 0; i < AHEAD / 2; i++) { acc += c[i].x; acc += c[i].y; }
#pragma unroll
            for (int i = 0; i < AHEAD / 2; i++) c[i] = nx[i];
        }
#pragma unroll
        for (int i = 0; i < AHEAD / 2; i++) { acc += c[i].x; acc += c[i].y; }
    }
    for (; t < len; t++) acc += buf[t];
    return acc;
}

// Two register sets taking turns (no copies); BAR keeps each batch of reads
// behind the adds that free its registers.
template <int AHEAD, bool BAR>
__device__ double acc_two_sets(double acc, const double* buf, int len) {
    int t = 0;
    if (len >= AHEAD) {
        double c[AHEAD], nx[AHEAD];
#pragma unroll
        for (int i = 0; i < AHEAD; i++) c[i] = buf[i];
        for (t = AHEAD; t + 2 * AHEAD <= len; t += 2 * AHEAD) {
#pragma unroll
            for (int i = 0; i < AHEAD; i++) nx[i] = buf[t + i];
#pragma unroll
            for (int i = 0; i < AHEAD; i++) acc += c[i];
            if (BAR) asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < AHEAD; i++) c[i] = buf[t + AHEAD + i];
#pragma unroll
            for (int i = 0; i < AHEAD; i++) acc += nx[i];
            if (BAR) asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < AHEAD; i++) acc += c[i];
    }
    for (; t < len; t++) acc += buf[t];
    return acc;
}

// Registers only, 16 adds per loop iteration.
__device__ double acc_regs16(double acc, const double* buf, int len) {
    double a[4];
    for (int i = 0; i < 4; i++) a[i] = buf[i];
    for (int t = 0; t + 16 <= len; t += 16) {
#pragma unroll
        for (int i = 0; i < 16; i++) acc += a[i & 3];
    }
    return acc;
}

// Blocks of BLK values: read a whole block, then add it (no overlap at all).
template <int BLK>
__device__ double acc_block(double acc, const double* buf, int len) {
    int t = 0;
    for (; t + BLK <= len; t += BLK) {
        double c[BLK];
#pragma unroll
        for (int i = 0; i < BLK; i++) c[i] = buf[t + i];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < BLK; i++) acc += c[i];
        asm volatile("" ::: "memory");
    }
    for (; t < len; t++) acc += buf[t];
    return acc;
}

// The sixteen lanes of a DPP row load four consecutive values each (64 values,
// two LDS instructions); lane 0 of the row then takes them in order through
// row broadcasts: one v_mov_b64_dpp and one add per value, no LDS access in the
// dependent chain.  Every row of the wavefront can run its own list.
template <int SRC>
__device__ __forceinline__ double row_bcast(double x) {
    return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + SRC, 0xf, 0xf, true);
}
template <int... I>
__device__ __forceinline__ double add64(double acc, const double (&c)[4], std::integer_sequence<int, I...>) {
    ((acc += row_bcast<I / 4>(c[I % 4])), ...);
    return acc;
}
// The same with the broadcast folded into the add: v_fmac_f64 has a VOP2 form,
// which takes a DPP row broadcast on its first source; acc + x * 1.0 rounds as
// acc + x does.
template <int SRC>
__device__ __forceinline__ void add_bcast(double& acc, double x, double one) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc) : "v"(x), "v"(one), "n"(SRC));
}
template <int... I>
__device__ __forceinline__ double fmac64(double acc, const double (&c)[4], double one, std::integer_sequence<int, I...>) {
    (add_bcast<I / 4>(acc, c[I % 4], one), ...);
    return acc;
}
__device__ double acc_rows_fmac(double acc, const double* buf, int len) {
    const int l = threadIdx.x & 15;
    double one = 1.0;
    asm volatile("" : "+v"(one));
    int t = 0;
    if (len >= 64) {
        double c[4], nx[4];
#pragma unroll
        for (int i = 0; i < 4; i++) c[i] = buf[4 * l + i];
        for (t = 64; t + 64 <= len; t += 64) {
#pragma unroll
            for (int i = 0; i < 4; i++) nx[i] = buf[t + 4 * l + i];
            acc = fmac64(acc, c, one, std::make_integer_sequence<int, 64>{});
#pragma unroll
            for (int i = 0; i < 4; i++) c[i] = nx[i];
        }
        acc = fmac64(acc, c, one, std::make_integer_sequence<int, 64>{});
    }
    for (; t < len; t++) acc += buf[t];
    return acc;
}

__device__ double acc_rows(double acc, const double* buf, int len) {
    const int l = threadIdx.x & 15;
    int t = 0;
    if (len >= 64) {
        double c[4], nx[4];
#pragma unroll
        for (int i = 0; i < 4; i++) c[i] = buf[4 * l + i];
        for (t = 64; t + 64 <= len; t += 64) {
#pragma unroll
            for (int i = 0; i < 4; i++) nx[i] = buf[t + 4 * l + i];
            acc = add64(acc, c, std::make_integer_sequence<int, 64>{});
#pragma unroll
            for (int i = 0; i < 4; i++) c[i] = nx[i];
        }
        acc = add64(acc, c, std::make_integer_sequence<int, 64>{});
    }
    for (; t < len; t++) acc += buf[t];
    return acc;
}

// Every lane loads one value of each 64-value block; lane values are then added
// in lane order through scalar registers.
__device__ double acc_readlane(double acc, const double* buf, int len) {
    const int lane = threadIdx.x & 63;
    int t = 0;
    double cur = (lane < len) ? buf[lane] : 0.;
    for (; t + 64 <= len; t += 64) {
        const double nxt = (t + 64 + lane < len) ? buf[t + 64 + lane] : 0.;
#pragma unroll
        for (int i = 0; i < 64; i++) {
            const int lo = __builtin_amdgcn_readlane(__double2loint(cur), i);
            const int hi = __builtin_amdgcn_readlane(__double2hiint(cur), i);
            acc += __hiloint2double(hi, lo);
        }
        cur = nxt;
    }
    for (int i = 0; t + i < len; i++) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(cur), i);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(cur), i);
        acc += __hiloint2double(hi, lo);
    }
    return acc;
}

// Values in registers only: a chain of n dependent adds (lower bound).
__device__ double acc_regs(double acc, const double* buf, int len) {
    const double a = buf[0], b = buf[1];
    for (int t = 0; t + 1 < len; t += 2) { acc += a; acc += b; }
    return acc;
}

template <int MODE>
__global__ __launch_bounds__(256) void k(const double* in, double* out, long long* cyc, int n,
                                         int reps, int lanes) {
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 7 * n; i += blockDim.x) lds[i] = in[i % n] * (1 + i / n);
    __syncthreads();
    double r = 0.;
    long long t0 = 0, t1 = 0;
    if (threadIdx.x < 64) {
        t0 = wall_clock64();
        for (int rep = 0; rep < reps; rep++) {
            const bool act = (threadIdx.x < lanes);
            const double* b = lds + (act ? threadIdx.x * n : 0);
            double a = r * 1e-300;
            if (MODE == 16) {
                const int row = threadIdx.x >> 4;
                a = acc_rows_fmac(a, lds + (row < lanes ? row : 0) * n, n);
            } else if (MODE == 15) {
                // rows 0..lanes-1 run lists 0..lanes-1 (at most 4)
                const int row = threadIdx.x >> 4;
                a = acc_rows(a, lds + (row < lanes ? row : 0) * n, n);
            } else if (MODE == 4) {
                a = acc_readlane(a, lds, n);
            } else if (act) {
                if (MODE == 0) a = acc_scalar<8>(a, b, n);
                if (MODE == 1) a = acc_pair<8>(a, b, n);
                if (MODE == 2) a = acc_pair<16>(a, b, n);
                if (MODE == 3) a = acc_scalar<16>(a, b, n);
                if (MODE == 5) a = acc_regs(a, b, n);
                if (MODE == 6) a = acc_pair<32>(a, b, n);
                if (MODE == 12) a = acc_regs16(a, b, n);
                if (MODE == 13) a = acc_block<32>(a, b, n);
                if (MODE == 14) a = acc_block<64>(a, b, n);
                if (MODE == 7) a = acc_two_sets<8, false>(a, b, n);
                if (MODE == 8) a = acc_two_sets<16, false>(a, b, n);
                if (MODE == 9) a = acc_two_sets<8, true>(a, b, n);
                if (MODE == 10) a = acc_two_sets<16, true>(a, b, n);
                if (MODE == 11) a = acc_two_sets<32, true>(a, b, n);
            }
            r = a;
        }
        t1 = wall_clock64();
    }
    if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, const double* din, double* dout, long long* dcyc, int n, int lanes,
         double expect) {
    const int reps = 50;
    k<MODE><<<1, 256, 7 * n * sizeof(double)>>>(din, dout, dcyc, n, reps, lanes);
    hipDeviceSynchronize();
    long long c;
    double r;
    hipMemcpy(&c, dcyc, sizeof c, hipMemcpyDeviceToHost);
    hipMemcpy(&r, dout, sizeof r, hipMemcpyDeviceToHost);
    printf("%-22s lanes=%d  %.2f us per sum, %.2f ns per value   result %.17g (%s)\n", name, lanes,
           c * 10.0 / reps / 1e3, c * 10.0 / reps / n, r, (MODE == 5 || MODE == 12 || r == expect) ? "ok" : "DIFF");
}

int main() {
    const int n = 1200;
    std::vector<double> h(n);
    srand(1);
    for (auto& x : h) x = (rand() / (double)RAND_MAX - 0.5) * 1e3;
    double expect = 0.;
    for (int rep = 0; rep < 50; rep++) {
        double a = expect * 1e-300;
        for (int i = 0; i < n; i++) a += h[i];
        expect = a;
    }
    double *din, *dout;
    long long* dcyc;
    hipMalloc(&din, n * sizeof(double));
    hipMalloc(&dout, 64 * sizeof(double));
    hipMalloc(&dcyc, sizeof(long long));
    hipMemcpy(din, h.data(), n * sizeof(double), hipMemcpyHostToDevice);
    for (int lanes : {1, 7}) {
        run<0>("scalar reads, 8 ahead", din, dout, dcyc, n, lanes, expect);
        run<3>("scalar reads, 16 ahead", din, dout, dcyc, n, lanes, expect);
        run<1>("paired reads, 8 ahead", din, dout, dcyc, n, lanes, expect);
        run<2>("paired reads, 16 ahead", din, dout, dcyc, n, lanes, expect);
        run<6>("paired reads, 32 ahead", din, dout, dcyc, n, lanes, expect);
        run<13>("blocks of 32", din, dout, dcyc, n, lanes, expect);
        run<14>("blocks of 64", din, dout, dcyc, n, lanes, expect);
        run<7>("two sets, 8", din, dout, dcyc, n, lanes, expect);
        run<8>("two sets, 16", din, dout, dcyc, n, lanes, expect);
        run<9>("two sets, 8, pinned", din, dout, dcyc, n, lanes, expect);
        run<10>("two sets, 16, pinned", din, dout, dcyc, n, lanes, expect);
        run<11>("two sets, 32, pinned", din, dout, dcyc, n, lanes, expect);
    }
    run<4>("lane loads + readlane", din, dout, dcyc, n, 1, expect);
    run<15>("row broadcasts", din, dout, dcyc, n, 1, expect);
    run<15>("row broadcasts", din, dout, dcyc, n, 4, expect);
    run<16>("row broadcasts in fmac", din, dout, dcyc, n, 1, expect);
    run<16>("row broadcasts in fmac", din, dout, dcyc, n, 4, expect);
    run<5>("registers only", din, dout, dcyc, n, 1, expect);
    run<12>("registers only, x16", din, dout, dcyc, n, 1, expect);
    return 0;
}
