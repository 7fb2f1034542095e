// Microbenchmark: what does a wave's OWN non-matrix work cost next to its fp32 MFMAs, and how many waves per SIMD does
// it take to keep the matrix pipe full?  (The co-residency benchmark, coresident.hip, measured the other wave; this
// one measures the issuing wave and the aggregate.)
//   * W waves per SIMD (one workgroup of 4 W waves per CU, 100 KB of LDS so that only one fits), each runs
//     ITER x { 4 x v_mfma_f32_32x32x2_f32 on 4 independent accumulators, each followed by N "filler" instructions }
//   * filler kinds: 0 = independent v_fma_f32, 1 = ds_read_b128 (conflict-free), 2 = global_load_dwordx4 (L2-resident
//     line), 3 = v_exp_f32 (transcendental), 4 = v_fma_f64
//   * output per (kind, N, W): shader cycles per MFMA per wave and the matrix-pipe utilisation of the SIMD
//     (= 64 cycles x MFMAs of all its waves / elapsed cycles)
// Everything inside the loop is inline asm, so the order of issue is the order written.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shadow mfma_shadow.hip && ./mfma_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA(ACC) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b))

template <int KIND, int N>
__device__ __forceinline__ void filler(float (&x)[8], double (&xd)[4], f32x4 (&d)[4], const float a, const float b,
                                       const unsigned lds_addr, const float *gp) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i & 7]) : "v"(a), "v"(b));
        else if constexpr (KIND == 1) asm volatile("ds_read_b128 %0, %1" : "=v"(d[i & 3]) : "v"(lds_addr));
        else if constexpr (KIND == 2) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d[i & 3]) : "v"(gp));
        else if constexpr (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i & 7]));
        else asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(xd[i & 3]) : "v"((double)a));
    }
}

template <int KIND, int N>
__global__ __launch_bounds__(1024) void shadow(const float *in, float *out, unsigned long long *cyc, int iters) {
    extern __shared__ float lds[];
    f32x16 acc0, acc1, acc2, acc3;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
    const float a = in[threadIdx.x & 255], b = in[256 + (threadIdx.x & 255)];
    float x[8];
    double xd[4];
    f32x4 d[4];
    for (int i = 0; i < 8; ++i) x[i] = a * (i + 1);
    for (int i = 0; i < 4; ++i) { xd[i] = a * (i + 1); d[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    lds[threadIdx.x] = a;
    const unsigned lds_addr = (threadIdx.x & 63) * 16;      // 64 lanes x 16 B: conflict-free
    const float *gp = in + (threadIdx.x & 63) * 4;
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (KIND == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MFMA(acc0); filler<KIND, N>(x, xd, d, a, b, lds_addr, gp);
        MFMA(acc1); filler<KIND, N>(x, xd, d, a, b, lds_addr, gp);
        MFMA(acc2); filler<KIND, N>(x, xd, d, a, b, lds_addr, gp);
        MFMA(acc3); filler<KIND, N>(x, xd, d, a, b, lds_addr, gp);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = clock64();
    float res = 0.f;
    for (int r = 0; r < 16; ++r) res += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    for (int i = 0; i < 8; ++i) res += x[i];
    for (int i = 0; i < 4; ++i) res += (float)xd[i] + d[i][0] + d[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int N>
static void run(const float *in, float *out, unsigned long long *cyc, int W) {
    const int iters = 4000, blocks = 256, threads = 256 * W;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(shadow<KIND, N>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL((shadow<KIND, N>), dim3(blocks), dim3(threads), 100 * 1024, 0, in, out, cyc, iters);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
    std::vector<unsigned long long> h(blocks * 4 * W);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0, mx = 0;
    for (auto v : h) { sum += (double)v; mx = mx > (double)v ? mx : (double)v; }
    const double per_wave = sum / h.size();                 // cycles of one wave for iters * 4 MFMAs
    // waves that win the arbitration finish early, so the utilisation is taken from the LAST wave to finish
    printf("  N=%2d W=%d: %7.1f cycles per MFMA per wave (mean), %7.1f (slowest wave): pipe utilisation %.3f\n", N, W,
           per_wave / (iters * 4.0), mx / (iters * 4.0), 64.0 * W * iters * 4.0 / mx);
}

// Two waves per SIMD with different jobs: waves 0-3 (one per SIMD) only issue MFMAs, waves 4-7 only filler
// instructions; each group is timed alone and together.
template <int KIND>
__global__ __launch_bounds__(512) void mix2(const float *in, float *out, unsigned long long *cyc, int itersA, int itersB) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6;
    const float a = in[threadIdx.x & 255], b = in[256 + (threadIdx.x & 255)];
    lds[threadIdx.x] = a;
    const unsigned lds_addr = (threadIdx.x & 63) * 16;
    const float *gp = in + (threadIdx.x & 63) * 4;
    float res = 0.f;
    __syncthreads();
    const unsigned long long t0 = clock64();
    if (wave < 4) {
        f32x16 acc0, acc1, acc2, acc3;
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
        for (int it = 0; it < itersA; ++it) { MFMA(acc0); MFMA(acc1); MFMA(acc2); MFMA(acc3); }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        for (int r = 0; r < 16; ++r) res += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    } else {
        float x[8];
        double xd[4];
        f32x4 d[4];
        for (int i = 0; i < 8; ++i) x[i] = a * (i + 1);
        for (int i = 0; i < 4; ++i) { xd[i] = a * (i + 1); d[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        for (int it = 0; it < itersB; ++it) {
            if constexpr (KIND == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (KIND == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            filler<KIND, 8>(x, xd, d, a, b, lds_addr, gp);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        for (int i = 0; i < 8; ++i) res += x[i];
        for (int i = 0; i < 4; ++i) res += (float)xd[i] + d[i][0] + d[i][3];
    }
    const unsigned long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND>
static void run_mix(const char *name, const float *in, float *out, unsigned long long *cyc) {
    const int blocks = 256, itA = 4000, itB = 4000;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mix2<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    double res[3][2];
    for (int c = 0; c < 3; ++c) {           // 0: MFMA waves alone, 1: filler waves alone, 2: together
        hipLaunchKernelGGL((mix2<KIND>), dim3(blocks), dim3(512), 100 * 1024, 0, in, out, cyc, c == 1 ? 0 : itA, c == 0 ? 0 : itB);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
        std::vector<unsigned long long> h(blocks * 8);
        (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double sa = 0, sb = 0;
        for (int i = 0; i < blocks * 8; ++i) ((i & 7) < 4 ? sa : sb) += (double)h[i];
        res[c][0] = sa / (blocks * 4); res[c][1] = sb / (blocks * 4);
    }
    printf("  %-28s MFMA waves: %6.1f -> %6.1f cycles per MFMA;  filler waves: %6.2f -> %6.2f cycles per instruction\n", name,
           res[0][0] / (itA * 4.0), res[2][0] / (itA * 4.0), res[1][1] / (itB * 8.0), res[2][1] / (itB * 8.0));
}

template <int KIND>
static void sweep(const char *name, const float *in, float *out, unsigned long long *cyc) {
    printf("%s\n", name);
    for (int W = 1; W <= 4; ++W) {
        if (W == 3) continue;
        run<KIND, 0>(in, out, cyc, W);
        run<KIND, 1>(in, out, cyc, W);
        run<KIND, 2>(in, out, cyc, W);
        run<KIND, 4>(in, out, cyc, W);
        if (KIND == 0 || KIND == 3) { run<KIND, 8>(in, out, cyc, W); run<KIND, 16>(in, out, cyc, W); }
    }
}

int main() {
    float hin[512];
    for (int i = 0; i < 512; ++i) hin[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *in, *out;
    unsigned long long *cyc;
    (void)hipMalloc(&in, sizeof(hin));
    (void)hipMemcpy(in, hin, sizeof(hin), hipMemcpyHostToDevice);
    (void)hipMalloc(&out, 256 * 1024 * 4);
    (void)hipMalloc(&cyc, 256 * 16 * 8);
    printf("cycles per v_mfma_f32_32x32x2_f32 (64 = pipe-bound) with N filler instructions issued behind EVERY MFMA, W waves per SIMD\n");
    printf("pure MFMA, W = 1..4\n");
    for (int W = 1; W <= 4; ++W) run<0, 0>(in, out, cyc, W);
    sweep<0>("filler v_fma_f32", in, out, cyc);
    sweep<1>("filler ds_read_b128", in, out, cyc);
    sweep<2>("filler global_load_dwordx4 (L2 hit)", in, out, cyc);
    sweep<3>("filler v_exp_f32", in, out, cyc);
    sweep<4>("filler v_fma_f64", in, out, cyc);
    printf("two waves per SIMD, one issues only MFMAs, the other only filler instructions (alone -> together)\n");
    run_mix<0>("v_fma_f32", in, out, cyc);
    run_mix<1>("ds_read_b128", in, out, cyc);
    run_mix<2>("global_load_dwordx4 (L2 hit)", in, out, cyc);
    run_mix<3>("v_exp_f32", in, out, cyc);
    run_mix<4>("v_fma_f64", in, out, cyc);
    return 0;
}
