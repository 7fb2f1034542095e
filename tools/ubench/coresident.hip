// Microbenchmark: how much does a NON-matrix workgroup slow down when the other workgroup of its CU runs a
// back-to-back fp32 MFMA loop (the situation of every prologue / epilogue / light item of the persistent rollout)?
// Two workgroups per CU (LDS-limited), roles by order of arrival on the CU: the first is the "matrix" workgroup
// (v_mfma_f32_32x32x2_f32, 4 independent accumulators, until the victim is done), the second the "victim", which
// times one of four bodies:  0 = VALU fma chain,  1 = transcendental chain (exp, as in the gate math),
// 2 = dependent global loads (pointer chase, L2-resident),  3 = LDS read-modify-write chain.
// Variants: matrix workgroup idle / busy, victim priority 0 / 2 / 3, matrix loop with an s_nop window every K MFMAs.
//   hipcc --offload-arch=gfx950 -O3 -o coresident coresident.hip && ./coresident
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Ctl {
    int arrivals[8 * 256];      // per (XCC, CU) arrival counters
    int done[8 * 256];          // victims finished on that CU
    unsigned long long ticks[8 * 256];
    unsigned long long mfma_ticks[8 * 256];     // matrix workgroup: time of its loop
    unsigned long long mfma_count[8 * 256];     //                   MFMAs per wave issued in that time
};

template <int BODY>
__device__ float victim_body(const int *chase, float *lds, int iters) {
    float x = 1.0f + threadIdx.x * 1e-3f;
    if (BODY == 0) {
        float a = 1.0001f, b = 0.9999f, c = 0.5f, d = 0.25f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) { x = fmaf(x, a, b); c = fmaf(c, b, a); d = fmaf(d, a, c); a = fmaf(a, 0.99999f, 1e-6f); }
        }
        x += c + d + a;
    } else if (BODY == 1) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) x = 1.0f / (1.0f + __expf(-x));
        }
    } else if (BODY == 2) {
        int p = threadIdx.x;
        for (int i = 0; i < iters; ++i) p = chase[p];
        x = (float)p;
    } else {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { lds[threadIdx.x] = x; x = lds[(threadIdx.x + 33) & 255] + 1.0f; }
        }
    }
    return x;
}

template <int BODY, int PACE>
__global__ __launch_bounds__(256, 2) void mix(Ctl *ctl, const int *chase, const float *in, float *out, int iters,
                                               int matrix_on, int victim_prio) {
    extern __shared__ float lds[];
    __shared__ int s_role, s_slot;
    if (threadIdx.x == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        s_slot = (int)((xcc & 7) * 256u + ((hwid >> 8) & 255u));
        s_role = atomicAdd(&ctl->arrivals[s_slot], 1) & 1;         // 0: matrix, 1: victim
    }
    __syncthreads();
    const int slot = s_slot;
    float res = 0.f;
    if (s_role == 0) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        float a = in[threadIdx.x], b = in[threadIdx.x + 256];
        __builtin_amdgcn_s_setprio(0);
        if (matrix_on) {
            int guard = 0;
            const unsigned long long m0 = wall_clock64();
            // pace: 0 = back to back; 1..4 = that many `s_nop 15` after EVERY MFMA; 9 = `s_sleep 1` after every MFMA;
            // 16 / 4 (legacy) = a 64-cycle s_nop window every 16 / 4 loop iterations
            while (__hip_atomic_load(&ctl->done[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && guard < 4000000) {
                for (int it = 0; it < 64; ++it) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                        if constexpr (PACE == 1) asm volatile("s_nop 15");
                        else if constexpr (PACE == 2) asm volatile("s_nop 15\n\ts_nop 15");
                        else if constexpr (PACE == 3) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15");
                        else if constexpr (PACE == 5) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 7");
                        else if constexpr (PACE == 9) asm volatile("s_sleep 1");
                    }
                    if constexpr (PACE == 16) { if ((it & 3) == 0) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"); }
                }
                a += 1e-9f;
                guard += 64;
            }
            if (threadIdx.x == 0) { ctl->mfma_ticks[slot] = wall_clock64() - m0; ctl->mfma_count[slot] = (unsigned long long)guard * 4; }
        }
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) res += acc[i][r];
    } else {
        if (victim_prio == 2) __builtin_amdgcn_s_setprio(2);
        if (victim_prio == 3) __builtin_amdgcn_s_setprio(3);
        __syncthreads();
        const unsigned long long t0 = wall_clock64();
        res = victim_body<BODY>(chase, lds, iters);
        __syncthreads();
        const unsigned long long t1 = wall_clock64();
        if (threadIdx.x == 0) {
            ctl->ticks[slot] = t1 - t0;
            __hip_atomic_store(&ctl->done[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = res;
}

static double g_rate = 0.0;    // matrix workgroup's MFMA rate of the last run (MFMAs per wave per microsecond)

template <int BODY, int PACE>
static double run(Ctl *ctl, const int *chase, const float *in, float *out, int iters, int matrix_on, int prio) {
    hipMemset(ctl, 0, sizeof(Ctl));
    // 70 KB of dynamic LDS: exactly two workgroups per CU
    hipFuncSetAttribute(reinterpret_cast<const void *>(mix<BODY, PACE>), hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024);
    hipLaunchKernelGGL((mix<BODY, PACE>), dim3(512), dim3(256), 70 * 1024, 0, ctl, chase, in, out, iters, matrix_on, prio);
    hipDeviceSynchronize();
    std::vector<unsigned long long> t(8 * 256);
    std::vector<int> arr(8 * 256);
    hipMemcpy(t.data(), ctl->ticks, t.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(arr.data(), ctl->arrivals, arr.size() * 4, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> mt(8 * 256), mc(8 * 256);
    hipMemcpy(mt.data(), ctl->mfma_ticks, mt.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(mc.data(), ctl->mfma_count, mc.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0, rate = 0;
    int n = 0, nr = 0;
    for (size_t i = 0; i < t.size(); ++i)
        if (arr[i] == 2 && t[i]) {
            sum += (double)t[i]; ++n;
            if (mt[i]) { rate += (double)mc[i] / ((double)mt[i] * 0.01); ++nr; }       // MFMAs per wave per microsecond
        }
    g_rate = nr ? rate / nr : 0.0;
    return n ? sum / n * 0.01 : -1.0;       // wall_clock64: 100 MHz -> microseconds
}

int main() {
    Ctl *ctl;
    hipMalloc(&ctl, sizeof(Ctl));
    std::vector<int> hch(256);
    for (int i = 0; i < 256; ++i) hch[i] = (i * 37 + 11) & 255;
    int *chase;
    hipMalloc(&chase, 1024);
    hipMemcpy(chase, hch.data(), 1024, hipMemcpyHostToDevice);
    float hin[512];
    for (int i = 0; i < 512; ++i) hin[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *in, *out;
    hipMalloc(&in, sizeof(hin));
    hipMemcpy(in, hin, sizeof(hin), hipMemcpyHostToDevice);
    hipMalloc(&out, 512 * 256 * 4);
    const char *names[4] = {"VALU fma chain", "exp / rcp chain", "dependent L2 loads", "LDS write-read chain"};
    const int iters[4] = {2000, 2000, 4000, 2000};
    printf("victim time in microseconds (mean over CUs that hosted exactly one matrix and one victim workgroup)\n");
    printf("(in brackets: MFMAs per microsecond per wave of the matrix workgroup; 16-pass MFMA at 2.4 GHz = 37.5 at most)\n");
    printf("%-22s %8s %14s %14s %14s %14s %14s %14s %14s %14s\n", "victim body", "alone", "mfma p0", "mfma p2", "p2 nop15x1",
           "p2 nop15x2", "p2 nop15x3", "p2 nop 56", "p2 s_sleep1", "p2 win/16");
    auto row = [&](auto bc, int b) {
        constexpr int B = decltype(bc)::value;
        double r[9], q[9];
        r[0] = run<B, 0>(ctl, chase, in, out, iters[b], 0, 0); q[0] = 0;
        r[1] = run<B, 0>(ctl, chase, in, out, iters[b], 1, 0); q[1] = g_rate;
        r[2] = run<B, 0>(ctl, chase, in, out, iters[b], 1, 2); q[2] = g_rate;
        r[3] = run<B, 1>(ctl, chase, in, out, iters[b], 1, 2); q[3] = g_rate;
        r[4] = run<B, 2>(ctl, chase, in, out, iters[b], 1, 2); q[4] = g_rate;
        r[5] = run<B, 3>(ctl, chase, in, out, iters[b], 1, 2); q[5] = g_rate;
        r[6] = run<B, 5>(ctl, chase, in, out, iters[b], 1, 2); q[6] = g_rate;
        r[7] = run<B, 9>(ctl, chase, in, out, iters[b], 1, 2); q[7] = g_rate;
        r[8] = run<B, 16>(ctl, chase, in, out, iters[b], 1, 2); q[8] = g_rate;
        printf("%-22s %8.1f", names[b], r[0]);
        for (int c = 1; c < 9; ++c) printf(" %7.1f (%4.1f)", r[c], q[c]);
        printf("\n");
    };
    row(std::integral_constant<int, 0>{}, 0);
    row(std::integral_constant<int, 1>{}, 1);
    row(std::integral_constant<int, 2>{}, 2);
    row(std::integral_constant<int, 3>{}, 3);
    return 0;
}
