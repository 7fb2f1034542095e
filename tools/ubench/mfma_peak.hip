// Microbenchmark: sustained v_mfma_f32_32x32x2_f32 rate with random operands, no memory traffic.
// hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(const float *in, float *out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a += 1e-9f;
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float *in, *out;
    float hin[512];
    for (int i = 0; i < 512; ++i) hin[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMalloc(&in, sizeof(hin));
    hipMemcpy(in, hin, sizeof(hin), hipMemcpyHostToDevice);
    hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
        const int grid = 256 * wgs_per_cu, iters = 20000;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop<8>, dim3(grid), dim3(256), 0, 0, in, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double flops = 2.0 * 32 * 32 * 2 * 8.0 * iters * 4 * grid;
            printf("wgs/CU %d rep %d: %.3f ms  %.1f TFLOP/s (long run: %.0f ms of MFMA)\n", wgs_per_cu, rep, ms,
                   flops / ms / 1e9, ms);
        }
    }
    return 0;
}
