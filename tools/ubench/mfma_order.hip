// Microbenchmark: in which order do the fp32 MFMA shapes accumulate their K products?
// The engine's bit-identity across tile plans rests on v_mfma_f32_32x32x2_f32 being acc = fma(a1, b1, fma(a0, b0, acc));
// this checks that, and asks the same of v_mfma_f32_16x16x4_f32 (K = 4 per instruction), the candidate for
// finer-grained conv-LSTM tiles at small batches: is it the plain chain k = 0, 1, 2, 3?
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o mfma_order mfma_order.hip && ./mfma_order
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A [32][K], B [K][32] -> C [32][32] with 32x32x2 steps; lane l: row / col = l % 32, k = l / 32
__global__ void k32(const float *A, const float *B, float *C, int K) {
    const int l = threadIdx.x, n = l & 31, kh = l >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k = 0; k < K; k += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[n * K + k + kh], B[(k + kh) * 32 + n], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + n] = acc[r];
}
// A [16][K], B [K][16] -> C [16][16] with 16x16x4 steps; lane l: row / col = l % 16, k = l / 16
__global__ void k16(const float *A, const float *B, float *C, int K) {
    const int l = threadIdx.x, n = l & 15, kq = l >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; k += 4)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[n * K + k + kq], B[(k + kq) * 16 + n], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[(4 * kq + r) * 16 + n] = acc[r];
}

int main() {
    const int K = 4800;
    float *hA = new float[32 * K], *hB = new float[K * 32], hC[32 * 32];
    srand(7);
    for (int i = 0; i < 32 * K; ++i) { hA[i] = (float)rand() / RAND_MAX * 2.f - 1.f; hB[i] = (float)rand() / RAND_MAX * 2.f - 1.f; }
    float *A, *B, *C;
    hipMalloc(&A, 32 * K * 4); hipMalloc(&B, K * 32 * 4); hipMalloc(&C, 32 * 32 * 4);
    // ---- 32x32x2
    hipMemcpy(A, hA, 32 * K * 4, hipMemcpyHostToDevice); hipMemcpy(B, hB, K * 32 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, A, B, C, K);
    hipMemcpy(hC, C, sizeof(hC), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            float acc = 0.f;
            for (int k = 0; k < K; ++k) acc = fmaf(hA[i * K + k], hB[k * 32 + j], acc);
            bad += memcmp(&acc, &hC[i * 32 + j], 4) != 0;
        }
    printf("32x32x2: %d of 1024 outputs differ from the sequential fmaf chain\n", bad);
    // ---- 16x16x4: rows 0..15 of A, columns 0..15 of B (re-packed [K][16])
    float *hB16 = new float[K * 16];
    for (int k = 0; k < K; ++k) for (int j = 0; j < 16; ++j) hB16[k * 16 + j] = hB[k * 32 + j];
    hipMemcpy(B, hB16, K * 16 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, A, B, C, K);
    hipMemcpy(hC, C, 16 * 16 * 4, hipMemcpyDeviceToHost);
    int bad_seq = 0, bad_tree = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float acc = 0.f, acc2 = 0.f;
            for (int k = 0; k < K; ++k) acc = fmaf(hA[i * K + k], hB16[k * 16 + j], acc);
            for (int k = 0; k < K; k += 4) {        // alternative: pairwise inside the instruction
                const float p = fmaf(hA[i * K + k + 1], hB16[(k + 1) * 16 + j], hA[i * K + k] * hB16[k * 16 + j]);
                const float q = fmaf(hA[i * K + k + 3], hB16[(k + 3) * 16 + j], hA[i * K + k + 2] * hB16[(k + 2) * 16 + j]);
                acc2 = acc2 + (p + q);
            }
            bad_seq += memcmp(&acc, &hC[i * 16 + j], 4) != 0;
            bad_tree += memcmp(&acc2, &hC[i * 16 + j], 4) != 0;
        }
    printf("16x16x4: %d of 256 outputs differ from the sequential fmaf chain, %d from a pairwise tree\n", bad_seq, bad_tree);
    return 0;
}
