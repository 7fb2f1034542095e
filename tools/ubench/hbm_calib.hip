// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on this chip for the access patterns of the rollout kernel
// (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access pattern before trusting an absolute").
// Every kernel moves exactly 1 GiB (far beyond the 256 MiB Infinity Cache) with one pattern:
//   w_row128   the conv-LSTM epilogue store: one dword per lane, a wave instruction covers two 128-byte pixel rows
//              512 bytes apart (lanes 0-31 / 32-63), 16 instructions fill a 4 KiB block (32 rows)
//   w_x4       16 bytes per lane, a wave instruction covers 1 KiB contiguous (staging-style)
//   r_x4       16 bytes per lane, 1 KiB contiguous per wave instruction (the staging loads)
//   r_row128   one dword per lane, two 128-byte rows per wave instruction (the cell-state load of the epilogue)
// Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (tools/pmc_calib.sh) and compare with 1 GiB.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr size_t kBytes = 1ull << 30;

__global__ __launch_bounds__(256) void w_row128(float *p) {
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, n = lane & 31, kh = lane >> 5;
    float *base = p + wave * 1024;                          // 32 rows x 128 B = 4 KiB per wave
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;   // the MFMA 32x32 C layout of the epilogue
        base[row * 32 + n] = (float)r;
    }
}
__global__ __launch_bounds__(256) void w_x4(f32x4 *p) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    p[i] = f32x4{1.f, 2.f, 3.f, 4.f};
}
__global__ __launch_bounds__(256) void r_x4(const f32x4 *p, float *sink) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const f32x4 v = p[i];
    if (v[0] + v[1] + v[2] + v[3] == 12345.f) sink[0] = 1.f;
}
__global__ __launch_bounds__(256) void r_row128(const float *p, float *sink) {
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, n = lane & 31, kh = lane >> 5;
    const float *base = p + wave * 1024;
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += base[((r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + n];
    if (s == 12345.f) sink[0] = 1.f;
}

int main() {
    float *buf, *sink;
    if (hipMalloc(&buf, kBytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 0, kBytes);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(w_row128, dim3(kBytes / 4096 / 4), dim3(256), 0, 0, buf);
        hipLaunchKernelGGL(w_x4, dim3(kBytes / 16 / 256), dim3(256), 0, 0, reinterpret_cast<f32x4 *>(buf));
        hipLaunchKernelGGL(r_x4, dim3(kBytes / 16 / 256), dim3(256), 0, 0, reinterpret_cast<const f32x4 *>(buf), sink);
        hipLaunchKernelGGL(r_row128, dim3(kBytes / 4096 / 4), dim3(256), 0, 0, buf, sink);
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("failed\n"); return 1; }
    printf("ok: 4 kernels x 3 launches, 1 GiB each\n");
    return 0;
}
