// vf_conv_first.h - the first convolution of the encoder (5 x 5, stride 2, on the 3-channel frame) on the vector ALUs.
//
// On the matrix pipe this layer is all overhead: K = 75 (padded to 200 by the 8-channel chunk), 16 - 32 output channels, a
// scalar staging path for the three channels that are not a whole quad, and 128-row items of ~17 us of which 1 us is MFMAs -
// 5 % of a 128 x 128 launch for 0.3 % of its FLOPs (profiles/r05_phase_stats_c5_shard.txt).  Here one item = 16 x 16 output
// pixels of one image, one thread per pixel:
//   1. the 35 x 35 x 3 input patch goes to LDS with dword buffer loads (rows of 105 consecutive floats; padding and the
//      image border through the buffer's out-of-range zeros);
//   2. every thread accumulates its pixel's CO outputs: the patch row of a kernel row is 15 floats read once from LDS, the
//      weights [tap][channel][CO] are wave-uniform and arrive as scalar operands (s_load from the constant address space), so
//      the inner loop is one v_fmac per MAC and nothing else;
//   3. bias, the exact integer LayerNorm partial of the tile (the contract of conv_epilogue<1, EPI_RAW_STATS>: one
//      [sum, sum of squares] pair per (sample, tile) in ConvParams::stats), CO / 4 16-byte stores per thread - sc1 stores and
//      an atomic-store partial when ConvParams::wt_out says the item publishes without a release fence.
// K order per output: (ky, kx, channel) ascending as one fmaf chain from 0, then + bias.  The per-layer launch
// (conv_first_kernel) and the persistent rollout run this same body: the same bits.
#pragma once
#include <hip/hip_runtime.h>
#include "vf_conv_mfma.h"

namespace vf {

constexpr int kFirstK = 5, kFirstStride = 2, kFirstCin = 3;
// LDS floats of a TH x TW tile: the patch (even row stride: every thread's row of 15 floats starts 8-byte aligned) + the
// reduction scratch
__host__ __device__ constexpr int first_patch_stride(const int TW) { return (((TW - 1) * kFirstStride + kFirstK) * kFirstCin + 1) & ~1; }
__host__ __device__ constexpr size_t first_lds_floats(const int TH, const int TW) {
    return (size_t)((TH - 1) * kFirstStride + kFirstK) * first_patch_stride(TW) + 32;
}

template <int CO, class PT>
__device__ __forceinline__ void conv_first_tile(const PT &p, const int bx_, float *smem) {
    static_assert(CO % 4 == 0 && CO <= 32, "first conv: 4 - 32 output channels");
    typedef const __attribute__((address_space(4))) float cfloat;
    const int bx = __builtin_amdgcn_readfirstlane(bx_);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_per_img = p.tilesY * p.tilesX;
    const int b = bx / tiles_per_img, tile_id = bx - b * tiles_per_img;
    const int ty0 = (tile_id / p.tilesX) * p.TH, tx0 = (tile_id % p.tilesX) * p.TW;
    const int PH = (p.TH - 1) * kFirstStride + kFirstK, PW3 = ((p.TW - 1) * kFirstStride + kFirstK) * kFirstCin;
    const int PS = first_patch_stride(p.TW);
    float *s_in = smem;
    long long *red = reinterpret_cast<long long *>(smem + (size_t)PH * PS);     // [4 waves][2] (8-byte aligned: PS is even)

    // ---- 1. the input patch: row r of the patch = image row iy0 + r, floats [ix0 * 3, ix0 * 3 + PW3) of it
    {
        const auto &sg = p.seg[0];
        const int iy0 = ty0 * kFirstStride - p.pad, jx0 = (tx0 * kFirstStride - p.pad) * kFirstCin;
        const int row_f = p.Win * kFirstCin;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(sg.ptr + (long long)b * sg.bstride), 0, (int)((unsigned)(p.Hin * row_f) * 4u), 0x00020000);
        const unsigned magic = 0xFFFFFFFFu / (unsigned)PW3 + 1u;
        const int total = PH * PW3;
        constexpr int U = 8;
        for (int i0 = tid; i0 < total; i0 += kConvThreads * U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * kConvThreads;
                const int r = (int)__umulhi((unsigned)i, magic), j = i - r * PW3;
                const int iy = iy0 + r, jx = jx0 + j;
                const bool ok = i < total && (unsigned)iy < (unsigned)p.Hin && (unsigned)jx < (unsigned)row_f;
                v[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    rsrc, ok ? (unsigned)(iy * row_f + jx) * 4u : 0xFFFFFFFFu, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * kConvThreads;
                if (i < total) {
                    const int r = (int)__umulhi((unsigned)i, magic), j = i - r * PW3;
                    s_in[r * PS + j] = v[u];
                }
            }
        }
    }
    __syncthreads();

    // ---- 2. one thread per output pixel
    const TileDiv div_tw(p.TW);
    const int py = div_tw.div(tid), px = tid - py * p.TW;
    const int y = ty0 + py, x = tx0 + px;
    const bool ok = py < p.TH && y < p.Hout && x < p.Wout;
    float acc[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[co] = 0.f;
    {
        cfloat *w = (cfloat *)(unsigned long long)p.Wp;
        const float *ap = s_in + (ok ? (py * kFirstStride) * PS + px * kFirstStride * kFirstCin : 0);
#pragma unroll 1
        for (int ky = 0; ky < kFirstK; ++ky) {
            float xin[kFirstK * kFirstCin + 1];
            const float *rp = ap + ky * PS;
#pragma unroll
            for (int j = 0; j < (kFirstK * kFirstCin + 1) / 2; ++j) {
                const float2 t = *reinterpret_cast<const float2 *>(rp + 2 * j);
                xin[2 * j] = t.x; xin[2 * j + 1] = t.y;
            }
            cfloat *wr = w + ky * (kFirstK * kFirstCin * CO);
#pragma unroll
            for (int j = 0; j < kFirstK * kFirstCin; ++j)
#pragma unroll
                for (int co = 0; co < CO; ++co) acc[co] = fmaf(xin[j], wr[j * CO + co], acc[co]);
        }
    }

    // ---- 3. bias, statistics, stores
    {
        cfloat *bias = (cfloat *)(unsigned long long)p.bias;
        float vmax = 0.f;
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            acc[co] += bias[co];
            vmax = fmaxf(vmax, fabsf(acc[co]));
        }
        long long ssum = 0, ssq = 0;
        if (__all(vmax < 128.f)) {          // (the float64 fast path of the exact statistics: vf_fused_top.h)
            StatSumD st;
#pragma unroll
            for (int co = 0; co < CO; ++co) st.add(ok ? acc[co] : 0.f);
            ssum = st.sum(); ssq = st.sumsq();
        } else if (ok) {
#pragma unroll
            for (int co = 0; co < CO; ++co) { ssum += stat_q(acc[co]); ssq += stat_q2(acc[co]); }
        }
        const bool wt = p.wt_out != 0;
        const long long img_elems = (long long)p.Hout * p.Wout * CO;
        const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(
            p.out + (long long)b * img_elems, 0, (int)((unsigned)img_elems * 4u), 0x00020000);
        const unsigned off = ok ? (unsigned)((y * p.Wout + x) * CO) * 4u : 0xFFFFFFFFu;
#pragma unroll
        for (int q = 0; q < CO / 4; ++q) {
            const f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
            const unsigned o = ok ? off + 16u * q : off;
            if (wt) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), r_out, o, 0, 16);
            else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), r_out, o, 0, 0);
        }
        const long long wsum = wave_sum(ssum), wsq = wave_sum(ssq);
        if (lane == 0) { red[2 * wave] = wsum; red[2 * wave + 1] = wsq; }
        __syncthreads();
        if (tid == 0) {
            long long su = 0, sq = 0;
            for (int w = 0; w < 4; ++w) { su += red[2 * w]; sq += red[2 * w + 1]; }
            long long *dst = p.stats + ((long long)b * p.stats_nparts + tile_id) * 2;
            if (wt) {
                __hip_atomic_store(dst, su, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(dst + 1, sq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                dst[0] = su; dst[1] = sq;
            }
        }
    }
}

template <int CO>
VF_GLOBAL VF_LAUNCH_BOUNDS(kConvThreads, 2) void conv_first_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    conv_first_tile<CO>(p, blockIdx.x, smem);
}

}  // namespace vf
