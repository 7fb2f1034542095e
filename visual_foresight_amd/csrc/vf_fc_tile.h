// vf_fc_tile.h - the CDNA kernel FC of the persistent rollout as ONE item per (row tile, K split).
//
// [B, 8192] x [8192, 250]: rows = samples, K = the flattened bottleneck state h5 (LayerNorm applied while staging), split 32
// ways over K (deterministic combine in cdna_finalize).  The generic tile (conv_tile<1, EPI_PARTIAL, 2>) cuts the 250
// columns into 8 groups of 32 and runs one item per (group, split): every one of the 8 items of a split stages the SAME
// 256 rows x 256 channels - 100 of an item's 134 us at 200 samples were prologue + staging (profiles/r04_cu_trace_200.txt),
// 0.87 ms per slot and launch; at 25 samples the 256 items per step (76 us each, nine tenths of their rows empty) were 4 % of
// all slot time and sat in the ticket order in front of every sample's decoder.  Here a workgroup keeps all 8 column groups
// (8 accumulator tiles per wave: wave w = rows 32 w .. 32 w + 31 of a 128-row tile) and stages a split's operand ONCE:
// 32 x ceil(B / 128) items per step instead of 256 x ceil(B / 256), 1/16 (B <= 128) to 1/8 of the staging work.
// Same packed weights, same (chunk, k8, j) order per output as the generic tile: the same bits.
#pragma once
#include <hip/hip_runtime.h>
#include "vf_conv_mfma.h"

namespace vf {

constexpr int kFcRows = 128;            // GEMM rows (samples) per item
constexpr int kFcGroups = 8;            // column groups of 32 per item (250 columns)
__host__ __device__ inline size_t fc_wide_lds_bytes() { return (size_t)(kFcRows * 36 + 2 * kFcRows) * 4 + 64; }

// p: the FC's ConvParams (one 1x1 "image" per sample, KC = 32, ncg = kFcGroups, chunks_per_split); bx = row tile, bz = split
template <class PT>
__device__ __forceinline__ void fc_wide_tile(const PT &p, const int bx_, const int bz_, float *smem) {
    const int bx = __builtin_amdgcn_readfirstlane(bx_), bz = __builtin_amdgcn_readfirstlane(bz_);
    constexpr int KCpad = 36, G = kFcGroups;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, kh = lane >> 5;
    float *lnTab = smem + kFcRows * KCpad;          // [rows][2]: mean, rstd
    const auto &sg = p.seg[0];
    const int b0 = bx * kFcRows;
    const int n_here = min(kFcRows, p.B - b0);

    // ---- LayerNorm statistics of this tile's samples: one thread per row (exact integer partials: any order, same bits)
    if (tid < kFcRows) {
        float mean = 0.f, rstd = 1.f;
        if (sg.ln_part && tid < n_here) {
            long long su = 0, sq = 0;
            const long long *pp = sg.ln_part + (long long)(b0 + tid) * sg.ln_bstride;
            for (int k = 0; k < sg.ln_nparts; ++k) { su += pp[2 * k]; sq += pp[2 * k + 1]; }
            ln_from_totals(su, sq, sg.ln_inv_n, mean, rstd);
        }
        lnTab[2 * tid] = mean; lnTab[2 * tid + 1] = rstd;
    }

    f32x16 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

    const int ncg = p.ncg;              // column groups of this layer (<= kFcGroups)
    const int Ntot = ncg * 32;
    const unsigned w_loff = (unsigned)(((kh * Ntot + n) * 4) * 4);          // this lane's column inside a (chunk, k8) block
    const unsigned wstep_b = (unsigned)(2 * Ntot * 4) * 4u;                 // bytes per (chunk, k8) block
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.Wp), 0, 0x7FFFFFFF, 0x00020000);
    const int ch_begin = bz * p.chunks_per_split, ch_end = min(ch_begin + p.chunks_per_split, sg.nchunk);
    auto load_b = [&](f32x4 (&D_)[G], const int ci, const int k8) {
        const unsigned so_ = (unsigned)(min(ci, sg.nchunk - 1) * 4 + k8) * wstep_b;
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (g < ncg)        // (fewer than eight column groups - arch 2: 150 columns - leave the last tiles idle)
                D_[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_loff, so_ + 512u * g, 0));
    };

    // ---- staging: thread = (row lane tid / 8, channel quad tid % 8); rows tid / 8 + 32 u
    const int q = tid & 7, pl = tid >> 3;
    const unsigned row_bytes = (unsigned)sg.bstride * 4u;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(sg.ptr + (long long)b0 * sg.bstride), 0,
        n_here > 0 ? (int)((unsigned)(n_here - 1) * row_bytes + (unsigned)sg.C * 4u) : 0, 0x00020000);
    const f32x4 *smem4 = reinterpret_cast<const f32x4 *>(smem);
    const int a4 = ((wave * 32 + n) * KCpad + kh * 4) >> 2;                 // this lane's A row, in float4 units

    f32x4 bP[G], bQ[G];
    for (int ci = ch_begin; ci < ch_end; ++ci) {
        const int c = ci * 32 + 4 * q;              // first channel of this thread's quad
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = pl + 32 * u;
            v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                a_rsrc, row < n_here ? (unsigned)row * row_bytes + (unsigned)c * 4u : 0xFFFFFFFFu, 0, 0));
        }
        load_b(bP, ci, 0);                          // the chunk's first weight block flies during the staging
        f32x4 gq = {1.f, 1.f, 1.f, 1.f}, bq = {0.f, 0.f, 0.f, 0.f};
        if (sg.ln_part) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cc = (c + j) % sg.gamma_mod;
                gq[j] = sg.gamma[cc]; bq[j] = sg.beta[cc];
            }
        }
        __syncthreads();                            // previous chunk consumed (first chunk: lnTab visible)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = pl + 32 * u;
            if (row < n_here) {
                if (sg.ln_part) {
                    const float mean = lnTab[2 * row], rstd = lnTab[2 * row + 1];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[u][j] = fmaf((v[u][j] - mean) * rstd, gq[j], bq[j]);
                }
                if (sg.relu) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[u][j] = fmaxf(v[u][j], 0.f);
                }
            }
            *reinterpret_cast<f32x4 *>(smem + row * KCpad + 4 * q) = v[u];
        }
        __syncthreads();
        // ---- K loop of the chunk: 4 k8 steps x 4 j x 8 column groups; the next step's weights one step ahead
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
            f32x4 (&cur)[G] = (k8 & 1) ? bQ : bP;
            f32x4 (&nxt)[G] = (k8 & 1) ? bP : bQ;
            const f32x4 a = smem4[a4 + k8 * 2];
            if (k8 < 3) load_b(nxt, ci, k8 + 1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < G; ++g)
                    if (g < ncg) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], cur[g][j], acc[g], 0, 0, 0);
        }
    }

    // ---- raw accumulators of this split: out[split][B][n_valid]
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int ch = g * 32 + n;
        if (ch >= p.n_valid) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (row < n_here) p.out[((long long)bz * p.B + b0 + row) * p.n_valid + ch] = acc[g][r];
        }
    }
}

}  // namespace vf
