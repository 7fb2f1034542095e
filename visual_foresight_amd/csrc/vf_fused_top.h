// vf_fused_top.h - the top transposed convolution of the decoder fused with the compositing of the next frame.
//
// Unfused, the top decoder tensor (64x64x32 floats per sample-step; 128x128x32 in arch 1) is written by the
// transposed conv and read back by the compositing tiles: 1 - 4 MB per sample-step of HBM traffic and two items
// with their fixed costs.  Fused, one item = one transposed-conv tile (128 input pixels -> 512 output pixels):
//   1. the K loop of conv_tile<4, EPI_CONVT_FUSED> leaves the tile in the accumulators;
//   2. bias, the tile's exact LayerNorm partial (integers), published as write-through atomic stores + a per-sample counter;
//   3. the haloed previous frame / distributions of the tile's output region and the sample's CDNA kernels go to LDS;
//   4. the item WAITS until the sample's other tiles have published their partials (they are adjacent tickets of the
//      same queue, drawn within microseconds of each other): LayerNorm needs the statistics of the whole image;
//   5. the outputs go through LDS ([256 px][36]: lane = channel on the way in, thread = pixel on the way out), four
//      cost-sum blocks (4 rows x 16 columns, one per wave) at a time, and composite_pixel() - the very code of the
//      stand-alone compositing tile - finishes every pixel.
// Deadlock: a waiting item only waits for tiles of its own sample; a queue hands tickets out in order, so all
// but the last group of every queue is completely drawn and finishes; at most kQueues x (tiles - 1) workgroups can
// therefore be waiting for undrawn mates, far fewer than are resident.  The wait is bounded like every other one.
// Bit-identity with the unfused path: same K loop, same bias add, same statistics, same composite_pixel on the same
// floats, cost sums per block in the same lane order (the per-layer launches stay unfused and are compared in tests).
#pragma once
#include <hip/hip_runtime.h>
#include "vf_conv_mfma.h"
#include "vf_small_kernels.h"

namespace vf {

constexpr unsigned kFusedSpinLimit = 1u << 24;
constexpr int kFusedCtlGoal = 8;        // == kCtlGoal of vf_persistent.h (checked there)

// LDS floats the fused epilogue needs for a tile of TH x TW input pixels (host + device)
__host__ __device__ inline size_t fused_top_lds_floats(int TH, int TW, int ND) {
    const size_t halo = (size_t)(2 * TH + 4) * (2 * TW + 4);
    return (size_t)256 * kCompEncPad + halo * comp_px_stride(ND) + (size_t)kTaps * kCompKernPad + 16 + 64;
}

template <int ND, bool FIRST, int K, class PT, class CT>
__device__ __forceinline__ void fused_top_body(const PT &p, const CT &c, f32x16 (&acc)[1][4], const int bx,
                                               long long *red, float *smem, const int *goal) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, kh = lane >> 5;
    const int tiles_per_img = p.tilesY * p.tilesX;
    const int b = bx / tiles_per_img, tile_id = bx - b * tiles_per_img;
    const int ty0 = (tile_id / p.tilesX) * p.TH, tx0 = (tile_id % p.tilesX) * p.TW;    // input coordinates
    const TileDiv div_tw(p.TW);

    [[maybe_unused]] const unsigned long long tf0 = VF_TS_NOW();
    // ---- 2. bias + exact statistics of this tile (as conv_epilogue<4, EPI_CONVT_RAW_STATS>)
    float bias_g[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias_g[g] = p.bias[g * 32 + n];
    long long ssum = 0, ssq = 0;
    float vmax = 0.f;
    unsigned okmask = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        const int yy = div_tw.div(row), xx = row - yy * p.TW;
        const bool ok = row < p.TH * p.TW && ty0 + yy < p.Hout && tx0 + xx < p.Wout;
        okmask |= ok ? (1u << r) : 0u;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float v = acc[0][g][r] + bias_g[g];
            acc[0][g][r] = v;
            vmax = fmaxf(vmax, fabsf(v));
        }
    }
    // The exact statistics are the integers trunc(v 2^32), trunc(v^2 2^32).  While every |v| of the wave is below 128 the
    // lane's 64 terms can be added as float64 integers (each below 2^46, the sum below 2^53: exact) and converted once -
    // 8 instead of ~24 instructions per value; a wave with a larger value (never seen) takes the per-value conversion.
    // Either way the same integers.
    if (__all(vmax < 128.f)) {
        StatSumD st;
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int g = 0; g < 4; ++g) st.add((okmask >> r) & 1u ? acc[0][g][r] : 0.f);
        ssum = st.sum(); ssq = st.sumsq();
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if ((okmask >> r) & 1u) { ssum += stat_q(acc[0][g][r]); ssq += stat_q2(acc[0][g][r]); }
    }
    const long long wsum = wave_sum(ssum), wsq = wave_sum(ssq);
    __syncthreads();                        // the operand tile is dead; `red` lies behind it
    if (lane == 0) { red[2 * wave] = wsum; red[2 * wave + 1] = wsq; }
    __syncthreads();
    if (tid == 0) {
        long long su = 0, sq = 0;
        for (int w = 0; w < 4; ++w) { su += red[2 * w]; sq += red[2 * w + 1]; }
        long long *dst = p.stats + ((long long)b * p.stats_nparts + tile_id) * 2;
        // The partial is the only thing the mates need from this tile: it leaves as two agent-scope atomic (write-through)
        // stores, drained before the counter is bumped - no release fence, which would write back whatever the other
        // workgroups of this XCD have dirtied in L2 (the frames of their compose passes) for 1.7 - 6.5 us (CDNA guide,
        // section 6 G16, recipe R1; the mates read the partials with agent-scope atomic loads)
        __hip_atomic_store(dst, su, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 1, sq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(p.fuse_ready + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();                        // red is dead from here on: the LDS below may cover it
    [[maybe_unused]] const unsigned long long tf1 = VF_TS_NOW();
    VF_TRACE_EVT(TR_TOP_STATS);

    // ---- LDS layout of the compositing part
    const int RH = 2 * p.TH, RW = 2 * p.TW;             // output region of this tile
    const int HW_ = RW + 4, HH_ = RH + 4;               // its halo tile
    const int oy0 = 2 * ty0, ox0 = 2 * tx0;
    float *s_enc = smem;                                 // [256][kCompEncPad]
    constexpr int PS = comp_px_stride(ND);
    float *s_px = s_enc + 256 * kCompEncPad;             // [HH_ * HW_][PS]: frame, distributions (composite_pixel)
    float *s_kern = s_px + HH_ * HW_ * PS;               // [kTaps][kCompKernPad]
    float *s_ln = s_kern + kTaps * kCompKernPad;         // [2]
    float *s_dscale = s_ln + 2;                          // [ND]
    int *s_flag = reinterpret_cast<int *>(s_dscale + ND + 1);
    const int nblocks = sum_blocks(c.H, c.W);

    // ---- 3. scale of the previous distributions, CDNA kernels, halo.  Three independent groups of loads: all of them are
    // requested before the first is used (the halo of up to 1024 pixels sits in registers across the reduction of the
    // block sums and the barrier), so the item waits for ONE memory latency here instead of three or more
    const float *pf = c.prev_frame + (long long)b * c.prev_frame_bstride;
    const float *pd = c.prev_distrib + (long long)b * c.prev_distrib_bstride;
    const TileDiv div_hw(HW_);
    constexpr int kHaloU = 4;
    float hv[kHaloU][3 + ND];
    unsigned hin = 0;
    auto halo_request = [&](const int i0) {
        hin = 0;
#pragma unroll
        for (int u = 0; u < kHaloU; ++u) {
            const int i = i0 + u * 256;
            const int ly = div_hw.div(i), lx = i - ly * HW_;
            const int y = oy0 + ly - 2, x = ox0 + lx - 2;
            const bool in = i < HH_ * HW_ && y >= 0 && y < c.H && x >= 0 && x < c.W;
            const long long o = (long long)y * c.W + x;
            hin |= in ? (1u << u) : 0u;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) hv[u][ch] = in ? pf[o * 3 + ch] : 0.f;
#pragma unroll
            for (int d = 0; d < ND; ++d) hv[u][3 + d] = in ? pd[o * ND + d] : 0.f;
        }
    };
    auto halo_store = [&](const int i0) {
#pragma unroll
        for (int u = 0; u < kHaloU; ++u) {
            const int i = i0 + u * 256;
            if (i < HH_ * HW_) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) s_px[i * PS + ch] = hv[u][ch];
#pragma unroll
                for (int d = 0; d < ND; ++d) s_px[i * PS + 3 + d] = (hin >> u) & 1u ? hv[u][3 + d] * s_dscale[d] : 0.f;
            }
        }
    };
    double dsum[(ND + 3) / 4];
#pragma unroll
    for (int j = 0; j < (ND + 3) / 4; ++j) {
        dsum[j] = 0.0;
        const int d = ((wave + 3) & 3) + 4 * j;
        if (d < ND && c.prev_sums) {
            const double *pp = c.prev_sums + ((long long)b * ND + d) * nblocks * 2;
            for (int k = lane; k < nblocks; k += 64) dsum[j] += pp[2 * k];
        }
    }
    float kv[(kTaps * K + 255) / 256];
#pragma unroll
    for (int j = 0; j < (kTaps * K + 255) / 256; ++j) {
        const int i = tid + 256 * j;
        kv[j] = i < kTaps * K ? c.kern[(long long)b * kTaps * K + i] : 0.f;
    }
    halo_request(tid);
#pragma unroll
    for (int j = 0; j < (ND + 3) / 4; ++j) {
        const int d = ((wave + 3) & 3) + 4 * j;
        if (d < ND) {
            float sc = 1.0f;
            if (c.prev_sums) sc = (float)(1.0 / wave_sum(dsum[j]));
            if (lane == 0) s_dscale[d] = sc;
        }
    }
#pragma unroll
    for (int j = 0; j < (kTaps * K + 255) / 256; ++j) {
        const int i = tid + 256 * j;
        if (i < kTaps * K) s_kern[(i / K) * kCompKernPad + i % K] = kv[j];
    }
    __syncthreads();
    halo_store(tid);
    for (int i0 = tid + kHaloU * 256; i0 < HH_ * HW_; i0 += kHaloU * 256) { halo_request(i0); halo_store(i0); }

    [[maybe_unused]] const unsigned long long tf2 = VF_TS_NOW();
    VF_TRACE_EVT(TR_TOP_HALO);
    // ---- 4. wait for the sample's other tiles, then the LayerNorm of the whole image
    if (wave == 0) {
        unsigned spins = 0;
        int ok = 1;
        if (lane == 0) {
            while (__hip_atomic_load(p.fuse_ready + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < tiles_per_img) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > kFusedSpinLimit ||
                    __hip_atomic_load(p.fuse_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = 0; break; }
            }
            VF_ACQUIRE_AGENT();
            *s_flag = ok;
        }
        __builtin_amdgcn_wave_barrier();
        ok = __shfl(ok, 0, 64);
        if (ok) {
            long long su = 0, sq = 0;
            const long long *pp = p.stats + (long long)b * p.stats_nparts * 2;
            for (int k = lane; k < tiles_per_img; k += 64) {
                su += __hip_atomic_load(pp + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sq += __hip_atomic_load(pp + 2 * k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            su = wave_sum(su); sq = wave_sum(sq);
            if (lane == 0) ln_from_totals(su, sq, c.ln_inv_n, s_ln[0], s_ln[1]);
        }
    }
    __syncthreads();
    if (*s_flag == 0) {                      // a mate never arrived: raise the launch's failure word and give up;
        if (tid == 0) {                      // word 1 of the control block tells the scheduler loop not to publish
            extern __shared__ __attribute__((aligned(16))) float smem_all[];
            atomicExch(const_cast<int *>(p.fuse_status), 1);
            reinterpret_cast<int *>(smem_all)[1] = 0;
        }
        return;
    }
    const float mean = s_ln[0], rstd = s_ln[1];
    [[maybe_unused]] const unsigned long long tf3 = VF_TS_NOW();
    VF_TRACE_EVT(TR_TOP_MATES);

    // ---- 5. four blocks (4 rows x 16 columns, one per wave) at a time through LDS
    const int nbx = RW / kSumBlockW, nby = RH / kSumBlockH, nblk = nbx * nby;
    for (int pass = 0; pass * 4 < nblk; ++pass) {
        if (pass) __syncthreads();           // the previous pass has read s_enc
        // (the LDS slots of the 64 accumulators do not depend on the pass; hoisted out of this loop they would stay in
        // registers across composite_pixel, which has none to spare for ND > 2 - laundering the row base keeps the
        // handful of index instructions inside the pass)
        int row0 = wave * 32 + 4 * kh;
        if constexpr (ND > 2) asm volatile("" : "+v"(row0));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + (r & 3) + 8 * (r >> 2);
            const int yy = div_tw.div(row), xx = row - yy * p.TW;
            if (row >= p.TH * p.TW) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int oy = 2 * yy + (g >> 1), ox = 2 * xx + (g & 1);
                const int blk = (oy >> 2) * nbx + (ox >> 4);
                if ((blk >> 2) == pass)
                    s_enc[((blk & 3) * 64 + (oy & 3) * 16 + (ox & 15)) * kCompEncPad + n] = acc[0][g][r];
            }
        }
        __syncthreads();
        const int blk = pass * 4 + wave;
        const int byl = blk / nbx, bxl = blk - byl * nbx;
        const int hy = byl * kSumBlockH + (lane >> 4), hx = bxl * kSumBlockW + (lane & 15);   // inside the region
        const int y = oy0 + hy, x = ox0 + hx;
        const bool valid = blk < nblk && y < c.H && x < c.W;
        double cost[2 * ND];
#pragma unroll
        for (int i = 0; i < 2 * ND; ++i) cost[i] = 0.0;
        float of[3] = {0.f, 0.f, 0.f}, od[ND];
#pragma unroll
        for (int d = 0; d < ND; ++d) od[d] = 0.f;
        if (valid)
            composite_pixel_values<ND, K, FIRST>(c, y, x, &s_enc[tid * kCompEncPad], mean, rstd, s_px, s_kern, HW_,
                                                 hy, hx, goal, cost, of, od);
        const int y_blk = oy0 + byl * kSumBlockH, x_blk = ox0 + bxl * kSumBlockW;
        const bool blk_ok = blk < nblk && y_blk < c.H && x_blk < c.W;
        // The block's 64 pixels leave as whole 16-byte pieces: the wave turns them over in its own 64 feature rows (dead now:
        // every lane has its pixel) - a block row is 16 pixels = 48 consecutive floats of the frame and 16 ND of the
        // distributions (top_fusable: whole blocks only, rows 16-byte aligned) - so that a lane has ONE frame store instead
        // of three 4-byte ones, and the stores can be sc1 (ConvParams::wt_out: the item then publishes without a release
        // fence, which would write back whatever the XCD's L2 holds dirty; 4-byte sc1 stores would cost six times as much per
        // byte).  Same values.
        {
            float *slab = s_enc + (wave * 64) * kCompEncPad;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) slab[lane * 3 + ch] = of[ch];
#pragma unroll
            for (int d = 0; d < ND; ++d) slab[192 + lane * ND + d] = od[d];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const bool wt = p.wt_out != 0;
            const unsigned fr_bytes = (unsigned)(c.H * c.W * 3) * 4u, di_bytes = (unsigned)(c.H * c.W * ND) * 4u;
            const __amdgpu_buffer_rsrc_t r_fr = __builtin_amdgcn_make_buffer_rsrc(
                c.out_frame + (long long)b * c.out_frame_bstride, 0, blk_ok ? (int)fr_bytes : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t r_di = __builtin_amdgcn_make_buffer_rsrc(
                c.out_distrib + (long long)b * c.out_distrib_bstride, 0, blk_ok ? (int)di_bytes : 0, 0x00020000);
            if (lane < 48) {
                const int row = lane / 12, q = lane - row * 12;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(slab + row * 48 + 4 * q);
                const unsigned off = (unsigned)(((y_blk + row) * c.W + x_blk) * 3 + 4 * q) * 4u;
                if (wt) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), r_fr, off, 0, 16);
                else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), r_fr, off, 0, 0);
            }
            for (int j = lane; j < 16 * ND; j += 64) {
                const int row = j / (4 * ND), q = j - row * (4 * ND);
                const f32x4 v = *reinterpret_cast<const f32x4 *>(slab + 192 + row * 16 * ND + 4 * q);
                const unsigned off = (unsigned)(((y_blk + row) * c.W + x_blk) * ND + 4 * q) * 4u;
                if (wt) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), r_di, off, 0, 16);
                else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), r_di, off, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 2 * ND; ++i) cost[i] = wave_sum(cost[i]);
        if (lane == 0 && blk_ok) {
            const int gblk = (y_blk / kSumBlockH) * sum_blocks_x(c.W) + x_blk / kSumBlockW;
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                double *dst = c.out_sums + (((long long)b * ND + d) * nblocks + gblk) * 2;
                if (p.wt_out != 0) {
                    __hip_atomic_store(dst, cost[2 * d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(dst + 1, cost[2 * d + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    dst[0] = cost[2 * d]; dst[1] = cost[2 * d + 1];
                }
            }
        }
    }
    // diagnostic build: row 25 = [stats + publish, halo / kernels to LDS, wait for mates + LayerNorm, compose passes]
    VF_TS_ADD(25, 0, tf1 - tf0); VF_TS_ADD(25, 1, tf2 - tf1); VF_TS_ADD(25, 2, tf3 - tf2); VF_TS_ADD(25, 4, VF_TS_NOW() - tf3);
    VF_TS_ADD(25, 3, 1);
}

// epilogue hook of conv_tile<4, fused_epi(ND, FIRST), 1>: the compositing parameters (a device address inside the
// schedule) are read through the constant address space, the goal pixels from the launch's LDS control block
template <int ND, bool FIRST, int K, class PT>
__device__ __forceinline__ void convt_fused_epilogue(const PT &p, f32x16 (&acc)[1][4], int bx, long long *red, float *smem) {
    typedef const __attribute__((address_space(4))) CompositeParams CT;
    const unsigned long long a = reinterpret_cast<unsigned long long>(p.fuse_comp);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    CT &c = *(CT *)(((unsigned long long)hi << 32) | lo);
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    const int *goal = reinterpret_cast<const int *>(smem_all) + kFusedCtlGoal + p.fuse_view * ND * 2;
    fused_top_body<ND, FIRST, K>(p, c, acc, bx, red, smem, goal);
}

}  // namespace vf
