// vf_conv_gsplit.h - the gate-split 128-row conv-LSTM tile of the persistent rollout (round 3, final form).
//
// Wave w multiplies gate w's weight slice with all four row blocks of the workgroup's 128 GEMM rows (vf_conv_mfma.h has the
// machine model and the first version, conv_tile<4, EPI_LSTM, MREP, PT, 0>, which still serves the 64-row plan and the
// debug-only 256-row variant).  This version differs in three ways:
//   * ONE register set for the weight slice: the k8 block q of the NEXT tap is requested into its registers as soon as
//     this tap's MFMAs on block q have been issued (~3000 cycles before its first use) - 16 VGPRs and the two parity
//     copies of the unrolled kernel row are gone (248 VGPRs, no spills, half the code);
//   * an 8 x 16-pixel tile shape (plan_geometry): 240 instead of 288 staged pixels per chunk;
//   * a chunk is staged in two batches of five 16-byte elements per thread; where an element comes from is worked out
//     once per item (ten registers), the LayerNorm gain / offset / statistics of the thread's channel quad are read once
//     per chunk, and the first batch of chunk c + 1 is requested before the K loop of chunk c.
// Measured against the first version (same box): C2 64.85 -> 64.31 ms, 1000 samples 368.0 -> 362.0 ms; the staging diet of
// the last bullet: 63.6 -> 63.3 ms.
//
// Also measured here and NOT adopted: two operand tiles in LDS with chunk c + 1 staged UNDER the K loop of chunk c (two
// elements per thread requested at the start of every kernel row and LayerNorm-ed / stored at its end, the late input of
// an early-started item prefetched after one non-blocking poll of its producer).  Bit-identical, staging per slot 8.6 ->
// 3.7 ms - and the K loops 32.1 -> 35.6 ms: a staging instruction inside a K loop is paid in matrix-pipe time (a wave's
// own VALU / VMEM instructions do not overlap with its MFMAs), one outside is mostly paid by a slot whose CU's pipe the
// other workgroup keeps busy anyway.  C2 64.78 vs 64.31 ms without the prefetch.
//
// Same values through the same expressions in the same (chunk, tap, k8, j) order as every other plan: the same bits.
#pragma once
#include <hip/hip_runtime.h>
#include "vf_conv_mfma.h"

namespace vf {

// MR = 4: 128 rows per workgroup (MR = 2, 64 rows, compiles too).  5 x 5 kernel, stride 1, 32-channel chunks of whole
// channel quads (vf_engine.hip plans this tile only then).
// RAW (arch 3): the tile ends with the GEMM and stores the raw gate pre-activations (gates_raw_epilogue)
template <int MR, class PT, bool RAW = false>
__device__ __forceinline__ void conv_lstm_gsplit2_tile(const PT &p, const int bx_, const int by_, float *smem) {
    [[maybe_unused]] constexpr bool kInLaunch = !std::is_same<PT, ConvParams>::value;
    const int bx = __builtin_amdgcn_readfirstlane(bx_), by = __builtin_amdgcn_readfirstlane(by_);
    constexpr int G = 4, KC = 32, KCpad = 36, K8 = 4, q4 = 8, q4_log2 = 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, kh = lane >> 5;
    const int LH = p.TH + 4, LW = p.TW + 4;
    const int tile_px = LH * LW;
    const int tile_floats = p.NI * tile_px * KCpad;
    float *lnTab = smem + tile_floats;                       // [2][NI][2]: mean, rstd
    long long *red = reinterpret_cast<long long *>(lnTab + 4 * p.NI);
    float *gbTab = lnTab + 4 * p.NI + 16;                    // LayerNorm gain / offset of every input channel
    const int gbC = (p.seg[0].C + (p.nseg > 1 ? p.seg[1].C : 0) + 3) & ~3;
    const int cg = by;
    const int tiles_per_img = p.tilesY * p.tilesX;
    int bimg0, ty0, tx0;
    if (p.NI == 1) {
        bimg0 = bx / tiles_per_img;
        const int tile_id = bx % tiles_per_img;
        ty0 = (tile_id / p.tilesX) * p.TH;
        tx0 = (tile_id % p.tilesX) * p.TW;
    } else {
        bimg0 = bx * p.NI; ty0 = 0; tx0 = 0;
    }

    const bool late = p.late_cnt != nullptr;
    ln_table(p, bimg0, lnTab, 0, late ? 1 : 2);
    for (int i = tid; i < gbC; i += kConvThreads) {
        const int sgi = i < p.seg[0].C ? 0 : 1;
        const int cc = sgi ? i - p.seg[0].C : i;
        float g = 1.f, b = 0.f;
        if (sgi < p.nseg && cc < p.seg[sgi].C && p.seg[sgi].ln_part) {
            const int m = cc % p.seg[sgi].gamma_mod;
            g = p.seg[sgi].gamma[m]; b = p.seg[sgi].beta[m];
        }
        gbTab[i] = g; gbTab[gbC + i] = b;
    }

    // ---- this lane's A rows (GEMM rows m * 32 + n), in float4 units inside an operand tile
    int ab4[MR];
    {
        const int px_per_img = p.TH * p.TW;
        const TileDiv div_rpi(p.RPI), div_tw(p.TW);
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const int row = m * 32 + n;
            const int img = div_rpi.div(row), rem = row - img * p.RPI;
            const bool ok = img < p.NI && rem < px_per_img;
            const int y = div_tw.div(rem), x = rem - y * p.TW;
            ab4[m] = ((ok ? (img * tile_px + y * LW + x) * KCpad : 0) + kh * 4) >> 2;
        }
    }
    f32x16 acc[MR][1];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][0][r] = 0.f;

    const int Ntot = p.ncg * G * 32;
    const int total_chunks = p.seg[0].nchunk + (p.nseg > 1 ? p.seg[1].nchunk : 0);
    const int gtN = total_chunks * 25;
    const unsigned gs_loff = (unsigned)(((kh * Ntot + (cg * G + wave) * 32 + n) * 4) * 4);
    const unsigned gs_wstep_b = (unsigned)(2 * Ntot * 4) * 4u;              // bytes per (tap, k8) block
    const __amdgpu_buffer_rsrc_t gs_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.Wp), 0, 0x7FFFFFFF, 0x00020000);
    // ONE register set for the weight slice: the k8 block q of the NEXT tap is requested into gsW[q] as soon as this tap's
    // MFMAs on gsW[q] have been issued - three quarters of a tap (~3000 cycles) before its first use
    f32x4 gsW[4];
    auto gs_loadq = [&](const int q_, const int gt) {       // (unconditional: behind the last tap the last slice again)
        const unsigned so_ = (unsigned)(min(gt, gtN - 1) * 4 + q_) * gs_wstep_b;
        gsW[q_] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(gs_rsrc, gs_loff, so_, 0));
    };
    const int ci0 = p.chunk_begin;          // (chunks in front of it multiply zeros: vf_conv_mfma.h, ConvParams::chunk_begin)
#pragma unroll
    for (int q_ = 0; q_ < 4; ++q_) gs_loadq(q_, ci0 * 25);

    // ---- staging of one chunk.  Element u of a thread = pixel (tid / 8) + 32 u, channel quad tid % 8.
    const unsigned magic_px = 0xFFFFFFFFu / (unsigned)tile_px + 1u, magic_lw = 0xFFFFFFFFu / (unsigned)LW + 1u;
    const int q = tid & (q4 - 1), pl = tid >> q4_log2;
    constexpr int ppp = kConvThreads >> q4_log2;            // 32 pixels per pass
    const int npix = p.NI * tile_px;
    const int y0 = ty0 - 2, x0 = tx0 - 2;
    const int n_here = min(p.NI, p.B - bimg0);
    const bool ni1 = p.NI == 1;
    struct ChunkSrc {                   // where chunk ci comes from (wave-uniform)
        __amdgpu_buffer_rsrc_t rsrc;
        unsigned cbyte, cstride, img_step;      // byte offset of this thread's channel quad, bytes per pixel, bytes per image
        int seg, gi;                    // input segment, index of the thread's quad in the gain / offset table
        bool has_ln, relu;
    };
    auto chunk_src = [&](const int ci) {
        ChunkSrc cs;
        cs.seg = (ci < p.seg[0].nchunk) ? 0 : 1;
        const auto &sg = p.seg[cs.seg];
        const int c = (cs.seg == 0 ? ci : ci - p.seg[0].nchunk) * KC + 4 * q;
        cs.cbyte = (unsigned)c * 4u;
        cs.cstride = (unsigned)sg.C * 4u;
        cs.img_step = (unsigned)sg.bstride * 4u;
        const unsigned img_bytes = (unsigned)(p.Hin * p.Win) * cs.cstride;
        cs.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(sg.ptr + (long long)bimg0 * sg.bstride), 0,
                                                    n_here > 0 ? (int)((unsigned)(n_here - 1) * cs.img_step + img_bytes) : 0,
                                                    0x00020000);
        cs.gi = (cs.seg == 0 ? 0 : p.seg[0].C) + c;
        cs.has_ln = sg.ln_part != nullptr;
        cs.relu = sg.relu != 0;
        return cs;
    };
    // Where element u of this thread comes from is the same for every chunk of the item: image pixel iy * Win + ix, or -1
    // for padding / rows past the tile.  Computed once (ten registers), so that a chunk's request is three instructions
    // per element - every instruction outside the K loops is paid at one issue slot per ~40 cycles while the other
    // workgroup of the CU is in its K loop (vf_conv_mfma.h, machine model).
    constexpr int kBatch = 5;           // 2 x 5 elements per thread and chunk at most (320 haloed pixels)
    int poff[2 * kBatch];
#pragma unroll
    for (int u = 0; u < 2 * kBatch; ++u) {
        const int pix = pl + u * ppp;
        int img = 0, r = pix;
        if (!ni1) { img = tile_px == 1 ? pix : (int)__umulhi((unsigned)pix, magic_px); r = pix - img * tile_px; }
        const int ly = (int)__umulhi((unsigned)r, magic_lw), lx = r - ly * LW;
        const int iy = y0 + ly, ix = x0 + lx;
        const bool ok = pix < npix && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win && img < n_here;
        poff[u] = ok ? iy * p.Win + ix : -1;
    }
    // request element u: returns its value (zeros for padding)
    auto st_issue = [&](const ChunkSrc &cs, const int u) {
        unsigned off = (unsigned)poff[u] * cs.cstride + cs.cbyte;
        if (!ni1) {
            const int pix = pl + u * ppp;
            off += (unsigned)(tile_px == 1 ? pix : (int)__umulhi((unsigned)pix, magic_px)) * cs.img_step;
        }
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(cs.rsrc, poff[u] >= 0 ? off : 0xFFFFFFFFu, 0, 0));
    };
    // LayerNorm gain / offset of this thread's channel quad and (one image per tile) the statistics: once per chunk
    struct ChunkLn { f32x4 g, b; float mean, rstd; };
    auto chunk_ln = [&](const ChunkSrc &cs) {
        ChunkLn L;
        L.g = *reinterpret_cast<const f32x4 *>(gbTab + cs.gi);
        L.b = *reinterpret_cast<const f32x4 *>(gbTab + gbC + cs.gi);
        L.mean = lnTab[2 * (cs.seg * p.NI)]; L.rstd = lnTab[2 * (cs.seg * p.NI) + 1];
        return L;
    };
    // LayerNorm / relu of element u and its LDS store into operand tile `dst`
    auto st_finish = [&](const ChunkSrc &cs, const ChunkLn &L, const int u, f32x4 v, float *dst) {
        const int pix = pl + u * ppp;
        if (pix >= npix) return;
        if (poff[u] >= 0) {
            if (cs.has_ln) {
                float mean = L.mean, rstd = L.rstd;
                if (!ni1) {
                    const int img = tile_px == 1 ? pix : (int)__umulhi((unsigned)pix, magic_px);
                    mean = lnTab[2 * (cs.seg * p.NI + img)]; rstd = lnTab[2 * (cs.seg * p.NI + img) + 1];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaf((v[j] - mean) * rstd, L.g[j], L.b[j]);
            }
            if (cs.relu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            }
        }
        *reinterpret_cast<f32x4 *>(dst + pix * KCpad + 4 * q) = v;
    };
    // A chunk is staged in two batches of five elements per thread.  The first batch of chunk c + 1 is REQUESTED before the
    // K loop of chunk c (five loads and their addresses, which had to be issued anyway; nothing is added inside the loop)
    // and sits in 20 registers under it, so that after the loop only the second batch still has a memory latency to
    // wait for - behind the LayerNorm and the stores of the first.
    f32x4 pv[kBatch];
    auto issue_batch = [&](const ChunkSrc &cs, const int u0, f32x4 (&v)[kBatch]) {
#pragma unroll
        for (int u = 0; u < kBatch; ++u) v[u] = st_issue(cs, u0 + u);
    };
    auto finish_batch = [&](const ChunkSrc &cs, const ChunkLn &L, const int u0, const f32x4 (&v)[kBatch], float *dst) {
#pragma unroll
        for (int u = 0; u < kBatch; ++u) st_finish(cs, L, u0 + u, v[u], dst);
    };

    // (yielding, vf_conv_mfma.h: the recurrent chunks of an early-started item step aside for chain-critical work of the
    // CU's other workgroup - once per kernel row, bounded per item)
    const bool yielding = late && p.cu_state != nullptr && p.yield_budget > 0;
    int ybudget = p.yield_budget;
    const int *yword = yielding ? cu_partner_word(p) : nullptr;
    // ---- K loop of one kernel row (vf_conv_mfma.h: the gate-split K loop), reading operand tile `a4`
    constexpr int GSZ = MR < 4 ? MR : 4, NG = MR / GSZ, NS = 4 * NG;
    const f32x4 *ar[MR];
    f32x4 aP4[GSZ], aQ4[GSZ];
    auto gs_fetch = [&](f32x4 (&A_)[GSZ], const int kx, const int sub) {
#pragma unroll
        for (int m_ = 0; m_ < GSZ; ++m_) A_[m_] = ar[(sub % NG) * GSZ + m_][kx * 9 + (sub / NG) * 2];
    };
    auto gs_tap = [&](auto kxc, const int gt_next) {
        constexpr int KX = decltype(kxc)::value;
        static_for<NS>([&](auto sc) {
            constexpr int S = decltype(sc)::value;
            f32x4 (&a_cur)[GSZ] = (S & 1) ? aQ4 : aP4;
            f32x4 (&a_nxt)[GSZ] = (S & 1) ? aP4 : aQ4;
            if constexpr (S + 1 < NS) gs_fetch(a_nxt, KX, S + 1);
            else if constexpr (KX < 4) gs_fetch(a_nxt, KX + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j_ = 0; j_ < 4; ++j_) {
#pragma unroll
                for (int m_ = 0; m_ < GSZ; ++m_)
                    acc[(S % NG) * GSZ + m_][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                        a_cur[m_][j_], gsW[S / NG][j_], acc[(S % NG) * GSZ + m_][0], 0, 0, 0);
                if constexpr (KX == 0 && S == 0) {
                    // the last k8 block of THIS tap: not requested at the end of the previous kernel row (the compiler
                    // drains the vector-memory counter at the head of the row loop - a load issued just before it would be
                    // waited for with the matrix pipe idle) but here, behind the row's first MFMAs, 44 MFMAs ahead of its use
                    if (j_ == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        gs_loadq(3, gt_next - 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (S % NG == NG - 1 && !(KX == 4 && S / NG == 3)) {
                gs_loadq(S / NG, gt_next);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };
    bool ychunk = false;                // this chunk is a recurrent chunk of a yielding item
    unsigned yseen = 0;                 // the partner's state word, requested one check ago
    auto gs_row = [&](const int gt_row) {
        gs_fetch(aP4, 0, 0);
        gs_tap(std::integral_constant<int, 0>{}, gt_row + 1);
        gs_tap(std::integral_constant<int, 1>{}, gt_row + 2);
        gs_tap(std::integral_constant<int, 2>{}, gt_row + 3);
        gs_tap(std::integral_constant<int, 3>{}, gt_row + 4);
        gs_tap(std::integral_constant<int, 4>{}, gt_row + 5);
    };

    if constexpr (kInLaunch) VF_TRACE_EVT(TR_MFMAS, (unsigned long long)(25 * K8 * 4 * MR));
    __builtin_amdgcn_s_setprio(0);
    const f32x4 *a4 = reinterpret_cast<const f32x4 *>(smem);
    bool pf = false;                    // batch 0 of this chunk was requested under the previous K loop
    for (int ci = ci0; ci < total_chunks; ++ci) {
        if (late && ci == p.seg[0].nchunk) {         // the recurrent chunks are done: now the layer input is needed
            const int b1 = p.NI == 1 ? bimg0 + 1 : min(bimg0 + p.NI, p.B);
            if constexpr (kInLaunch) VF_TRACE_EVT(TR_LATE);
            if (!late_wait(p, bimg0, b1, reinterpret_cast<int *>(red))) return;
            if constexpr (kInLaunch) VF_TRACE_EVT(TR_LATE_END);
            ln_table(p, bimg0, lnTab, 1, 2);
        }
        __syncthreads();                // previous chunk fully consumed (and lnTab / gbTab visible)
        if constexpr (kInLaunch) VF_TRACE_EVT(TR_STAGE);
        {
            const ChunkSrc cs = chunk_src(ci);
            if (!pf) issue_batch(cs, 0, pv);
            f32x4 v1[kBatch];
            issue_batch(cs, kBatch, v1);
            const ChunkLn L = chunk_ln(cs);
            finish_batch(cs, L, 0, pv, smem);
            finish_batch(cs, L, kBatch, v1, smem);
        }
        if constexpr (kInLaunch) VF_TRACE_EVT(TR_ST_WRITTEN);
        __syncthreads();
        pf = ci + 1 < total_chunks && !(late && ci + 1 == p.seg[0].nchunk);
        if (pf) issue_batch(chunk_src(ci + 1), 0, pv);
        if constexpr (kInLaunch) VF_TRACE_EVT(TR_KLOOP);
        ychunk = yielding && ci < p.seg[0].nchunk;
        for (int ky = 0; ky < 5; ++ky) {
            if (ychunk) yseen = yield_peek_issue(yword);
#pragma unroll
            for (int m = 0; m < MR; ++m) ar[m] = a4 + ab4[m] + ky * LW * 9;
            gs_row(ci * 25 + ky * 5);
            if (ychunk) yield_to_partner(yword, yseen, ybudget);
        }
    }
    __builtin_amdgcn_s_setprio(2);
    if constexpr (kInLaunch) VF_TRACE_EVT(TR_EPI);
    if constexpr (RAW) gates_raw_epilogue<MR>(p, acc, bx, by, smem);
    else lstm_gsplit_epilogue<MR>(p, acc, bx, by, smem);
}

template <int MR>
VF_GLOBAL VF_LAUNCH_BOUNDS(kConvThreads, 2) void conv_lstm_gsplit2_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    conv_lstm_gsplit2_tile<MR>(p, blockIdx.x, blockIdx.y, smem);
}
VF_GLOBAL VF_LAUNCH_BOUNDS(kConvThreads, 2) void conv_gates_raw_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    conv_lstm_gsplit2_tile<4, ConvParams, true>(p, blockIdx.x, blockIdx.y, smem);
}

}  // namespace vf
