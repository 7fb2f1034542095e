// vf_conv_bf16x6.h - conv-LSTM gate tile with fp32 emulated on the bf16 matrix cores.
//
// Opt-in precision mode (vf_config.precision = 1); the default path is the exact fp32 MFMA tile
// of vf_conv_mfma.h.  gfx950 runs bf16 MFMA at 16x the fp32 MFMA rate, so fp32-class accuracy
// can be bought with several bf16 products: every fp32 operand is split exactly into three bf16
// pieces (x = x1 + x2 + x3, 3 x 8 mantissa bits), and
//     a * b  ~=  a1 b1 + a1 b2 + a2 b1 + a2 b2 + a1 b3 + a3 b1
// (the three dropped terms are below 2^-24 |a b|, i.e. below one fp32 rounding).  Each bf16 x bf16
// product is exact in fp32 and the MFMA accumulates in fp32, so the result carries the error of
// an fp32 dot product with a different summation order - measured 7.5e-7 vs 4.0e-7 (relative to
// max |c|, K = 4800) for a plain fp32 GEMM.  Six v_mfma_f32_32x32x16_bf16 (32 cycles, K = 16) replace
// eight v_mfma_f32_32x32x2_f32 (64 cycles, K = 2): 2.67x fewer matrix-pipe cycles per MAC.
//
// Structure = the G == 4 path of conv_tile: haloed input tile of one 16-channel chunk in LDS, here
// as three bf16 planes [pixel][16 + 8 pad] (48-B rows: the 16-lane groups of ds_read_b128 hit 16
// distinct slots); LayerNorm (+relu) of the producer and the 3-way split happen while staging.
// Weights are split on the host and packed [chunk][tap][cg][gate][plane][k-half][32][8]; wave w
// fetches gate w's 3 planes one tap ahead and parks them in the other LDS buffer (one barrier per tap).
#pragma once
#include <hip/hip_runtime.h>
#include "vf_conv_mfma.h"

namespace vf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

constexpr int kBfKC = 16;           // channels per LDS chunk
constexpr int kBfRowUnits = 3;      // 16-B units per pixel row of one plane: 16 channels + 8 pad

// x = p0 + p1 + p2 exactly (round-to-nearest-even pieces of the running residual)
__device__ __forceinline__ void split3(const float x, __bf16 &p0, __bf16 &p1, __bf16 &p2) {
    p0 = (__bf16)x;
    const float r1 = x - (float)p0;
    p1 = (__bf16)r1;
    p2 = (__bf16)(r1 - (float)p1);
}

// LDS bytes of the tile for a layer geometry (host + device)
__host__ __device__ inline size_t bf16x6_lds_bytes(int NI, int LH, int LW) {
    const size_t a_units = (size_t)3 * NI * LH * LW * kBfRowUnits;
    return a_units * 16 + ((size_t)4 * NI + 16) * 4 + (size_t)2 * 4 * 3 * 64 * 16;
}

template <int MREP, class PT>
__device__ __forceinline__ void conv_lstm_bf16x6_tile(const PT &p, const int bx_, const int by_, float *smem) {
    const int bx = __builtin_amdgcn_readfirstlane(bx_), by = __builtin_amdgcn_readfirstlane(by_);   // (see conv_tile)
    static_assert(MREP == 1, "128-row tiles (the K loop below fetches one row block per wave)");
    constexpr int G = 4;
    constexpr int WROWS = MREP * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, kh = lane >> 5;
    const int LH = (p.TH - 1) * p.stride + p.KH, LW = (p.TW - 1) * p.stride + p.KW;
    const int tile_px = LH * LW;
    const int plane_units = p.NI * tile_px * kBfRowUnits;          // 16-B units per bf16 plane
    u16x8 *aP = reinterpret_cast<u16x8 *>(smem);                   // [3][pixel][3 units]
    float *lnTab = smem + (size_t)3 * plane_units * 4;             // [2][NI][2]
    long long *red = reinterpret_cast<long long *>(lnTab + 4 * p.NI);
    u16x8 *bsm = reinterpret_cast<u16x8 *>(lnTab + 4 * p.NI + 16);  // [2 buf][4 gates][3 planes][64]
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

    const bool late = p.late_cnt != nullptr;     // early-started item: see ConvParams::late_cnt
    ln_table(p, bimg0, lnTab, 0, late ? 1 : 2);

    const int px_per_img = p.TH * p.TW;
    int abase[MREP];                    // 16-B unit of this lane's row inside a plane (+ k-half)
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
        const int row = wave * WROWS + m * 32 + n;
        const int img = row / p.RPI, rem = row % p.RPI;
        const bool ok = img < p.NI && rem < px_per_img;
        const int y = rem / p.TW, x = rem % p.TW;
        abase[m] = (ok ? (img * tile_px + y * p.stride * LW + x * p.stride) * kBfRowUnits : 0) + kh;
    }

    f32x16 acc[MREP][G];
#pragma unroll
    for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][g][r] = 0.f;

    const int ntaps = p.KH * p.KW;
    const int total_chunks = p.seg[0].nchunk + (p.nseg > 1 ? p.seg[1].nchunk : 0);
    // packed weights, 16-B units: [chunk][tap][cg][gate][plane][64 lanes]
    const u16x8 *wgate = reinterpret_cast<const u16x8 *>(p.Wp16) + ((long long)cg * G + wave) * 3 * 64 + lane;
    const long long wtap = (long long)p.ncg * G * 3 * 64;          // units per (chunk, tap)
    const int gtN = total_chunks * ntaps;
    const unsigned w_loff = (unsigned)((((cg * G + wave) * 3) * 64 + lane) * 16);         // this lane's 16-B unit, plane 0
    const unsigned w_tap_b = (unsigned)(p.ncg * G * 3 * 64 * 16);                          // bytes per (chunk, tap)
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(p.Wp16), 0, 0x7FFFFFFF, 0x00020000);
    u16x8 breg[3];
#define VF_LOADB16(GT_)                                                                         \
    _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) breg[pl] = wgate[(long long)(GT_) * wtap + pl * 64];
#define VF_WRITEB16(BUF_)                                                                       \
    _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) bsm[(((BUF_) * G + wave) * 3 + pl) * 64 + lane] = breg[pl];
    VF_LOADB16(0)

    const int items = p.NI * tile_px * 2;       // (pixel, 8-channel half) pairs

    for (int ci = 0; ci < total_chunks; ++ci) {
        const int s = (ci < p.seg[0].nchunk) ? 0 : 1;
        const auto &sg = p.seg[s];
        const int c0 = (s == 0 ? ci : ci - p.seg[0].nchunk) * kBfKC;

        if (late && ci == p.seg[0].nchunk) {         // the recurrent chunks are done: now the layer input is needed
            const int b1 = p.NI == 1 ? bimg0 + 1 : min(bimg0 + p.NI, p.B);
            if (!late_wait(p, bimg0, b1, reinterpret_cast<int *>(red))) return;
            ln_table(p, bimg0, lnTab, 1, 2);
        }
        __syncthreads();
        if constexpr (!std::is_same<PT, ConvParams>::value) VF_TRACE_EVT(TR_STAGE);
        for (int it = tid; it < items; it += kConvThreads) {
            const int pix = it >> 1, oct = it & 1;
            const int img = pix / tile_px, r = pix - img * tile_px;
            const int ly = r / LW, lx = r - ly * LW;
            const int iy = ty0 * p.stride - p.pad + ly, ix = tx0 * p.stride - p.pad + lx;
            const int b = bimg0 + img;
            const int c = c0 + 8 * oct;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
            if (b < p.B && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win && c < sg.C) {
                const float *src = sg.ptr + (long long)b * sg.bstride + ((long long)iy * p.Win + ix) * sg.C + c;
                const f32x4 lo = *reinterpret_cast<const f32x4 *>(src);
                const f32x4 hi = *reinterpret_cast<const f32x4 *>(src + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
                if (sg.ln_part) {
                    const float mean = lnTab[2 * (s * p.NI + img)];
                    const float rstd = lnTab[2 * (s * p.NI + img) + 1];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int cc = (c + j) % sg.gamma_mod;
                        v[j] = fmaf((v[j] - mean) * rstd, sg.gamma[cc], sg.beta[cc]);
                    }
                }
                if (sg.relu) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
                }
            }
            bf16x8 q0, q1, q2;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                __bf16 a, bq, cq;
                split3(v[j], a, bq, cq);
                q0[j] = a; q1[j] = bq; q2[j] = cq;
            }
            const int u = pix * kBfRowUnits + oct;
            aP[u] = __builtin_bit_cast(u16x8, q0);
            aP[plane_units + u] = __builtin_bit_cast(u16x8, q1);
            aP[2 * plane_units + u] = __builtin_bit_cast(u16x8, q2);
        }
        if (ci == 0) { VF_WRITEB16(0) }
        __syncthreads();
        if constexpr (!std::is_same<PT, ConvParams>::value) {
            if (ci == 0) VF_TRACE_EVT(TR_MFMAS, (unsigned long long)(ntaps * 6 * G * MREP));
            VF_TRACE_EVT(TR_KLOOP);
        }

        // ---- K loop.  A tap is only 24 MFMAs of 32 cycles, so everything else in it is paid dearly (a wave's own VALU /
        // VMEM instructions do not overlap with its MFMAs, tools/ubench/mfma_shadow.hip): the next tap's weight planes
        // come through raw buffer loads (lane offset in one VGPR, tap offset in an SGPR), the operand fetches use one
        // address register per plane set up once per kernel row plus immediates, the weight fetches one per tap, and the
        // A planes of the NEXT tap are requested before this tap's barrier (they do not depend on it).  Same terms in the
        // same order as rounds 1-2: the same bits.
        for (int ky = 0; ky < p.KH; ++ky) {
            const u16x8 *arow[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) arow[pl] = aP + pl * plane_units + abase[0] + ky * LW * kBfRowUnits;
            bf16x8 a[3][MREP], an[3][MREP], bw[3][G];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) a[pl][0] = __builtin_bit_cast(bf16x8, arow[pl][0]);
            for (int kx = 0; kx < p.KW; ++kx) {
                const int gt = ci * ntaps + ky * p.KW + kx;
                const int buf = gt & 1;
                const bool more = gt + 1 < gtN;
                if (more) {
                    const unsigned so = (unsigned)(gt + 1) * w_tap_b;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        breg[pl] = __builtin_bit_cast(u16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_loff, so + (unsigned)pl * 1024u, 0));
                }
                const u16x8 *brow = bsm + buf * (G * 3 * 64) + lane;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int g = 0; g < G; ++g) bw[pl][g] = __builtin_bit_cast(bf16x8, brow[(g * 3 + pl) * 64]);
                __builtin_amdgcn_sched_barrier(0);
                // smallest terms first: a3 b1, a1 b3, a2 b2, a2 b1, a1 b2, a1 b1
#define VF_T(PA_, PB_)                                                                          \
                _Pragma("unroll") for (int g = 0; g < G; ++g)                                   \
                    acc[0][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA_][0], bw[PB_][g], acc[0][g], 0, 0, 0);
                VF_T(2, 0) VF_T(0, 2) VF_T(1, 1)
                __builtin_amdgcn_sched_barrier(0);
                if (kx + 1 < p.KW) {        // the next tap's operands: independent of the barrier below
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) an[pl][0] = __builtin_bit_cast(bf16x8, arow[pl][(kx + 1) * kBfRowUnits]);
                }
                __builtin_amdgcn_sched_barrier(0);
                VF_T(1, 0) VF_T(0, 1) VF_T(0, 0)
#undef VF_T
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
                    u16x8 *bw_ = bsm + (buf ^ 1) * (G * 3 * 64) + wave * 3 * 64 + lane;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) bw_[pl * 64] = breg[pl];
                }
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[pl][0] = an[pl][0];
                __syncthreads();
            }
        }
    }
#undef VF_LOADB16
#undef VF_WRITEB16
    if constexpr (!std::is_same<PT, ConvParams>::value) VF_TRACE_EVT(TR_EPI);
    conv_epilogue<G, EPI_LSTM, MREP>(p, acc, bx, by, 0, red);
}

// Measured alternatives (B=1024, per-layer launches, fp32-equivalent TF/s of this kernel): this
// version 219.9; B loads free 226.7; no per-tap barrier 238.6 (wrong results, timing only); a 3-stage
// pipeline (third B buffer, fragments of tap t+1 fetched under the MFMAs of tap t) 225.7 but 222 VGPRs,
// which spills once the tile is an out-of-line body of the persistent kernel.  219.9 x 6 = 1.32 PF/s of
// bf16 MFMA is where tuned bf16 GEMMs on random data land on this chip (power-limited clocks).
template <int MREP>
VF_GLOBAL VF_LAUNCH_BOUNDS(kConvThreads, 2) void conv_lstm_bf16x6_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    conv_lstm_bf16x6_tile<MREP>(p, blockIdx.x, blockIdx.y, smem);
}

}  // namespace vf
