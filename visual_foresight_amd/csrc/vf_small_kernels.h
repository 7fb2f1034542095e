// vf_small_kernels.h - the non-GEMM kernels of one predictor step (all fp32 VALU, HBM/LDS bound):
//   set_context      uint8 context frames -> float/255, context copies
//   sa_kernel        state FC + the action/state part of enc3 as a per-sample bias
//   cdna_finalize    sum K-split partials of the CDNA FC, relu-shift, L1-normalise each 5x5 kernel
//   composite        LN9+relu -> rgb/mask heads -> softmax -> CDNA warp -> compositing of the next
//                    frame and designated-pixel distributions + expected-distance partial sums
//   scores / export  reduce per-step sums to costs (mean over tasks or trade-off weights, mean over
//                    latent draws); hand predictions out in the reference layout (camera axis)
//   register         bilinear warp by a flow field + designated-pixel re-localisation + warp error
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vf_conv_mfma.h"

namespace vf {

constexpr int kMaxDesig = 4;
constexpr int kDnaKern = 5;
constexpr int kTaps = kDnaKern * kDnaKern;
constexpr int kCompTile = 16;           // composite: 16x16 pixels per workgroup
// Cost sums are kept per BLOCK of 4 rows x 16 columns - the pixels one wave of a compositing tile handles - and
// added up in block order by their readers, so the decomposition of an image into workgroup tiles does not show
// in the sums (any tiling whose waves cover whole blocks gives the same bits).
constexpr int kSumBlockH = 4, kSumBlockW = 16;
__host__ __device__ constexpr int sum_blocks_x(int W) { return (W + kSumBlockW - 1) / kSumBlockW; }
__host__ __device__ constexpr int sum_blocks(int H, int W) { return ((H + kSumBlockH - 1) / kSumBlockH) * sum_blocks_x(W); }
constexpr float kReluShift = 1e-12f;

// ------------------------------------------------------------------------------------------
// frames_u8 [nc][ncam][HW3] -> frames_f [ncam][nc][HW3] / 255; distrib [nc][ncam][HWD] -> [ncam][nc][HWD];
// states and context actions are shared by the views and copied as they are
VF_GLOBAL void set_context_kernel(const uint8_t *frames_u8, float *frames_f, int nc, int ncam, int hw3,
                                   const float *src_d, float *dst_d, int hwd,
                                   const float *src_s, float *dst_s, int n_s,
                                   const float *src_a, float *dst_a, int n_a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nc * ncam * hw3) {
        const int e = i % hw3, v = (i / hw3) % ncam, t = i / (hw3 * ncam);
        frames_f[((long long)v * nc + t) * hw3 + e] = (float)frames_u8[i] / 255.0f;
    }
    if (i < nc * ncam * hwd) {
        const int e = i % hwd, v = (i / hwd) % ncam, t = i / (hwd * ncam);
        dst_d[((long long)v * nc + t) * hwd + e] = src_d[i];
    }
    if (i < n_s) dst_s[i] = src_s[i];
    if (i < n_a) dst_a[i] = src_a[i];
}

// ------------------------------------------------------------------------------------------
struct SaParams {
    const float *action; long long action_bstride;     // [adim] per sample (0 stride: shared)
    const float *state;  long long state_bstride;      // [sdim]
    int adim, sdim, B;
    const float *w_state, *b_state;     // [adim+sdim][sdim], [sdim]
    const float *w_sa;                  // enc3 rows for the smeared inputs: [adim+sdim][n_out]
    int n_out;                          // 64
    float *state_out; long long state_out_bstride;      // may be null
    float *sbias;                       // [B][n_out]
};

// one sample, executed by the 64 lanes of one wave; sa = 32 floats of LDS scratch for this wave
template <class PT>
__device__ __forceinline__ void sa_sample(const PT &p, const int b, const int t, float *sa) {
    const int nsa = p.adim + p.sdim;
    if (t < p.adim) sa[t] = p.action[(long long)b * p.action_bstride + t];
    else if (t < nsa) sa[t] = p.state[(long long)b * p.state_bstride + (t - p.adim)];
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    if (t < p.n_out) {
        float acc = 0.f;
        for (int k = 0; k < nsa; ++k) acc = fmaf(sa[k], p.w_sa[k * p.n_out + t], acc);
        p.sbias[(long long)b * p.n_out + t] = acc;
    }
    if (p.state_out && t < p.sdim) {
        float acc = 0.f;
        for (int k = 0; k < nsa; ++k) acc = fmaf(sa[k], p.w_state[k * p.sdim + t], acc);
        p.state_out[(long long)b * p.state_out_bstride + t] = acc + p.b_state[t];
    }
}

VF_GLOBAL void sa_kernel(const SaParams p) {
    __shared__ float sa[32];
    sa_sample(p, blockIdx.x, threadIdx.x, sa);
}

// ------------------------------------------------------------------------------------------
// kern[b][tap][k] = v / sum_tap v,  v = relu(fc - shift) + shift,  fc = bias + sum_split partial
struct FinParams {
    const float *partial; int nsplit; int B; int K;
    const float *bias; float *kern;
};

// one sample per workgroup call; scratch: kTaps*16 + 16 floats of LDS
template <class PT>
__device__ __forceinline__ void cdna_finalize_sample(const PT &p, const int b, float *scratch) {
    const int t = threadIdx.x;
    const int n = kTaps * p.K;
    float *v = scratch, *norm = scratch + kTaps * 16;
    if (t < n) {
        float acc = p.bias[t];
        for (int z = 0; z < p.nsplit; ++z) acc += p.partial[((long long)z * p.B + b) * n + t];
        v[t] = fmaxf(acc - kReluShift, 0.f) + kReluShift;
    }
    __syncthreads();
    if (t < p.K) {
        float s = 0.f;
        for (int tap = 0; tap < kTaps; ++tap) s += v[tap * p.K + t];
        norm[t] = s;
    }
    __syncthreads();
    if (t < n) p.kern[(long long)b * n + t] = v[t] / norm[t % p.K];
}

VF_GLOBAL void cdna_finalize_kernel(const FinParams p) {
    __shared__ float scratch[kTaps * 16 + 16];
    cdna_finalize_sample(p, blockIdx.x, scratch);
}

// ------------------------------------------------------------------------------------------
// Per-layer conditioning of arch 2 (savp_arch.py, Savp2Config): the published SAVP generator tiles the vector
// v = [action, latent, state] over the image and concatenates it to the input of EVERY conv-LSTM (arXiv:1804.01523,
// appendix A).  A spatially constant input channel needs no GEMM rows: with zero padding its contribution to gate column
// `col` at pixel (y, x) is  sum over the taps (dy, dx) that fall inside the image of  t[dy][dx][col],
// t[tap][col] = sum_c W[tap][c][col] v[c],  and which taps fall inside only depends on whether y (x) is one of the two
// first / two last rows (columns): 5 x 5 border classes.  One item per (conv-LSTM, sample) computes the 25 class biases for
// all 4C gate columns; the conv-LSTM epilogue adds the row of its pixel's class.  No K growth (as real channels the 17
// values would cost every cell one more 32-channel chunk: +17 ... +50 % matrix work), same sums as the concatenated
// convolution up to fp32 association.
struct CondParams {
    const float *action; long long action_bstride;     // [adim] per sample (latent channels included; 0 stride: shared)
    const float *state;  long long state_bstride;      // [sdim]
    int adim, sdim, B;
    const float *w;         // [25 taps][adim + sdim][C4]: the conditioning rows of the layer's canonical weights
    int C4;                 // 4 * cell channels (gate-major columns)
    float *out;             // [B][25 classes][C4]
};
constexpr int kCondClasses = kTaps;
// border class of coordinate y in an image of H rows (5-tap kernel, pad 2): 0, 1 | 2 = interior | 3, 4
__host__ __device__ constexpr int cond_class(int y, int H) { return y < 2 ? y : (y >= H - 2 ? y - (H - 5) : 2); }
// first / one-past-last kernel tap that reads inside the image for class r
__host__ __device__ constexpr int cond_tap_lo(int r) { return r < 2 ? 2 - r : 0; }
__host__ __device__ constexpr int cond_tap_hi(int r) { return r > 2 ? 7 - r : 5; }

// samples [b0, b1) (at most kCondPerItem) per workgroup call; sv = kCondPerItem * 32 floats of LDS scratch.  A weight is
// loaded once for all samples of the call; every sample's sums run in the same order as in a one-sample call (same bits).
constexpr int kCondPerItem = 4;
template <class PT>
__device__ __forceinline__ void cond_bias_sample(const PT &p, const int b0, const int b1, float *sv) {
    const int t = threadIdx.x, nsa = p.adim + p.sdim, ns = b1 - b0;
    if (t < kCondPerItem * 32) {
        const int s = t >> 5, c = t & 31;
        float v = 0.f;
        if (s < ns) {
            if (c < p.adim) v = p.action[(long long)(b0 + s) * p.action_bstride + c];
            else if (c < nsa) v = p.state[(long long)(b0 + s) * p.state_bstride + (c - p.adim)];
        }
        sv[t] = v;
    }
    __syncthreads();
    for (int col = t; col < p.C4; col += 256) {
        float tt[kCondPerItem][kTaps];
#pragma unroll
        for (int tap = 0; tap < kTaps; ++tap) {
            float acc[kCondPerItem];
#pragma unroll
            for (int s = 0; s < kCondPerItem; ++s) acc[s] = 0.f;
            for (int c = 0; c < nsa; ++c) {
                const float w = p.w[((long long)tap * nsa + c) * p.C4 + col];
#pragma unroll
                for (int s = 0; s < kCondPerItem; ++s) acc[s] = fmaf(sv[s * 32 + c], w, acc[s]);
            }
#pragma unroll
            for (int s = 0; s < kCondPerItem; ++s) tt[s][tap] = acc[s];
        }
#pragma unroll
        for (int s = 0; s < kCondPerItem; ++s) {
            if (s >= ns) break;
            // row sums over the columns of class rx, then over the rows of class ry (fixed order: deterministic)
            float S[5][5];
#pragma unroll
            for (int dy = 0; dy < 5; ++dy)
#pragma unroll
                for (int rx = 0; rx < 5; ++rx) {
                    float a = 0.f;
#pragma unroll
                    for (int dx = 0; dx < 5; ++dx)
                        if (dx >= cond_tap_lo(rx) && dx < cond_tap_hi(rx)) a += tt[s][dy * 5 + dx];
                    S[dy][rx] = a;
                }
            float *o = p.out + (long long)(b0 + s) * kCondClasses * p.C4 + col;
#pragma unroll
            for (int ry = 0; ry < 5; ++ry)
#pragma unroll
                for (int rx = 0; rx < 5; ++rx) {
                    float a = 0.f;
#pragma unroll
                    for (int dy = 0; dy < 5; ++dy)
                        if (dy >= cond_tap_lo(ry) && dy < cond_tap_hi(ry)) a += S[dy][rx];
                    o[(long long)(ry * 5 + rx) * p.C4] = a;
                }
        }
    }
}

VF_GLOBAL VF_LAUNCH_BOUNDS(256) void cond_bias_kernel(const CondParams p) {
    __shared__ float sv[kCondPerItem * 32];
    const int b0 = blockIdx.x * kCondPerItem;
    cond_bias_sample(p, b0, min(b0 + kCondPerItem, p.B), sv);
}

// ------------------------------------------------------------------------------------------
struct CompositeParams {
    int B, H, W, ND, K;                 // K = num_masks (K+1 mask channels, K-1 kernels used)
    const float *enc6;                  // raw convT3 output [B][H][W][32]
    const long long *ln_part; int ln_nparts; float ln_inv_n;     // exact statistics of enc6 (vf_conv_mfma.h)
    const float *gamma, *beta;          // LN9 [32]
    const float *w_rgb, *b_rgb;         // [32][3], [3]
    const float *w_mask, *b_mask;       // [32][K+1], [K+1]
    const float *kern;                  // [B][25][K]
    const float *prev_frame; long long prev_frame_bstride;      // [H][W][3]
    const float *prev_distrib; long long prev_distrib_bstride;  // [H][W][ND]
    const double *prev_sums;            // [B][ND][blocks][2] partial sums of prev_distrib, or null
    const float *first_frame;           // arch 1 (savp_arch.py): first context frame [H][W][3], shared by every
    const float *first_distrib;         //   sample, and its distributions [H][W][ND]; null = CDNA compositing
    float *out_frame; long long out_frame_bstride;
    float *out_distrib; long long out_distrib_bstride;
    double *out_sums;                   // [B][ND][blocks][2]: sum d, sum d * dist(goal) per 4x16-pixel block
    int goal[kMaxDesig][2];             // (row, col); read by the per-layer kernel only - the persistent
                                        // kernel takes the goals from its launch arguments, so a schedule
                                        // does not depend on them
    int CF;                             // feature channels of enc6: 0 / 32, or 64 (the public CDNA decoder table, cdna_arch.py
                                        // decoder='public': the stand-alone compositing tile only, two 32-channel rounds)
};

// LDS floats needed by composite_tile<ND, K>
constexpr int kCompEncPad = 36;         // floats per pixel row of the staged feature tile (32 + 4: conflict-free b128 reads)
// The haloed previous frame and distributions sit in LDS as ONE record per pixel - [r, g, b, d0 | d1, d2, d3, -] - and
// the sample's CDNA kernels as one 12-float row per tap, so the 25 taps of a pixel cost 25 x (1 or 2) + 25 x 3
// ds_read_b128 / b32 instead of 25 x (3 + ND) + 25 x 9 ds_read_b32: the compositing is LDS-instruction bound.
__host__ __device__ constexpr int comp_px_stride(int nd) { return nd <= 1 ? 4 : 8; }
constexpr int kCompKernPad = 12;        // floats per tap row of the CDNA kernels in LDS (K <= 10 used, 16-byte aligned rows)
template <int ND, int K>
__host__ __device__ constexpr int composite_small_floats() {
    return (kCompTile + 4) * (kCompTile + 4) * comp_px_stride(ND) + kTaps * kCompKernPad + 2 + ND + 2 /*pad*/ + 2 * 4 * 2 * ND;
}
template <int ND, int K>
__host__ __device__ constexpr int composite_lds_floats() {
    return ((composite_small_floats<ND, K>() + 3) & ~3) + kCompTile * kCompTile * kCompEncPad;
}

// The two 1x1 heads of one pixel, accumulated over 32 feature channels [c_off, c_off + 32) of the LDS feature row `feat`
// (LN + relu applied on the way).  The 512 head / LayerNorm weights are the same for every lane: read through the CONSTANT
// address space they are scalar loads feeding the VALU as SGPR operands - as plain global pointers the compiler issued 132
// vector loads (and as many waits) per pixel, which was half of the compositing time.
template <int K, class PT>
__device__ __forceinline__ void composite_head_bias(const PT &p, float (&o_rgb)[3], float (&o_m)[K + 1]) {
    typedef const __attribute__((address_space(4))) float cfloat;
    cfloat *brgb_ = (cfloat *)(unsigned long long)p.b_rgb, *bmask_ = (cfloat *)(unsigned long long)p.b_mask;
#pragma unroll
    for (int j = 0; j < 3; ++j) o_rgb[j] = brgb_[j];
#pragma unroll
    for (int j = 0; j < K + 1; ++j) o_m[j] = bmask_[j];
}
template <int K, class PT>
__device__ __forceinline__ void composite_heads(const PT &p, const float *feat, const float mean, const float rstd,
                                                const int c_off, float (&o_rgb)[3], float (&o_m)[K + 1]) {
    constexpr int NM = K + 1;
    typedef const __attribute__((address_space(4))) float cfloat;
    auto as_const = [](const float *q) { return (cfloat *)(unsigned long long)q; };
    cfloat *gam_ = as_const(p.gamma) + c_off, *bet_ = as_const(p.beta) + c_off;
    cfloat *wrgb_ = as_const(p.w_rgb) + c_off * 3, *wmask_ = as_const(p.w_mask) + c_off * NM;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(feat);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const f32x4 raw = src[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = q * 4 + e;
            const float f = fmaxf(fmaf((raw[e] - mean) * rstd, gam_[c], bet_[c]), 0.f);
#pragma unroll
            for (int j = 0; j < 3; ++j) o_rgb[j] = fmaf(f, wrgb_[c * 3 + j], o_rgb[j]);
#pragma unroll
            for (int j = 0; j < NM; ++j) o_m[j] = fmaf(f, wmask_[c * NM + j], o_m[j]);
        }
    }
}
template <int ND, int K, bool FIRST, class PT>
__device__ __forceinline__ void composite_finish(const PT &p, const int b, const int y, const int x, float (&o_rgb)[3],
                                                 float (&o_m)[K + 1], const float *s_px, const float *s_kern, const int halo_w,
                                                 const int hy, const int hx, const int *goal, double (&cost)[2 * ND]);
template <int ND, int K, bool FIRST, class PT>
__device__ __forceinline__ void composite_values(const PT &p, const int y, const int x, float (&o_rgb)[3],
                                                 float (&o_m)[K + 1], const float *s_px, const float *s_kern, const int halo_w,
                                                 const int hy, const int hx, const int *goal, double (&cost)[2 * ND],
                                                 float (&of)[3], float (&od)[ND]);

// One output pixel (y, x) of sample b: LN + relu of its 32 features (a row of the LDS feature tile), the two 1x1
// heads, softmax, the effective 5x5 flow kernel over the haloed previous frame / distributions in LDS
// (halo_w = pixels per halo row, (hy, hx) = the pixel's position inside the halo tile, halo 2), outputs and cost terms.
// Shared by the stand-alone compositing tile and the fused transposed-conv + compositing tile: same expressions,
// same bits.
template <int ND, int K, bool FIRST, class PT>
__device__ __forceinline__ void composite_pixel(const PT &p, const int b, const int y, const int x, const float *feat,
                                                const float mean, const float rstd, const float *s_px,
                                                const float *s_kern, const int halo_w,
                                                const int hy, const int hx, const int *goal, double (&cost)[2 * ND]) {
    constexpr int NM = K + 1;
    static_assert(K <= 10 && ND <= 4, "LDS record layouts");
    float o_rgb[3], o_m[NM];
    composite_head_bias<K>(p, o_rgb, o_m);
    composite_heads<K>(p, feat, mean, rstd, 0, o_rgb, o_m);
    composite_finish<ND, K, FIRST>(p, b, y, x, o_rgb, o_m, s_px, s_kern, halo_w, hy, hx, goal, cost);
}
// ... its next-frame pixel and distributions returned instead of stored (the fused top turns a block over in LDS and
// stores whole 16-byte pieces write-through: vf_fused_top.h) - the same expressions, the same bits
template <int ND, int K, bool FIRST, class PT>
__device__ __forceinline__ void composite_pixel_values(const PT &p, const int y, const int x, const float *feat,
                                                       const float mean, const float rstd, const float *s_px,
                                                       const float *s_kern, const int halo_w, const int hy, const int hx,
                                                       const int *goal, double (&cost)[2 * ND], float (&of)[3], float (&od)[ND]) {
    constexpr int NM = K + 1;
    float o_rgb[3], o_m[NM];
    composite_head_bias<K>(p, o_rgb, o_m);
    composite_heads<K>(p, feat, mean, rstd, 0, o_rgb, o_m);
    composite_values<ND, K, FIRST>(p, y, x, o_rgb, o_m, s_px, s_kern, halo_w, hy, hx, goal, cost, of, od);
}

// the part of a pixel behind its head outputs: softmax, effective flow kernel, next frame / distributions, cost terms
template <int ND, int K, bool FIRST, class PT>
__device__ __forceinline__ void composite_finish(const PT &p, const int b, const int y, const int x, float (&o_rgb)[3],
                                                 float (&o_m)[K + 1], const float *s_px, const float *s_kern, const int halo_w,
                                                 const int hy, const int hx, const int *goal, double (&cost)[2 * ND]) {
    float of[3], od[ND];
    composite_values<ND, K, FIRST>(p, y, x, o_rgb, o_m, s_px, s_kern, halo_w, hy, hx, goal, cost, of, od);
    const long long o = (long long)y * p.W + x;
    float *fo = p.out_frame + (long long)b * p.out_frame_bstride + o * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) fo[c] = of[c];
    float *dout = p.out_distrib + (long long)b * p.out_distrib_bstride + o * ND;
#pragma unroll
    for (int d = 0; d < ND; ++d) dout[d] = od[d];
}
template <int ND, int K, bool FIRST, class PT>
__device__ __forceinline__ void composite_values(const PT &p, const int y, const int x, float (&o_rgb)[3],
                                                 float (&o_m)[K + 1], const float *s_px, const float *s_kern, const int halo_w,
                                                 const int hy, const int hx, const int *goal, double (&cost)[2 * ND],
                                                 float (&of)[3], float (&od)[ND]) {
    constexpr int NM = K + 1;
    constexpr int PS = comp_px_stride(ND);
    auto load_px = [&](const int sp, float (&fr)[3], float (&di)[ND]) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(s_px + sp * PS);
        fr[0] = a[0]; fr[1] = a[1]; fr[2] = a[2]; di[0] = a[3];
        if constexpr (ND > 1) {
            const f32x4 c2 = *reinterpret_cast<const f32x4 *>(s_px + sp * PS + 4);
#pragma unroll
            for (int d = 1; d < ND; ++d) di[d] = c2[d - 1];
        }
    };
    float mx = o_m[0];
#pragma unroll
    for (int j = 1; j < NM; ++j) mx = fmaxf(mx, o_m[j]);
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < NM; ++j) { o_m[j] = __expf(o_m[j] - mx); den += o_m[j]; }
    const float inv = 1.0f / den;
#pragma unroll
    for (int j = 0; j < NM; ++j) o_m[j] *= inv;

    // ---- per-pixel effective flow kernel: keff[tap] = sum_k mask[k+2] * kern[tap][k]
    // (arch 1: mask 2 weighs the first context frame, the warps use masks 3.. and kernels 0..K-3)
    const int ctr = (hy + 2) * halo_w + (hx + 2);
    {
        float fr[3], di[ND];
        load_px(ctr, fr, di);
#pragma unroll
        for (int c = 0; c < 3; ++c) of[c] = fmaf(o_m[0], fr[c], o_m[1] * sigmoidf_(o_rgb[c]));
        // (K == 6, arch 2: the published generator composes the distributions with the PREVIOUS distribution in the
        // scratch layer's slot too - the scratch image has no distribution of its own)
#pragma unroll
        for (int d = 0; d < ND; ++d) od[d] = (K == 6 ? o_m[0] + o_m[1] : o_m[0]) * di[d];
    }
    if constexpr (FIRST) {
        const long long o1 = (long long)y * p.W + x;
#pragma unroll
        for (int c = 0; c < 3; ++c) of[c] = fmaf(o_m[2], p.first_frame[o1 * 3 + c], of[c]);
#pragma unroll
        for (int d = 0; d < ND; ++d) od[d] = fmaf(o_m[2], p.first_distrib[o1 * ND + d], od[d]);
        // the warp loop below pairs kernel k with o_m[k + 2]: shift the warp masks down by one and
        // retire the last kernel (its product with 0 leaves the sum unchanged)
#pragma unroll
        for (int j = 2; j < K; ++j) o_m[j] = o_m[j + 1];
        o_m[K] = 0.f;
    }
#pragma unroll
    for (int dy = 0; dy < kDnaKern; ++dy) {
#pragma unroll
        for (int dx = 0; dx < kDnaKern; ++dx) {
            const int tap = dy * kDnaKern + dx;
            float kr[kCompKernPad];
#pragma unroll
            for (int q = 0; q < (K - 1 + 3) / 4; ++q) {
                const f32x4 kq = *reinterpret_cast<const f32x4 *>(s_kern + tap * kCompKernPad + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) kr[4 * q + e] = kq[e];
            }
            float ke = 0.f;
#pragma unroll
            for (int k = 0; k < K - 1; ++k) ke = fmaf(o_m[k + 2], kr[k], ke);
            const int sp = (hy + dy) * halo_w + (hx + dx);
            float fr[3], di[ND];
            load_px(sp, fr, di);
#pragma unroll
            for (int c = 0; c < 3; ++c) of[c] = fmaf(ke, fr[c], of[c]);
#pragma unroll
            for (int d = 0; d < ND; ++d) od[d] = fmaf(ke, di[d], od[d]);
        }
    }
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const float ry = (float)(y - goal[2 * d]), rx = (float)(x - goal[2 * d + 1]);
        const float dist = sqrtf(fmaf(ry, ry, rx * rx));
        cost[2 * d] = (double)od[d];
        cost[2 * d + 1] = (double)od[d] * (double)dist;
    }
}

// one 16x16 pixel tile of one sample.  FIRST (arch 1, savp_arch.py): the first context frame is one more compositing
// layer - a template parameter, because a run-time branch around the mask bookkeeping costs the CDNA path 3.5 us per
// tile (measured).
template <int ND, int K, bool FIRST, class PT, int CFT = 32>
__device__ __forceinline__ void composite_tile(const PT &p, const int tile, const int b, const int *goal,
                                               float *smem) {
    constexpr int TS = kCompTile, HS = TS + 4;
    constexpr int NM = K + 1;
    constexpr int PS = comp_px_stride(ND);
    float *s_px = smem;                                 // [HS*HS][PS]: frame, distributions
    float *s_kern = s_px + HS * HS * PS;                // [kTaps][kCompKernPad]
    float *s_ln = s_kern + kTaps * kCompKernPad;        // [2]
    float *s_dscale = s_ln + 2;                         // [ND]
    float *s_enc = smem + ((composite_small_floats<ND, K>() + 3) & ~3);     // [TS*TS][kCompEncPad]

    const int tid = threadIdx.x;
    const int tilesX = (p.W + TS - 1) / TS;
    const int nblocks = sum_blocks(p.H, p.W);
    const int ty0 = (tile / tilesX) * TS, tx0 = (tile % tilesX) * TS;
    constexpr int CF = CFT;                     // feature channels per pixel (64: the public decoder table, two rounds below)

    // The tile's 32-channel features are fetched cooperatively - a wave instruction reads 1 KiB of consecutive
    // pixels - and transposed through LDS; one thread reading the 128 bytes of "its" pixel straight from memory
    // touches 64 cache lines per load instruction.  Issued first: the loads fly during the prologue below.
    f32x4 ev[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = tid + 256 * k, px = i >> 3, q = i & 7;
        const int yy = ty0 + (px >> 4), xx = tx0 + (px & 15);
        ev[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (yy < p.H && xx < p.W)
            ev[k] = *reinterpret_cast<const f32x4 *>(p.enc6 + (((long long)b * p.H + yy) * p.W + xx) * CF + q * 4);
    }

    // prologue reductions, lanes over the partials (32 LayerNorm partials and 64 tiles per sample at 128x128):
    // wave 0 the exact LayerNorm statistics, wave (d + 1) & 3 the mass of distribution d (fixed order: lanes
    // stride the tiles, then the xor butterfly - deterministic)
    {
        const int lane_ = tid & 63, wave_ = tid >> 6;
        if (wave_ == 0) {
            long long su = 0, sq = 0;
            const long long *pp = p.ln_part + (long long)b * p.ln_nparts * 2;
            for (int k = lane_; k < p.ln_nparts; k += 64) { su += pp[2 * k]; sq += pp[2 * k + 1]; }
            su = wave_sum(su); sq = wave_sum(sq);
            if (lane_ == 0) ln_from_totals(su, sq, p.ln_inv_n, s_ln[0], s_ln[1]);
        }
        for (int d = (wave_ + 3) & 3; d < ND; d += 4) {
            float sc = 1.0f;
            if (p.prev_sums) {
                double su = 0.0;
                const double *pp = p.prev_sums + ((long long)b * ND + d) * nblocks * 2;
                for (int k = lane_; k < nblocks; k += 64) su += pp[2 * k];
                su = wave_sum(su);
                sc = (float)(1.0 / su);
            }
            if (lane_ == 0) s_dscale[d] = sc;
        }
    }
    for (int i = tid; i < kTaps * K; i += 256) s_kern[(i / K) * kCompKernPad + i % K] = p.kern[(long long)b * kTaps * K + i];
    __syncthreads();

    const float *pf = p.prev_frame + (long long)b * p.prev_frame_bstride;
    const float *pd = p.prev_distrib + (long long)b * p.prev_distrib_bstride;
    for (int i = tid; i < HS * HS; i += 256) {
        const int ly = i / HS, lx = i % HS;
        const int y = ty0 + ly - 2, x = tx0 + lx - 2;
        const bool in = y >= 0 && y < p.H && x >= 0 && x < p.W;
        const long long o = (long long)y * p.W + x;
#pragma unroll
        for (int c = 0; c < 3; ++c) s_px[i * PS + c] = in ? pf[o * 3 + c] : 0.f;
#pragma unroll
        for (int d = 0; d < ND; ++d) s_px[i * PS + 3 + d] = in ? pd[o * ND + d] * s_dscale[d] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = tid + 256 * k;
        *reinterpret_cast<f32x4 *>(&s_enc[(i >> 3) * kCompEncPad + (i & 7) * 4]) = ev[k];
    }
    __syncthreads();

    const int ly = tid / TS, lx = tid % TS;
    const int y = ty0 + ly, x = tx0 + lx;
    const bool valid = y < p.H && x < p.W;
    double cost[2 * ND];
#pragma unroll
    for (int i = 0; i < 2 * ND; ++i) cost[i] = 0.0;

    if constexpr (CF == 32) {
        if (valid)
            composite_pixel<ND, K, FIRST>(p, b, y, x, &s_enc[tid * kCompEncPad], s_ln[0], s_ln[1], s_px, s_kern, HS,
                                          ly, lx, goal, cost);
    } else {
        // 64 feature channels (cdna_arch.py decoder='public'): the heads accumulate over two 32-channel rounds through the
        // same LDS feature tile - channel order 0 .. 63, i.e. the fma chain of the one-round form continued
        float o_rgb[3], o_m[NM];
        composite_head_bias<K>(p, o_rgb, o_m);
        composite_heads<K>(p, &s_enc[tid * kCompEncPad], s_ln[0], s_ln[1], 0, o_rgb, o_m);     // (a pixel outside the image reads zeros)
#pragma unroll
        for (int c_off = 32; c_off < CF; c_off += 32) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = tid + 256 * k, px = i >> 3, q = i & 7;
                const int yy = ty0 + (px >> 4), xx = tx0 + (px & 15);
                f32x4 e = {0.f, 0.f, 0.f, 0.f};
                if (yy < p.H && xx < p.W)
                    e = *reinterpret_cast<const f32x4 *>(p.enc6 + (((long long)b * p.H + yy) * p.W + xx) * CF + c_off + q * 4);
                *reinterpret_cast<f32x4 *>(&s_enc[(i >> 3) * kCompEncPad + (i & 7) * 4]) = e;
            }
            __syncthreads();
            composite_heads<K>(p, &s_enc[tid * kCompEncPad], s_ln[0], s_ln[1], c_off, o_rgb, o_m);
        }
        if (valid) composite_finish<ND, K, FIRST>(p, b, y, x, o_rgb, o_m, s_px, s_kern, HS, ly, lx, goal, cost);
    }
    // ---- cost sums of this wave's block (4 rows x 16 columns): lanes in a fixed butterfly, one entry per block
#pragma unroll
    for (int i = 0; i < 2 * ND; ++i) cost[i] = wave_sum(cost[i]);
    const int lane = tid & 63, wave = tid >> 6;
    const int by = ty0 / kSumBlockH + wave, bxk = tx0 / kSumBlockW;
    if (lane == 0 && by * kSumBlockH < p.H && tx0 < p.W) {
        const int blk = by * sum_blocks_x(p.W) + bxk;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            double *dst = p.out_sums + (((long long)b * ND + d) * nblocks + blk) * 2;
            dst[0] = cost[2 * d]; dst[1] = cost[2 * d + 1];
        }
    }
}

template <int ND, int K>
VF_GLOBAL VF_LAUNCH_BOUNDS(256) void composite_kernel(const CompositeParams p) {
    __shared__ __attribute__((aligned(16))) float smem[composite_lds_floats<ND, K>()];
    if (K == 6 || p.first_frame) composite_tile<ND, K, true>(p, blockIdx.x, blockIdx.y, &p.goal[0][0], smem);
    else if constexpr (K != 6) {
        if (p.CF > 32) composite_tile<ND, K, false, CompositeParams, 64>(p, blockIdx.x, blockIdx.y, &p.goal[0][0], smem);
        else composite_tile<ND, K, false>(p, blockIdx.x, blockIdx.y, &p.goal[0][0], smem);
    }
}

// ------------------------------------------------------------------------------------------
// Cost finalisation.  Per rolled sequence r and task (view v, pixel d):
//     e[r][v*ND+d] = sum_t w_t (S1/S0)(t, v, r, d) / sum_t w_t,      w = (1, ..., 1, finalweight)
// (reference pixel_cost_controller.py:168-187); n_draws consecutive sequences are the latent draws
// of one action and are averaged; the action's score is the plain mean over tasks (reference :153)
// or, with use_weights, the trade-off weighted sum (register_gtruth_controller.py:88-94).
// Scores leave the device in float64 (the reference's host cost is float64 under NumPy >= 2,
// SURVEY section 7): no fp32 rounding can merge two distinct costs into a tie before the argsort.
// sums[t]: [ncam][Bcap][ND][blocks][2] (ntiles below = 4x16-pixel blocks per image).  A non-zero *status (a tile of the rollout gave up
// waiting for its producers) poisons every score with NaN, so a failed rollout cannot feed CEM.
constexpr int kMaxCam = 4;
struct TaskWeights { int use; float w[kMaxCam * kMaxDesig]; };

// One WAVE per action: the lanes stride the blocks of a (step, task, draw) and a fixed butterfly adds them up
// (deterministic; every reader of the sums uses this order).
VF_GLOBAL void scores_kernel(const double *sums, long long step_stride, long long view_stride, int n_actions,
                              int n_draws, int T, int ND, int ncam, int ntiles, float finalweight,
                              const TaskWeights tw, const int *status, double *scores, double *scores_per_task) {
    const int a = blockIdx.x, lane = threadIdx.x;
    if (a >= n_actions) return;
    const int ntask = ncam * ND;
    const bool poisoned = status && *status != 0;
    double total = 0.0;
    for (int v = 0; v < ncam; ++v)
        for (int d = 0; d < ND; ++d) {
            double over_draws = 0.0;
            for (int j = 0; j < n_draws; ++j) {
                const long long b = (long long)a * n_draws + j;
                double acc = 0.0, wsum = 0.0;
                for (int t = 0; t < T; ++t) {
                    const double *pp = sums + (long long)t * step_stride + (long long)v * view_stride +
                                       (b * ND + d) * ntiles * 2;
                    double s0 = 0.0, s1 = 0.0;
                    for (int k = lane; k < ntiles; k += 64) { s0 += pp[2 * k]; s1 += pp[2 * k + 1]; }
                    s0 = wave_sum(s0); s1 = wave_sum(s1);
                    const double w = (t == T - 1) ? (double)finalweight : 1.0;
                    acc += w * (s1 / s0);
                    wsum += w;
                }
                over_draws += acc / wsum;
            }
            const double sc = over_draws / n_draws;
            const int col = v * ND + d;
            if (scores_per_task && lane == 0)
                scores_per_task[(long long)a * ntask + col] = poisoned ? __builtin_nan("") : sc;
            total += tw.use ? (double)tw.w[col] * sc : sc;
        }
    const double out = tw.use ? total : total / ntask;
    if (lane == 0) scores[a] = poisoned ? __builtin_nan("") : out;
}

// predictions out in the reference layout: dst[bb][t][view][hw][C] <- src[view][Bcap][t][hw][C]
// (reference vpred_model_interface.py:78,88 stacks the views on axis 2)
VF_GLOBAL void export_frames_kernel(const float *src, long long view_stride, int first, int count, int T,
                                     int ncam, int HWC, float *dst) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per_sample = (long long)T * ncam * HWC;
    if (i >= (long long)count * per_sample) return;
    const int bb = (int)(i / per_sample);
    const long long r = i - (long long)bb * per_sample;
    const int t = (int)(r / ((long long)ncam * HWC));
    const int v = (int)((r / HWC) % ncam);
    const int e = (int)(r % HWC);
    dst[i] = src[(long long)v * view_stride + ((long long)(first + bb) * T + t) * HWC + e];
}

// normalised distributions out: dst[bb][t][view][hw][d] = src[view][b][t][hw][d] / S0(t, view, b, d)
VF_GLOBAL void export_distrib_kernel(const float *src, long long view_stride, const double *sums,
                                      long long step_stride, long long sums_view_stride, int first, int count,
                                      int T, int ncam, int HW, int ND, int ntiles, float *dst) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per_sample = (long long)T * ncam * HW * ND;
    if (i >= (long long)count * per_sample) return;
    const int bb = (int)(i / per_sample);
    const long long r = i - (long long)bb * per_sample;
    const int t = (int)(r / ((long long)ncam * HW * ND));
    const int v = (int)((r / ((long long)HW * ND)) % ncam);
    const long long e = r % ((long long)HW * ND);
    const int d = (int)(e % ND);
    const int b = first + bb;
    const double *pp = sums + (long long)t * step_stride + (long long)v * sums_view_stride +
                       ((long long)b * ND + d) * ntiles * 2;
    double s0 = 0.0;
    for (int k = 0; k < ntiles; ++k) s0 += pp[2 * k];
    dst[i] = (float)((double)src[(long long)v * view_stride + ((long long)b * T + t) * HW * ND + e] / s0);
}

// ------------------------------------------------------------------------------------------
// Registration (reference register_gtruth_controller.py:54-173).  A flow field (dx, dy) per
// reference pixel - the output of the plug-in registration network - maps reference pixel
// (r, c) to the point (x, y) = (c + dx, r + dy) of the current frame ("warp_pts", :64-66).
__device__ __forceinline__ float bilinear_clamped(const float *img, int H, int W, float x, float y, int ch) {
    x = fminf(fmaxf(x, 0.f), (float)(W - 1));
    y = fminf(fmaxf(y, 0.f), (float)(H - 1));
    const int x0 = (int)floorf(x), y0 = (int)floorf(y);
    const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
    const float fx = x - (float)x0, fy = y - (float)y0;
    const float a = img[((long long)y0 * W + x0) * 3 + ch], b = img[((long long)y0 * W + x1) * 3 + ch];
    const float c = img[((long long)y1 * W + x0) * 3 + ch], d = img[((long long)y1 * W + x1) * 3 + ch];
    const float top = fmaf(fx, b - a, a), bot = fmaf(fx, d - c, c);
    return fmaf(fy, bot - top, top);
}

// warped[cam][r][c][:] = bilinear(current[cam], warp_pts[cam][r][c]);  pts[cam][r][c] = (x, y)
VF_GLOBAL void warp_image_kernel(const float *cur, const float *flow, int ncam, int H, int W, float *warped,
                                  float *pts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncam * H * W) return;
    const int cam = i / (H * W), r = (i / W) % H, c = i % W;
    const float x = (float)c + flow[2 * i], y = (float)r + flow[2 * i + 1];
    if (pts) { pts[2 * i] = x; pts[2 * i + 1] = y; }
    if (warped)
        for (int ch = 0; ch < 3; ++ch)
            warped[3 * (long long)i + ch] = bilinear_clamped(cur + (long long)cam * H * W * 3, H, W, x, y, ch);
}

// One workgroup (128 threads) per (camera, task): re-localise the task's pixel and measure the
// warp error.  region == 0: the flow and the L2 photometric error at the pixel itself (:129-135,
// 163-170); region > 0: median flow and mean squared error over the (2*region+1)^2 window
// clipped to [0, size - clip_sub] (:139-161; the reference clips the start window with
// clip_sub = 1 and the goal window with 0).  desig = (row, col) float; an empty window gives NaN.
constexpr int kRegMaxWin = 11 * 11;
VF_GLOBAL VF_LAUNCH_BOUNDS(128) void register_kernel(const float *cur, const float *ref, const float *flow,
                                                       const int *pix, int ncam, int ntask, int H, int W,
                                                       int region, int clip_sub, float *desig, float *err) {
    __shared__ float s_x[kRegMaxWin], s_y[kRegMaxWin];
    __shared__ double s_red[2];
    __shared__ float s_med[4];
    const int cam = blockIdx.x / ntask, task = blockIdx.x % ntask, tid = threadIdx.x;
    const int pr = pix[(cam * ntask + task) * 2], pc = pix[(cam * ntask + task) * 2 + 1];
    const float *cur_c = cur + (long long)cam * H * W * 3, *ref_c = ref + (long long)cam * H * W * 3;
    const float *flow_c = flow + (long long)cam * H * W * 2;
    float *out_d = desig + (cam * ntask + task) * 2;
    if (region == 0) {
        if (tid == 0) {
            const int r = min(max(pr, 0), H - 1), c = min(max(pc, 0), W - 1);
            const float x = (float)c + flow_c[(r * W + c) * 2], y = (float)r + flow_c[(r * W + c) * 2 + 1];
            double acc = 0.0;
            for (int ch = 0; ch < 3; ++ch) {
                const float df = ref_c[(r * W + c) * 3 + ch] - bilinear_clamped(cur_c, H, W, x, y, ch);
                acc += (double)df * (double)df;
            }
            out_d[0] = y; out_d[1] = x;
            err[cam * ntask + task] = (float)sqrt(acc);
        }
        return;
    }
    const int hi_r = H - clip_sub, hi_c = W - clip_sub;
    const int r0 = min(max(pr - region, 0), hi_r), r1 = min(max(pr + region + 1, 0), hi_r);
    const int c0 = min(max(pc - region, 0), hi_c), c1 = min(max(pc + region + 1, 0), hi_c);
    const int wh = max(r1 - r0, 0), ww = max(c1 - c0, 0), n = wh * ww;
    if (n == 0) {
        if (tid == 0) { out_d[0] = out_d[1] = __builtin_nanf(""); err[cam * ntask + task] = __builtin_nanf(""); }
        return;
    }
    double sq = 0.0;
    if (tid < n) {
        const int r = r0 + tid / ww, c = c0 + tid % ww;
        const float x = (float)c + flow_c[(r * W + c) * 2], y = (float)r + flow_c[(r * W + c) * 2 + 1];
        s_x[tid] = x; s_y[tid] = y;
        for (int ch = 0; ch < 3; ++ch) {
            const float df = ref_c[(r * W + c) * 3 + ch] - bilinear_clamped(cur_c, H, W, x, y, ch);
            sq += (double)df * (double)df;
        }
    }
    sq = wave_sum(sq);
    if ((tid & 63) == 0) s_red[tid >> 6] = sq;
    __syncthreads();
    // medians by rank: element i has rank #{j : v_j < v_i or (v_j == v_i and j < i)}
    if (tid < n) {
        const float vx = s_x[tid], vy = s_y[tid];
        int rx = 0, ry = 0;
        for (int j = 0; j < n; ++j) {
            rx += (s_x[j] < vx) || (s_x[j] == vx && j < tid);
            ry += (s_y[j] < vy) || (s_y[j] == vy && j < tid);
        }
        const int lo = (n - 1) / 2, hi = n / 2;
        if (rx == lo) s_med[0] = vx;
        if (rx == hi) s_med[1] = vx;
        if (ry == lo) s_med[2] = vy;
        if (ry == hi) s_med[3] = vy;
    }
    __syncthreads();
    if (tid == 0) {
        out_d[0] = 0.5f * (s_med[2] + s_med[3]);        // row <- median y
        out_d[1] = 0.5f * (s_med[0] + s_med[1]);        // col <- median x
        err[cam * ntask + task] = (float)((s_red[0] + s_red[1]) / (double)(3 * n));
    }
}

}  // namespace vf
