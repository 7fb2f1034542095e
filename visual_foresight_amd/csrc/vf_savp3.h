// vf_savp3.h - the non-GEMM items of arch 3, the published SAVP generator (video_prediction/savp3_arch.py; Lee et al. 2018,
// arXiv:1804.01523 appendix A; the reference only instantiates the class, visual_mpc/video_prediction/vpred_model_interface.py:52-58).
//
// What differs from the CDNA-core networks (arch 0 - 2) and why it needs its own items:
//   * INSTANCE normalisation - per sample AND channel over H x W - follows every convolution and sits INSIDE the conv-LSTM
//     cell (over the 4C gate pre-activations and over the new cell state).  A statistic of the whole image stands between a
//     GEMM tile's accumulators and the values that depend on them, so the gate math cannot live in the GEMM epilogue: the
//     GEMM tiles of arch 3 store raw outputs (EPI_RAW / the raw gate-split epilogue) and ONE element-wise item per (sample,
//     channel group) owns all pixels of its channels - it computes the statistics itself (fixed order, float64: the result
//     does not depend on the GEMM's tiling or on the batch), normalises, runs the cell update, takes the statistics of the
//     new cell state, normalises again and writes c and h.  No cross-item reduction, no extra dependency hop per statistic.
//   * conv + 2x2 average pool is ONE stride-2 convolution with the box-filtered kernel (k + 1) x (k + 1) (packed on the
//     host): 2.8x (5x5) / 2.25x (3x3) fewer MACs than the literal pair, same sums up to fp32 association.
//   * bilinear 2x up-sampling + 3x3 conv: the up-sampled tensor is materialised by an element-wise item (the 3x3 conv at the
//     up-sampled resolution has the MAC count of any fused form - 36 (tap, parity) blocks per source pixel - so fusing would
//     only save the round trip through memory).
//   * the conditioning vector v = [a, s, rnn_z(z)] is tile-concatenated to the input of every conv and every conv-LSTM: as in
//     arch 2 it never becomes GEMM rows - a spatially constant input contributes a bias that only depends on the pixel's
//     border class (5 x 5 classes) - but here the tables follow each layer's geometry (pooled conv: 3 classes per axis,
//     up-sampled conv: the bilinear kernel attenuates the outermost up-sampled row to 0.75) through a per-layer coefficient
//     table f[class][tap]:  bias[ry][rx] = sum_ty f[ry][ty] sum_tx f[rx][tx] (W_cond . v)[ty][tx].
//   * dependent masks: the mask head convolves [h_masks | the seven compositing layers], so the layers (CDNA warps with
//     SYMMETRIC padding, previous, first, scratch) are materialised (EW_TRANSFORM) before the 3x3 mask conv and composed
//     after it (EW_COMPOSE).
#pragma once
#include <hip/hip_runtime.h>
#include "vf_conv_mfma.h"
#include "vf_small_kernels.h"

namespace vf {

enum EwOp { EW_SA3 = 0, EW_COND3, EW_INORM, EW_INCELL, EW_UPSAMPLE, EW_TRANSFORM, EW_COMPOSE, EW_TOP3 };

constexpr int kTransCh = 32;        // floats per pixel of the materialised compositing layers (21 used: 4 warps, previous, first, scratch)
constexpr int kMaskCh = 8;          // floats per pixel of the mask logits (7 used)
constexpr int kScrCh = 4;           // floats per pixel of the raw scratch image (3 used)
constexpr int kNumWarp3 = 4;        // CDNA kernels of the published generator

struct Sa3Params {
    const float *action; long long action_bs;   // [adim] = [a_env | z] per sample (0 stride: shared)
    const float *state; long long state_bs;     // [sdim]
    int adim, sdim, zdim;
    const float *w_state, *b_state;             // [a_env + sdim][sdim], [sdim]
    const float *w_rnnz, *b_rnnz;               // [2 zdim][4 zdim], [4 zdim]  (BasicLSTMCell: gate order i, j, f, o)
    float *rnn_state;                           // [B][2 zdim] = (c, h), updated in place
    int first;                                  // 1: the previous rnn state is zero
    float *condvec;                             // [B][a_env + sdim + zdim] = [a, s, rnn_z]
    float *state_out; long long state_out_bs;   // may be null
};

struct Cond3Params {
    const float *condvec;       // [B][ncond]
    int ncond;
    const float *w;             // [KH * KH][ncond][Ctot]: the conditioning rows of the layer's (effective) kernel
    int Ctot, KH;
    float f[5][8];              // f[border class][tap]: weight with which a tap of the constant input reaches a pixel of that class
    float *out;                 // [B][25][Ctot]
};

struct NormParams {
    const float *in; long long in_bs;       // INORM: raw conv output [HW][C]; INCELL: raw gates [HW][4C]
    const float *cond; long long cond_bs;   // border-class biases [25][C] / [25][4C], or null
    const float *g0, *b0;                   // INORM: gain / offset [C]; INCELL: of the gate norm [4C]
    const float *g1, *b1;                   // INCELL: of the cell-state norm [C]
    const float *cprev; long long cprev_bs; // INCELL: previous cell state [HW][C], null = zero
    float *out; long long out_bs;           // INORM: normalised (may alias in); INCELL: h
    float *out2; int split;                 // INORM, split > 0: channels >= split go to out2 (both outputs `split` channels
                                            // per pixel: the two heads' hidden layers leave one fused conv)
    float *tab;                             // INORM, non-null: statistics only - (scale, shift) of every channel -> tab[B][C][2];
                                            // the consumer (EW_TOP3) normalises while staging
    float *cout; long long cout_bs;         // INCELL: new cell state (may alias cprev)
    int H, W, C, cpi, relu;
    float eps;
};

struct UpParams {
    const float *in0; long long in0_bs; int C0;
    const float *in1; long long in1_bs; int C1;     // second source (skip connection), C1 = 0: none
    float *out; long long out_bs;                   // [2h][2w][C0 + C1]
    int h, w, rows;                                 // source size; output rows per item
};

struct TopParams {
    int H, W, ND;
    const float *prev_frame; long long prev_frame_bs;
    const float *prev_distrib; long long prev_distrib_bs;
    const double *prev_sums;                // [B][ND][blocks][2] partial sums of prev_distrib, or null (already normalised)
    const float *first_frame, *first_distrib;
    const float *kern;                      // [B][25][4]
    const float *scr_raw;                   // [B][HW][kScrCh]
    float *trans;                           // [B][HW][kTransCh]
    float *transd;                          // [B][HW][4 ND]
    const float *mlog;                      // [B][HW][kMaskCh]
    float *out_frame; long long out_frame_bs;
    float *out_distrib; long long out_distrib_bs;
    double *out_sums;
    int goal[kMaxDesig][2];                 // per-layer launches only (the persistent kernel takes them from its launch arguments)
    // EW_TOP3 (the heads' second phase as ONE item per 8 x 16-pixel tile): the raw hidden layers of both heads and their
    // instance-norm tables, the heads' output convs in the layout the VALU loops read
    const float *hmhs_raw;                  // [B][HW][64]: hm 0..31 | hs 32..63
    const float *norm_tab;                  // [B][64][2]: scale, shift
    const float *w_scr, *b_scr;             // [9 taps][32][4], [4]
    const float *w_msk, *b_msk;             // [9 taps][kT3MaskIn][8], [8]: input channels hm 0..31, layers 32..52, zeros beyond
};

struct EwParams {
    int op, B;
    int gx;             // items per sample (spi == 0)
    int spi;            // > 0: an item covers spi consecutive samples (EW_SA3, EW_COND3)
    int wt;             // the item's outputs leave as 16-byte sc1 / atomic stores: it publishes without a release fence
                        // (persistent schedule only, set by build_schedule; ConvParams::wt_out has the reasoning)
    union {
        Sa3Params sa;
        Cond3Params cond;
        NormParams norm;
        UpParams up;
        TopParams top;
    };
};

constexpr int kEwBatch = 4;         // pixels whose loads an element-wise item keeps in flight per thread
constexpr int kEwLdsFloats = 20 * 20 * 8 + 25 * 4 + 16 + 2 * 4 * 8 * 32;   // largest: EW_TRANSFORM halo; reductions [4][8][32] doubles

// dynamic LDS of an element-wise item (host + device)
__host__ __device__ constexpr int top3_lds_floats(int nd);
__host__ __device__ constexpr size_t ew_lds_bytes(const int op, const int nd) {
    return (size_t)(op == EW_TOP3 ? top3_lds_floats(nd) : kEwLdsFloats) * 4 + 16;
}

// border class of coordinate y in an image of H rows: 0, 1 | 2 = interior | 3, 4   (H >= 4)
__device__ __forceinline__ int cls5(const int y, const int H) { return y < 2 ? y : (y >= H - 2 ? y - (H - 5) : 2); }

__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ void st4(float *p, const f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }
// An output tensor of one sample as a raw buffer (wave-uniform base; byte offsets below 2^31): a 16-byte store that is plain
// or - write-through items - sc1
struct OutBuf { __amdgpu_buffer_rsrc_t r; };
__device__ __forceinline__ OutBuf out_buf(const float *base) {
    OutBuf o;
    o.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7FFFFFFF, 0x00020000);
    return o;
}
__device__ __forceinline__ void st4o(const OutBuf &o, const bool wt, const unsigned byte_off, const f32x4 v) {
    if (wt) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), o.r, byte_off, 0, 16);
    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), o.r, byte_off, 0, 0);
}

// ------------------------------------------------------------------------------------------ state FC + rnn_z + conditioning vector
// one sample per wave; s = 128 floats of LDS scratch for this wave
template <class PT>
__device__ __forceinline__ void sa3_sample(const PT &p, const int b, const int t, float *s) {
    const int a_env = p.adim - p.zdim, nsa = a_env + p.sdim, nz = p.zdim, ncond = nsa + nz;
    if (t < a_env) s[t] = p.action[(long long)b * p.action_bs + t];
    else if (t < nsa) s[t] = p.state[(long long)b * p.state_bs + (t - a_env)];
    if (t < nz) {
        s[32 + t] = p.action[(long long)b * p.action_bs + a_env + t];
        s[32 + nz + t] = p.first ? 0.f : p.rnn_state[(long long)b * 2 * nz + nz + t];
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    if (t < 4 * nz) {
        float acc = 0.f;
        for (int k = 0; k < 2 * nz; ++k) acc = fmaf(s[32 + k], p.w_rnnz[k * 4 * nz + t], acc);
        s[64 + t] = acc + p.b_rnnz[t];
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    if (t < nz) {
        const float gi = s[64 + t], gj = s[64 + nz + t], gf = s[64 + 2 * nz + t], go = s[64 + 3 * nz + t];
        const float c_old = p.first ? 0.f : p.rnn_state[(long long)b * 2 * nz + t];
        float c_new, h_new;
        lstm_cell(gi, gj, gf, go, c_old, c_new, h_new);
        p.rnn_state[(long long)b * 2 * nz + t] = c_new;
        p.rnn_state[(long long)b * 2 * nz + nz + t] = h_new;
        p.condvec[(long long)b * ncond + nsa + t] = h_new;
    }
    if (t < nsa) p.condvec[(long long)b * ncond + t] = s[t];
    if (p.state_out && t < p.sdim) {
        float acc = 0.f;
        for (int k = 0; k < nsa; ++k) acc = fmaf(s[k], p.w_state[k * p.sdim + t], acc);
        p.state_out[(long long)b * p.state_out_bs + t] = acc + p.b_state[t];
    }
}

// ------------------------------------------------------------------------------------------ border-class biases of one layer
// samples [b0, b1) (at most 4) per call; sv = 128 floats of LDS.  Separable: row sums over the taps a column class sees, then
// over the rows a row class sees - in a fixed order (deterministic; independent of the batch the sample is rolled in).
constexpr int kCond3PerItem = 4;
template <class PT>
__device__ __forceinline__ void cond3_item(const PT &p, const int b0, const int b1, float *sv) {
    const int t = threadIdx.x, ns = b1 - b0, nc = p.ncond, KH = p.KH;
    if (t < kCond3PerItem * 32) {
        const int s = t >> 5, c = t & 31;
        sv[t] = (s < ns && c < nc) ? p.condvec[(long long)(b0 + s) * nc + c] : 0.f;
    }
    __syncthreads();
    // all four samples of the item per pass over the layer's conditioning weights (they are the item's memory traffic: up to
    // 2.5 MB per layer), eight weight loads in flight per thread
    static_assert(kCond3PerItem == 4, "cond3_item: four samples per pass");
    for (int col = t; col < p.Ctot; col += 256) {
        float acc[4][25];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 25; ++i) acc[s][i] = 0.f;
        for (int ty = 0; ty < KH; ++ty) {
            float tt[4][6];
#pragma unroll
            for (int tx = 0; tx < 6; ++tx) {
                float a[4] = {0.f, 0.f, 0.f, 0.f};
                if (tx < KH) {
                    const float *wp = p.w + ((long long)(ty * KH + tx) * nc) * p.Ctot + col;
                    for (int c0 = 0; c0 < nc; c0 += 8) {
                        float w[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) w[j] = c0 + j < nc ? wp[(long long)(c0 + j) * p.Ctot] : 0.f;
#pragma unroll
                        for (int j = 0; j < 8; ++j)
#pragma unroll
                            for (int s = 0; s < 4; ++s) a[s] = fmaf(sv[s * 32 + c0 + j], w[j], a[s]);
                    }
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) tt[s][tx] = a[s];
            }
            float fy[5];
#pragma unroll
            for (int ry = 0; ry < 5; ++ry) fy[ry] = p.f[ry][ty];
#pragma unroll
            for (int rx = 0; rx < 5; ++rx) {
                float r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int tx = 0; tx < 6; ++tx) {
                    const float fx = p.f[rx][tx];       // (zero beyond the kernel)
#pragma unroll
                    for (int s = 0; s < 4; ++s) r[s] = fmaf(fx, tt[s][tx], r[s]);
                }
#pragma unroll
                for (int ry = 0; ry < 5; ++ry)
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[s][ry * 5 + rx] = fmaf(fy[ry], r[s], acc[s][ry * 5 + rx]);
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s >= ns) break;
            float *o = p.out + (long long)(b0 + s) * 25 * p.Ctot + col;
#pragma unroll
            for (int i = 0; i < 25; ++i) o[(long long)i * p.Ctot] = acc[s][i];
        }
    }
}

// ------------------------------------------------------------------------------------------ reductions of the norm items
// v[i] <- sum over all threads of the workgroup that share this thread's index modulo nq (nq = 1, 2, 4, 8): lanes by a
// fixed xor butterfly, the four waves in order through LDS ([4][8][NV] doubles).  Deterministic.
template <int NV>
__device__ __forceinline__ void ew_reduce(double (&v)[NV], const int nq, double *lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        for (int off = 32; off >= nq; off >>= 1) v[i] += __shfl_xor(v[i], off, 64);
    __syncthreads();                        // (the previous use of lds is over)
    if (lane < nq)
#pragma unroll
        for (int i = 0; i < NV; ++i) lds[(wave * 8 + lane) * NV + i] = v[i];
    __syncthreads();
    const int q = lane & (nq - 1);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double a = 0.0;
        for (int w = 0; w < 4; ++w) a += lds[(w * 8 + q) * NV + i];
        v[i] = a;
    }
}

__device__ __forceinline__ void in_scale_shift(const double su, const double sq, const double inv_n, const float eps,
                                               const float g, const float b, float &scale, float &shift) {
    const double m = su * inv_n;
    double var = sq * inv_n - m * m;
    var = var < 0.0 ? 0.0 : var;
    const double sc = (double)g / sqrt(var + (double)eps);
    scale = (float)sc;
    shift = (float)((double)b - m * sc);
}

// relu(IN(x + class bias)) of the channels [grp * cpi, (grp + 1) * cpi) of sample b, all pixels
template <class PT>
__device__ __forceinline__ void inorm_item(const PT &p, const int b_, const int grp, const bool wt, float *smem) {
    const int tid = threadIdx.x, b = __builtin_amdgcn_readfirstlane(b_);
    const int nq = p.cpi >> 2, nq_log2 = 31 - __builtin_clz((unsigned)nq);
    const int q = tid & (nq - 1), ps = tid >> nq_log2, ppp = kConvThreads >> nq_log2;
    const int HW = p.H * p.W, C = p.C, c0 = grp * p.cpi + 4 * q;
    const float *in = p.in + (long long)b * p.in_bs + c0;
    const float *cond = p.cond ? p.cond + (long long)b * p.cond_bs + c0 : nullptr;
    const int Co = p.split > 0 ? p.split : C;       // channels per pixel of the output tensor(s)
    const bool second = p.split > 0 && c0 >= p.split;
    const OutBuf o1 = out_buf(p.out + (long long)b * p.out_bs), o2 = out_buf((p.split > 0 ? p.out2 : p.out) + (long long)b * p.out_bs);
    const unsigned c_off = (unsigned)(second ? c0 - p.split : c0) * 4u;
    const TileDiv div_w(p.W);
    auto value = [&](const int px) {
        f32x4 v = ld4(in + (long long)px * C);
        if (cond) {
            const int y = div_w.div(px), x = px - y * p.W;
            const f32x4 cb = ld4(cond + (cls5(y, p.H) * 5 + cls5(x, p.W)) * C);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += cb[j];
        }
        return v;
    };
    double st[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) st[i] = 0.0;
    // (the items are latency chains: kEwBatch pixels' loads are issued back to back before anything is computed on them)
    for (int px0 = ps; px0 < HW; px0 += kEwBatch * ppp) {
        f32x4 v[kEwBatch];
#pragma unroll
        for (int u = 0; u < kEwBatch; ++u) { const int px = px0 + u * ppp; v[u] = px < HW ? value(px) : f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int u = 0; u < kEwBatch; ++u) {
            if (px0 + u * ppp >= HW) break;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const double d = (double)v[u][j]; st[j] += d; st[4 + j] += d * d; }
        }
    }
    ew_reduce<8>(st, nq, reinterpret_cast<double *>(smem));
    float sc[4], sh[4];
    const double inv_n = 1.0 / (double)HW;
#pragma unroll
    for (int j = 0; j < 4; ++j) in_scale_shift(st[j], st[4 + j], inv_n, p.eps, p.g0[c0 + j], p.b0[c0 + j], sc[j], sh[j]);
    if (p.tab != nullptr) {             // statistics only: the consumer applies them
        if (ps == 0) {
            const OutBuf ot = out_buf(p.tab + (long long)b * C * 2);
            st4o(ot, wt, (unsigned)(c0 * 2) * 4u, f32x4{sc[0], sh[0], sc[1], sh[1]});
            st4o(ot, wt, (unsigned)(c0 * 2 + 4) * 4u, f32x4{sc[2], sh[2], sc[3], sh[3]});
        }
        return;
    }
    for (int px0 = ps; px0 < HW; px0 += kEwBatch * ppp) {
        f32x4 v[kEwBatch];
#pragma unroll
        for (int u = 0; u < kEwBatch; ++u) { const int px = px0 + u * ppp; v[u] = px < HW ? value(px) : f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int u = 0; u < kEwBatch; ++u) {
            const int px = px0 + u * ppp;
            if (px >= HW) break;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[u][j] = fmaf(v[u][j], sc[j], sh[j]);
                if (p.relu) v[u][j] = fmaxf(v[u][j], 0.f);
            }
            const unsigned off = (unsigned)(px * Co) * 4u + c_off;
            if (second) st4o(o2, wt, off, v[u]); else st4o(o1, wt, off, v[u]);
        }
    }
}

// The conv-LSTM cell behind its gate GEMM, channels [grp * cpi, ..) of sample b:  g = IN(raw gates + class bias);
// c_new = c_prev * sigmoid(f + 1) + sigmoid(i) * tanh(j);  c = IN(c_new);  h = tanh(c) * sigmoid(o).
// Three passes over the item's own elements; c_new and sigmoid(o) rest in the c / h buffers between passes 2 and 3
// (a thread re-reads only what it wrote itself).
template <class PT>
__device__ __forceinline__ void incell_item(const PT &p, const int b_, const int grp, const bool wt, float *smem) {
    const int tid = threadIdx.x, b = __builtin_amdgcn_readfirstlane(b_);
    const int nq = p.cpi >> 2, nq_log2 = 31 - __builtin_clz((unsigned)nq);
    const int q = tid & (nq - 1), ps = tid >> nq_log2, ppp = kConvThreads >> nq_log2;
    const int HW = p.H * p.W, C = p.C, C4 = 4 * C, c0 = grp * p.cpi + 4 * q;
    const float *in = p.in + (long long)b * p.in_bs + c0;
    const float *cond = p.cond ? p.cond + (long long)b * p.cond_bs + c0 : nullptr;
    const float *cprev = p.cprev ? p.cprev + (long long)b * p.cprev_bs + c0 : nullptr;
    float *hout = p.out + (long long)b * p.out_bs + c0;
    float *cout = p.cout + (long long)b * p.cout_bs + c0;
    const OutBuf o_h = out_buf(p.out + (long long)b * p.out_bs), o_c = out_buf(p.cout + (long long)b * p.cout_bs);
    const TileDiv div_w(p.W);
    double *red = reinterpret_cast<double *>(smem);
    auto gates = [&](const int px, f32x4 (&g)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = ld4(in + (long long)px * C4 + k * C);
        if (cond) {
            const int y = div_w.div(px), x = px - y * p.W;
            const float *cb = cond + (cls5(y, p.H) * 5 + cls5(x, p.W)) * C4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 a = ld4(cb + k * C);
#pragma unroll
                for (int j = 0; j < 4; ++j) g[k][j] += a[j];
            }
        }
    };
    // ---- pass 1: statistics of the four gate maps of every channel
    double st[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) st[i] = 0.0;
    for (int px0 = ps; px0 < HW; px0 += 2 * ppp) {      // (two pixels = eight + eight 16-byte loads in flight)
        f32x4 g[2][4];
        const bool two = px0 + ppp < HW;
        gates(px0, g[0]);
        gates(two ? px0 + ppp : px0, g[1]);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) { const double d = (double)g[u][k][j]; st[k * 4 + j] += d; st[16 + k * 4 + j] += d * d; }
        }
    }
    ew_reduce<32>(st, nq, red);
    const double inv_n = 1.0 / (double)HW;
    float sc[4][4], sh[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            in_scale_shift(st[k * 4 + j], st[16 + k * 4 + j], inv_n, p.eps, p.g0[k * C + c0 + j], p.b0[k * C + c0 + j],
                           sc[k][j], sh[k][j]);
    // ---- pass 2: the cell update; statistics of the new cell state
    double ct[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ct[i] = 0.0;
    for (int px0 = ps; px0 < HW; px0 += 2 * ppp) {
        f32x4 g[2][4], cp[2];
        const bool two = px0 + ppp < HW;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int px = (u == 1 && two) ? px0 + ppp : px0;
            gates(px, g[u]);
            cp[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (cprev) cp[u] = ld4(cprev + (long long)px * C);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
            const int px = px0 + u * ppp;
            f32x4 cn, so;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gi = fmaf(g[u][0][j], sc[0][j], sh[0][j]), gj = fmaf(g[u][1][j], sc[1][j], sh[1][j]);
                const float gf = fmaf(g[u][2][j], sc[2][j], sh[2][j]), go = fmaf(g[u][3][j], sc[3][j], sh[3][j]);
                cn[j] = fmaf(cp[u][j], sigmoidf_(gf + 1.0f), sigmoidf_(gi) * tanhf_(gj));
                so[j] = sigmoidf_(go);
                const double d = (double)cn[j];
                ct[j] += d; ct[4 + j] += d * d;
            }
            st4(cout + (long long)px * C, cn);
            st4(hout + (long long)px * C, so);
        }
    }
    ew_reduce<8>(ct, nq, red);
    float csc[4], csh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) in_scale_shift(ct[j], ct[4 + j], inv_n, p.eps, p.g1[c0 + j], p.b1[c0 + j], csc[j], csh[j]);
    // ---- pass 3: normalised cell state, hidden state
    for (int px0 = ps; px0 < HW; px0 += kEwBatch * ppp) {
        f32x4 cn[kEwBatch], so[kEwBatch];
#pragma unroll
        for (int u = 0; u < kEwBatch; ++u) {
            const int px = px0 + u * ppp < HW ? px0 + u * ppp : px0;
            cn[u] = ld4(cout + (long long)px * C);
            so[u] = ld4(hout + (long long)px * C);
        }
#pragma unroll
        for (int u = 0; u < kEwBatch; ++u) {
            const int px = px0 + u * ppp;
            if (px >= HW) break;
            f32x4 hn;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                cn[u][j] = fmaf(cn[u][j], csc[j], csh[j]);
                hn[j] = tanhf_(cn[u][j]) * so[u][j];
            }
            // (the final values: sc1 stores in a write-through item; pass 2 parked its intermediates with plain ones)
            st4o(o_c, wt, (unsigned)(px * C + c0) * 4u, cn[u]);
            st4o(o_h, wt, (unsigned)(px * C + c0) * 4u, hn);
        }
    }
}

// ------------------------------------------------------------------------------------------ bilinear 2x up-sampling of concat[in0, in1]
// out(2i + a, 2j + b) = sum over the two source rows / columns the transposed convolution with the kernel
// [.25, .75, .75, .25] (stride 2, SAME) reaches: a = 0: .25 s[i-1] + .75 s[i];  a = 1: .75 s[i] + .25 s[i+1];  zero outside.
template <class PT>
__device__ __forceinline__ void upsample_item(const PT &p, const int b_, const int band, const bool wt) {
    const int b = __builtin_amdgcn_readfirstlane(b_);
    const int C = p.C0 + p.C1, Cq = C >> 2, OW = 2 * p.w, OH = 2 * p.h;
    const int y_begin = band * p.rows, n_rows = min(p.rows, OH - y_begin);
    const int total = n_rows * OW * Cq;
    const TileDiv div_cq(Cq), div_ow(OW);
    const OutBuf ob = out_buf(p.out + (long long)b * p.out_bs);
    // two output elements per thread and pass: their eight source loads are issued back to back (clamped addresses, zeros
    // by select: no branch between the loads)
    for (int e0 = threadIdx.x; e0 < total; e0 += 2 * kConvThreads) {
        f32x4 v[2][4];
        float wgt[2][4];
        unsigned dst[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = min(e0 + u * kConvThreads, total - 1);
            const int pq = div_cq.div(e), cq = e - pq * Cq;
            const int yl = div_ow.div(pq), X = pq - yl * OW, Y = y_begin + yl;
            const int i = Y >> 1, a = Y & 1, j = X >> 1, bb = X & 1;
            const int r0 = i - 1 + a, r1 = i + a, q0 = j - 1 + bb, q1 = j + bb;
            const float wy0 = a ? 0.75f : 0.25f, wy1 = a ? 0.25f : 0.75f, wx0 = bb ? 0.75f : 0.25f, wx1 = bb ? 0.25f : 0.75f;
            const int c = 4 * cq;
            const float *src; int Cs;
            if (c < p.C0) { src = p.in0 + (long long)b * p.in0_bs + c; Cs = p.C0; }
            else { src = p.in1 + (long long)b * p.in1_bs + (c - p.C0); Cs = p.C1; }
            const bool r0ok = r0 >= 0, r1ok = r1 < p.h, q0ok = q0 >= 0, q1ok = q1 < p.w;
            const int r0c = max(r0, 0), r1c = min(r1, p.h - 1), q0c = max(q0, 0), q1c = min(q1, p.w - 1);
            v[u][0] = ld4(src + (long long)(r0c * p.w + q0c) * Cs);
            v[u][1] = ld4(src + (long long)(r0c * p.w + q1c) * Cs);
            v[u][2] = ld4(src + (long long)(r1c * p.w + q0c) * Cs);
            v[u][3] = ld4(src + (long long)(r1c * p.w + q1c) * Cs);
            wgt[u][0] = (r0ok && q0ok) ? wy0 * wx0 : 0.f;
            wgt[u][1] = (r0ok && q1ok) ? wy0 * wx1 : 0.f;
            wgt[u][2] = (r1ok && q0ok) ? wy1 * wx0 : 0.f;
            wgt[u][3] = (r1ok && q1ok) ? wy1 * wx1 : 0.f;
            dst[u] = (unsigned)((Y * OW + X) * C + c) * 4u;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (e0 + u * kConvThreads >= total) break;
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                o[k] = fmaf(wgt[u][3], v[u][3][k], fmaf(wgt[u][2], v[u][2][k], fmaf(wgt[u][1], v[u][1][k], wgt[u][0] * v[u][0][k])));
            st4o(ob, wt, dst[u], o);
        }
    }
}

// ------------------------------------------------------------------------------------------ the compositing layers
__device__ __forceinline__ int reflect_sym(int v, const int n) {
    v = v < 0 ? -v - 1 : (v >= n ? 2 * n - 1 - v : v);
    return min(max(v, 0), n - 1);
}

// mass of the previous distributions (they are stored un-normalised with their block sums): s_dscale[d] = 1 / sum
template <class PT>
__device__ __forceinline__ void top_dscale(const PT &p, const int b, float *s_dscale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nblocks = sum_blocks(p.H, p.W);
    for (int d = wave; d < p.ND; d += 4) {
        float sc = 1.0f;
        if (p.prev_sums) {
            double su = 0.0;
            const double *pp = p.prev_sums + ((long long)b * p.ND + d) * nblocks * 2;
            for (int k = lane; k < nblocks; k += 64) su += pp[2 * k];
            su = wave_sum(su);
            sc = (float)(1.0 / su);
        }
        if (lane == 0) s_dscale[d] = sc;
    }
}

// one 16 x 16 tile of sample b: [warp_0..3(previous frame), previous, first, sigmoid(scratch)] -> trans, warp_k(previous
// distributions) -> transd.  The warps read the SYMMETRICALLY padded image (the published apply_cdna_kernels).
template <int ND, class PT>
__device__ __forceinline__ void transform_item(const PT &p, const int tile, const int b, float *smem) {
    constexpr int TS = kCompTile, HS = TS + 4, PS = comp_px_stride(ND);
    float *s_px = smem;                         // [HS * HS][PS]
    float *s_kern = s_px + HS * HS * PS;        // [25][4]
    float *s_dscale = s_kern + kTaps * 4;       // [ND]
    const int tid = threadIdx.x;
    const int tilesX = (p.W + TS - 1) / TS;
    const int ty0 = (tile / tilesX) * TS, tx0 = (tile % tilesX) * TS;
    top_dscale(p, b, s_dscale);
    if (tid < kTaps * kNumWarp3) s_kern[tid] = p.kern[(long long)b * kTaps * kNumWarp3 + tid];
    __syncthreads();
    const float *pf = p.prev_frame + (long long)b * p.prev_frame_bs;
    const float *pd = p.prev_distrib + (long long)b * p.prev_distrib_bs;
    for (int i = tid; i < HS * HS; i += kConvThreads) {
        const int ly = i / HS, lx = i - ly * HS;
        const int y = reflect_sym(ty0 + ly - 2, p.H), x = reflect_sym(tx0 + lx - 2, p.W);
        const long long o = (long long)y * p.W + x;
#pragma unroll
        for (int c = 0; c < 3; ++c) s_px[i * PS + c] = pf[o * 3 + c];
#pragma unroll
        for (int d = 0; d < ND; ++d) s_px[i * PS + 3 + d] = pd[o * ND + d] * s_dscale[d];
    }
    __syncthreads();
    const int ly = tid / TS, lx = tid - ly * TS;
    const int y = ty0 + ly, x = tx0 + lx;
    if (y >= p.H || x >= p.W) return;
    float wf[kNumWarp3][3], wd[kNumWarp3][ND];
#pragma unroll
    for (int k = 0; k < kNumWarp3; ++k) {
#pragma unroll
        for (int c = 0; c < 3; ++c) wf[k][c] = 0.f;
#pragma unroll
        for (int d = 0; d < ND; ++d) wd[k][d] = 0.f;
    }
    // (one kernel row at a time: fully unrolled, the 25 taps' LDS reads are hoisted and the item needs all 256 VGPRs)
#pragma unroll 1
    for (int dy = 0; dy < kDnaKern; ++dy)
#pragma unroll
        for (int dx = 0; dx < kDnaKern; ++dx) {
            const f32x4 kr = ld4(s_kern + (dy * kDnaKern + dx) * 4);
            const int sp = (ly + dy) * HS + (lx + dx);
            const f32x4 a = ld4(s_px + sp * PS);
            float di[ND];
            di[0] = a[3];
            if constexpr (ND > 1) {
                const f32x4 c2 = ld4(s_px + sp * PS + 4);
#pragma unroll
                for (int d = 1; d < ND; ++d) di[d] = c2[d - 1];
            }
#pragma unroll
            for (int k = 0; k < kNumWarp3; ++k) {
#pragma unroll
                for (int c = 0; c < 3; ++c) wf[k][c] = fmaf(kr[k], a[c], wf[k][c]);
#pragma unroll
                for (int d = 0; d < ND; ++d) wd[k][d] = fmaf(kr[k], di[d], wd[k][d]);
            }
        }
    const long long o = (long long)y * p.W + x;
    const long long ob = (long long)b * p.H * p.W + o;
    float t[kTransCh];
#pragma unroll
    for (int k = 0; k < kNumWarp3; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) t[k * 3 + c] = wf[k][c];
    const f32x4 ctr = ld4(s_px + ((ly + 2) * HS + lx + 2) * PS);
    const f32x4 scr = ld4(p.scr_raw + ob * kScrCh);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        t[12 + c] = ctr[c];
        t[15 + c] = p.first_frame[o * 3 + c];
        t[18 + c] = sigmoidf_(scr[c]);
    }
#pragma unroll
    for (int c = 21; c < kTransCh; ++c) t[c] = 0.f;
    float *to = p.trans + ob * kTransCh;
#pragma unroll
    for (int k = 0; k < kTransCh / 4; ++k) st4(to + 4 * k, f32x4{t[4 * k], t[4 * k + 1], t[4 * k + 2], t[4 * k + 3]});
    float *tdo = p.transd + ob * (kNumWarp3 * ND);
#pragma unroll
    for (int k = 0; k < kNumWarp3; ++k)
#pragma unroll
        for (int d = 0; d < ND; ++d) tdo[k * ND + d] = wd[k][d];
}

// one 16 x 16 tile of sample b: softmax of the mask logits, next frame, next (un-normalised) distributions, cost sums per
// 4 x 16-pixel block (the layout scores_kernel / export_distrib_kernel and the next step's top_dscale read)
template <int ND, class PT>
__device__ __forceinline__ void compose_item(const PT &p, const int tile, const int b, const int *goal, float *smem) {
    constexpr int TS = kCompTile;
    float *s_dscale = smem;
    const int tid = threadIdx.x;
    const int tilesX = (p.W + TS - 1) / TS;
    const int nblocks = sum_blocks(p.H, p.W);
    const int ty0 = (tile / tilesX) * TS, tx0 = (tile % tilesX) * TS;
    top_dscale(p, b, s_dscale);
    __syncthreads();
    const int ly = tid / TS, lx = tid - ly * TS;
    const int y = ty0 + ly, x = tx0 + lx;
    double cost[2 * ND];
#pragma unroll
    for (int i = 0; i < 2 * ND; ++i) cost[i] = 0.0;
    if (y < p.H && x < p.W) {
        const long long o = (long long)y * p.W + x;
        const long long ob = (long long)b * p.H * p.W + o;
        const f32x4 l0 = ld4(p.mlog + ob * kMaskCh), l1 = ld4(p.mlog + ob * kMaskCh + 4);
        float m[7] = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2]};
        float mx = m[0];
#pragma unroll
        for (int j = 1; j < 7; ++j) mx = fmaxf(mx, m[j]);
        float den = 0.f;
#pragma unroll
        for (int j = 0; j < 7; ++j) { m[j] = __expf(m[j] - mx); den += m[j]; }
        const float inv = 1.0f / den;
#pragma unroll
        for (int j = 0; j < 7; ++j) m[j] *= inv;
        float t[24];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const f32x4 v = ld4(p.trans + ob * kTransCh + 4 * k);
            t[4 * k] = v[0]; t[4 * k + 1] = v[1]; t[4 * k + 2] = v[2]; t[4 * k + 3] = v[3];
        }
        float *fo = p.out_frame + (long long)b * p.out_frame_bs + o * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float a = m[0] * t[c];
#pragma unroll
            for (int j = 1; j < 7; ++j) a = fmaf(m[j], t[3 * j + c], a);
            fo[c] = a;
        }
        const float *wd = p.transd + ob * (kNumWarp3 * ND);
        const float *pd = p.prev_distrib + (long long)b * p.prev_distrib_bs + o * ND;
        float *dout = p.out_distrib + (long long)b * p.out_distrib_bs + o * ND;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const float prev = pd[d] * s_dscale[d];
            float a = m[0] * wd[d];
#pragma unroll
            for (int k = 1; k < kNumWarp3; ++k) a = fmaf(m[k], wd[k * ND + d], a);
            a = fmaf(m[4], prev, a);
            a = fmaf(m[5], p.first_distrib[o * ND + d], a);
            a = fmaf(m[6], prev, a);            // the scratch layer's slot carries the previous distribution
            dout[d] = a;
            const float ry = (float)(y - goal[2 * d]), rx = (float)(x - goal[2 * d + 1]);
            const float dist = sqrtf(fmaf(ry, ry, rx * rx));
            cost[2 * d] = (double)a;
            cost[2 * d + 1] = (double)a * (double)dist;
        }
    }
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

// ------------------------------------------------------------------------------------------ the heads' second phase, fused
// One item per 8 x 16-pixel tile of one sample does what four phases did - scratch conv, the seven compositing layers, the
// mask conv over [h_masks | layers], softmax + composition - without any of their tensors touching memory:
//   1. hs = relu(IN(raw hs)) over the tile + 2 pixels (normalised while staging, statistics from EW_INORM's table), the
//      previous frame / distributions over the tile + 3 pixels (symmetric padding), the sample's CDNA kernels;
//   2. over the tile + 1 pixel (what the mask conv's 3 x 3 window reaches): scratch = sigmoid(conv3x3(hs)), the four warps,
//      previous, first -> the 21 layer channels in LDS (zeros outside the image: the mask conv's zero padding); the warped
//      distributions of the tile's own pixels;
//   3. hm = relu(IN(raw hm)) over the tile + 1 pixel into the LDS hs occupied;
//   4. mask logits = conv3x3([hm | layers]) on the VALU, two threads per pixel (taps 0-4 / 5-8), weights as scalar operands;
//   5. softmax, next frame, next distributions, cost sums per 4 x 16 block (the layout of compose_item).
// 7 x 9 x (32 + 21 + 32 x 3 / 7 ...) FMAs per pixel on the VALU is ~1 ms of the whole chip per C5 launch: the four phases it
// replaces were bound by their 60 000+ items per step and four trips through memory, not by arithmetic.
constexpr int kT3H = 8, kT3W = 16;
constexpr int kT3R1H = kT3H + 2, kT3R1W = kT3W + 2, kT3R1 = kT3R1H * kT3R1W;    // 10 x 18: layers / hm
constexpr int kT3R2H = kT3H + 4, kT3R2W = kT3W + 4, kT3R2 = kT3R2H * kT3R2W;    // 12 x 20: hs
constexpr int kT3RPH = kT3H + 6, kT3RPW = kT3W + 6, kT3RP = kT3RPH * kT3RPW;    // 14 x 22: previous frame
constexpr int kT3FeatPad = 36, kT3LayPad = 28, kT3MaskIn = 56;
__host__ __device__ constexpr int top3_lds_floats(const int nd) {
    return kT3R2 * kT3FeatPad + kT3R1 * kT3LayPad + kT3RP * comp_px_stride(nd) + kT3H * kT3W * kNumWarp3 * nd + kTaps * 4 + 128 + 8;
}

template <int ND, class PT>
__device__ __forceinline__ void top3_item(const PT &p, const int tile, const int b_, const int *goal, const bool wt, float *smem) {
    const int b = __builtin_amdgcn_readfirstlane(b_);
    constexpr int PS = comp_px_stride(ND);
    typedef const __attribute__((address_space(4))) float cfloat;
    float *s_a = smem;                                  // [kT3R2][36]: hs, later hm ([kT3R1][36])
    float *s_t = s_a + kT3R2 * kT3FeatPad;              // [kT3R1][28]: the layers
    float *s_p = s_t + kT3R1 * kT3LayPad;               // [kT3RP][PS]: previous frame + distributions; later the partial logits
    float *s_wd = s_p + kT3RP * PS;                     // [128][4][ND]: warped distributions of the tile's pixels
    float *s_kern = s_wd + kT3H * kT3W * kNumWarp3 * ND;
    float *s_tab = s_kern + kTaps * 4;                  // [64][2]
    float *s_ds = s_tab + 128;                          // [ND]
    const int tid = threadIdx.x;
    const int tilesX = (p.W + kT3W - 1) / kT3W;
    const int nblocks = sum_blocks(p.H, p.W);
    const int ty0 = (tile / tilesX) * kT3H, tx0 = (tile % tilesX) * kT3W;
    const long long HW = (long long)p.H * p.W;
    top_dscale(p, b, s_ds);
    if (tid < kTaps * kNumWarp3) s_kern[tid] = p.kern[(long long)b * kTaps * kNumWarp3 + tid];
    if (tid < 128) s_tab[tid] = p.norm_tab[(long long)b * 128 + tid];
    __syncthreads();
    // ---- 1. hs over R2 (normalised + relu, zeros outside the image), previous frame over RP (symmetric padding)
    const float *raw = p.hmhs_raw + (long long)b * HW * 64;
    for (int e = tid; e < kT3R2 * 8; e += kConvThreads) {
        const int px = e >> 3, q = e & 7;
        const int ry = px / kT3R2W, rx = px - ry * kT3R2W;
        const int y = ty0 - 2 + ry, x = tx0 - 2 + rx;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) {
            v = ld4(raw + ((long long)y * p.W + x) * 64 + 32 + 4 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaf(v[j], s_tab[2 * (32 + 4 * q + j)], s_tab[2 * (32 + 4 * q + j) + 1]), 0.f);
        }
        st4(s_a + px * kT3FeatPad + 4 * q, v);
    }
    {
        const float *pf = p.prev_frame + (long long)b * p.prev_frame_bs;
        const float *pd = p.prev_distrib + (long long)b * p.prev_distrib_bs;
        for (int i = tid; i < kT3RP; i += kConvThreads) {
            const int ly = i / kT3RPW, lx = i - ly * kT3RPW;
            const int y = reflect_sym(ty0 - 3 + ly, p.H), x = reflect_sym(tx0 - 3 + lx, p.W);
            const long long o = (long long)y * p.W + x;
#pragma unroll
            for (int c = 0; c < 3; ++c) s_p[i * PS + c] = pf[o * 3 + c];
#pragma unroll
            for (int d = 0; d < ND; ++d) s_p[i * PS + 3 + d] = pd[o * ND + d] * s_ds[d];
        }
    }
    __syncthreads();
    // ---- 2. the layers over R1: one thread per pixel
    if (tid < kT3R1) {
        const int ry = tid / kT3R1W, rx = tid - ry * kT3R1W;
        const int y = ty0 - 1 + ry, x = tx0 - 1 + rx;
        float t[24];
#pragma unroll
        for (int c = 0; c < 24; ++c) t[c] = 0.f;
        if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) {
            // scratch = sigmoid(conv3x3(hs) + b): R1 pixel (ry, rx) = R2 pixel (ry + 1, rx + 1); taps reach R2 (ry + dy, rx + dx)
            cfloat *ws = (cfloat *)(unsigned long long)p.w_scr, *bs = (cfloat *)(unsigned long long)p.b_scr;
            float a0 = bs[0], a1 = bs[1], a2 = bs[2];
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap - dy * 3;
                const float *ap = s_a + ((ry + dy) * kT3R2W + rx + dx) * kT3FeatPad;
                cfloat *wt = ws + tap * 32 * 4;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const f32x4 av = ld4(ap + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = 4 * q + e;
                        a0 = fmaf(av[e], wt[c * 4 + 0], a0); a1 = fmaf(av[e], wt[c * 4 + 1], a1); a2 = fmaf(av[e], wt[c * 4 + 2], a2);
                    }
                }
            }
            t[18] = sigmoidf_(a0); t[19] = sigmoidf_(a1); t[20] = sigmoidf_(a2);
            // warps of the previous frame (and, for the tile's own pixels, of the distributions): R1 (ry, rx) = RP (ry + 2, rx + 2)
            const bool inner = ry >= 1 && ry <= kT3H && rx >= 1 && rx <= kT3W;
            float wd[kNumWarp3][ND];
#pragma unroll
            for (int k = 0; k < kNumWarp3; ++k)
#pragma unroll
                for (int d = 0; d < ND; ++d) wd[k][d] = 0.f;
#pragma unroll 1
            for (int dy = 0; dy < kDnaKern; ++dy)
#pragma unroll
                for (int dx = 0; dx < kDnaKern; ++dx) {
                    const f32x4 kr = ld4(s_kern + (dy * kDnaKern + dx) * 4);
                    const int sp = (ry + dy) * kT3RPW + rx + dx;
                    const f32x4 a = ld4(s_p + sp * PS);
                    float di[ND];
                    di[0] = a[3];
                    if constexpr (ND > 1) {
                        const f32x4 c2 = ld4(s_p + sp * PS + 4);
#pragma unroll
                        for (int d = 1; d < ND; ++d) di[d] = c2[d - 1];
                    }
#pragma unroll
                    for (int k = 0; k < kNumWarp3; ++k) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) t[k * 3 + c] = fmaf(kr[k], a[c], t[k * 3 + c]);
#pragma unroll
                        for (int d = 0; d < ND; ++d) wd[k][d] = fmaf(kr[k], di[d], wd[k][d]);
                    }
                }
            const f32x4 ctr = ld4(s_p + ((ry + 2) * kT3RPW + rx + 2) * PS);
            const long long o = (long long)y * p.W + x;
#pragma unroll
            for (int c = 0; c < 3; ++c) { t[12 + c] = ctr[c]; t[15 + c] = p.first_frame[o * 3 + c]; }
            if (inner) {
                float *wo = s_wd + ((ry - 1) * kT3W + rx - 1) * kNumWarp3 * ND;
#pragma unroll
                for (int k = 0; k < kNumWarp3; ++k)
#pragma unroll
                    for (int d = 0; d < ND; ++d) wo[k * ND + d] = wd[k][d];
            }
        }
        float *to = s_t + tid * kT3LayPad;
#pragma unroll
        for (int k = 0; k < 6; ++k) st4(to + 4 * k, f32x4{t[4 * k], t[4 * k + 1], t[4 * k + 2], t[4 * k + 3]});
    }
    __syncthreads();
    // ---- 3. hm over R1 into the LDS hs occupied
    for (int e = tid; e < kT3R1 * 8; e += kConvThreads) {
        const int px = e >> 3, q = e & 7;
        const int ry = px / kT3R1W, rx = px - ry * kT3R1W;
        const int y = ty0 - 1 + ry, x = tx0 - 1 + rx;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) {
            v = ld4(raw + ((long long)y * p.W + x) * 64 + 4 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaf(v[j], s_tab[2 * (4 * q + j)], s_tab[2 * (4 * q + j) + 1]), 0.f);
        }
        st4(s_a + px * kT3FeatPad + 4 * q, v);
    }
    __syncthreads();
    // ---- 4. mask logits: pixel = tid % 128, taps 0 - 4 (waves 0, 1) / 5 - 8 (waves 2, 3)
    const int px = tid & 127, half = tid >> 7;
    const int iy = px >> 4, ix = px & 15;
    float m[7];
    {
        cfloat *wm = (cfloat *)(unsigned long long)p.w_msk, *bm = (cfloat *)(unsigned long long)p.b_msk;
#pragma unroll
        for (int j = 0; j < 7; ++j) m[j] = half == 0 ? bm[j] : 0.f;
        const int t0 = half == 0 ? 0 : 5, t1 = half == 0 ? 5 : 9;
        for (int tap = t0; tap < t1; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
            const int r1 = (iy + dy) * kT3R1W + ix + dx;
            const float *ap = s_a + r1 * kT3FeatPad, *lp = s_t + r1 * kT3LayPad;
            cfloat *wt = wm + tap * kT3MaskIn * 8;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 av = ld4(ap + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < 7; ++j) m[j] = fmaf(av[e], wt[(4 * q + e) * 8 + j], m[j]);
            }
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const f32x4 lv = ld4(lp + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (4 * q + e >= 21) continue;
#pragma unroll
                    for (int j = 0; j < 7; ++j) m[j] = fmaf(lv[e], wt[(32 + 4 * q + e) * 8 + j], m[j]);
                }
            }
        }
    }
    float *s_part = s_p;                // (the previous-frame halo is no longer read)
    if (half == 1) {
        st4(s_part + px * 8, f32x4{m[0], m[1], m[2], m[3]});
        st4(s_part + px * 8 + 4, f32x4{m[4], m[5], m[6], 0.f});
    }
    __syncthreads();
    // ---- 5. softmax, composition, cost sums: thread t < 128 = pixel (t / 16, t % 16); wave = one 4 x 16 block
    double cost[2 * ND];
#pragma unroll
    for (int i = 0; i < 2 * ND; ++i) cost[i] = 0.0;
    const int y = ty0 + iy, x = tx0 + ix;
    float of[3] = {0.f, 0.f, 0.f}, od[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) od[d] = 0.f;
    const bool live = half == 0 && y < p.H && x < p.W;
    const long long o = (long long)y * p.W + x;
    if (live) {
        const f32x4 q0 = ld4(s_part + px * 8), q1 = ld4(s_part + px * 8 + 4);
        m[0] += q0[0]; m[1] += q0[1]; m[2] += q0[2]; m[3] += q0[3]; m[4] += q1[0]; m[5] += q1[1]; m[6] += q1[2];
        float mx = m[0];
#pragma unroll
        for (int j = 1; j < 7; ++j) mx = fmaxf(mx, m[j]);
        float den = 0.f;
#pragma unroll
        for (int j = 0; j < 7; ++j) { m[j] = __expf(m[j] - mx); den += m[j]; }
        const float inv = 1.0f / den;
#pragma unroll
        for (int j = 0; j < 7; ++j) m[j] *= inv;
        const float *lp = s_t + ((iy + 1) * kT3R1W + ix + 1) * kT3LayPad;
        float t[24];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const f32x4 v = ld4(lp + 4 * k);
            t[4 * k] = v[0]; t[4 * k + 1] = v[1]; t[4 * k + 2] = v[2]; t[4 * k + 3] = v[3];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float a = m[0] * t[c];
#pragma unroll
            for (int j = 1; j < 7; ++j) a = fmaf(m[j], t[3 * j + c], a);
            of[c] = a;
        }
        const float *wd = s_wd + px * kNumWarp3 * ND;
        const float *pd = p.prev_distrib + (long long)b * p.prev_distrib_bs + o * ND;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const float prev = pd[d] * s_ds[d];
            float a = m[0] * wd[d];
#pragma unroll
            for (int k = 1; k < kNumWarp3; ++k) a = fmaf(m[k], wd[k * ND + d], a);
            a = fmaf(m[4], prev, a);
            a = fmaf(m[5], p.first_distrib[o * ND + d], a);
            a = fmaf(m[6], prev, a);
            od[d] = a;
            const float ry = (float)(y - goal[2 * d]), rx = (float)(x - goal[2 * d + 1]);
            const float dist = sqrtf(fmaf(ry, ry, rx * rx));
            cost[2 * d] = (double)a;
            cost[2 * d + 1] = (double)a * (double)dist;
        }
    }
    if (!wt) {
        if (live) {
            float *fo = p.out_frame + (long long)b * p.out_frame_bs + o * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) fo[c] = of[c];
            float *dout = p.out_distrib + (long long)b * p.out_distrib_bs + o * ND;
#pragma unroll
            for (int d = 0; d < ND; ++d) dout[d] = od[d];
        }
    } else if (half == 0) {
        // write-through item (whole 4 x 16 blocks: the host checks H % 4 == 0, W % 16 == 0): the wave turns its block over in
        // the dead feature tile - a block row is 48 consecutive floats of the frame and 16 ND of the distributions - and
        // every lane stores 16-byte sc1 pieces (vf_fused_top.h has the same turn-over)
        const int lane = tid & 63, wave = tid >> 6;
        float *slab = s_a + wave * (64 * (3 + ND));
        const int y_blk = ty0 + wave * kSumBlockH;
        const bool blk_ok = y_blk < p.H && tx0 < p.W;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 3; ++c) slab[lane * 3 + c] = of[c];
#pragma unroll
        for (int d = 0; d < ND; ++d) slab[192 + lane * ND + d] = od[d];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const __amdgpu_buffer_rsrc_t r_fr = __builtin_amdgcn_make_buffer_rsrc(
            p.out_frame + (long long)b * p.out_frame_bs, 0, blk_ok ? (int)((unsigned)(p.H * p.W * 3) * 4u) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_di = __builtin_amdgcn_make_buffer_rsrc(
            p.out_distrib + (long long)b * p.out_distrib_bs, 0, blk_ok ? (int)((unsigned)(p.H * p.W * ND) * 4u) : 0, 0x00020000);
        if (lane < 48) {
            const int row = lane / 12, q = lane - row * 12;
            const f32x4 v = ld4(slab + row * 48 + 4 * q);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), r_fr,
                                                   (unsigned)(((y_blk + row) * p.W + tx0) * 3 + 4 * q) * 4u, 0, 16);
        }
        for (int j = lane; j < 16 * ND; j += 64) {
            const int row = j / (4 * ND), q = j - row * (4 * ND);
            const f32x4 v = ld4(slab + 192 + row * 16 * ND + 4 * q);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), r_di,
                                                   (unsigned)(((y_blk + row) * p.W + tx0) * ND + 4 * q) * 4u, 0, 16);
        }
    }
    if (half == 0) {
#pragma unroll
        for (int i = 0; i < 2 * ND; ++i) cost[i] = wave_sum(cost[i]);
        const int lane = tid & 63, wave = tid >> 6;
        const int by = ty0 / kSumBlockH + wave, bxk = tx0 / kSumBlockW;
        if (lane == 0 && by * kSumBlockH < p.H && tx0 < p.W) {
            const int blk = by * sum_blocks_x(p.W) + bxk;
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                double *dst = p.out_sums + (((long long)b * ND + d) * nblocks + blk) * 2;
                if (wt) {
                    __hip_atomic_store(dst, cost[2 * d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(dst + 1, cost[2 * d + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    dst[0] = cost[2 * d]; dst[1] = cost[2 * d + 1];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ dispatch of one item
// `idx` = item index within its sample (spi == 0) / ignored (spi > 0: the item covers samples [b0, b1)); goal = this view's
// goal pixels (EW_COMPOSE)
template <int ND, class PT>
__device__ __forceinline__ void ew_item(const PT &p, const int idx, const int b0, const int b1, const int *goal, float *smem) {
    switch (p.op) {
        case EW_SA3: {
            const int wave = threadIdx.x >> 6, b = b0 + wave;
            if (b < b1) sa3_sample(p.sa, b, threadIdx.x & 63, smem + 128 * wave);
            break;
        }
        case EW_COND3: cond3_item(p.cond, b0, b1, smem); break;
        case EW_INORM: inorm_item(p.norm, b0, idx, p.wt != 0, smem); break;
        case EW_INCELL: incell_item(p.norm, b0, idx, p.wt != 0, smem); break;
        case EW_UPSAMPLE: upsample_item(p.up, b0, idx, p.wt != 0); break;
        case EW_TRANSFORM: transform_item<ND>(p.top, idx, b0, smem); break;
        case EW_TOP3: top3_item<ND>(p.top, idx, b0, goal, p.wt != 0, smem); break;
        default: compose_item<ND>(p.top, idx, b0, goal, smem); break;
    }
}

// per-layer launch: one workgroup per item
template <int ND>
VF_GLOBAL VF_LAUNCH_BOUNDS(kConvThreads) void ew_kernel(const EwParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];        // ew_lds_bytes(op, ND) of dynamic LDS
    const int item = blockIdx.x;
    int b0, b1, idx = 0;
    if (p.spi > 0) { b0 = item * p.spi; b1 = min(b0 + p.spi, p.B); }
    else { b0 = item / p.gx; b1 = b0 + 1; idx = item - b0 * p.gx; }
    ew_item<ND>(p, idx, b0, b1, &p.top.goal[0][0], smem);
}

}  // namespace vf
