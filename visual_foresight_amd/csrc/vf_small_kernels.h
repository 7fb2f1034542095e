// vf_small_kernels.h - the non-GEMM kernels of one predictor step (all fp32 VALU, HBM/LDS bound):
//   set_context      uint8 context frames -> float/255, context copies
//   sa_kernel        state FC + the action/state part of enc3 as a per-sample bias
//   cdna_finalize    sum K-split partials of the CDNA FC, relu-shift, L1-normalise each 5x5 kernel
//   composite        LN9+relu -> rgb/mask heads -> softmax -> CDNA warp -> compositing of the next
//                    frame and designated-pixel distributions + expected-distance partial sums
//   scores / export  reduce per-step sums to costs; hand predictions out in the reference layout
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vf_conv_mfma.h"

namespace vf {

constexpr int kMaxDesig = 4;
constexpr int kDnaKern = 5;
constexpr int kTaps = kDnaKern * kDnaKern;
constexpr int kCompTile = 16;           // composite: 16x16 pixels per workgroup
constexpr float kReluShift = 1e-12f;

// ------------------------------------------------------------------------------------------
__global__ void set_context_kernel(const uint8_t *frames_u8, float *frames_f, int n_frames,
                                   const float *src_a, float *dst_a, int n_a,
                                   const float *src_b, float *dst_b, int n_b,
                                   const float *src_c, float *dst_c, int n_c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_frames) frames_f[i] = (float)frames_u8[i] / 255.0f;
    if (i < n_a) dst_a[i] = src_a[i];
    if (i < n_b) dst_b[i] = src_b[i];
    if (i < n_c) dst_c[i] = src_c[i];
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

__global__ void sa_kernel(const SaParams p) {
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

__global__ void cdna_finalize_kernel(const FinParams p) {
    __shared__ float scratch[kTaps * 16 + 16];
    cdna_finalize_sample(p, blockIdx.x, scratch);
}

// ------------------------------------------------------------------------------------------
struct CompositeParams {
    int B, H, W, ND, K;                 // K = num_masks (K+1 mask channels, K-1 kernels used)
    const float *enc6;                  // raw convT3 output [B][H][W][32]
    const double *ln_part; int ln_nparts; float ln_inv_n;
    const float *gamma, *beta;          // LN9 [32]
    const float *w_rgb, *b_rgb;         // [32][3], [3]
    const float *w_mask, *b_mask;       // [32][K+1], [K+1]
    const float *kern;                  // [B][25][K]
    const float *prev_frame; long long prev_frame_bstride;      // [H][W][3]
    const float *prev_distrib; long long prev_distrib_bstride;  // [H][W][ND]
    const double *prev_sums;            // [B][ND][ntiles][2] partial sums of prev_distrib, or null
    float *out_frame; long long out_frame_bstride;
    float *out_distrib; long long out_distrib_bstride;
    double *out_sums;                   // [B][ND][ntiles][2]: sum d, sum d * dist(goal)
    int goal[kMaxDesig][2];             // (row, col)
};

// LDS floats needed by composite_tile<ND, K>
template <int ND, int K>
__host__ __device__ constexpr int composite_lds_floats() {
    return (kCompTile + 4) * (kCompTile + 4) * (3 + ND) + kTaps * K + 2 + ND + 2 /*pad*/ + 2 * 4 * 2 * ND;
}

// one 16x16 pixel tile of one sample
template <int ND, int K, class PT>
__device__ __forceinline__ void composite_tile(const PT &p, const int tile, const int b, float *smem) {
    constexpr int TS = kCompTile, HS = TS + 4;
    constexpr int NM = K + 1;
    float *s_frame = smem;                              // [HS*HS*3]
    float *s_dist = s_frame + HS * HS * 3;              // [HS*HS*ND]
    float *s_kern = s_dist + HS * HS * ND;              // [kTaps*K]
    float *s_ln = s_kern + kTaps * K;                   // [2]
    float *s_dscale = s_ln + 2;                         // [ND]
    // doubles: keep 8-byte alignment (all counts above are even except possibly ND)
    double (*s_red)[2 * ND] = reinterpret_cast<double (*)[2 * ND]>(
        smem + ((HS * HS * (3 + ND) + kTaps * K + 2 + ND + 1) & ~1));

    const int tid = threadIdx.x;
    const int tilesX = (p.W + TS - 1) / TS;
    const int ntiles = tilesX * ((p.H + TS - 1) / TS);
    const int ty0 = (tile / tilesX) * TS, tx0 = (tile % tilesX) * TS;

    if (tid == 0) {
        double su = 0.0, sq = 0.0;
        const double *pp = p.ln_part + (long long)b * p.ln_nparts * 2;
        for (int k = 0; k < p.ln_nparts; ++k) { su += pp[2 * k]; sq += pp[2 * k + 1]; }
        const double m = su * (double)p.ln_inv_n;
        double var = sq * (double)p.ln_inv_n - m * m;
        var = var < 0.0 ? 0.0 : var;
        s_ln[0] = (float)m;
        s_ln[1] = (float)(1.0 / sqrt(var + (double)kLnEps));
    }
    if (tid >= 64 && tid < 64 + ND) {
        const int d = tid - 64;
        float sc = 1.0f;
        if (p.prev_sums) {
            double su = 0.0;
            const double *pp = p.prev_sums + ((long long)b * ND + d) * ntiles * 2;
            for (int k = 0; k < ntiles; ++k) su += pp[2 * k];
            sc = (float)(1.0 / su);
        }
        s_dscale[d] = sc;
    }
    for (int i = tid; i < kTaps * K; i += 256) s_kern[i] = p.kern[(long long)b * kTaps * K + i];
    __syncthreads();

    const float *pf = p.prev_frame + (long long)b * p.prev_frame_bstride;
    const float *pd = p.prev_distrib + (long long)b * p.prev_distrib_bstride;
    for (int i = tid; i < HS * HS; i += 256) {
        const int ly = i / HS, lx = i % HS;
        const int y = ty0 + ly - 2, x = tx0 + lx - 2;
        const bool in = y >= 0 && y < p.H && x >= 0 && x < p.W;
        const long long o = (long long)y * p.W + x;
#pragma unroll
        for (int c = 0; c < 3; ++c) s_frame[i * 3 + c] = in ? pf[o * 3 + c] : 0.f;
#pragma unroll
        for (int d = 0; d < ND; ++d) s_dist[i * ND + d] = in ? pd[o * ND + d] * s_dscale[d] : 0.f;
    }
    __syncthreads();

    const int ly = tid / TS, lx = tid % TS;
    const int y = ty0 + ly, x = tx0 + lx;
    const bool valid = y < p.H && x < p.W;
    double cost[2 * ND];
#pragma unroll
    for (int i = 0; i < 2 * ND; ++i) cost[i] = 0.0;

    if (valid) {
        // ---- LN9 + relu of this pixel's 32 features, then the two 1x1 heads
        const f32x4 *src = reinterpret_cast<const f32x4 *>(p.enc6 + (((long long)b * p.H + y) * p.W + x) * 32);
        float o_rgb[3], o_m[NM];
#pragma unroll
        for (int j = 0; j < 3; ++j) o_rgb[j] = p.b_rgb[j];
#pragma unroll
        for (int j = 0; j < NM; ++j) o_m[j] = p.b_mask[j];
        const float mean = s_ln[0], rstd = s_ln[1];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 raw = src[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = q * 4 + e;
                const float f = fmaxf(fmaf((raw[e] - mean) * rstd, p.gamma[c], p.beta[c]), 0.f);
#pragma unroll
                for (int j = 0; j < 3; ++j) o_rgb[j] = fmaf(f, p.w_rgb[c * 3 + j], o_rgb[j]);
#pragma unroll
                for (int j = 0; j < NM; ++j) o_m[j] = fmaf(f, p.w_mask[c * NM + j], o_m[j]);
            }
        }
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
        float of[3], od[ND];
        const int ctr = (ly + 2) * HS + (lx + 2);
#pragma unroll
        for (int c = 0; c < 3; ++c)
            of[c] = fmaf(o_m[0], s_frame[ctr * 3 + c], o_m[1] * sigmoidf_(o_rgb[c]));
#pragma unroll
        for (int d = 0; d < ND; ++d) od[d] = o_m[0] * s_dist[ctr * ND + d];
#pragma unroll
        for (int dy = 0; dy < kDnaKern; ++dy) {
#pragma unroll
            for (int dx = 0; dx < kDnaKern; ++dx) {
                const int tap = dy * kDnaKern + dx;
                float ke = 0.f;
#pragma unroll
                for (int k = 0; k < K - 1; ++k) ke = fmaf(o_m[k + 2], s_kern[tap * K + k], ke);
                const int sp = (ly + dy) * HS + (lx + dx);
#pragma unroll
                for (int c = 0; c < 3; ++c) of[c] = fmaf(ke, s_frame[sp * 3 + c], of[c]);
#pragma unroll
                for (int d = 0; d < ND; ++d) od[d] = fmaf(ke, s_dist[sp * ND + d], od[d]);
            }
        }
        const long long o = (long long)y * p.W + x;
        float *fo = p.out_frame + (long long)b * p.out_frame_bstride + o * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) fo[c] = of[c];
        float *dout = p.out_distrib + (long long)b * p.out_distrib_bstride + o * ND;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            dout[d] = od[d];
            const float ry = (float)(y - p.goal[d][0]), rx = (float)(x - p.goal[d][1]);
            const float dist = sqrtf(fmaf(ry, ry, rx * rx));
            cost[2 * d] = (double)od[d];
            cost[2 * d + 1] = (double)od[d] * (double)dist;
        }
    }
    // ---- deterministic workgroup reduction of the cost sums
#pragma unroll
    for (int i = 0; i < 2 * ND; ++i) cost[i] = wave_sum(cost[i]);
    const int lane = tid & 63, wave = tid >> 6;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 2 * ND; ++i) s_red[wave][i] = cost[i];
    }
    __syncthreads();
    if (tid < 2 * ND) {
        const double s = s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid];
        const int d = tid >> 1;
        p.out_sums[(((long long)b * ND + d) * ntiles + tile) * 2 + (tid & 1)] = s;
    }
}

template <int ND, int K>
__global__ __launch_bounds__(256) void composite_kernel(const CompositeParams p) {
    __shared__ __attribute__((aligned(16))) float smem[composite_lds_floats<ND, K>()];
    composite_tile<ND, K>(p, blockIdx.x, blockIdx.y, smem);
}

// ------------------------------------------------------------------------------------------
// score_b = mean_p sum_t w_t (S1/S0) / sum_t w_t; sums[t]: [B][ND][ntiles][2]
__global__ void scores_kernel(const double *sums, long long step_stride, int B, int T, int ND, int ntiles,
                              float finalweight, float *scores, float *scores_per_task) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double total = 0.0;
    for (int d = 0; d < ND; ++d) {
        double acc = 0.0, wsum = 0.0;
        for (int t = 0; t < T; ++t) {
            const double *pp = sums + (long long)t * step_stride + ((long long)b * ND + d) * ntiles * 2;
            double s0 = 0.0, s1 = 0.0;
            for (int k = 0; k < ntiles; ++k) { s0 += pp[2 * k]; s1 += pp[2 * k + 1]; }
            const double w = (t == T - 1) ? (double)finalweight : 1.0;
            acc += w * (s1 / s0);
            wsum += w;
        }
        const double sc = acc / wsum;
        if (scores_per_task) scores_per_task[(long long)b * ND + d] = (float)sc;
        total += sc;
    }
    scores[b] = (float)(total / ND);
}

// normalised distributions out: dst[b][t][h][w][d] = src / S0(b,t,d)
__global__ void export_distrib_kernel(const float *src, const double *sums, long long step_stride,
                                      int first, int count, int T, int HW, int ND, int ntiles,
                                      float *dst) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per_sample = (long long)T * HW * ND;
    if (i >= (long long)count * per_sample) return;
    const int bb = (int)(i / per_sample);
    const long long r = i - (long long)bb * per_sample;
    const int t = (int)(r / ((long long)HW * ND));
    const int d = (int)(r % ND);
    const int b = first + bb;
    const double *pp = sums + (long long)t * step_stride + ((long long)b * ND + d) * ntiles * 2;
    double s0 = 0.0;
    for (int k = 0; k < ntiles; ++k) s0 += pp[2 * k];
    dst[i] = (float)((double)src[(long long)b * per_sample + r] / s0);
}

}  // namespace vf
