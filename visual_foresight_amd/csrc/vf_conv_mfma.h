// vf_conv_mfma.h - fp32 MFMA implicit-GEMM convolution for gfx950 (MI355X), one kernel
// template for every dense layer of the CDNA predictor:
//
//   conv-LSTM gate conv 5x5 (+ fused cell update)         G = 4 (i, j, f, o)    EPI_LSTM
//   strided encoder convs 5x5/2, 3x3/2, 1x1               G = 1                 EPI_BIAS_RELU / EPI_RAW_STATS
//   transposed convs 3x3*2 as a 2x2 conv over the input   G = 4 (output parity) EPI_CONVT_*
//   CDNA kernel FC [B,8192]x[8192,250], split over K      G = 1                 EPI_PARTIAL
//
// GEMM view: rows = output pixels (256 per workgroup: 4 waves x 2 MFMA row blocks of 32),
// columns = one group of 32 output channels x G gates, K = taps x input channels.
// A operand: the haloed input tile of one channel chunk staged in LDS ([pixel][KC+4] floats:
// the +4 pad makes the 16-lane groups of ds_read_b128 hit 16 distinct 16-B slots); LayerNorm
// (+relu) of the producing layer is applied while staging, so normalised tensors are never
// materialised.  B operand: weights pre-packed [chunk][tap][k8][khalf][N][4] and read straight
// from L2/L1 with one 16-B load per lane per gate (all four waves read the same lines).
// One ds_read_b128 / global b128 feeds four v_mfma_f32_32x32x2_f32 (K order inside an 8-channel
// step is permuted identically on both operands: lane half h supplies channels 4h..4h+3).
//
// Numerics: v_mfma_f32_32x32x2_f32 is an exact-fp32 fma chain (no reduced precision).  The library
// is built with -ffp-contract=off and every fused multiply-add is written as fmaf(), so the same
// tile body gives the same bits whether it is compiled into a per-layer kernel or into the
// persistent rollout kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

// -DVF_HOST_SELFTEST (tools/sanitize/): the file is compiled for the host only and no kernel may be
// emitted (a translation unit with kernels references the device binary); kernels then degrade to
// plain device functions, which the host pass parses but never generates.
#ifdef VF_HOST_SELFTEST
#define VF_GLOBAL __device__
#define VF_LAUNCH_BOUNDS(...)
#else
#define VF_GLOBAL __global__
#define VF_LAUNCH_BOUNDS(...) __launch_bounds__(__VA_ARGS__)
#endif

namespace vf {

constexpr int kPollSleep = 16;      // s_sleep units (64 cycles) between two polls of a dependency counter (2, 4: no gain)

#ifdef VF_TILE_STATS
// Diagnostic build only (hipcc -DVF_TILE_STATS, read by tools/tile_stats.py): where a conv-LSTM item spends
// its time.  [layer key][0] entry->first MFMA, [1] K loop, [2] epilogue, [3] items, [4] staging of later chunks;
// row 15: [5] ticket fetch, [6] publish, [7] items of the persistent scheduler.
__device__ unsigned long long g_tile_clk[32][8];
#define VF_TS_NOW() (threadIdx.x == 0 ? wall_clock64() : 0ull)
#define VF_TS_ADD(KEY_, SLOT_, DT_) do { if (threadIdx.x == 0) atomicAdd(&g_tile_clk[KEY_][SLOT_], (unsigned long long)(DT_)); } while (0)
#else
#define VF_TS_NOW() 0ull
#define VF_TS_ADD(KEY_, SLOT_, DT_) do { } while (0)
#endif

#ifdef VF_TRACE
// Diagnostic build only (hipcc -DVF_TRACE, read by tools/trace_cu.py): a per-workgroup event log of the persistent
// rollout - what every resident workgroup is doing when (ticket, dependency wait, prologue, staging, K loop, mid-item
// wait, epilogue, publish) - from which the co-residency statistics of a CU (both workgroups in their K loops / one /
// none, and the MFMA rate in each case) are reconstructed.  One 64-bit word per event: (wall_clock64 << 8) | code, or
// (value << 8) | code for the value-carrying codes.  The event count lives in word 5 of the LDS control block.
#ifndef VF_TRACE_MAX
#define VF_TRACE_MAX 8192
#endif
constexpr int kTraceMax = VF_TRACE_MAX;       // events per workgroup (-DVF_TRACE_MAX=65536 for the long launches of arch 3)
constexpr int kTraceWgs = 512;
__device__ unsigned long long g_trace[(size_t)kTraceWgs * kTraceMax];
__device__ unsigned g_trace_n[kTraceWgs];
enum TraceCode { TR_TICKET = 1, TR_DONE = 3, TR_STAGE = 10, TR_KLOOP = 11, TR_LATE = 12, TR_LATE_END = 13, TR_EPI = 14,
                 TR_MFMAS = 15 /* value: MFMAs per wave per chunk */, TR_ST_LOADED = 16, TR_ST_WRITTEN = 17, TR_YIELD = 18, TR_YIELD_END = 19, TR_TOP_STATS = 24, TR_TOP_HALO = 25, TR_TOP_MATES = 26, TR_HWID = 20 /* value: xcc << 16 | hw_id */, TR_PHASE = 21 /* value: index of the phase the item belongs to */,
                 TR_RUN = 32 /* + phase type */ };
__device__ __forceinline__ void vf_trace(const unsigned code, const unsigned long long val = ~0ull) {
    if (threadIdx.x == 0 && blockIdx.x < kTraceWgs) {
        extern __shared__ __attribute__((aligned(16))) float smem_all[];
        int *ctl = reinterpret_cast<int *>(smem_all);
        const int n = ctl[5];
        if (n < kTraceMax) {
            g_trace[(size_t)blockIdx.x * kTraceMax + n] = ((val == ~0ull ? wall_clock64() : val) << 8) | code;
            ctl[5] = n + 1;
        }
    }
}
#define VF_TRACE_EVT(...) vf_trace(__VA_ARGS__)
#else
#define VF_TRACE_EVT(...) do { } while (0)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));

// the acquire behind an observed completion counter (TIMING ONLY, wrong results: -DVF_TIMING_NO_ACQUIRE drops it - what the
// invalidations cost a launch, EXPERIMENTS.md)
#ifdef VF_TIMING_NO_ACQUIRE
#define VF_ACQUIRE_AGENT() do { } while (0)
#else
#define VF_ACQUIRE_AGENT() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#endif
constexpr int kConvThreads = 256;
constexpr int kConvRows = 256;      // GEMM rows per workgroup at MREP = 2 (128 at MREP = 1)
constexpr float kLnEps = 1e-12f;

enum Epilogue {
    EPI_LSTM = 0,             // c,h update; h_out + LayerNorm partial sums
    EPI_BIAS_RELU = 1,        // relu(acc + bias [+ per-sample bias])
    EPI_RAW_STATS = 2,        // acc + bias, + LayerNorm partial sums
    EPI_CONVT_RELU = 3,       // depth-to-space, relu(acc + bias)
    EPI_CONVT_RAW_STATS = 4,  // depth-to-space, acc + bias, + LayerNorm partial sums
    EPI_PARTIAL = 5,          // raw accumulators of one K split
    EPI_CONVT_FUSED = 6,      // top transposed conv whose tile is composed into the next frame right away
                              // (vf_fused_top.h); 6 + 2 * (designated pixels - 1) + (arch 1 / 2 first-frame layer)
                              // + 8 for the six-kernel compositing of arch 2 (four CDNA warps + previous + first + scratch)
    EPI_CONV_PAIR = 30,       // a conv whose whole-image tiles feed a 1x1 conv in the same item (conv_pair_epilogue: enc2 ->
                              // enc3, the 8 x 8 bottleneck; G = 2: both 32-channel groups of the first conv in one workgroup)
    EPI_RAW = 31              // acc + bias, nothing else (arch 3: instance norm needs the whole image's statistics first; the
                              // element-wise items of vf_savp3.h normalise)
};
__host__ __device__ constexpr bool is_top_fused(int epi) { return epi >= EPI_CONVT_FUSED && epi < EPI_CONV_PAIR; }
__host__ __device__ constexpr int fused_epi(int nd, bool first, bool k6 = false) {
    return EPI_CONVT_FUSED + 2 * (nd - 1) + (first ? 1 : 0) + (k6 ? 8 : 0);
}

struct ConvSeg {
    const float *ptr;       // NHWC activations of this input segment
    long long bstride;      // floats between samples (0: one image broadcast to every sample)
    int C;                  // channels (= pixel stride)
    int nchunk;             // ceil(C / KC)
    const long long *ln_part;  // LayerNorm partial sums [B][ln_nparts][2] (sum, sumsq; Q31.32 integers) or null
    long long ln_bstride;   // entries between samples (0: shared statistics)
    int ln_nparts;
    float ln_inv_n;         // 1 / (elements normalised together)
    const float *gamma;     // [gamma_mod]
    const float *beta;
    int gamma_mod;
    int relu;               // relu after (optional) LayerNorm
};

struct ConvParams {
    ConvSeg seg[2];
    int nseg;
    int B;
    int Hin, Win;           // input spatial size
    int Hout, Wout;         // GEMM row grid (transposed conv: == Hin, Win)
    int KH, KW, stride, pad;
    int KC;                 // channels per LDS chunk (8, 16 or 32)
    int NI, TH, TW, RPI;    // images per workgroup, tile shape, GEMM rows reserved per image
    int tilesY, tilesX;     // tiles per image
    int ncg;                // groups of 32 output channels
    int Cout;               // real output channels per gate
    const float *Wp;        // packed weights
    const unsigned short *Wp16;  // split-bf16 packed weights (vf_conv_bf16x6.h) or null
    const float *bias;      // packed [ncg][G][32]
    const float *sbias;     // optional per-sample bias [B][sbias_ld]
    int sbias_ld;
    float *out;
    float *cstate;          // EPI_LSTM: new cell state [B][H][W][C]
    const float *cstate_in; // EPI_LSTM: previous cell state (may alias cstate)
    long long cin_bstride;  // floats between samples of cstate_in (0: shared)
    long long *stats;       // LayerNorm partial sums of the output [B][stats_nparts][2] (Q31.32 integers)
    int stats_nparts;
    int chunks_per_split;   // K split (blockIdx.z)
    int n_valid;            // EPI_PARTIAL: valid output columns
    int tile_variant;       // EPI_LSTM: 0 = conv_tile (fp32), 1 = split-bf16 (vf_conv_bf16x6.h)
    // EPI_CONVT_FUSED only (vf_fused_top.h): the compositing parameters of the same step (device address inside the
    // schedule), the per-sample "LayerNorm partials published" counters, the launch's failure word, view, pixels
    const void *fuse_comp;
    // EPI_CONV_PAIR only: the parameters of the 1x1 conv that consumes this conv's tile inside the same item (device
    // address inside the schedule)
    const void *fuse_next;
    int *fuse_ready;
    const int *fuse_status;
    int fuse_view, fuse_nd;
    // Two-input tiles inside the persistent launch: "early start".  seg[0] is the input that exists EARLY - the
    // recurrent h(s-1) of a conv-LSTM, the encoder skip tensor of a decoder's transposed conv - seg[1] the one produced
    // by the previous layer.  The item is released as soon as seg[0] exists, runs its chunks, and only then waits
    // for the producer of seg[1] (completion counters late_cnt: one per sample, or counter 0 when late_mode = 1, done
    // at late_expect) - the wait that used to idle the slot in front of the item now overlaps part of its K loop.
    // null: every input is complete at entry (per-layer launches).
    const int *late_cnt;
    int late_expect, late_mode;
    int *late_status;       // the launch's sticky failure word
    // Cooperative CU-level priority of the persistent launch (vf_persistent.h, "yielding"): the table of per-workgroup
    // state words (null: off) and how many polls an early-started item may spend yielding to its CU partner
    int *cu_state;
    int yield_budget;
    // Write-through publish (persistent launch, conv-LSTM tiles with 16-byte epilogue stores): c / h leave the tile as sc1
    // stores and the LayerNorm partial as agent-scope atomic stores, so the item needs NO release fence before its
    // completion counter (CDNA guide section 6 G16, recipe R1; a release = buffer_wbl2 costs 1.7 us clean and 6.5 us and
    // more with the tens of KB a tile has just dirtied in its XCD's L2).  0: plain stores + release fence.
    int wt_out;
    // arch 2: the per-sample border-class biases of the tiled conditioning vector, [B][25][4 Cout] (cond_bias_sample,
    // vf_small_kernels.h), added to the gate pre-activations in the conv-LSTM epilogue; null: none
    const float *cond_bias;
    // gate-split tile only: first channel chunk of the K loop.  A conv-LSTM's recurrent input is all zeros at the first step
    // of a rollout - its chunks (segment 0) contribute nothing and are skipped (arch 3; 0 = every chunk)
    int chunk_begin;
};

constexpr unsigned kLateSpinLimit = 1u << 26;   // polls before a mid-item wait gives up (as kSpinLimit)

// ---- Cooperative CU-level priority ("yielding") of the persistent launch.
// Machine model: a CU is time-sliced - next to a K loop the other workgroup's instructions are issued about
// once per 40 cycles whatever their s_setprio, and two K loops share the matrix pipe half and half.  For a batch that fills
// the chip that is harmless (every cycle of the pipe is somebody's throughput).  For a small shard it is not: the RECURRENT
// half of an early-started conv-LSTM item is work for later (its layer input does not exist yet), while the partner
// workgroup's epilogue / staging / light item / input-half K loop is on some sample's dependency chain - and the filler
// slows the chain down 2x (K loop) to 5x (everything else).  Hardware priorities do not arbitrate the matrix pipe, so the
// priority is cooperative: every workgroup publishes one word - 1 while it runs chain-critical work, 0 while it waits,
// polls or runs a recurrent half - and a recurrent half looks at its PARTNER's word once per kernel row (a scalar load
// issued at the head of the row, consumed at its end: nothing is added between the MFMAs) and sleeps while it is 1.
// Bounded: an item yields at most ConvParams::yield_budget polls in total, so a stale word costs time, never progress, and
// the recurrent half never falls behind its own layer input by more than the budget.  Timing only: results cannot change.
constexpr int kYieldSleep = 8;                  // s_sleep units (64 cycles) per yield poll
// words of the persistent kernel's LDS control block that hold the ADDRESSES of this workgroup's state word and of its CU
// partner's (two words each, behind the goal pixels; written once at kernel entry, valid after the first barrier)
constexpr int kCtlMyState = 40, kCtlPartnerState = 42;
__device__ __forceinline__ int *ctl_state_word(const int which) {
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    return *reinterpret_cast<int *const *>(smem_all + which);
}
// publish this workgroup's state (one lane; fire and forget)
template <class PT>
__device__ __forceinline__ void cu_publish(const PT &p, const int critical) {
    if (p.cu_state != nullptr && threadIdx.x == 0)
        __hip_atomic_store(ctl_state_word(kCtlMyState), critical, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the partner's state word as a wave-uniform address
template <class PT>
__device__ __forceinline__ const int *cu_partner_word(const PT &) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(ctl_state_word(kCtlPartnerState));
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return reinterpret_cast<const int *>(((unsigned long long)hi << 32) | lo);
}
// scalar, L2-coherent load of the partner's word: issue now ...
__device__ __forceinline__ unsigned yield_peek_issue(const int *word) {
    unsigned v;
    asm volatile("s_load_dword %0, %1, 0x0 glc" : "=s"(v) : "s"(word) : "memory");
    return v;       // NOT valid before yield_peek_wait
}
// ... consume later (the wait also covers the wave's LDS reads in flight: they are the next row's operands, needed next)
__device__ __forceinline__ unsigned yield_peek_wait(unsigned v) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v) : : "memory");
    return v;
}
// end of a kernel row of a recurrent half: sleep while the partner is on a dependency chain (bounded by `budget`)
__device__ __forceinline__ void yield_to_partner(const int *word, unsigned seen, int &budget) {
    seen = yield_peek_wait(seen);
    if (seen == 0u || budget <= 0) return;
    VF_TRACE_EVT(TR_YIELD);
    do {
        __builtin_amdgcn_s_sleep(kYieldSleep);
        --budget;
        seen = yield_peek_wait(yield_peek_issue(word));
    } while (seen != 0u && budget > 0);
    VF_TRACE_EVT(TR_YIELD_END);
}

// Mid-item wait of a conv-LSTM tile for the producer of its layer input (samples [b0, b1)).  One wave polls
// (relaxed agent-scope loads, s_sleep), acquires, and the result crosses the workgroup through `flag` (LDS).
// Returns false - uniformly - when the producer never arrived: the launch's status word is raised and word 1 of
// the persistent kernel's LDS control block tells its scheduler loop not to publish this item.
template <class PT>
__device__ __forceinline__ bool late_wait(const PT &p, const int b0, const int b1, int *flag) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    cu_publish(p, 0);                   // (yielding) waiting: the partner's recurrent half need not step aside
    if (wave == 0) {
        unsigned spins = 0;
        bool ok;
        do {
            ok = true;
            if (p.late_mode == 1) {
                if (lane == 0)
                    ok = __hip_atomic_load(p.late_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= p.late_expect;
            } else {
                for (int b = b0 + lane; b < b1; b += 64)
                    ok = ok && (__hip_atomic_load(p.late_cnt + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= p.late_expect);
            }
            ok = __all(ok);
            if (!ok) {
                __builtin_amdgcn_s_sleep(kPollSleep);
                if (++spins > kLateSpinLimit ||
                    __hip_atomic_load(p.late_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    if (lane == 0) {
                        extern __shared__ __attribute__((aligned(16))) float smem_all[];
                        atomicExch(p.late_status, 1);
                        reinterpret_cast<int *>(smem_all)[1] = 0;
                    }
                    break;
                }
            }
        } while (!ok);
        if (lane == 0) {
            *flag = ok ? 1 : 0;
            VF_ACQUIRE_AGENT();
        }
    }
    cu_publish(p, 1);                   // the layer input is there: from here on this item is on its sample's chain
    __syncthreads();
    return *flag != 0;
}


// Gate non-linearities.  The quotients use the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division: a
// correctly rounded division is a ten-instruction sequence (two v_div_scale, v_rcp, four fmas, v_div_fmas, v_div_fixup),
// five of them per cell were a third of the conv-LSTM epilogue's instructions, and an epilogue next to a K loop runs
// at about one VALU instruction per 40 cycles (tools/ubench/mfma_shadow.hip).  The exponential is the hardware one
// (v_exp_f32) either way, so the gate math was never correctly rounded; tools/precision_check.py has the distances
// to the float64 oracle.  Every tile plan and launch strategy shares these two functions: results stay bit-identical
// among themselves.
#ifdef VF_GATE_DIV
__device__ __forceinline__ float rcpf_(float x) { return 1.0f / x; }
#else
__device__ __forceinline__ float rcpf_(float x) { return __builtin_amdgcn_rcpf(x); }
#endif
__device__ __forceinline__ float sigmoidf_(float x) { return rcpf_(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    // tanh via exp; saturates cleanly
    float e = __expf(-2.0f * fabsf(x));
#ifdef VF_GATE_DIV
    float t = (1.0f - e) / (1.0f + e);
#else
    float t = (1.0f - e) * rcpf_(1.0f + e);
#endif
    return copysignf(t, x);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ long long wave_sum(long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// LayerNorm statistics are EXACT: every element contributes the integers trunc(v * 2^32) and
// trunc(v^2 * 2^32) (a deterministic function of the element alone; resolution 2.3e-10, far below
// fp32 rounding), and integer addition is associative.  The per-sample totals - and therefore every
// result of the network - do not depend on how a layer is cut into tiles, waves and lanes, which is
// what allows the tile plan to follow the batch size without giving up bit-identical scores across
// chunkings, rank counts and launch strategies.  Range: |v| < 46340 (outputs here are O(1) .. O(100)).
constexpr double kStatScale = 4294967296.0;         // 2^32
__device__ __forceinline__ long long stat_q(const float v) { return (long long)((double)v * kStatScale); }
__device__ __forceinline__ long long stat_q2(const float v) { return (long long)((double)v * (double)v * kStatScale); }
// The same integers summed in float64: trunc(v * 2^32) and trunc(v^2 * 2^32) are integers below 2^32 * max(|v|, v^2),
// and float64 adds integers exactly below 2^53 - so for bounded outputs (|v| <= 1: a conv-LSTM's h = tanh(c) * sigmoid(o))
// a lane can add its terms as doubles (4 instructions each) and convert the lane total to an integer once, instead
// of one emulated float64 -> int64 conversion (~10 instructions) per term.  Same integers, hence the same bits.
struct StatSumD {
    double s = 0.0, q = 0.0;
    __device__ __forceinline__ void add(const float v) {
        const double d = (double)v;
        s += __builtin_trunc(d * kStatScale);
        q += __builtin_trunc(d * d * kStatScale);
    }
    __device__ __forceinline__ long long sum() const { return (long long)s; }
    __device__ __forceinline__ long long sumsq() const { return (long long)q; }
};
// mean / rstd from the integer totals of n = 1 / inv_n elements
__device__ __forceinline__ void ln_from_totals(const long long su, const long long sq, const float inv_n,
                                               float &mean, float &rstd) {
    const double m = (double)su * (1.0 / kStatScale) * (double)inv_n;
    double var = (double)sq * (1.0 / kStatScale) * (double)inv_n - m * m;
    var = var < 0.0 ? 0.0 : var;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)kLnEps));
}

// n / d for the small non-negative indices of a tile (GEMM rows, halo pixels) through a multiply-high: exact for
// n < 70000 and d in 2..600 (checked exhaustively); d == 1 bypasses the multiply.  The divisors are per-layer
// constants that live in scalar registers, so this replaces a ~30-instruction division by one v_mul_hi_u32.
struct TileDiv {
    unsigned magic; int d;
    __device__ __forceinline__ explicit TileDiv(const int d_) : magic(0xFFFFFFFFu / (unsigned)d_ + 1u), d(d_) {}
    __device__ __forceinline__ int div(const int n) const { return d == 1 ? n : (int)__umulhi((unsigned)n, magic); }
};

// mean / rstd of every (input segment, image) of a tile from the producers' partial sums -> lnTab[nseg * NI][2].
// Few entries (conv tiles: 1-8): one WAVE per entry, lanes over the partials (up to 32 per sample at 128x128), so the
// prologue costs one load latency instead of a serial chain; many entries (the FC: one per GEMM row): one thread
// each.  The partials are integers: any summation order gives the same bits.
// (segments [s_begin, s_end): an early-started two-input tile fills the entries of its late segment after the wait)
template <class PT>
__device__ __forceinline__ void ln_table(const PT &p, const int bimg0, float *lnTab, const int s_begin = 0,
                                         const int s_end = 2) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = min(s_end, p.nseg) * p.NI, i0 = s_begin * p.NI;
    if (p.nseg * p.NI <= 8) {
        for (int i = i0 + wave; i < n; i += kConvThreads / 64) {
            const int s = i / p.NI, img = i - s * p.NI;
            const auto &sg = p.seg[s];
            float mean = 0.f, rstd = 1.f;
            const int b = bimg0 + img;
            if (sg.ln_part && b < p.B) {
                long long su = 0, sq = 0;
                const long long *pp = sg.ln_part + (long long)b * sg.ln_bstride;
                for (int k = lane; k < sg.ln_nparts; k += 64) { su += pp[2 * k]; sq += pp[2 * k + 1]; }
                su = wave_sum(su); sq = wave_sum(sq);
                ln_from_totals(su, sq, sg.ln_inv_n, mean, rstd);
            }
            if (lane == 0) { lnTab[2 * i] = mean; lnTab[2 * i + 1] = rstd; }
        }
    } else {
        for (int i = i0 + tid; i < n; i += kConvThreads) {
            const int s = i / p.NI, img = i % p.NI;
            const auto &sg = p.seg[s];
            float mean = 0.f, rstd = 1.f;
            const int b = bimg0 + img;
            if (sg.ln_part && b < p.B) {
                long long su = 0, sq = 0;
                const long long *pp = sg.ln_part + (long long)b * sg.ln_bstride;
                for (int k = 0; k < sg.ln_nparts; ++k) { su += pp[2 * k]; sq += pp[2 * k + 1]; }
                ln_from_totals(su, sq, sg.ln_inv_n, mean, rstd);
            }
            lnTab[2 * i] = mean;
            lnTab[2 * i + 1] = rstd;
        }
    }
}

// One conv-LSTM cell: gate pre-activations (bias added) + previous cell state -> new cell state, new hidden state.
// The ONE place the gate math lives: every tile plan, both precision modes and both launch strategies call it with the
// same values, hence produce the same bits.
__device__ __forceinline__ void lstm_cell(const float gi, const float gj, const float gf, const float go,
                                          const float c_old, float &c_new, float &h_new) {
    c_new = fmaf(c_old, sigmoidf_(gf + 1.0f), sigmoidf_(gi) * tanhf_(gj));
    h_new = tanhf_(c_new) * sigmoidf_(go);
}

// Where GEMM row `row` of a tile lands, for the cell update of the conv-LSTM epilogues.  One image per workgroup
// (NI == 1, every conv-LSTM tile of a 16x16 or larger layer): the sample is wave-uniform, so the three state pointers are
// scalar bases and a lane only computes a 32-bit in-image offset; several whole images per workgroup (the 8x8 layer):
// the sample index is per lane and its stride is applied with one wide multiply.
struct LstmRowAddr {
    bool ok;
    const float *cin;
    float *cst, *hout;
    unsigned off;
};
template <bool NI1, class PT>
__device__ __forceinline__ LstmRowAddr lstm_row_addr(const PT &p, const int row, const int ch, const int bimg0,
                                                      const int ty0, const int tx0, const long long img_elems,
                                                      const TileDiv &div_rpi, const TileDiv &div_tw) {
    LstmRowAddr a;
    if constexpr (NI1) {
        const int yy = div_tw.div(row);
        const int y = ty0 + yy, x = tx0 + row - yy * p.TW;
        a.ok = row < p.TH * p.TW && bimg0 < p.B && y < p.Hout && x < p.Wout;
        a.off = (unsigned)((y * p.Wout + x) * p.Cout + ch);
        a.cin = p.cstate_in + (long long)bimg0 * p.cin_bstride;
        a.cst = p.cstate + (long long)bimg0 * img_elems;
        a.hout = p.out + (long long)bimg0 * img_elems;
    } else {
        const int img = div_rpi.div(row), rem = row - img * p.RPI;
        const int yy = div_tw.div(rem);
        const int y = ty0 + yy, x = tx0 + rem - yy * p.TW;
        const int b = bimg0 + img;
        a.ok = img < p.NI && rem < p.TH * p.TW && b < p.B && y < p.Hout && x < p.Wout;
        a.off = (unsigned)((y * p.Wout + x) * p.Cout + ch);
        a.cin = p.cstate_in + (long long)b * p.cin_bstride;
        a.cst = p.cstate + (long long)b * img_elems;
        a.hout = p.out + (long long)b * img_elems;
    }
    return a;
}

// Shared epilogue of the fp32 and the split-bf16 tiles: accumulators (MFMA 32x32 C layout) ->
// bias / activation / cell update / stores + deterministic LayerNorm partial sums.
constexpr int kEpiVecFloats = 4 * 32 * 36;     // LDS floats of the vectorised light-layer epilogue (wave-private 32 x 36 slabs)

template <int G, int EPI, int MREP, class PT>
__device__ __forceinline__ void conv_epilogue(const PT &p, f32x16 (&acc)[MREP][G], const int bx, const int by,
                                              const int bz, long long *red, float *smem = nullptr) {
    constexpr int WROWS = MREP * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, kh = lane >> 5;
    const int cg = by;
    const int tiles_per_img = p.tilesY * p.tilesX;
    const int px_per_img = p.TH * p.TW;
    int bimg0, ty0, tx0, tile_id;
    if (p.NI == 1) {
        bimg0 = bx / tiles_per_img;
        tile_id = bx % tiles_per_img;
        ty0 = (tile_id / p.tilesX) * p.TH;
        tx0 = (tile_id % p.tilesX) * p.TW;
    } else {
        bimg0 = bx * p.NI;
        tile_id = 0; ty0 = 0; tx0 = 0;
    }
    const int ch = cg * 32 + n;             // output channel of this lane
    float bias_g[G];
#pragma unroll
    for (int g = 0; g < G; ++g) bias_g[g] = (EPI == EPI_PARTIAL) ? 0.f : p.bias[(cg * G + g) * 32 + n];

    long long ssum = 0, ssq = 0;            // exact LayerNorm partials over this lane's outputs
    [[maybe_unused]] StatSumD hstat;        // (conv-LSTM: the same integers, summed in float64)
    [[maybe_unused]] const long long img_elems = (long long)p.Hout * p.Wout * p.Cout;
    const TileDiv div_rpi(p.RPI), div_tw(p.TW);

    if constexpr (EPI == EPI_LSTM) {
        auto cells = [&](auto ni1) {
#pragma unroll
            for (int m = 0; m < MREP; ++m) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wave * WROWS + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const LstmRowAddr a = lstm_row_addr<decltype(ni1)::value>(p, row, ch, bimg0, ty0, tx0, img_elems,
                                                                              div_rpi, div_tw);
                    if (!a.ok) continue;
                    float c_new, h_new;
                    lstm_cell(acc[m][0][r] + bias_g[0], acc[m][1 % G][r] + bias_g[1 % G],
                              acc[m][2 % G][r] + bias_g[2 % G], acc[m][3 % G][r] + bias_g[3 % G], a.cin[a.off],
                              c_new, h_new);
                    a.cst[a.off] = c_new;
                    a.hout[a.off] = h_new;
                    hstat.add(h_new);
                }
            }
        };
        if (p.NI == 1) cells(std::true_type{}); else cells(std::false_type{});
    }
    // Every other layer type.  One image per workgroup (NI1): the sample is wave-uniform, so the output pointer is a
    // scalar base plus a 32-bit in-image offset per lane; several images: the sample's stride enters through one wide
    // multiply.  (An epilogue next to a K loop pays ~40 cycles per VALU instruction: 64-bit address chains per stored
    // element were most of a transposed conv's item time.)
    // The light layers of the persistent launch / the per-layer kernels (one row block per wave): the outputs leave the
    // tile as 16-byte stores.  The MFMA layout keeps ONE channel per lane (16 rows of it), so a wave first turns its 32 x 32
    // block over in a private LDS slab ([row][36]: conflict-free both ways) and every lane then owns four consecutive channels
    // of four pixels: 4 memory instructions and 4 pixel geometries per gate instead of 16 - and, with ConvParams::wt_out, as
    // sc1 (write-through) stores, which is what lets the item publish without a release fence (16-byte sc1 stores cost what
    // plain ones do, 4-byte ones six times as much per byte: CDNA guide, section 6 G16).  Same values, same statistics.
    // (EPI_RAW also with two row blocks per wave - the 256-row plan of arch 3's full-resolution layers: the same per block)
    constexpr bool kVec = (MREP == 1 && (EPI == EPI_BIAS_RELU || EPI == EPI_RAW_STATS || EPI == EPI_CONVT_RELU ||
                                          EPI == EPI_CONVT_RAW_STATS)) || EPI == EPI_RAW;
    if constexpr (kVec) {
        constexpr bool kT = EPI == EPI_CONVT_RELU || EPI == EPI_CONVT_RAW_STATS;
        constexpr bool kRelu = EPI == EPI_BIAS_RELU || EPI == EPI_CONVT_RELU;
        constexpr bool kStats = EPI == EPI_RAW_STATS || EPI == EPI_CONVT_RAW_STATS;
        const bool ni1 = p.NI == 1;
        const int n_here = ni1 ? (bimg0 < p.B ? 1 : 0) : min(p.NI, p.B - bimg0);
        // (EPI_RAW with G > 1, arch 3: the item's G "gates" are G consecutive 32-channel groups of a plain conv with
        // kGS * Cout output channels - gate g of column group cg = channels [(cg G + g) 32, +32) of every pixel, the order
        // of the plain layer's packed weights and bias)
        constexpr int kGS = (EPI == EPI_RAW) ? G : 1;
        const long long out_elems = (long long)(kT ? 4 : kGS) * p.Hout * p.Wout * p.Cout;
        const unsigned out_bytes = (unsigned)out_elems * 4u;
        const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(
            p.out + (long long)bimg0 * out_elems, 0, n_here > 0 ? (int)((unsigned)n_here * out_bytes) : 0, 0x00020000);
        const bool wt = p.wt_out != 0;
        // ---- values (bias, per-sample bias, relu) and the exact statistics, lane = channel
        const bool ch_ok = ch < p.Cout;
#pragma unroll
        for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave * WROWS + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            int img = 0, rem = row;
            if (!ni1) { img = div_rpi.div(row); rem = row - img * p.RPI; }
            const int yy = div_tw.div(rem);
            const int y = ty0 + yy, x = tx0 + rem - yy * p.TW;
            const bool ok = img < n_here && rem < px_per_img && y < p.Hout && x < p.Wout && ch_ok;
            float sb = 0.f;
            if (p.sbias && ok) sb = p.sbias[(long long)(bimg0 + img) * p.sbias_ld + ch];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float v = acc[m][g][r] + bias_g[g];
                if (p.sbias) v += sb;
                if (kRelu) v = fmaxf(v, 0.f);
                acc[m][g][r] = v;
                if constexpr (kStats) { if (ok) { ssum += stat_q(v); ssq += stat_q2(v); } }
            }
        }
        // ---- where this lane's four pixels (rows lane / 8 + 8 k of the wave's block) go: channels 4 (lane % 8) ..
        const int pl = lane >> 3, cq = lane & 7;
        unsigned off_o[MREP][4];
#pragma unroll
        for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = wave * WROWS + m * 32 + pl + 8 * k;
            int img = 0, rem = row;
            if (!ni1) { img = div_rpi.div(row); rem = row - img * p.RPI; }
            const int yy = div_tw.div(rem);
            const int y = ty0 + yy, x = tx0 + rem - yy * p.TW;
            const bool ok = img < n_here && rem < px_per_img && y < p.Hout && x < p.Wout && (kGS > 1 || cg * 32 + 4 * cq < p.Cout);
            const unsigned in_img = kT ? (unsigned)(((2 * y) * (2 * p.Wout) + 2 * x) * p.Cout + cg * 32 + 4 * cq) * 4u
                                       : (unsigned)((y * p.Wout + x) * (kGS * p.Cout) + cg * (kGS * 32) + 4 * cq) * 4u;
            off_o[m][k] = ok ? (unsigned)img * out_bytes + in_img : 0xFFFFFFFFu;
        }
        __syncthreads();                    // every wave is done reading the operand tile: its LDS becomes the slabs
        float *T = smem + wave * (32 * 36);
#pragma unroll
        for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * kh) * 36 + n] = acc[m][g][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const unsigned par = kT ? (unsigned)(((g >> 1) * (2 * p.Wout) + (g & 1)) * p.Cout) * 4u
                                    : (kGS > 1 ? (unsigned)(g * 32) * 4u : 0u);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 q = *reinterpret_cast<const f32x4 *>(T + (pl + 8 * k) * 36 + 4 * cq);
                const unsigned off = off_o[m][k] == 0xFFFFFFFFu ? off_o[m][k] : off_o[m][k] + par;
                if (wt) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, q), r_out, off, 0, 16);
                else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, q), r_out, off, 0, 0);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    if constexpr (EPI != EPI_LSTM && !kVec) {
        auto rows = [&](auto ni1c) {
            constexpr bool NI1 = decltype(ni1c)::value;
            constexpr bool kT = EPI == EPI_CONVT_RELU || EPI == EPI_CONVT_RAW_STATS;
            const long long out_elems = (long long)(kT ? 4 : 1) * p.Hout * p.Wout * p.Cout;    // floats per output image
#pragma unroll
            for (int m = 0; m < MREP; ++m) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wave * WROWS + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    int img = 0, rem = row;
                    if constexpr (!NI1) { img = div_rpi.div(row); rem = row - img * p.RPI; }
                    const int yy = div_tw.div(rem);
                    const int y = ty0 + yy, x = tx0 + rem - yy * p.TW;
                    const int b = bimg0 + img;
                    const bool ok = img < p.NI && rem < px_per_img && b < p.B && y < p.Hout && x < p.Wout;
                    if (!ok) continue;
                    if constexpr (EPI == EPI_BIAS_RELU || EPI == EPI_RAW_STATS) {
                        if (ch < p.Cout) {
                            float v = acc[m][0][r] + bias_g[0];
                            if (p.sbias) v += p.sbias[(long long)b * p.sbias_ld + ch];
                            if (EPI == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
                            float *ob = p.out + (long long)b * out_elems;
                            ob[(unsigned)((y * p.Wout + x) * p.Cout + ch)] = v;
                            if constexpr (EPI == EPI_RAW_STATS) { ssum += stat_q(v); ssq += stat_q2(v); }
                        }
                    } else if constexpr (kT) {
                        if (ch < p.Cout) {
                            float *ob = p.out + (long long)b * out_elems;
                            const unsigned o00 = (unsigned)(((2 * y) * (2 * p.Wout) + 2 * x) * p.Cout + ch);
#pragma unroll
                            for (int g = 0; g < G; ++g) {
                                float v = acc[m][g][r] + bias_g[g];
                                if (EPI == EPI_CONVT_RELU) v = fmaxf(v, 0.f);
                                ob[o00 + (unsigned)(((g >> 1) * (2 * p.Wout) + (g & 1)) * p.Cout)] = v;
                                if constexpr (EPI == EPI_CONVT_RAW_STATS) { ssum += stat_q(v); ssq += stat_q2(v); }
                            }
                        }
                    } else {    // EPI_PARTIAL: [split][B][n_valid]
                        if (ch < p.n_valid)
                            p.out[((long long)bz * p.B + b) * p.n_valid + ch] = acc[m][0][r];
                    }
                }
            }
        };
        if (p.NI == 1) rows(std::true_type{}); else rows(std::false_type{});
    }

    if constexpr (EPI == EPI_LSTM) { ssum = hstat.sum(); ssq = hstat.sumsq(); }
    if constexpr (EPI == EPI_LSTM || EPI == EPI_RAW_STATS || EPI == EPI_CONVT_RAW_STATS) {
        // exact integer reduction: lane -> wave (xor butterfly) -> waves -> one partial per tile
        const long long wsum = wave_sum(ssum), wsq = wave_sum(ssq);
        __syncthreads();
        if (lane == 0) { red[2 * wave] = wsum; red[2 * wave + 1] = wsq; }
        __syncthreads();
        // (write-through items: the partial leaves as agent-scope atomic stores, like every other output of the tile)
        auto put = [&](long long *dst, const long long su, const long long sq) {
            if (p.wt_out != 0) {
                __hip_atomic_store(dst, su, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(dst + 1, sq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                dst[0] = su; dst[1] = sq;
            }
        };
        if (p.NI == 1) {
            if (tid == 0 && bimg0 < p.B) {
                long long su = 0, sq = 0;
                for (int w = 0; w < 4; ++w) { su += red[2 * w]; sq += red[2 * w + 1]; }
                put(p.stats + ((long long)bimg0 * p.stats_nparts + tile_id * p.ncg + cg) * 2, su, sq);
            }
        } else {
            // RPI is a multiple of WROWS here: wave w owns image slot (w*WROWS)/RPI entirely or
            // shares it with its neighbours; sum the waves of each image in fixed order
            const int waves_per_img = p.RPI / WROWS;
            if (lane == 0 && (wave % waves_per_img) == 0) {
                const int img = wave / waves_per_img;
                const int b = bimg0 + img;
                if (img < p.NI && b < p.B) {
                    long long su = 0, sq = 0;
                    for (int w = 0; w < waves_per_img; ++w) { su += red[2 * (wave + w)]; sq += red[2 * (wave + w) + 1]; }
                    put(p.stats + ((long long)b * p.stats_nparts + cg) * 2, su, sq);
                }
            }
        }
    }
}

// Epilogue of the row-split conv-LSTM tiles (conv_tile<4, EPI_LSTM, 1, PT, RB> with RB = 2 or 1 row blocks
// per workgroup): wave w holds GA = RB gates (2: {i,j} or {f,o}; 1: a single gate) of row block w % RB, so
// the gate pre-activations cross through LDS (xch: [RB][4 gates][16][64 lanes] floats, the idle weight
// buffers) and every wave finishes 4 * RB of its row block's 16 accumulator rows with all four gates at hand.
// Same expressions on the same values, hence the same bits, as conv_epilogue.
template <int RB, class PT>
__device__ __forceinline__ void lstm_split_epilogue(const PT &p, f32x16 (&acc)[1][RB], const int bx, const int by,
                                                    long long *red, float *xch) {
    constexpr int GA = RB;                  // gates per wave
    constexpr int RSTEP = 4 * RB;           // accumulator rows finished per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, kh = lane >> 5;
    const int rb = wave % RB, gg = wave / RB;       // row block, gate group
    const int cg = by;
    const int tiles_per_img = p.tilesY * p.tilesX;
    int bimg0, ty0, tx0, tile_id;
    if (p.NI == 1) {
        bimg0 = bx / tiles_per_img;
        tile_id = bx % tiles_per_img;
        ty0 = (tile_id / p.tilesX) * p.TH;
        tx0 = (tile_id % p.tilesX) * p.TW;
    } else {
        bimg0 = bx * p.NI;
        tile_id = 0; ty0 = 0; tx0 = 0;
    }
    const int ch = cg * 32 + n;
    __syncthreads();                        // the operand tiles are no longer read: their LDS becomes xch
#pragma unroll
    for (int g = 0; g < GA; ++g) {
        const int gate = gg * GA + g;
        const float bias = p.bias[(cg * 4 + gate) * 32 + n];
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[((rb * 4 + gate) * 16 + r) * 64 + lane] = acc[0][g][r] + bias;
    }
    __syncthreads();
    StatSumD hstat;
    const long long img_elems = (long long)p.Hout * p.Wout * p.Cout;
    const TileDiv div_rpi(p.RPI), div_tw(p.TW);
    auto cells = [&](auto ni1) {
#pragma unroll
        for (int rr = 0; rr < RSTEP; ++rr) {
            const int r = gg * RSTEP + rr;
            const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            const LstmRowAddr a = lstm_row_addr<decltype(ni1)::value>(p, row, ch, bimg0, ty0, tx0, img_elems, div_rpi,
                                                                      div_tw);
            if (!a.ok) continue;
            const float gi = xch[((rb * 4 + 0) * 16 + r) * 64 + lane], gj = xch[((rb * 4 + 1) * 16 + r) * 64 + lane];
            const float gf = xch[((rb * 4 + 2) * 16 + r) * 64 + lane], go = xch[((rb * 4 + 3) * 16 + r) * 64 + lane];
            float c_new, h_new;
            lstm_cell(gi, gj, gf, go, a.cin[a.off], c_new, h_new);
            a.cst[a.off] = c_new;
            a.hout[a.off] = h_new;
            hstat.add(h_new);
        }
    };
    if (p.NI == 1) cells(std::true_type{}); else cells(std::false_type{});
    const long long wsum = wave_sum(hstat.sum()), wsq = wave_sum(hstat.sumsq());
    if (lane == 0) { red[2 * wave] = wsum; red[2 * wave + 1] = wsq; }
    __syncthreads();
    if (tid == 0) {
        // wave w carries part of row block w % RB, which lies in image slot (w % RB) * 32 / RPI
        for (int img = 0; img < p.NI; ++img) {
            if (bimg0 + img >= p.B) continue;
            long long su = 0, sq = 0;
            for (int w = 0; w < 4; ++w)
                if (((w % RB) * 32) / p.RPI == img) { su += red[2 * w]; sq += red[2 * w + 1]; }
            long long *dst = p.stats + ((long long)(bimg0 + img) * p.stats_nparts +
                                        (p.NI == 1 ? tile_id * p.ncg + cg : cg)) * 2;
            dst[0] = su; dst[1] = sq;
        }
    }
}

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{})
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// Epilogue of the gate-split 128-row conv-LSTM tile (conv_tile<4, EPI_LSTM, 1, PT, 0>): wave w holds gate w of all four
// row blocks.  The gate pre-activations cross through LDS (xch: [4 row blocks][4 gates][16][64 lanes] floats = 64 KiB
// over the dead operand tile) and wave w finishes row block w with all four gates at hand; the reduction scratch lies
// behind xch.  Same expressions on the same values, hence the same bits, as conv_epilogue.
// (MR = 8, the 256-row tile: the same in two rounds of four row blocks - wave w finishes row blocks w and 4 + w)
constexpr int kGsXchFloats = 4 * 4 * 16 * 64;
// arch 2: LDS floats of a 128-row gate-split tile whose epilogue adds the conditioning biases through its class tables
// (exchange buffer, reduction scratch, [25][128] values, [128] classes)
constexpr int kGsCondFloats = kGsXchFloats + 16 + 25 * 128 + 128;
template <int MR, class PT>
__device__ __forceinline__ void lstm_gsplit_epilogue(const PT &p, f32x16 (&acc)[MR][1], const int bx, const int by,
                                                     float *smem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, kh = lane >> 5;
    const int cg = by;
    const int tiles_per_img = p.tilesY * p.tilesX;
    int bimg0, ty0, tx0, tile_id;
    if (p.NI == 1) {
        bimg0 = bx / tiles_per_img;
        tile_id = bx % tiles_per_img;
        ty0 = (tile_id / p.tilesX) * p.TH;
        tx0 = (tile_id % p.tilesX) * p.TW;
    } else {
        bimg0 = bx * p.NI;
        tile_id = 0; ty0 = 0; tx0 = 0;
    }
    const int ch = cg * 32 + n;
    float *xch = smem;
    // (the reduction scratch lies behind the part of xch this tile height uses: 16 KiB per row block of a round)
    long long *red = reinterpret_cast<long long *>(smem + (MR < 4 ? MR : 4) * 4 * 16 * 64);
    const bool wt = p.wt_out != 0;          // write-through publish: sc1 stores, no release fence behind them

    // ---- Wave w finishes row block w: 32 pixels x 32 channels.  After the exchange a lane owns FOUR pixels x FOUR
    // consecutive channels (pixel lane / 8 + 8 k, channels 4 (lane % 8) ..): the gates come out of xch as float4 (the
    // MFMA layout keeps a pixel's 32 channels in 32 consecutive lanes), the previous cell state is one 16-byte load and
    // c / h two 16-byte stores per pixel - 12 memory instructions per lane instead of 48, and the pixel geometry is
    // computed 4 times instead of 16.  The three state tensors of this workgroup's images are raw buffers (wave-uniform
    // base and size): a pixel outside the image / batch gets the offset ~0, its load returns zero and its stores are
    // dropped - no branch anywhere, so the loads are issued before the gate exchange and have landed when the gate
    // math needs them.  Per cell the same expressions on the same values as conv_epilogue, and the statistics are
    // exact integers: the same bits.
    const long long img_elems = (long long)p.Hout * p.Wout * p.Cout;
    const TileDiv div_rpi(p.RPI), div_tw(p.TW);
    const int n_here = min(p.NI, p.B - bimg0);
    const unsigned img_bytes = (unsigned)img_elems * 4u;
    const unsigned cin_step = (unsigned)p.cin_bstride * 4u;     // 0: one state shared by every sample (context steps)
    const int span_out = n_here > 0 ? (int)((unsigned)(n_here - 1) * img_bytes + img_bytes) : 0;
    const int span_cin = n_here > 0 ? (int)((unsigned)(n_here - 1) * cin_step + img_bytes) : 0;
    const __amdgpu_buffer_rsrc_t r_cin = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.cstate_in + (long long)bimg0 * p.cin_bstride), 0, span_cin, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_cst =
        __builtin_amdgcn_make_buffer_rsrc(p.cstate + (long long)bimg0 * img_elems, 0, span_out, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_out =
        __builtin_amdgcn_make_buffer_rsrc(p.out + (long long)bimg0 * img_elems, 0, span_out, 0x00020000);
    const int pl = lane >> 3, cq = lane & 7;            // pixel lane, channel quad
    const bool ni1 = p.NI == 1;
    const float bias = p.bias[(cg * 4 + wave) * 32 + n];
    StatSumD hstat;
    // rounds of RBR row blocks through xch; in a round a wave finishes NK of the four 8-pixel groups of ONE row block:
    // MR 4 / 8: row block (round * 4 + wave), all four groups; MR 2: row block (wave & 1), groups 2 (wave >> 1) + {0, 1}
    constexpr int RBR = MR < 4 ? MR : 4, NK = RBR, ROUNDS = MR / RBR;
#pragma unroll
    for (int half = 0; half < ROUNDS; ++half) {
        const int rbl = MR < 4 ? (wave & (RBR - 1)) : wave;                 // row block within the round
        const int k0 = MR < 4 ? NK * (wave / RBR) : 0;                      // first pixel group of this wave
        const int rb = half * RBR + rbl;                 // the row block this wave finishes in this round
        unsigned off_o[NK];
        f32x4 c_old[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int row = rb * 32 + pl + 8 * (k0 + k);
            int img = 0, rem = row;
            if (!ni1) { img = div_rpi.div(row); rem = row - img * p.RPI; }
            const int yy = div_tw.div(rem);
            const int y = ty0 + yy, x = tx0 + rem - yy * p.TW;
            const bool ok = img < n_here && rem < p.TH * p.TW && y < p.Hout && x < p.Wout;
            const unsigned in_img = (unsigned)((y * p.Wout + x) * p.Cout + cg * 32 + 4 * cq) * 4u;
            off_o[k] = ok ? (unsigned)img * img_bytes + in_img : 0xFFFFFFFFu;
            const unsigned off_c = ok ? (unsigned)img * cin_step + in_img : 0xFFFFFFFFu;
            c_old[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_cin, off_c, 0, 0));
        }
        __syncthreads();                    // the operand tile / the previous round's gates are no longer read
        if (MR == 4 && ni1 && p.cond_bias != nullptr) {
            // arch 2, one image per 128-row tile: + the conditioning bias of each row's border class through two small LDS
            // tables behind the exchange buffer (kGsCondFloats, planned by init_layer) - this lane's 25 class values of its
            // gate column and the class of each of the 128 rows - instead of a row geometry and a 4-byte global load per
            // accumulator element (64 per lane: 10 us of a 36-us epilogue at the C5 shard).  The same addends: the same bits.
            const int C4 = 4 * p.Cout, col = wave * p.Cout + ch;
            float *ctab = reinterpret_cast<float *>(red + 8);           // [25][128]
            int *ccls = reinterpret_cast<int *>(ctab + 25 * 128);       // [128]
            const float *cb = p.cond_bias + (long long)bimg0 * (25 * C4) + col;
            for (int cls = kh; cls < 25; cls += 2) ctab[cls * 128 + wave * 32 + n] = cb[(long long)cls * C4];
            if (tid < 128) {
                const int yy = div_tw.div(tid);
                const int y = ty0 + yy, x = tx0 + tid - yy * p.TW;
                const bool ok = n_here > 0 && tid < p.TH * p.TW && y < p.Hout && x < p.Wout;
                const int cy = y < 2 ? y : (y >= p.Hout - 2 ? y - (p.Hout - 5) : 2);
                const int cx = x < 2 ? x : (x >= p.Wout - 2 ? x - (p.Wout - 5) : 2);
                ccls[tid] = ok ? cy * 5 + cx : -1;
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < RBR; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cl = ccls[(half * RBR + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh];
                    const float add = cl >= 0 ? ctab[cl * 128 + wave * 32 + n] : 0.f;
                    acc[half * RBR + m][0][r] = acc[half * RBR + m][0][r] + add;
                }
        } else if (p.cond_bias != nullptr) {
            // arch 2: + the conditioning bias of each row's border class (lane = gate column wave * Cout + ch; the class
            // is uniform over each half of the wave, the 32 lanes of a half read 128 consecutive bytes)
            const int C4 = 4 * p.Cout, col = wave * p.Cout + ch;
            const float *cb = p.cond_bias + (long long)bimg0 * (25 * C4) + col;
#pragma unroll
            for (int m = 0; m < RBR; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (half * RBR + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    int img = 0, rem = row;
                    if (!ni1) { img = div_rpi.div(row); rem = row - img * p.RPI; }
                    const int yy = div_tw.div(rem);
                    const int y = ty0 + yy, x = tx0 + rem - yy * p.TW;
                    const bool ok = img < n_here && rem < p.TH * p.TW && y < p.Hout && x < p.Wout;
                    const int cy = y < 2 ? y : (y >= p.Hout - 2 ? y - (p.Hout - 5) : 2);
                    const int cx = x < 2 ? x : (x >= p.Wout - 2 ? x - (p.Wout - 5) : 2);
                    const float add = ok ? cb[(long long)(img * 25 + cy * 5 + cx) * C4] : 0.f;
                    acc[half * RBR + m][0][r] = acc[half * RBR + m][0][r] + add;
                }
        }
#pragma unroll
        for (int m = 0; m < RBR; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) xch[((m * 4 + wave) * 16 + r) * 64 + lane] = acc[half * RBR + m][0][r] + bias;
        __syncthreads();
        // GEMM row R = pl + 8 k of the block sits in accumulator row r = (R & 3) + 4 (R >> 3) of lane half (R >> 2) & 1
        const float *xw = xch + ((rbl * 4) * 16 + (pl & 3)) * 64 + 32 * ((pl >> 2) & 1) + 4 * cq;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            f32x4 gate[4], cn, hn;
#pragma unroll
            for (int g = 0; g < 4; ++g) gate[g] = *reinterpret_cast<const f32x4 *>(xw + (g * 16 + 4 * (k0 + k)) * 64);
            const bool live = off_o[k] != 0xFFFFFFFFu;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float c_new, h_new;
                lstm_cell(gate[0][e], gate[1][e], gate[2][e], gate[3][e], c_old[k][e], c_new, h_new);
                cn[e] = c_new; hn[e] = h_new;
                hstat.add(live ? h_new : 0.f);          // (a dropped pixel adds the integer 0)
            }
            if (wt) {       // (aux 16 = sc1)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, cn), r_cst, off_o[k], 0, 16);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, hn), r_out, off_o[k], 0, 16);
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, cn), r_cst, off_o[k], 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, hn), r_out, off_o[k], 0, 0);
            }
        }
    }
    // exact integer reduction (the 256-row tile always holds ONE image - plan_geometry - so only the per-tile total is
    // needed there; otherwise wave w finished (part of) row block w, or w & 1 in the 64-row tile)
    const long long wsum = wave_sum(hstat.sum()), wsq = wave_sum(hstat.sumsq());
    if (lane == 0) { red[2 * wave] = wsum; red[2 * wave + 1] = wsq; }
    __syncthreads();
    if (tid == 0) {
        for (int img = 0; img < p.NI; ++img) {
            if (bimg0 + img >= p.B) continue;
            long long su = 0, sq = 0;
            for (int w = 0; w < 4; ++w) {
                const int wrb = MR < 4 ? (w & (MR - 1)) : w;
                if (p.NI == 1 || (wrb * 32) / p.RPI == img) { su += red[2 * w]; sq += red[2 * w + 1]; }
            }
            long long *dst = p.stats + ((long long)(bimg0 + img) * p.stats_nparts +
                                        (p.NI == 1 ? tile_id * p.ncg + cg : cg)) * 2;
            if (wt) {
                __hip_atomic_store(dst, su, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(dst + 1, sq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                dst[0] = su; dst[1] = sq;
            }
        }
    }
}

// Raw epilogue of the gate-split 128-row tile (arch 3, vf_savp3.h): the published SAVP cell normalises the gate
// pre-activations per sample and channel over the whole image before the cell update, so the tile's job ends with the
// GEMM - the gates leave as they are, [pixel][4C] with gate-major columns, for the element-wise cell item.  Same
// exchange through LDS as lstm_gsplit_epilogue (wave w holds gate w of all four row blocks and finishes row block w):
// a lane stores four pixels x four consecutive channels x four gates as sixteen 16-byte stores.
template <int MR, class PT>
__device__ __forceinline__ void gates_raw_epilogue(const PT &p, f32x16 (&acc)[MR][1], const int bx, const int by, float *smem) {
    static_assert(MR == 4, "raw gate epilogue: 128-row tile");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    float *xch = smem;
    const bool wt = p.wt_out != 0;
    const int C4 = 4 * p.Cout;
    const long long img_elems = (long long)p.Hout * p.Wout * C4;
    const TileDiv div_rpi(p.RPI), div_tw(p.TW);
    const int n_here = min(p.NI, p.B - bimg0);
    const unsigned img_bytes = (unsigned)img_elems * 4u;
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(
        p.out + (long long)bimg0 * img_elems, 0, n_here > 0 ? (int)((unsigned)n_here * img_bytes) : 0, 0x00020000);
    const int pl = lane >> 3, cq = lane & 7;
    const bool ni1 = p.NI == 1;
    unsigned off_o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = wave * 32 + pl + 8 * k;
        int img = 0, rem = row;
        if (!ni1) { img = div_rpi.div(row); rem = row - img * p.RPI; }
        const int yy = div_tw.div(rem);
        const int y = ty0 + yy, x = tx0 + rem - yy * p.TW;
        const bool ok = img < n_here && rem < p.TH * p.TW && y < p.Hout && x < p.Wout;
        off_o[k] = ok ? (unsigned)img * img_bytes + (unsigned)((y * p.Wout + x) * C4 + cg * 32 + 4 * cq) * 4u : 0xFFFFFFFFu;
    }
    __syncthreads();                        // the operand tile is no longer read: its LDS becomes xch
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[((m * 4 + wave) * 16 + r) * 64 + lane] = acc[m][0][r];
    __syncthreads();
    const float *xw = xch + ((wave * 4) * 16 + (pl & 3)) * 64 + 32 * ((pl >> 2) & 1) + 4 * cq;
    const unsigned gstep = (unsigned)p.Cout * 4u;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(xw + (g * 16 + 4 * k) * 64);
            const unsigned off = off_o[k] == 0xFFFFFFFFu ? off_o[k] : off_o[k] + g * gstep;
            if (wt) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), r_out, off, 0, 16);
            else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), r_out, off, 0, 0);
        }
}

// Epilogue of conv_tile<2, EPI_CONV_PAIR, 1>: conv -> relu -> 1x1 conv (+ per-sample bias) -> relu in ONE item.
// The first conv's tiles hold whole images (the 8 x 8 bottleneck: enc2, 3x3 / 2) and its two 32-channel groups are
// the two "gates" of this workgroup (same packed weights, same K order per output as the two items of the stand-alone
// layer), so the tile in the accumulators is the COMPLETE input of the 1x1 conv that follows (enc3): it goes to LDS in
// the layout that conv's staging would have produced - [32-channel chunk][GEMM row][KC + 4], the relu'd fp32 values
// the stand-alone layer stores and reloads - and the second GEMM runs its K loop in the stand-alone order (chunk ->
// k8 -> j) on the same operands: the same bits, one item, one dependency hop and one round trip through memory less
// per sample-step.  The geometry of both layers is the same row grid (checked on the host: NI, RPI, tile = image).
template <class PT>
__device__ __forceinline__ void conv_pair_epilogue(const PT &p, f32x16 (&acc)[1][2], const int bx, float *smem) {
    typedef const __attribute__((address_space(4))) ConvParams QT;
    const unsigned long long qa = reinterpret_cast<unsigned long long>(p.fuse_next);
    const unsigned qlo = __builtin_amdgcn_readfirstlane((unsigned)qa);
    const unsigned qhi = __builtin_amdgcn_readfirstlane((unsigned)(qa >> 32));
    QT &q = *(QT *)(((unsigned long long)qhi << 32) | qlo);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, kh = lane >> 5;
    constexpr int KCP = 36;                 // 32-channel chunks of the second conv, padded like every operand tile
    const int bimg0 = bx * p.NI;            // (whole images per tile: tilesY * tilesX == 1)
    float *T = smem;                        // [2 chunks][128 rows][KCP] over the dead operand tile of the first conv

    // the second conv's weights: 2 chunks x 4 k8 steps x 2 column groups, requested before anything else
    const int Ntot = q.ncg * 32;            // (= 64: the stand-alone layer runs two channel-group items on these columns)
    const float *wl = q.Wp + ((long long)kh * Ntot + n) * 4;
    const long long wstep = (long long)2 * Ntot * 4;
    f32x4 wb[2][4][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8)
#pragma unroll
            for (int g = 0; g < 2; ++g)
                wb[c][k8][g] = *reinterpret_cast<const f32x4 *>(wl + (long long)(c * 4 + k8) * wstep + g * 128);

    // ---- 1. relu(acc + bias) -> T (exactly what conv_epilogue<1, EPI_BIAS_RELU> stores for the stand-alone layer)
    __syncthreads();                        // every wave is done reading the operand tile
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const float bias = p.bias[g * 32 + n];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            T[(g * 128 + row) * KCP + n] = fmaxf(acc[0][g][r] + bias, 0.f);
        }
    }
    __syncthreads();

    // ---- 2. the 1x1 conv: rows = the same GEMM rows, K = 64 input channels in two chunks, columns = 2 x 32
    f32x16 acc2[2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[g][r] = 0.f;
    const f32x4 *T4 = reinterpret_cast<const f32x4 *>(T);
    const int arow = wave * 32 + n;         // this lane's A row; channels 8 k8 + 4 kh .. + 3 of the chunk
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
            const f32x4 a = T4[((c * 128 + arow) * KCP + k8 * 8 + kh * 4) >> 2];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 2; ++g)
                    acc2[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], wb[c][k8][g][j], acc2[g], 0, 0, 0);
        }

    // ---- 3. bias + per-sample bias + relu -> the second conv's output (conv_epilogue<1, EPI_BIAS_RELU> of that layer)
    const int px_per_img = q.TH * q.TW;
    const long long out_elems = (long long)q.Hout * q.Wout * q.Cout;
    const TileDiv div_rpi(q.RPI), div_tw(q.TW);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int ch = g * 32 + n;
        const float bias = q.bias[g * 32 + n];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            const int img = div_rpi.div(row), rem = row - img * q.RPI;
            const int y = div_tw.div(rem), x = rem - y * q.TW;
            const int b = bimg0 + img;
            if (!(img < q.NI && rem < px_per_img && b < q.B && y < q.Hout && x < q.Wout && ch < q.Cout)) continue;
            float v = acc2[g][r] + bias;
            if (q.sbias) v += q.sbias[(long long)b * q.sbias_ld + ch];
            v = fmaxf(v, 0.f);
            (q.out + (long long)b * out_elems)[(unsigned)((y * q.Wout + x) * q.Cout + ch)] = v;
        }
    }
}

// One workgroup tile.  (bx, by, bz) = (row tile, channel group, K split); smem = the workgroup's
// dynamic LDS (conv_lds_bytes).  Called by the per-layer kernel below and, item by item, by the
// persistent rollout kernel (vf_persistent.h).
// MREP = MFMA row blocks (of 32 GEMM rows) per wave: the workgroup covers 4 * MREP * 32 rows.
// PT = ConvParams (kernel argument) or ConvParams in the constant address space (persistent kernel).
// RB < 4 (conv-LSTM, B through LDS only): the workgroup covers RB row blocks of 32 - 64 or 32 rows instead of
// 128 - and wave w takes row block w % RB and RB of the four gates, for batches so small that the per-sample
// dependency chain, not the throughput, bounds a rollout; same chunking and K order, i.e. the same bits.
// epilogue of EPI_CONVT_FUSED, defined in vf_fused_top.h (it needs the compositing code)
template <int ND, bool FIRST, int K, class PT>
__device__ __forceinline__ void convt_fused_epilogue(const PT &p, f32x16 (&acc)[1][4], int bx, long long *red, float *smem);

template <int G, int EPI, int MREP, class PT, int RB = 4>
__device__ __forceinline__ void conv_tile(const PT &p, const int bx_, const int by_, const int bz_,
                                          float *smem) {
    // the item coordinates are wave-uniform but reach an out-of-line tile body in vector registers: made scalar,
    // the tile origin, sample index and every bound derived from them are computed on the scalar unit (VALU
    // instructions do not overlap with the issuing wave's MFMAs - tools/ubench/mfma_shadow.hip - so they are not free)
    const int bx = __builtin_amdgcn_readfirstlane(bx_), by = __builtin_amdgcn_readfirstlane(by_),
              bz = __builtin_amdgcn_readfirstlane(bz_);
    constexpr bool SPLIT = RB == 1 || RB == 2;
    [[maybe_unused]] constexpr bool kInLaunch = !std::is_same<PT, ConvParams>::value;     // tile of the persistent rollout
    // gate-split conv-LSTM tiles: wave w = gate w of ALL row blocks of the workgroup - RB 0: 4 * MREP row blocks (128 /
    // 256 rows), RB -2: two row blocks (64 rows, the plan of narrow phases)
    constexpr bool GSPLIT = RB <= 0;
    static_assert(RB == 4 || (G == 4 && EPI == EPI_LSTM && ((SPLIT && MREP == 1) || RB == 0 || (RB == -2 && MREP == 1))),
                  "the row-split and gate-split tiles are conv-LSTM tiles");
    constexpr int MR = RB == 0 ? 4 * MREP : (RB < 0 ? -RB : MREP);  // MFMA row blocks (accumulator tiles along the rows) per wave
    constexpr int WROWS = MR * 32;      // GEMM rows per wave
    constexpr int GA = GSPLIT ? 1 : (SPLIT ? RB : G);   // gates (accumulator tiles along the columns) per wave
    // Where the weight operand B comes from:
    //  * 128- and 64-row conv-LSTM tiles: through LDS - wave w fetches gate w's slice one tap ahead, all four waves
    //    read all four gates, one barrier per tap (a quarter of the L2 loads of the direct path);
    //  * 32-row conv-LSTM tile (RB 1): every wave multiplies only ITS gate's slice, so LDS staging shares nothing
    //    and costs a barrier per tap, while a tap's 16 MFMAs are too short to cover the L2 latency of a one-tap
    //    look-ahead.  B goes straight from L2 into a register ring of one kernel ROW (5 taps x K8 float4): every
    //    load is issued five taps before its use, no barrier inside a chunk, same K order (same bits).  (For the
    //    64-row tile the ring measured 1-3 % slower than LDS: profiles/r03_tile_plan_sweep.txt.)
    //  * 256-row conv-LSTM tile: straight from L2 with a one-step look-ahead (its input tile needs the LDS, and it
    //    must keep the 32-channel chunks of the other plans so that every plan accumulates in the same K order);
    //  * light layers: straight from L2 through a ring of 4 (5) K steps, see kGRing below.
    //  * gate-split 128-row conv-LSTM tile (RB 0): wave w multiplies gate w's slice with all 128 rows, so nothing
    //    is shared and a tap is long (64 MFMAs): the slice of the NEXT tap goes straight from L2 into registers, no
    //    LDS staging, no barrier and no VALU instruction inside a kernel row (see the K loop).
    constexpr bool kBRing = SPLIT && RB == 1;
    constexpr bool kBLds = (EPI == EPI_LSTM) && MREP == 1 && !kBRing && !GSPLIT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, kh = lane >> 5;
    const int wrow0 = GSPLIT ? 0 : (SPLIT ? (wave % RB) * 32 : wave * WROWS);     // first GEMM row of this wave
    const int gbase = GSPLIT ? wave : (SPLIT ? (wave / RB) * GA : 0);            // first gate of this wave
    const int KC = p.KC, KCpad = KC + 4, K8 = KC >> 3;
    const int LH = (p.TH - 1) * p.stride + p.KH, LW = (p.TW - 1) * p.stride + p.KW;
    const int tile_px = LH * LW;
    const int tile_floats = p.NI * tile_px * KCpad;
    float *lnTab = smem + tile_floats;                       // [2][NI][2]: mean, rstd
    long long *red = reinterpret_cast<long long *>(lnTab + 4 * p.NI);   // [4 waves][2]
    const int cg = by;
    const int tiles_per_img = p.tilesY * p.tilesX;

    int bimg0, ty0, tx0, tile_id;
    if (p.NI == 1) {
        bimg0 = bx / tiles_per_img;
        tile_id = bx % tiles_per_img;
        ty0 = (tile_id / p.tilesX) * p.TH;
        tx0 = (tile_id % p.tilesX) * p.TW;
    } else {
        bimg0 = bx * p.NI;
        tile_id = 0; ty0 = 0; tx0 = 0;
    }

    [[maybe_unused]] const unsigned long long ts0 = VF_TS_NOW();
    [[maybe_unused]] unsigned long long ts1 = 0, ts_stage = 0;
    [[maybe_unused]] int ts_key = (((p.seg[0].C + (p.nseg > 1 ? p.seg[1].C : 0)) >> 5) & 7) + (p.Hout >= 32 ? 0 : 8);
    // light layers: 16 enc0 / enc00, 17 enc3, 18 enc1, 19 enc2, 20 convt1, 21 convt2 (two inputs), 22 FC, 23 fused top, 24 unfused top
    if constexpr (EPI == EPI_RAW_STATS) ts_key = 16;
    else if constexpr (EPI == EPI_BIAS_RELU) ts_key = p.KH == 1 ? 17 : (p.seg[0].C == 32 ? 18 : 19);
    else if constexpr (EPI == EPI_CONVT_RELU) ts_key = p.nseg == 1 ? 20 : 21;
    else if constexpr (EPI == EPI_PARTIAL) ts_key = 22;
    else if constexpr (is_top_fused(EPI)) ts_key = 23;
    else if constexpr (EPI == EPI_CONVT_RAW_STATS) ts_key = 24;
    // ---- LayerNorm statistics of the producing layers (this workgroup's samples only); an early-started conv-LSTM
    // item reads them only once the producer of its layer input is known to be done (chunk loop below)
    const bool late = p.late_cnt != nullptr;
    ln_table(p, bimg0, lnTab, 0, late ? 1 : 2);
    // (yielding, above: the recurrent chunks of an early-started conv-LSTM item step aside for chain-critical work of the
    // CU's other workgroup - once per kernel row, bounded per item; the small-shard tiles only)
    [[maybe_unused]] const bool yielding = kInLaunch && EPI == EPI_LSTM && (GSPLIT || (SPLIT && RB == 1)) && late &&
                                           p.cu_state != nullptr && p.yield_budget > 0;
    [[maybe_unused]] int ybudget = p.yield_budget;
    [[maybe_unused]] const int *yword = yielding ? cu_partner_word(p) : nullptr;

    // ---- this lane's A rows (GEMM rows wave*WROWS + m*32 + n)
    const int px_per_img = p.TH * p.TW;
    int abase[MR];
    {
        const TileDiv div_rpi(p.RPI), div_tw(p.TW);     // (multiply-high instead of four integer divisions per row block)
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const int row = wrow0 + m * 32 + n;
            const int img = div_rpi.div(row), rem = row - img * p.RPI;
            const bool ok = img < p.NI && rem < px_per_img;
            const int y = div_tw.div(rem), x = rem - y * p.TW;
            abase[m] = (ok ? (img * tile_px + y * p.stride * LW + x * p.stride) * KCpad : 0) + kh * 4;
        }
    }

    f32x16 acc[MR][GA];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int g = 0; g < GA; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][g][r] = 0.f;

    const int ntaps = p.KH * p.KW;
    const int Ntot = p.ncg * G * 32;
    const int total_chunks = p.seg[0].nchunk + (p.nseg > 1 ? p.seg[1].nchunk : 0);
    const int ch_begin = bz * p.chunks_per_split;
    const int ch_end = min(ch_begin + p.chunks_per_split, total_chunks);
    const float *wlane = p.Wp + ((long long)kh * Ntot + (cg * G) * 32 + n) * 4;
    const long long wstep = (long long)2 * Ntot * 4;        // floats per (chunk, tap, k8) block

    const int q4 = KC >> 2;
    const int q4_log2 = 31 - __builtin_clz((unsigned)q4);
    const unsigned magic_px = 0xFFFFFFFFu / (unsigned)tile_px + 1u, magic_lw = 0xFFFFFFFFu / (unsigned)LW + 1u;
    const int items = p.NI * tile_px * q4;

    // ---- G == 4 (conv-LSTM, transposed conv): the B operand goes through LDS.  Per tap the
    // workgroup needs K8 blocks of [4 gates][2 k-halves][32 columns] float4; wave w fetches only
    // gate w's slice (a quarter of the global loads of the direct path, whose 4 waves each pull
    // the whole block through L1) one tap ahead, parks it in registers during the tap's MFMAs,
    // then writes it to the other LDS buffer; one barrier per tap.
    // conv-LSTM tiles: LayerNorm gain and offset of every input channel, [gamma: gbC][beta: gbC] (recurrent
    // segment first), filled once per item - the staging loop below reads a thread's channel quad with two
    // ds_read_b128 per chunk instead of a modulo and two global loads per element and channel
    constexpr bool kGbTab = EPI == EPI_LSTM;
    const int gbC = kGbTab ? ((p.seg[0].C + (p.nseg > 1 ? p.seg[1].C : 0) + 3) & ~3) : 0;
    float *gbTab = lnTab + 4 * p.NI + 16;
    if constexpr (kGbTab) {
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
    }
    f32x4 *bsm = reinterpret_cast<f32x4 *>(gbTab + 2 * gbC);    // [2][K8][4][64]
    const float *wgate = p.Wp + ((long long)kh * Ntot + (cg * G + (wave & (G - 1))) * 32 + n) * 4;
    const int gt0 = ch_begin * ntaps, gtN = ch_end * ntaps;
    f32x4 breg[4];
#define VF_LOADB(GT_)                                                                           \
    _Pragma("unroll") for (int q = 0; q < 4; ++q)                                               \
        if (q < K8) breg[q] = *reinterpret_cast<const f32x4 *>(wgate + ((long long)(GT_) * K8 + q) * wstep);
#define VF_WRITEB(BUF_)                                                                         \
    _Pragma("unroll") for (int q = 0; q < 4; ++q)                                               \
        if (q < K8) bsm[(((BUF_) * K8 + q) * 4 + wave) * 64 + lane] = breg[q];
    if constexpr (kBLds) {
        if (gt0 < gtN) { VF_LOADB(gt0) }
    }
    // gate-split tile: two register sets for the weight slice of the current / the next tap, filled by raw buffer loads
    // (lane offset in one VGPR, (tap, k8) offset in an SGPR); the first tap's slice is requested here and lands during
    // the prologue and the first staging
    [[maybe_unused]] f32x4 gsA[GSPLIT ? 4 : 1], gsB[GSPLIT ? 4 : 1];
    [[maybe_unused]] int gs_par = 0;
    [[maybe_unused]] const unsigned gs_loff = (unsigned)(((kh * Ntot + (cg * G + wave) * 32 + n) * 4) * 4);
    [[maybe_unused]] const unsigned gs_wstep_b = (unsigned)wstep * 4u;
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t gs_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.Wp), 0, 0x7FFFFFFF, 0x00020000);
    if constexpr (GSPLIT) {
        if (gt0 < gtN) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                gsA[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    gs_rsrc, gs_loff, (unsigned)(gt0 * 4 + q) * gs_wstep_b, 0));
        }
    }
    constexpr int kRing = 5;                // taps in flight = one row of the 5x5 kernel
    [[maybe_unused]] f32x4 bring[kBRing ? kRing : 1][kBRing ? 4 : 1][GA];
    [[maybe_unused]] const float *wring = p.Wp + ((long long)kh * Ntot + (cg * G + gbase) * 32 + n) * 4;
#define VF_LOADRING(SLOT_, GT_)                                                                 \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_)                                            \
        _Pragma("unroll") for (int g_ = 0; g_ < GA; ++g_)                                       \
            bring[SLOT_][q_][g_] = *reinterpret_cast<const f32x4 *>(                            \
                wring + ((long long)(GT_) * 4 + q_) * wstep + g_ * 128);
    if constexpr (kBRing) {
        // (the split plans exist for 5x5 kernels with 32-channel chunks only: half_ok / quarter_ok in vf_engine.hip)
#pragma unroll
        for (int d = 0; d < kRing; ++d)
            if (gt0 + d < gtN) { VF_LOADRING(d, gt0 + d) }
    }

    // Light layers (everything but the conv-LSTM tiles): a K step is 4 - 9 MFMAs (0.1 - 0.25 us) while the weight
    // operand comes straight from L2 (~1 us away), so with the one-step look-ahead of the pipelined loop below every
    // step waits for its own loads.  Their B operand therefore travels through a register ring of D steps (D = 4, or
    // 5 for the 5x5 layers with 8- / 16-channel chunks; D divides the steps of a chunk, so slots are static) that is
    // refilled D steps ahead - across chunk boundaries: the weights of the next chunk do not depend on its staging.
    constexpr bool kGRing = EPI != EPI_LSTM;
    constexpr bool kConvT = (EPI == EPI_CONVT_RELU || EPI == EPI_CONVT_RAW_STATS || is_top_fused(EPI));
    [[maybe_unused]] f32x4 gring[kGRing ? (G == 1 ? 5 : 4) : 1][G];
    [[maybe_unused]] const int nit_g = ntaps * K8;
    [[maybe_unused]] const int ring_d = (nit_g & 3) == 0 ? 4 : ((G == 1 && nit_g % 5 == 0) ? 5 : 0);
    // live output parities of transposed-conv tap `tap_` (bit g = parity g): see the pipelined loop below
#define VF_TAPLIVE(TAP_) (kConvT ? ((((TAP_) >= p.KW) ? 15 : 3) & ((((TAP_) % p.KW) != 0) ? 15 : 5)) : 15)
    // ring slot SLOT_ <- B of step IT_ of chunk CI_ (a transposed conv fetches its dead parity blocks - zeros - too:
    // unconditional loads keep the ring in plain registers; only the MFMAs on them are skipped)
    // (raw buffer loads: the lane's offset in one VGPR, the (chunk, step, gate) offset on the scalar unit - no 64-bit
    // vector address arithmetic inside the K loops of the light layers either)
    [[maybe_unused]] const unsigned gr_loff = (unsigned)(((kh * Ntot + (cg * G) * 32 + n) * 4) * 4);
    [[maybe_unused]] const unsigned gr_wstep_b = (unsigned)wstep * 4u;
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t gr_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.Wp), 0, 0x7FFFFFFF, 0x00020000);
#define VF_GLOAD(SLOT_, CI_, IT_)                                                               \
    {                                                                                           \
        const unsigned so_ = (unsigned)((CI_) * nit_g + (IT_)) * gr_wstep_b;                    \
        _Pragma("unroll") for (int g_ = 0; g_ < G; ++g_)                                        \
            gring[SLOT_][g_] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128( \
                gr_rsrc, gr_loff, so_ + 512u * g_, 0));                                          \
    }
    if constexpr (kGRing) {
        if (ch_begin < ch_end) {
            if (ring_d == 4) {
#pragma unroll
                for (int d = 0; d < 4; ++d) VF_GLOAD(d, ch_begin, d)
            } else if (G == 1 && ring_d == 5) {
#pragma unroll
                for (int d = 0; d < 5; ++d) VF_GLOAD(d, ch_begin, d)
            } else if (G == 2) {            // (EPI_CONV_PAIR on 16-channel chunks: 18 steps per chunk, rings of 2)
#pragma unroll
                for (int d = 0; d < 2; ++d) VF_GLOAD(d, ch_begin, d)
            }
        }
    }

    // Wave priority: everything that is NOT a conv-LSTM K loop - the light tiles, the prologues and epilogues - is a
    // latency chain that others wait for; it runs at priority 2 (set by the persistent kernel) and the long matrix
    // loops step down to 0, so a co-resident light item or epilogue gets its few instructions issued first.
    if constexpr (EPI == EPI_LSTM) __builtin_amdgcn_s_setprio(0);
    constexpr int kStageU = 4;      // staging runs in batches of kStageU elements per thread (below)
    for (int ci = ch_begin; ci < ch_end; ++ci) {
        const int s = (ci < p.seg[0].nchunk) ? 0 : 1;
        const auto &sg = p.seg[s];
        const int c0 = (s == 0 ? ci : ci - p.seg[0].nchunk) * KC;
        const bool vec_ok = (sg.C & 3) == 0;

        if (late && ci == p.seg[0].nchunk) {         // the chunks of the early input are done: now the late one is needed
            const int b1 = p.NI == 1 ? bimg0 + 1 : min(bimg0 + p.NI, p.B);
            if constexpr (kInLaunch) VF_TRACE_EVT(TR_LATE);
            if (!late_wait(p, bimg0, b1, reinterpret_cast<int *>(red))) return;
            if constexpr (kInLaunch) VF_TRACE_EVT(TR_LATE_END);
            ln_table(p, bimg0, lnTab, 1, 2);
        }
        __syncthreads();        // previous chunk fully consumed (and lnTab visible on entry)
        if constexpr (kInLaunch) VF_TRACE_EVT(TR_STAGE);
        [[maybe_unused]] const unsigned long long ts_s0 = VF_TS_NOW();
        // kConvThreads is a multiple of q4, so a thread stages the same channel quad of every pixel it visits: its
        // channel range and LayerNorm gain / offset are loop invariants.  The light layers load them once per chunk
        // (no per-element modulo and loads: -1.5 % at the C5 shard); in the conv-LSTM tiles the up-front loads cost
        // more than they save (+1.2 % at C2, measured), so those keep fetching them per element.
        constexpr bool kHoistLn = EPI != EPI_LSTM || MREP > 1;
        const int q = tid & (q4 - 1);
        const int c = c0 + 4 * q;
        const int nvalid = min(4, sg.C - c);
        const bool lean = vec_ok && sg.C % KC == 0;     // whole channel quads in whole chunks: the lean loop below
        [[maybe_unused]] float gam[4] = {1.f, 1.f, 1.f, 1.f}, bet[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (kHoistLn) {
            if (sg.ln_part && c < sg.C && !(kGbTab && lean)) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cc = (c + j) % sg.gamma_mod;
                    gam[j] = sg.gamma[cc]; bet[j] = sg.beta[cc];
                }
            }
        }
        // Staging runs in batches of kStageU elements per thread: the geometry of all of them first, then all their
        // global loads back to back, then LayerNorm / relu / the LDS stores - one exposed load latency per batch
        // instead of one per element (a light layer's item is mostly this chain: 17 elements per thread in enc1,
        // 64 over the eight chunks of an FC item).
        // (all ~9 loads of a thread in flight at once, or the element geometry computed once per item instead of per
        // chunk: measured in the gate-split tile, no gain - the extra live registers spill)
        // Segments with whole channel quads in whole chunks (every layer but the 3-channel frame input): the lean loop.  A starved
        // wave - the other workgroup of the CU is in its K loop - gets about one VALU instruction issued per 40
        // cycles (tools/ubench/mfma_shadow.hip), so the staging time IS its instruction count: ~30 per element here
        // (pixel -> row / column by one multiply-high, unsigned bounds checks, 32-bit in-image offset on a scalar or
        // per-image base, gain / offset from the LDS table) against ~110 in the general loop below.  Same values,
        // same expressions on them: the same bits.
        bool staged = false;
        {
            if (lean) {
                staged = true;
                auto stage_fast = [&](auto ni1c) {
                    constexpr bool NI1 = decltype(ni1c)::value;
                    const int pl = tid >> q4_log2, ppp = kConvThreads >> q4_log2;       // pixel lane, pixels per pass
                    const bool has_ln = sg.ln_part != nullptr;
                    f32x4 gq = {1.f, 1.f, 1.f, 1.f}, bq = {0.f, 0.f, 0.f, 0.f};
                    if (has_ln) {
                        if constexpr (kGbTab) {
                            const int gi = (s == 0 ? 0 : p.seg[0].C) + c;
                            gq = *reinterpret_cast<const f32x4 *>(gbTab + gi);
                            bq = *reinterpret_cast<const f32x4 *>(gbTab + gbC + gi);
                        } else {
                            gq = f32x4{gam[0], gam[1], gam[2], gam[3]};
                            bq = f32x4{bet[0], bet[1], bet[2], bet[3]};
                        }
                    }
                    const int y0 = ty0 * p.stride - p.pad, x0 = tx0 * p.stride - p.pad;
                    const int npix = p.NI * tile_px;
                    const unsigned Cs = (unsigned)sg.C;
                    float mean1 = 0.f, rstd1 = 1.f;
                    if (NI1 && has_ln) { mean1 = lnTab[2 * s * p.NI]; rstd1 = lnTab[2 * s * p.NI + 1]; }
                    // The images of this workgroup as ONE raw buffer (base and size are wave-uniform): a load whose
                    // offset lies beyond it returns zeros, so padding pixels and samples past the batch cost one
                    // select of the offset - no branch, no zero-initialised registers
                    const unsigned img_bytes = (unsigned)(p.Hin * p.Win) * Cs * 4u;
                    const unsigned img_step = (unsigned)sg.bstride * 4u;        // 0: one image broadcast to every sample
                    const int n_here = min(p.NI, p.B - bimg0);                   // images of this workgroup inside the batch
                    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<float *>(sg.ptr + (long long)bimg0 * sg.bstride), 0,
                        n_here > 0 ? (int)((unsigned)(n_here - 1) * img_step + img_bytes) : 0, 0x00020000);
                    float *dst = smem + 4 * q;
                    for (int pix0 = pl; pix0 < npix; pix0 += ppp * kStageU) {
                        f32x4 v[kStageU];
                        bool oks[kStageU];
                        [[maybe_unused]] int imgs[kStageU];
#pragma unroll
                        for (int u = 0; u < kStageU; ++u) {
                            const int pix = pix0 + u * ppp;
                            int img = 0, r = pix;
                            if constexpr (!NI1) {
                                img = tile_px == 1 ? pix : (int)__umulhi((unsigned)pix, magic_px);
                                r = pix - img * tile_px;
                            }
                            const int ly = LW == 1 ? r : (int)__umulhi((unsigned)r, magic_lw), lx = r - ly * LW;
                            const int iy = y0 + ly, ix = x0 + lx;
                            const bool ok = pix < npix && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win &&
                                            img < n_here;
                            imgs[u] = img; oks[u] = ok;
                            unsigned off = ((unsigned)(iy * p.Win + ix) * Cs + (unsigned)c) * 4u;
                            if constexpr (!NI1) off += (unsigned)img * img_step;
                            v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? off : 0xFFFFFFFFu, 0, 0));
                        }
                        if constexpr (kInLaunch) { if (pix0 == pl) VF_TRACE_EVT(TR_ST_LOADED); }
#pragma unroll
                        for (int u = 0; u < kStageU; ++u) {
                            const int pix = pix0 + u * ppp;
                            if (pix >= npix) break;
                            if (oks[u]) {
                                if (has_ln) {
                                    float mean = mean1, rstd = rstd1;
                                    if constexpr (!NI1) {
                                        mean = lnTab[2 * (s * p.NI + imgs[u])];
                                        rstd = lnTab[2 * (s * p.NI + imgs[u]) + 1];
                                    }
#pragma unroll
                                    for (int j = 0; j < 4; ++j) v[u][j] = fmaf((v[u][j] - mean) * rstd, gq[j], bq[j]);
                                }
                                if (sg.relu) {
#pragma unroll
                                    for (int j = 0; j < 4; ++j) v[u][j] = fmaxf(v[u][j], 0.f);
                                }
                            }
                            *reinterpret_cast<f32x4 *>(dst + pix * KCpad) = v[u];
                        }
                    }
                };
                if (p.NI == 1) stage_fast(std::true_type{}); else stage_fast(std::false_type{});
            }
        }
        for (int it0 = tid; !staged && it0 < items; it0 += kConvThreads * kStageU) {
            // q4 is a power of two; tile_px and LW divide through a multiply-high (exact for every index a tile
            // can hold: checked exhaustively for dividends < 70000, divisors 2..600; divisor 1 - the FC - bypasses it)
            f32x4 v[kStageU];
            int pixs[kStageU], imgs[kStageU];
            bool oks[kStageU];
#pragma unroll
            for (int u = 0; u < kStageU; ++u) {
                const int it = it0 + u * kConvThreads;
                const int pix = it >> q4_log2;
                const int img = tile_px == 1 ? pix : (int)__umulhi((unsigned)pix, magic_px), r = pix - img * tile_px;
                const int ly = LW == 1 ? r : (int)__umulhi((unsigned)r, magic_lw), lx = r - ly * LW;
                const int iy = ty0 * p.stride - p.pad + ly, ix = tx0 * p.stride - p.pad + lx;
                const int b = bimg0 + img;
                pixs[u] = pix; imgs[u] = img;
                oks[u] = it < items && b < p.B && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win && c < sg.C;
                v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (oks[u]) {
                    const float *src = sg.ptr + (long long)b * sg.bstride +
                                       ((long long)iy * p.Win + ix) * sg.C + c;
                    if (vec_ok) {
                        v[u] = *reinterpret_cast<const f32x4 *>(src);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[u][j] = (j < nvalid) ? src[j] : 0.f;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < kStageU; ++u) {
                if (it0 + u * kConvThreads >= items) break;
                if (oks[u]) {
                    if (sg.ln_part) {
                        const float mean = lnTab[2 * (s * p.NI + imgs[u])];
                        const float rstd = lnTab[2 * (s * p.NI + imgs[u]) + 1];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if constexpr (kHoistLn) {
                                v[u][j] = fmaf((v[u][j] - mean) * rstd, gam[j], bet[j]);
                            } else {
                                const int cc = (c + j) % sg.gamma_mod;
                                v[u][j] = fmaf((v[u][j] - mean) * rstd, sg.gamma[cc], sg.beta[cc]);
                            }
                        }
                    }
                    if (sg.relu) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[u][j] = fmaxf(v[u][j], 0.f);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[u][j] = (j < nvalid) ? v[u][j] : 0.f;
                }
                *reinterpret_cast<f32x4 *>(&smem[pixs[u] * KCpad + 4 * q]) = v[u];
            }
        }
        if constexpr (kBLds) {
            if (ci == ch_begin) { VF_WRITEB(0) }
        }
        if constexpr (kInLaunch) VF_TRACE_EVT(TR_ST_WRITTEN);
        __syncthreads();
#ifdef VF_TILE_STATS
        if (ci == ch_begin) ts1 = VF_TS_NOW(); else ts_stage += VF_TS_NOW() - ts_s0;
#endif
        if constexpr (kInLaunch) {
            if (ci == ch_begin) VF_TRACE_EVT(TR_MFMAS, (unsigned long long)(ntaps * K8 * 4 * GA * MR));
            VF_TRACE_EVT(TR_KLOOP);
        }

        const f32x4 *smem4 = reinterpret_cast<const f32x4 *>(smem);
        int ab4[MR];                                                // in float4 units
#pragma unroll
        for (int m = 0; m < MR; ++m) ab4[m] = abase[m] >> 2;
        const int kcp4 = KCpad >> 2;
        [[maybe_unused]] f32x4 aP[GSPLIT ? 1 : MR], aQ[GSPLIT ? 1 : MR];
        [[maybe_unused]] f32x4 bP[GA], bQ[GA];
#define VF_MFMA(A_, B_)                                                                         \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                         \
            _Pragma("unroll") for (int g = 0; g < GA; ++g) {                                    \
                _Pragma("unroll") for (int m = 0; m < (GSPLIT ? 1 : MR); ++m)                   \
                    acc[m][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(A_[m][j], B_[g][j], acc[m][g], 0, 0, 0); \
            }                                                                                   \
        }

        if constexpr (GSPLIT) {
            // ---- K loop of the gate-split tile (5x5 kernel, 32-channel chunks): one kernel ROW = 5 taps = 20 k8 steps
            // of 16 MFMAs, statically unrolled.  Per step the wave reads its four A row blocks (one ds_read_b128 each,
            // immediate offsets off four row pointers that are set up once per kernel row), one step ahead of the
            // MFMAs that use them and across tap boundaries; per tap it issues the four buffer loads of the NEXT tap's
            // weight slice into the idle one of two register sets (a tap is 64 MFMAs = 4096 cycles, several times the
            // L2 latency).  No barrier, no LDS store, no VALU instruction inside a row: what a K loop costs the matrix
            // pipe besides its MFMAs is 4 VMEM + 16 LDS issues per tap.  Same (chunk, tap, k8, j) order per output as
            // every other plan: the same bits.
            // (MREP 2: eight row blocks per wave, worked through in two groups of four per k8 step - a "substep" is one
            // group's 16 MFMAs; the operands of substep s + 1 are fetched before the MFMAs of substep s are issued)
            constexpr int GSZ = MR < 4 ? MR : 4;            // row blocks per group (the 64-row tile has one group of two)
            constexpr int NG = MR / GSZ, NS = 4 * NG;       // row-block groups, substeps per tap
            const f32x4 *ar[MR];
            f32x4 aP4[GSZ], aQ4[GSZ];
            auto gs_fetch = [&](f32x4 (&A_)[GSZ], const int kx, const int sub) {
#pragma unroll
                for (int m_ = 0; m_ < GSZ; ++m_) A_[m_] = ar[(sub % NG) * GSZ + m_][kx * 9 + (sub / NG) * 2];
            };
            auto gs_loadb = [&](f32x4 (&D_)[4], const int gt) {
                // (unconditional: behind the item's last tap the last slice is simply fetched again - a branch around
                // the loads makes the compiler copy the register set on the path that skips them)
                const unsigned so_ = (unsigned)(min(gt, gtN - 1) * 4) * gs_wstep_b;
#pragma unroll
                for (int q_ = 0; q_ < 4; ++q_)
                    D_[q_] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        gs_rsrc, gs_loff, so_ + (unsigned)q_ * gs_wstep_b, 0));
            };
            // one tap: the operands of its substep 0 are already in aP; cur = this tap's weight slice, nxt <- the next tap's
            auto gs_tap = [&](auto kxc, f32x4 (&cur)[4], f32x4 (&nxt)[4], const int gt_next) {
                constexpr int KX = decltype(kxc)::value;
                static_for<NS>([&](auto sc) {
                    constexpr int S = decltype(sc)::value;
                    f32x4 (&a_cur)[GSZ] = (S & 1) ? aQ4 : aP4;
                    f32x4 (&a_nxt)[GSZ] = (S & 1) ? aP4 : aQ4;
                    if constexpr (S + 1 < NS) gs_fetch(a_nxt, KX, S + 1);
                    else if constexpr (KX < 4) gs_fetch(a_nxt, KX + 1, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (S == 0) {
                        gs_loadb(nxt, gt_next);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const f32x4 bq = cur[S / NG];
#pragma unroll
                    for (int j_ = 0; j_ < 4; ++j_) {
#pragma unroll
                        for (int m_ = 0; m_ < GSZ; ++m_)
                            acc[(S % NG) * GSZ + m_][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                a_cur[m_][j_], bq[j_], acc[(S % NG) * GSZ + m_][0], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            };
            const bool ychunk = yielding && ci < p.seg[0].nchunk;
            unsigned yseen = 0;
            auto gs_row = [&](f32x4 (&X_)[4], f32x4 (&Y_)[4], const int gt_row) {
                gs_fetch(aP4, 0, 0);
                gs_tap(std::integral_constant<int, 0>{}, X_, Y_, gt_row + 1);
                gs_tap(std::integral_constant<int, 1>{}, Y_, X_, gt_row + 2);
                gs_tap(std::integral_constant<int, 2>{}, X_, Y_, gt_row + 3);
                gs_tap(std::integral_constant<int, 3>{}, Y_, X_, gt_row + 4);
                gs_tap(std::integral_constant<int, 4>{}, X_, Y_, gt_row + 5);
            };
            for (int ky = 0; ky < 5; ++ky) {
                const int gt_row = ci * 25 + ky * 5;
                if (ychunk) yseen = yield_peek_issue(yword);
#pragma unroll
                for (int m = 0; m < MR; ++m) ar[m] = smem4 + ab4[m] + ky * LW * 9;
                if (gs_par == 0) gs_row(gsA, gsB, gt_row); else gs_row(gsB, gsA, gt_row);
                gs_par ^= 1;                // five taps: the slice of the next row's first tap sits in the other set
                if (ychunk) yield_to_partner(yword, yseen, ybudget);
            }
        } else if constexpr (kBLds) {
          if (K8 == 4) {
            // ---- K loop, B through LDS, 32-channel chunks (every conv-LSTM plan of vf_engine.hip): the k8 steps are
            // unrolled with immediate LDS offsets and the next tap's weight slice comes through a raw buffer load
            // (lane offset in one VGPR, tap offset in an SGPR), so a tap costs three VALU instructions besides its 64
            // MFMAs.  A wave's own VALU / VMEM instructions do not overlap with its MFMAs (each costs the matrix pipe
            // 17 / 37 cycles when the wave has the SIMD to itself, tools/ubench/mfma_shadow.hip): the generic loop
            // below spends ~20 VALU instructions per tap on 64-bit addresses and K8-dependent selects, which is what
            // held a K loop whose partner workgroup is outside its own to 0.78 of the pipe (tools/trace_cu.py).
            // Same (chunk, tap, k8, j) order: the same bits.
            const unsigned wl_off = (unsigned)(((kh * Ntot + (cg * G + (wave & (G - 1))) * 32 + n) * 4) * 4);
            const unsigned wstep_b = (unsigned)wstep * 4u;                      // bytes per (tap, k8) block
            const __amdgpu_buffer_rsrc_t wrsrc =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.Wp), 0, 0x7FFFFFFF, 0x00020000);
            const f32x4 *a_lane = smem4 + ab4[0];
            const f32x4 *b_rd0 = bsm + gbase * 64 + lane;
            f32x4 *b_wr0 = bsm + wave * 64 + lane;
            int ky = 0, kx = 0;
            for (int tap = 0; tap < ntaps; ++tap) {
                const int gt = ci * ntaps + tap;
                const int buf = (gt - gt0) & 1;
                const bool more = gt + 1 < gtN;
                const f32x4 *ap = a_lane + (ky * LW + kx) * kcp4;
                const f32x4 *bp = b_rd0 + buf * (4 * 4 * 64);
#define VF_FETCH_4(A_, B_, Q_)                                                                  \
                {                                                                               \
                    A_[0] = ap[(Q_) * 2];                                                       \
                    _Pragma("unroll") for (int g = 0; g < GA; ++g) B_[g] = bp[((Q_) * 4 + g) * 64]; \
                }
                // (sched_barrier: the operands of step q + 1 are fetched BEFORE the MFMAs of step q are issued - left
                // to itself the scheduler sinks every fetch behind the previous step's MFMAs to save registers and
                // exposes the LDS latency four times per tap)
                VF_FETCH_4(aP, bP, 0)
                VF_FETCH_4(aQ, bQ, 1)
                __builtin_amdgcn_sched_barrier(0);
                // (the weight loads of the next tap are issued behind the first operand fetches: any wait the compiler
                // attaches to those fetches then never covers loads that were only just issued)
                if (more) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        breg[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                            wrsrc, wl_off, (unsigned)((gt + 1) * 4 + q) * wstep_b, 0));
                }
                __builtin_amdgcn_sched_barrier(0);
                VF_MFMA(aP, bP)
                __builtin_amdgcn_sched_barrier(0);
                VF_FETCH_4(aP, bP, 2)
                __builtin_amdgcn_sched_barrier(0);
                VF_MFMA(aQ, bQ)
                __builtin_amdgcn_sched_barrier(0);
                VF_FETCH_4(aQ, bQ, 3)
                __builtin_amdgcn_sched_barrier(0);
                VF_MFMA(aP, bP)
                VF_MFMA(aQ, bQ)
                __builtin_amdgcn_sched_barrier(0);
#undef VF_FETCH_4
                if (more) {
                    f32x4 *bw = b_wr0 + (buf ^ 1) * (4 * 4 * 64);
#pragma unroll
                    for (int q = 0; q < 4; ++q) bw[q * 4 * 64] = breg[q];
                }
                __syncthreads();
                if (++kx == p.KW) { kx = 0; ++ky; }
            }
          } else {
            // ---- K loop, B through LDS: taps outer (one barrier each), k8 inner (ping-pong)
#define VF_FETCH_L(A_, B_, Q_)                                                                  \
            {                                                                                   \
                _Pragma("unroll") for (int m = 0; m < MREP; ++m)                                \
                    A_[m] = smem4[ab4[m] + ao + (Q_) * 2];                            \
                _Pragma("unroll") for (int g = 0; g < GA; ++g)                                  \
                    B_[g] = bsm[((buf * K8 + (Q_)) * 4 + gbase + g) * 64 + lane];               \
            }
            for (int ky = 0; ky < p.KH; ++ky) {
                for (int kx = 0; kx < p.KW; ++kx) {
                    const int gt = ci * ntaps + ky * p.KW + kx;
                    const int buf = (gt - gt0) & 1;
                    const int ao = (ky * LW + kx) * kcp4;
                    const bool more = gt + 1 < gtN;
                    if (more) { VF_LOADB(gt + 1) }
                    VF_FETCH_L(aP, bP, 0)
                    int q = 0;
                    for (; q + 2 <= K8; q += 2) {
                        VF_FETCH_L(aQ, bQ, q + 1)
                        VF_MFMA(aP, bP)
                        if (q + 2 < K8) VF_FETCH_L(aP, bP, q + 2)
                        VF_MFMA(aQ, bQ)
                    }
                    if (q < K8) VF_MFMA(aP, bP)
                    if (more) { VF_WRITEB(buf ^ 1) }
                    __syncthreads();
                }
            }
#undef VF_FETCH_L
          }
        } else if constexpr (kBRing) {
            // ---- K loop of the row-split tiles: B from the register ring, A double-buffered from LDS, no barrier
            f32x4 aC[4], aN[4];
            const int a0 = ab4[0];
#pragma unroll
            for (int q = 0; q < 4; ++q) aC[q] = smem4[a0 + q * 2];
            const bool ychunk = yielding && ci < p.seg[0].nchunk;
            for (int ky = 0; ky < 5; ++ky) {
                unsigned yseen = 0;
                if (ychunk) yseen = yield_peek_issue(yword);
#pragma unroll
                for (int kx = 0; kx < kRing; ++kx) {
                    const int gt = ci * ntaps + ky * 5 + kx;
                    // A of the next tap of this chunk (the last tap re-reads its own: harmless, no branch)
                    const int kyn = kx == 4 ? min(ky + 1, 4) : ky, kxn = kx == 4 ? (ky == 4 ? 4 : 0) : kx + 1;
                    const int aon = (kyn * LW + kxn) * kcp4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) aN[q] = smem4[a0 + aon + q * 2];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int g = 0; g < GA; ++g)
                                acc[0][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(aC[q][j], bring[kx][q][g][j], acc[0][g], 0, 0, 0);
                    if (gt + kRing < gtN) { VF_LOADRING(kx, gt + kRing) }
#pragma unroll
                    for (int q = 0; q < 4; ++q) aC[q] = aN[q];
                }
                if (ychunk) yield_to_partner(yword, yseen, ybudget);
            }
        } else if (kGRing && (G > 1 || ring_d != 0)) {      // (transposed convs: 4 taps x K8 steps, always ring_d == 4)
            // ---- K loop of the light layers: B from the register ring (refilled ring_d steps ahead), A one step
            // ahead from LDS; same (tap, k8, j) order as the pipelined loop below
            auto kloop = [&](auto dc) {
                constexpr int D = decltype(dc)::value;
                f32x4 aC[MREP], aN[MREP];
                int ky = 0, kx = 0, k8 = 0;         // position of the NEXT A fetch
                auto fetch_a = [&](f32x4 (&A_)[MREP]) {
                    const int ao_ = (ky * LW + kx) * kcp4 + k8 * 2;
#pragma unroll
                    for (int m = 0; m < MREP; ++m) A_[m] = smem4[ab4[m] + ao_];
                    if (++k8 == K8) { k8 = 0; if (++kx == p.KW) { kx = 0; ++ky; } }
                };
                fetch_a(aC);
                for (int it0 = 0; it0 < nit_g; it0 += D) {
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        const int it = it0 + j;
                        if (it + 1 < nit_g) fetch_a(aN);
                        const int live = VF_TAPLIVE(it / K8);
#pragma unroll
                        for (int g = 0; g < G; ++g) {
                            if (g == 0 || (live >> g) & 1) {
#pragma unroll
                                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                                    for (int m = 0; m < MREP; ++m)
                                        acc[m][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(aC[m][jj], gring[j][g][jj], acc[m][g], 0, 0, 0);
                            }
                        }
                        const int nx = it + D;
                        if (nx < nit_g) { VF_GLOAD(j, ci, nx) }
                        else if (ci + 1 < ch_end) { VF_GLOAD(j, ci + 1, nx - nit_g) }
#pragma unroll
                        for (int m = 0; m < MREP; ++m) aC[m] = aN[m];
                    }
                }
            };
            if constexpr (G == 1) {
                if (ring_d == 4) kloop(std::integral_constant<int, 4>{});
                else kloop(std::integral_constant<int, 5>{});
            } else if constexpr (G == 2) {
                if (ring_d == 4) kloop(std::integral_constant<int, 4>{});
                else kloop(std::integral_constant<int, 2>{});       // an even number of steps per chunk (host: pairable)
            } else {
                kloop(std::integral_constant<int, 4>{});
            }
        } else if constexpr (!(kGRing && G > 1)) {
            // ---- K loop over (tap, k8), software pipelined: operands of step it+1 are fetched
            // (A: LDS b128, B: L1/L2 b128) before the MFMAs of step it are issued.
            // Transposed convs: of the 16 (tap, output parity) blocks of the 2x2 view only 9 carry weights -
            // input row tap 0 only reaches even output rows, column tap 0 only even output columns - so a step
            // fetches and multiplies only its tap's live parities (LIVE_: bit g = parity g).  Skipped products are
            // exact zeros: the sums are unchanged.
            const float *wchunk = wlane + (long long)ci * ntaps * K8 * wstep;
            const int nit = ntaps * K8;
            int ky = 0, kx = 0, k8 = 0;
            [[maybe_unused]] int liveP = 15, liveQ = 15;
#define VF_FETCH(A_, B_, LIVE_, IT_)                                                            \
            {                                                                                   \
                const int ao_ = (ky * LW + kx) * kcp4 + k8 * 2;                                 \
                _Pragma("unroll") for (int m = 0; m < MREP; ++m) A_[m] = smem4[ab4[m] + ao_]; \
                const float *wp_ = wchunk + (long long)(IT_) * wstep;                             \
                if constexpr (kConvT) {                                                         \
                    LIVE_ = (ky ? 15 : 3) & (kx ? 15 : 5);                                      \
                    B_[0] = *reinterpret_cast<const f32x4 *>(wp_);                              \
                    if (LIVE_ & 2) B_[1 % G] = *reinterpret_cast<const f32x4 *>(wp_ + 128);     \
                    if (LIVE_ & 4) B_[2 % G] = *reinterpret_cast<const f32x4 *>(wp_ + 256);     \
                    if (LIVE_ & 8) B_[3 % G] = *reinterpret_cast<const f32x4 *>(wp_ + 384);     \
                } else {                                                                        \
                    _Pragma("unroll") for (int g = 0; g < G; ++g)                               \
                        B_[g] = *reinterpret_cast<const f32x4 *>(wp_ + g * 128);                \
                }                                                                               \
                if (++k8 == K8) { k8 = 0; if (++kx == p.KW) { kx = 0; ++ky; } }                 \
            }
#define VF_MFMA_G(A_, B_, G_)                                                                   \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                     \
                _Pragma("unroll") for (int m = 0; m < MREP; ++m)                                \
                    acc[m][(G_) % G] = __builtin_amdgcn_mfma_f32_32x32x2f32(A_[m][j], B_[(G_) % G][j], acc[m][(G_) % G], 0, 0, 0); \
            }
#define VF_MFMA_L(A_, B_, LIVE_)                                                                \
            if constexpr (kConvT) {                                                             \
                VF_MFMA_G(A_, B_, 0)                                                            \
                if (LIVE_ & 2) { VF_MFMA_G(A_, B_, 1) }                                         \
                if (LIVE_ & 4) { VF_MFMA_G(A_, B_, 2) }                                         \
                if (LIVE_ & 8) { VF_MFMA_G(A_, B_, 3) }                                         \
            } else {                                                                            \
                VF_MFMA(A_, B_)                                                                 \
            }
            VF_FETCH(aP, bP, liveP, 0)
            int it = 0;
            for (; it + 2 <= nit; it += 2) {
                VF_FETCH(aQ, bQ, liveQ, it + 1)
                VF_MFMA_L(aP, bP, liveP)
                if (it + 2 < nit) VF_FETCH(aP, bP, liveP, it + 2)
                VF_MFMA_L(aQ, bQ, liveQ)
            }
            if (it < nit) { VF_MFMA_L(aP, bP, liveP) }
#undef VF_FETCH
#undef VF_MFMA_G
#undef VF_MFMA_L
        }
#undef VF_MFMA
    }
#undef VF_LOADB
#undef VF_WRITEB
#undef VF_LOADRING
#undef VF_GLOAD
#undef VF_TAPLIVE

    if constexpr (EPI == EPI_LSTM) __builtin_amdgcn_s_setprio(2);
    if constexpr (kInLaunch) VF_TRACE_EVT(TR_EPI);
    [[maybe_unused]] const unsigned long long ts2 = VF_TS_NOW();
    // (xch = the double-buffered B area: 32 KiB at 32-channel chunks, disjoint from lnTab / red)
    if constexpr (GSPLIT) lstm_gsplit_epilogue<MR>(p, acc, bx, by, smem);
    // (the 32-row tile: wave w = gate w of its one row block - the gate-split exchange with one row block per round, and
    // with it the vectorised cell update: 16-byte loads and stores instead of 48 scalar ones per lane)
    else if constexpr (SPLIT && RB == 1) lstm_gsplit_epilogue<1>(p, acc, bx, by, smem);
    else if constexpr (SPLIT) lstm_split_epilogue<RB>(p, acc, bx, by, red, reinterpret_cast<float *>(bsm));
    else if constexpr (is_top_fused(EPI))
        convt_fused_epilogue<((EPI - EPI_CONVT_FUSED) & 7) / 2 + 1, ((EPI - EPI_CONVT_FUSED) & 1) != 0,
                             (EPI - EPI_CONVT_FUSED) >= 8 ? 6 : 10>(p, acc, bx, red, smem);
    else if constexpr (EPI == EPI_CONV_PAIR) conv_pair_epilogue(p, acc, bx, smem);
    else conv_epilogue<G, EPI, MREP>(p, acc, bx, by, bz, red, smem);
#ifdef VF_TILE_STATS
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long ts3 = VF_TS_NOW();
        VF_TS_ADD(ts_key, 0, ts1 - ts0); VF_TS_ADD(ts_key, 1, ts2 - ts1); VF_TS_ADD(ts_key, 2, ts3 - ts2);
        VF_TS_ADD(ts_key, 3, 1); VF_TS_ADD(ts_key, 4, ts_stage);
    }
#endif
}

template <int G, int EPI, int MREP>
VF_GLOBAL VF_LAUNCH_BOUNDS(kConvThreads, 2) void conv_mfma_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    conv_tile<G, EPI, MREP>(p, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// RB = 2 / 1: the 64- / 32-row tiles
template <int RB>
VF_GLOBAL VF_LAUNCH_BOUNDS(kConvThreads, 2) void conv_lstm_split_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    conv_tile<4, EPI_LSTM, 1, ConvParams, RB>(p, blockIdx.x, blockIdx.y, 0, smem);
}

// the gate-split 64-row tile
VF_GLOBAL VF_LAUNCH_BOUNDS(kConvThreads, 2) void conv_lstm_gsplit64_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    conv_tile<4, EPI_LSTM, 1, ConvParams, -2>(p, blockIdx.x, blockIdx.y, 0, smem);
}

// the gate-split tiles: 128 (MREP 1) / 256 (MREP 2) rows per workgroup
template <int MREP>
VF_GLOBAL VF_LAUNCH_BOUNDS(kConvThreads, 2) void conv_lstm_gsplit_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    conv_tile<4, EPI_LSTM, MREP, ConvParams, 0>(p, blockIdx.x, blockIdx.y, 0, smem);
}

}  // namespace vf
