// vf_persistent.h - the whole rollout (all steps, all layers, all samples) as ONE persistent launch.
//
// Why: every layer launch of the per-layer path ends in a tail (at B=200 the 800/400/200
// workgroups of a layer fill 512 workgroup slots 1.56 / 0.78 / 0.39 times), and the next layer
// cannot start before the last workgroup of the previous one retires - although a sample's next
// layer only needs THAT sample's previous layer.  Here the launch is a list of phases (one per
// former kernel launch) cut into items (one per former workgroup).  Resident workgroups draw
// items from one ticket counter in phase order; before running an item they wait until the
// producer phases have finished the tiles of the samples the item covers (per-phase per-sample
// completion counters).  So the head of layer k+1 overlaps the tail of layer k, and step s+1
// overlaps step s.  Deadlock-free by construction: an item only waits for items with smaller
// tickets, and tickets are only drawn by running workgroups.
//
// Visibility between workgroups follows the agent-scope release/acquire recipe of the CDNA guide
// (section 6, guideline 16): producer = every wave drains its stores, barrier, one lane issues
// fence(release, agent) and then the relaxed agent-scope counter increments; consumer = one wave
// polls the counters relaxed, one lane issues fence(acquire, agent), barrier, plain loads.
#pragma once
#include <hip/hip_runtime.h>
#include "vf_conv_mfma.h"
#include "vf_small_kernels.h"
#include "vf_conv_bf16x6.h"
#include "vf_conv_gsplit.h"
#include "vf_fused_top.h"
#include "vf_fc_tile.h"
#include "vf_conv_first.h"
#include "vf_savp3.h"

namespace vf {

enum PhaseType {
    PH_LSTM = 0, PH_CONV_RELU, PH_CONV_RAW, PH_CONVT_RELU, PH_CONVT_RAW, PH_FC_PARTIAL,
    PH_SA, PH_CDNA_FIN, PH_COMPOSITE,
    PH_TOP_FUSED,           // top transposed conv + compositing in one item (vf_fused_top.h)
    PH_CONV_PAIR,           // enc2 + enc3 in one item (conv_pair_epilogue, vf_conv_mfma.h)
    PH_COND,                // arch 2: border-class biases of the tiled conditioning vector for one conv-LSTM (cond_bias_sample)
    // arch 3 (the published SAVP generator, vf_savp3.h): GEMM tiles that store raw outputs, element-wise items
    PH_CONV_RAW3,           // conv, acc + bias (EPI_RAW)
    PH_GATES_RAW,           // conv-LSTM gate GEMM on the gate-split 128-row tile, raw gate pre-activations (gates_raw_epilogue)
    PH_EW,                  // state FC / class biases / instance norm / cell / up-sampling / compositing layers (EwParams::op)
    PH_CONV_RAW3G2,         // two 32-channel groups of a conv as the two "gates" of one item (the hidden layers of the mask and
                            // scratch heads; every 64-channel conv): conv_tile<2, EPI_RAW>, output [pixel][.. 2 x 32 ..]
    PH_CONV_RAW3G4          // four 32-channel groups per item (convs of 128 / 256 channels: the input tile is staged once for
                            // 128 output channels instead of once per 32): conv_tile<4, EPI_RAW, 1>
};
__host__ __device__ constexpr bool ph_is_conv(const int t) {
    return t <= PH_CONVT_RAW || t == PH_CONV_RAW3 || t == PH_GATES_RAW || t == PH_CONV_RAW3G2 || t == PH_CONV_RAW3G4;
}

constexpr int kMaxDeps = 3;
constexpr int kQueues = 8;                  // one ticket queue per XCD
constexpr int kTicketStride = 32;           // ints between ticket heads (128 B)
constexpr int kCuKeys = 8 * 256;            // (XCC id, SE / SH / CU id of HW_REG_HW_ID) keys of the per-CU state table
constexpr int kCuWords = 4;                 // ints per key: [arrivals, state of slot 0, state of slot 1, -]
constexpr int kSaPerItem = 4;               // samples per PH_SA item (one per wave)
constexpr unsigned kSpinLimit = 1u << 26;   // polls before a waiting item gives up (~ seconds)

struct PhaseDep {
    int cnt_base;       // first completion counter of the producer phase
    int expect;         // value of a counter once the producer is done with that sample
    int mode;           // 0: counter of the same sample, 1: counter 0 (shared / whole-phase producer)
};

struct PhaseDesc {
    int type;
    int first_ticket, n_items;          // position in the global phase order (bookkeeping, statistics)
    int first_q[kQueues], n_q[kQueues]; // ticket range of this phase in each XCD's queue
    int q_gy, q_inner;                  // dealing rule: item = (unit * q_inner + inner) * q_gy + cg  ->  queue
                                        // (unit % (nq / q_gy)) * q_gy + cg, so a channel group's weight slice and a
                                        // sample's tiles stay inside one XCD's L2
    int q_full;                         // queue positions [0, q_full) of this phase follow that rule; the units that do
                                        // not fill a whole round of the nq / q_gy queue groups (25 samples on 8 XCDs: one)
                                        // are dealt tile by tile behind them, so the queues differ by one ITEM, not by
                                        // one sample (a queue only gets help once it is exhausted: an XCD with one
                                        // sample more than the others held a small shard back by 5 %)
    int gx, gy;             // conv phases: items = gx * gy * gz, channel group (gy) fastest
    int B;                  // samples this phase covers
    int NI, tiles_per_img;  // conv phases: how an item maps to samples
    int whole;              // 1: completion is counted once per item on counter 0
    int mrep;               // MFMA row blocks per wave of this conv phase (1 or 2; 0 / -1: the 64- / 32-row conv-LSTM
                            // tiles; 3 / 4 / 5: the gate-split 128- / 256- / 64-row conv-LSTM tiles of vf_conv_mfma.h;
                            // 6: the gate-split 128-row tile of vf_conv_gsplit.h)
    int prec;               // conv-LSTM tile: 0 exact fp32, 1 split-bf16
    int view;               // camera view this phase belongs to (selects the goal pixels of PH_COMPOSITE)
    int cnt_base;
    int aux_base;           // PH_TOP_FUSED: first of the per-sample "LayerNorm partial published" counters
    int ndep;
    PhaseDep dep[kMaxDeps];
    int has_late;           // two-input conv phases: the producer of segment 1 is awaited INSIDE the item, after the chunks
    PhaseDep late;          // of segment 0 (ConvParams::late_cnt, "early start"); dep[] then only holds segment 0's producer
    ConvParams conv;
    ConvParams conv2;       // PH_CONV_PAIR: the 1x1 conv behind `conv` (conv.fuse_next points here, on the device)
    SaParams sa;
    FinParams fin;
    CompositeParams comp;
    CondParams cond;
    EwParams ew;            // PH_EW
};

constexpr int kCtlWords = 64;               // LDS control block: [0..3] scheduler, [8..8+32) goal pixels, [40..44) state words
constexpr int kCtlGoal = 8;
static_assert(kCtlGoal == kFusedCtlGoal, "vf_fused_top.h reads the goal pixels from the control block");
static_assert(kCtlGoal + kMaxCam * kMaxDesig * 2 <= kCtlMyState && kCtlPartnerState + 2 <= kCtlWords, "control block layout");

struct Schedule {
    const PhaseDesc *phases;
    int n_phases;
    int total_items;
    int *ticket;            // [kQueues][kTicketStride] ticket heads, one per XCD (own cache lines)
    int total_q[kQueues];   // items per queue
    int nq;                 // XCD queues in use: kQueues, or 1 (plain phase order)
    int *counters;          // completion counters
    int *status;            // [1] sticky: set non-zero when an item gave up waiting; cleared by the host
                            //     only after it has been read (vf_device_status)
    unsigned long long *stats;  // optional [n_phases][2]: summed wait / run time per phase (wall clock ticks)
    int goal[kMaxCam * kMaxDesig * 2];      // goal pixels [view][desig][row, col]: launch arguments, so
                                            // the device schedule does not depend on them
    int nd;                 // designated pixels per view
    int *cu_tab;            // [kCuKeys][kCuWords] per-CU state words of the cooperative priority scheme ("yielding",
                            // vf_conv_mfma.h) or null: off.  Zeroed with the ticket heads before every launch.
};

__device__ __forceinline__ int ld_relaxed(const int *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// samples [b0, b1) covered by item `local` of phase P
__device__ __forceinline__ void item_samples(const PhaseDesc &P, int local, int &b0, int &b1) {
    switch (P.type) {
        case PH_SA: b0 = local * kSaPerItem; b1 = min(b0 + kSaPerItem, P.B); break;
        case PH_COND: b0 = local * kCondPerItem; b1 = min(b0 + kCondPerItem, P.B); break;
        case PH_CDNA_FIN: b0 = local; b1 = local + 1; break;
        case PH_COMPOSITE: b0 = local / P.gx; b1 = b0 + 1; break;      // gx = tiles per image
        case PH_FC_PARTIAL: b0 = 0; b1 = P.B; break;
        case PH_EW:
            if (P.ew.spi > 0) { b0 = local * P.ew.spi; b1 = min(b0 + P.ew.spi, P.B); }
            else { b0 = local / P.gx; b1 = b0 + 1; }        // gx = items per sample
            break;
        default: {
            const int bx = local / P.gy;    // channel group fastest: a sample's items are adjacent
            if (P.NI == 1) { b0 = bx / P.tiles_per_img; b1 = b0 + 1; }
            else { b0 = bx * P.NI; b1 = min(b0 + P.NI, P.B); }
        }
    }
}

// Out-of-line tile bodies: each keeps its own register allocation instead of being merged into
// one giant function (inlining all of them costs ~60 VGPRs of pressure and spills).  Two things
// keep them as fast as the stand-alone kernels: the LDS workspace is re-derived from the
// `extern __shared__` symbol (a pointer argument would be generic and turn every LDS access
// into a FLAT instruction), and the parameter block is copied out of the CONSTANT address space
// (scalar loads into SGPRs; pointers loaded from constant memory are known to be global).
#define VF_CONST_AS __attribute__((address_space(4)))

// Function arguments travel in VGPRs, so the compiler cannot know the pointer is wave-uniform;
// readfirstlane makes it scalar, and the constant address space makes every field access an
// s_load and every pointer field a known-global pointer.
template <class T>
__device__ __forceinline__ const VF_CONST_AS T &const_params(const T *generic_ptr) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(generic_ptr);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return *(const VF_CONST_AS T *)(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ float *tile_lds() {
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    return smem_all + kCtlWords;    // the control block comes first
}

template <int G, int EPI, int MREP>
static __device__ __noinline__ __attribute__((not_tail_called)) void conv_tile_call(const ConvParams *p, int bx, int by, int bz) {
    conv_tile<G, EPI, MREP>(const_params(p), bx, by, bz, tile_lds());
}
template <int RB>
static __device__ __noinline__ __attribute__((not_tail_called)) void lstm_split_tile_call(const ConvParams *p, int bx, int by) {
    conv_tile<4, EPI_LSTM, 1, const VF_CONST_AS ConvParams, RB>(const_params(p), bx, by, 0, tile_lds());
}
template <int MREP>
static __device__ __noinline__ __attribute__((not_tail_called)) void lstm_gsplit_tile_call(const ConvParams *p, int bx, int by) {
    conv_tile<4, EPI_LSTM, MREP, const VF_CONST_AS ConvParams, 0>(const_params(p), bx, by, 0, tile_lds());
}
template <int MR>
static __device__ __noinline__ __attribute__((not_tail_called)) void lstm_gsplit2_tile_call(const ConvParams *p, int bx, int by) {
    conv_lstm_gsplit2_tile<MR>(const_params(p), bx, by, tile_lds());
}
static __device__ __noinline__ __attribute__((not_tail_called)) void lstm_gsplit64_tile_call(const ConvParams *p, int bx, int by) {
    conv_tile<4, EPI_LSTM, 1, const VF_CONST_AS ConvParams, -2>(const_params(p), bx, by, 0, tile_lds());
}
template <int MREP>
static __device__ __noinline__ __attribute__((not_tail_called)) void lstm_bf16x6_tile_call(const ConvParams *p, int bx, int by) {
    conv_lstm_bf16x6_tile<MREP>(const_params(p), bx, by, tile_lds());
}
template <int CO>
static __device__ __noinline__ __attribute__((not_tail_called)) void conv_first_tile_call(const ConvParams *p, int bx) {
    conv_first_tile<CO>(const_params(p), bx, tile_lds());
}
static __device__ __noinline__ __attribute__((not_tail_called)) void fc_wide_tile_call(const ConvParams *p, int bx, int bz) {
    fc_wide_tile(const_params(p), bx, bz, tile_lds());
}
template <int ND, bool FIRST, int K, int CF = 32>
static __device__ __noinline__ __attribute__((not_tail_called)) void composite_tile_call(const CompositeParams *p, int tile, int b, int view) {
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    const int *goal = reinterpret_cast<const int *>(smem_all) + kCtlGoal + view * ND * 2;
    composite_tile<ND, K, FIRST, const VF_CONST_AS CompositeParams, CF>(const_params(p), tile, b, goal, tile_lds());
}
static __device__ __noinline__ __attribute__((not_tail_called)) void gates_raw_tile_call(const ConvParams *p, int bx, int by) {
    conv_lstm_gsplit2_tile<4, const VF_CONST_AS ConvParams, true>(const_params(p), bx, by, tile_lds());
}
// the element-wise items of arch 3 (vf_savp3.h), one out-of-line body per operation (own register allocation each)
#define VF_EW_BODY(NAME_, CALL_)                                                                                        \
    static __device__ __noinline__ __attribute__((not_tail_called)) void NAME_(const EwParams *p_, int idx, int b0, int b1) { \
        const VF_CONST_AS EwParams &p = const_params(p_);                                                               \
        (void)idx; (void)b0; (void)b1;                                                                                  \
        CALL_;                                                                                                          \
    }
VF_EW_BODY(ew_cond3_call, cond3_item(p.cond, b0, b1, tile_lds()))
VF_EW_BODY(ew_inorm_call, inorm_item(p.norm, b0, idx, p.wt != 0, tile_lds()))
VF_EW_BODY(ew_incell_call, incell_item(p.norm, b0, idx, p.wt != 0, tile_lds()))
VF_EW_BODY(ew_upsample_call, upsample_item(p.up, b0, idx, p.wt != 0))
#undef VF_EW_BODY
static __device__ __noinline__ __attribute__((not_tail_called)) void ew_sa3_call(const EwParams *p_, int b0, int b1) {
    const VF_CONST_AS EwParams &p = const_params(p_);
    const int wave = threadIdx.x >> 6, b = b0 + wave;
    if (b < b1) sa3_sample(p.sa, b, threadIdx.x & 63, tile_lds() + 128 * wave);
}
template <int ND>
static __device__ __noinline__ __attribute__((not_tail_called)) void ew_transform_call(const EwParams *p_, int idx, int b) {
    transform_item<ND>(const_params(p_).top, idx, b, tile_lds());
}
template <int ND>
static __device__ __noinline__ __attribute__((not_tail_called)) void ew_top3_call(const EwParams *p_, int idx, int b, int view) {
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    const int *goal = reinterpret_cast<const int *>(smem_all) + kCtlGoal + view * ND * 2;
    top3_item<ND>(const_params(p_).top, idx, b, goal, const_params(p_).wt != 0, tile_lds());
}
template <int ND>
static __device__ __noinline__ __attribute__((not_tail_called)) void ew_compose_call(const EwParams *p_, int idx, int b, int view) {
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    const int *goal = reinterpret_cast<const int *>(smem_all) + kCtlGoal + view * ND * 2;
    compose_item<ND>(const_params(p_).top, idx, b, goal, tile_lds());
}
static __device__ __noinline__ __attribute__((not_tail_called)) void small_item_call(const PhaseDesc *P, int type, int b0, int b1) {
    float *smem = tile_lds();
    if (type == PH_SA) {
        const int wave = threadIdx.x >> 6, b = b0 + wave;
        if (b < b1) sa_sample(const_params(&P->sa), b, threadIdx.x & 63, smem + 32 * wave);
    } else if (type == PH_COND) {
        cond_bias_sample(const_params(&P->cond), b0, b1, smem);
    } else {
        cdna_finalize_sample(const_params(&P->fin), b0, smem);
    }
}

// Two resident workgroups per CU (one wave of each per SIMD): 256 VGPRs per lane for every tile body.
template <int ND>
VF_GLOBAL VF_LAUNCH_BOUNDS(kConvThreads, 2) void rollout_persistent_kernel(
    const PhaseDesc *__restrict__ phases, const Schedule sched) {
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    // all LDS in one dynamic array: the control block first, the tile workspace after it
    int *s_ctl = reinterpret_cast<int *>(smem_all);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int ph = 0;
    __builtin_amdgcn_s_setprio(2);      // default priority of this launch; conv-LSTM K loops step down to 0 (vf_conv_mfma.h)
    if (tid < kMaxCam * kMaxDesig * 2) s_ctl[kCtlGoal + tid] = sched.goal[tid];    // visible after the first barrier
#ifdef VF_TRACE
    if (tid == 0) {
        s_ctl[5] = 0;
        unsigned hwid, xcc_;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));
        VF_TRACE_EVT(TR_HWID, ((unsigned long long)(xcc_ & 15u) << 32) | hwid);
    }
#endif

    // XCD-aware ticketing.  A phase's items are dealt to sched.nq queues so that all items of one output-channel
    // group and all tiles of one sample land in the same queue; a workgroup draws from the queue of the XCD it
    // runs on, so the weight slice of that channel group (0.8 - 2.4 MB) and the halo rows shared by a sample's
    // neighbouring tiles stay in that XCD's 4 MB L2 instead of every L2 streaming everything.  The XCD id only
    // picks the queue (speed, never correctness): any workgroup may run any item, and a workgroup whose queue is
    // exhausted steals from the others.  Every queue is in phase order, so the undone item of the lowest phase
    // always has its producers done and sits at the head of its queue: deadlock-free as before.
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int nq = sched.nq;
    const int q_own = (int)(xcc & (unsigned)(nq - 1));

    // Yielding (vf_conv_mfma.h): the two workgroups of a CU find each other through the CU's entry of sched.cu_tab - the
    // first to arrive takes state word 1, the second word 2 - and remember "mine" / "the partner's" in the control block.
    // (A key shared by more than two workgroups, or a CU with one, only costs the scheme its effect: yields are bounded.)
    const bool yield_on = sched.cu_tab != nullptr;
    if (yield_on && tid == 0) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        const int key = (int)((xcc & 7u) * 256u + ((hwid >> 8) & 0xFFu));
        int *entry = sched.cu_tab + key * kCuWords;
        const int slot = atomicAdd(entry, 1) & 1;
        *reinterpret_cast<int **>(s_ctl + kCtlMyState) = entry + 1 + slot;
        *reinterpret_cast<int **>(s_ctl + kCtlPartnerState) = entry + 1 + (slot ^ 1);
    }
    // this workgroup's state word: 1 = on some sample's dependency chain, 0 = waiting / polling / recurrent half
    auto publish_state = [&](const int critical) {
        // (the word's address lives in the LDS control block, not in a register: the tile calls clobber every VGPR)
        if (yield_on && tid == 0)
            __hip_atomic_store(ctl_state_word(kCtlMyState), critical, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    for (;;) {
        [[maybe_unused]] const unsigned long long ts_top = VF_TS_NOW();
        // hipcc 7.2 does not put an `s_waitcnt lgkmcnt(0)` in front of a loop-head barrier for an LDS store that is
        // still pending on the back edge (a draw moved to the end of this body once let the other waves read the
        // previous ticket, profiles/r03_tile_plan_sweep.txt).  The wait is therefore explicit - measured free - so
        // that the scheduler does not depend on where the compiler places its waits; tools/lint_barriers.py
        // (tests/test_kernel_lint.py, CPU suite) checks that it is still there and that no other loop-head barrier of
        // the code object is reached with an LDS store in flight.
        // (the comment travels into the assembly listing: the lint finds THIS wait by its marker and checks that the
        // barrier is the next instruction, instead of guessing which barrier of the kernel is the loop head)
        asm volatile("s_waitcnt lgkmcnt(0) ; vf_sched_loop_head" ::: "memory");
        __syncthreads();                    // previous item fully retired (LDS reusable)
        VF_TRACE_EVT(TR_TICKET);
        if (tid == 0) {
            int t = -1, qq = q_own;
            for (int tries = 0; tries < nq; ++tries) {
                const int cand = atomicAdd(sched.ticket + qq * kTicketStride, 1);
                if (cand < sched.total_q[qq]) { t = cand; break; }
                qq = (qq + 1) & (nq - 1);
            }
            s_ctl[0] = t;
            s_ctl[2] = qq;
        }
        __syncthreads();
        const int t = s_ctl[0];
        if (t < 0) break;
        const int qq = s_ctl[2];
        const bool own = qq == q_own;
        if (!own) ph = 0;                   // stolen ticket (tail of the launch): look its phase up from the start
        while (t >= phases[ph].first_q[qq] + phases[ph].n_q[qq]) ++ph;
        const int ph_run = ph;
        const PhaseDesc &P = phases[ph_run];
        if (!own) ph = 0;                   // the cursor is only monotone within one queue
        // queue position -> item of the phase
        int local;
        {
            const int lq = t - P.first_q[qq];
            const int per = nq / P.q_gy, qb = qq / P.q_gy, cg = qq - qb * P.q_gy;
            int unit, inner;
            if (lq < P.q_full) {            // whole rounds of units: a sample's tiles stay in this queue
                const int grp = lq / P.q_inner;
                inner = lq - grp * P.q_inner;
                unit = grp * per + qb;
            } else {                        // the remaining units, dealt tile by tile over the queue groups
                const int i = (lq - P.q_full) * per + qb;
                const int u = i / P.q_inner;
                inner = i - u * P.q_inner;
                unit = (P.q_full / P.q_inner) * per + u;
            }
            local = (unit * P.q_inner + inner) * P.q_gy + cg;
        }

        int b0, b1;
        item_samples(P, local, b0, b1);
        const unsigned long long t_start = sched.stats ? wall_clock64() : 0ull;
        VF_TS_ADD(15, 5, VF_TS_NOW() - ts_top);         // ticket fetch + phase lookup
        VF_TS_ADD(15, 7, 1);

        // ---- wait for the producers (wave 0 polls: lane i watches sample b0 + i, strided)
        if (wave == 0 && P.ndep > 0) {
            unsigned spins = 0;
            bool ok;
            do {
                ok = true;
                for (int d = 0; d < P.ndep; ++d) {
                    const PhaseDep &dp = P.dep[d];
                    if (dp.mode == 1) {
                        if (lane == 0) ok = ok && (ld_relaxed(sched.counters + dp.cnt_base) >= dp.expect);
                    } else {
                        for (int b = b0 + lane; b < b1; b += 64)
                            ok = ok && (ld_relaxed(sched.counters + dp.cnt_base + b) >= dp.expect);
                    }
                }
                ok = __all(ok);
                if (!ok) {
                    __builtin_amdgcn_s_sleep(kPollSleep);
                    if (++spins > kSpinLimit || ld_relaxed(sched.status) != 0) {
                        if (lane == 0) atomicExch(sched.status, 1);
                        break;
                    }
                }
            } while (!ok);
            if (lane == 0) {
                s_ctl[1] = ok ? 1 : 0;
                VF_ACQUIRE_AGENT();
            }
        } else if (tid == 0) {
            s_ctl[1] = 1;
        }
        __syncthreads();
        if (s_ctl[1] == 0) break;           // a producer never arrived: abandon the rollout
        // chain-critical from here to the publish - except the part of an early-started item in front of its mid-item
        // wait (recurrent half / skip-tensor chunks: work for later) and the CDNA FC (needed at the end of the step only)
        publish_state((P.has_late || P.type == PH_FC_PARTIAL) ? 0 : 1);
        const unsigned long long t_run = sched.stats ? wall_clock64() : 0ull;
        VF_TRACE_EVT(TR_RUN + (unsigned)P.type);
        VF_TRACE_EVT(TR_PHASE, (unsigned long long)ph_run);

        // ---- run the item
        {
            // conv items: channel group fastest, then row tile (so all items of one sample are
            // neighbours in ticket order); FC: (cg, split) fastest over its single row tile
            const int by = local % P.gy, bx = local / P.gy;
            switch (P.type) {
                case PH_LSTM:
                    // Production plans (lstm_plan / plan_geometry in vf_engine.hip) select four fp32 tiles: the 128- and
                    // 64-row gate-split tiles, the 32-row tile, and - only for a geometry the gate-split tile cannot stage
                    // - the 128-row tile with its weights through LDS.  The tiles that lost in rounds 2-3 (first-generation
                    // gate-split 128 / 256 rows, 256 rows from L2, 64 rows through LDS) are A/B material of
                    // -DVF_DEBUG_KNOBS builds: compiled in here they cost the launch 33 spilled VGPRs and 924 B of scratch.
                    if (P.prec == 1) lstm_bf16x6_tile_call<1>(&P.conv, bx, by);         // 128-row tiles only
                    else if (P.mrep == 6) lstm_gsplit2_tile_call<4>(&P.conv, bx, by);    // gate-split 128-row tile, final form
                    else if (P.mrep == 5) lstm_gsplit64_tile_call(&P.conv, bx, by);        // gate-split 64-row tile
                    else if (P.mrep < 0) lstm_split_tile_call<1>(&P.conv, bx, by);
#ifdef VF_DEBUG_KNOBS
                    else if (P.mrep == 0) lstm_split_tile_call<2>(&P.conv, bx, by);
                    else if (P.mrep == 3) lstm_gsplit_tile_call<1>(&P.conv, bx, by);       // gate-split 128-row tile, first form
                    else if (P.mrep == 4) lstm_gsplit_tile_call<2>(&P.conv, bx, by);       // gate-split 256-row tile
                    else if (P.mrep == 2) conv_tile_call<4, EPI_LSTM, 2>(&P.conv, bx, by, 0);
#endif
                    else conv_tile_call<4, EPI_LSTM, 1>(&P.conv, bx, by, 0);
                    break;
                case PH_CONV_RELU: conv_tile_call<1, EPI_BIAS_RELU, 1>(&P.conv, bx, by, 0); break;
                case PH_CONV_RAW:
                    // (mrep 8: the first conv of the encoder - 3-channel frame, 5 x 5 / 2 - on the vector ALUs, vf_conv_first.h)
                    if (P.mrep == 8) { if (P.conv.Cout == 16) conv_first_tile_call<16>(&P.conv, bx); else conv_first_tile_call<32>(&P.conv, bx); }
                    else conv_tile_call<1, EPI_RAW_STATS, 1>(&P.conv, bx, by, 0);
                    break;
                case PH_CONVT_RELU: conv_tile_call<4, EPI_CONVT_RELU, 1>(&P.conv, bx, by, 0); break;
                case PH_CONVT_RAW: conv_tile_call<4, EPI_CONVT_RAW_STATS, 1>(&P.conv, bx, by, 0); break;
                case PH_FC_PARTIAL:
                    // (mrep 7: all eight column groups in one item per (row tile, K split), vf_fc_tile.h - the plan of
                    // every persistent schedule; the generic tile serves a geometry that one cannot hold)
                    if (P.mrep == 7) fc_wide_tile_call(&P.conv, bx % P.gx, bx / P.gx);
                    else conv_tile_call<1, EPI_PARTIAL, 2>(&P.conv, bx % P.gx, by, bx / P.gx);
                    break;
                case PH_CONV_PAIR: conv_tile_call<2, EPI_CONV_PAIR, 1>(&P.conv, bx, 0, 0); break;
                // (K = 6: the compositing of arch 2 - four CDNA warps + previous + first frame + scratch - always with
                // the first-frame layer; K = 10 otherwise)
                case PH_TOP_FUSED:
                    if (P.comp.K == 6) conv_tile_call<4, fused_epi(ND, true, true), 1>(&P.conv, bx, 0, 0);
                    else if (P.comp.first_frame) conv_tile_call<4, fused_epi(ND, true), 1>(&P.conv, bx, 0, 0);
                    else conv_tile_call<4, fused_epi(ND, false), 1>(&P.conv, bx, 0, 0);
                    break;
                case PH_COMPOSITE:
                    if (P.comp.K == 6) composite_tile_call<ND, true, 6>(&P.comp, local % P.gx, b0, P.view);
                    else if (P.comp.first_frame) composite_tile_call<ND, true, 10>(&P.comp, local % P.gx, b0, P.view);
                    else if (P.comp.CF > 32) composite_tile_call<ND, false, 10, 64>(&P.comp, local % P.gx, b0, P.view);  // (public decoder)
                    else composite_tile_call<ND, false, 10>(&P.comp, local % P.gx, b0, P.view);
                    break;
                // (mrep 2: the 256-row plan of arch 3's full-resolution layers - half the items, each twice the GEMM rows)
                case PH_CONV_RAW3:
                    if (P.mrep == 2) conv_tile_call<1, EPI_RAW, 2>(&P.conv, bx, by, 0);
                    else conv_tile_call<1, EPI_RAW, 1>(&P.conv, bx, by, 0);
                    break;
                case PH_CONV_RAW3G2:
                    if (P.mrep == 2) conv_tile_call<2, EPI_RAW, 2>(&P.conv, bx, by, 0);
                    else conv_tile_call<2, EPI_RAW, 1>(&P.conv, bx, by, 0);
                    break;
                case PH_CONV_RAW3G4: conv_tile_call<4, EPI_RAW, 1>(&P.conv, bx, by, 0); break;
                case PH_GATES_RAW: gates_raw_tile_call(&P.conv, bx, by); break;
                case PH_EW: {
                    const int idx = P.ew.spi > 0 ? 0 : local - b0 * P.gx;
                    switch (P.ew.op) {
                        case EW_SA3: ew_sa3_call(&P.ew, b0, b1); break;
                        case EW_COND3: ew_cond3_call(&P.ew, idx, b0, b1); break;
                        case EW_INORM: ew_inorm_call(&P.ew, idx, b0, b1); break;
                        case EW_INCELL: ew_incell_call(&P.ew, idx, b0, b1); break;
                        case EW_UPSAMPLE: ew_upsample_call(&P.ew, idx, b0, b1); break;
                        case EW_TRANSFORM: ew_transform_call<ND>(&P.ew, idx, b0); break;
                        case EW_TOP3: ew_top3_call<ND>(&P.ew, idx, b0, P.view); break;
                        default: ew_compose_call<ND>(&P.ew, idx, b0, P.view); break;
                    }
                    break;
                }
                default: small_item_call(&P, P.type, b0, b1); break;
            }
        }

        // ---- publish: drain this wave's stores, barrier, one release, then the counters
        [[maybe_unused]] const unsigned long long ts_pub = VF_TS_NOW();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s_ctl[1] == 0) break;           // the item itself gave up (fused top: a mate never arrived): its outputs
                                            // were never written, so nothing is published and the rollout is abandoned
        if (sched.stats && tid == 0) {
            const unsigned long long t_end = wall_clock64();
            atomicAdd(sched.stats + 2 * ph_run, t_run - t_start);
            atomicAdd(sched.stats + 2 * ph_run + 1, t_end - t_run);
        }
        if (wave == 0) {
            // write-through items (ConvParams::wt_out: every global store of the tile was an sc1 / atomic store, drained
            // above by every wave) need no L2 write-back in front of their counters - CDNA guide section 6 G16, recipe R1
            const bool wt_item = ((ph_is_conv(P.type) || P.type == PH_TOP_FUSED) && P.conv.wt_out != 0) || (P.type == PH_EW && P.ew.wt != 0);
            if (lane == 0 && !wt_item) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_wave_barrier();
            if (P.whole) {
                if (lane == 0)
                    __hip_atomic_fetch_add(sched.counters + P.cnt_base, 1, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
            } else {
                // lane 0 performed the release; its counter increments must not be overtaken, so
                // the same lane publishes every covered sample
                if (lane == 0)
                    for (int b = b0; b < b1; ++b)
                        __hip_atomic_fetch_add(sched.counters + P.cnt_base + b, 1, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        publish_state(0);                   // back to the ticket counter / a dependency poll
        VF_TS_ADD(15, 6, VF_TS_NOW() - ts_pub);         // drain + barrier + release fence + counters (tid 0's view)
        VF_TRACE_EVT(TR_DONE);
    }
    publish_state(0);                       // (also on the abandon paths: never leave a stale "critical" behind)
#ifdef VF_TRACE
    if (tid == 0 && blockIdx.x < kTraceWgs) g_trace_n[blockIdx.x] = (unsigned)s_ctl[5];
#endif
}

}  // namespace vf
