// vf_engine.hip - host side of libvf_hip.so: device buffers, weight re-packing for the MFMA
// kernels, the per-step launch sequence of the CDNA predictor and the C ABI of include/vf_hip.h.
//
// Layer table and semantics: visual_foresight_amd/video_prediction/cdna_arch.py (the reference
// repo holds no network code).  Boundary semantics replaced here:
// visual_mpc/video_prediction/setup_predictor.py:98-114,164-200 and pred_util.py:4-48 of the
// reference (context slicing, /255, batch-1 context broadcast, per-sample action batch).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/vf_hip.h"
#include "vf_conv_mfma.h"
#include "vf_small_kernels.h"
#include "vf_conv_bf16x6.h"
#include "vf_persistent.h"

namespace vf {

static thread_local std::string g_last_error;

static int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

#define VF_HIP_CHECK(expr)                                                                  \
    do {                                                                                    \
        hipError_t err_ = (expr);                                                           \
        if (err_ != hipSuccess)                                                             \
            return fail(VF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(err_));   \
    } while (0)

static const int kLstmSizes[7] = {32, 32, 64, 64, 128, 64, 32};
static const int kMaxSubBatches = 8;

// ------------------------------------------------------------------ canonical tensor table
struct TensorDesc {
    std::string name;
    int shape[4];
    int rank;
    size_t offset;
    size_t size() const {
        size_t n = 1;
        for (int i = 0; i < rank; ++i) n *= (size_t)shape[i];
        return n;
    }
};

static std::vector<TensorDesc> tensor_table(const vf_config &c) {
    std::vector<TensorDesc> t;
    size_t off = 0;
    auto add = [&](const std::string &name, std::vector<int> shape) {
        TensorDesc d;
        d.name = name;
        d.rank = (int)shape.size();
        for (int i = 0; i < 4; ++i) d.shape[i] = i < d.rank ? shape[i] : 1;
        d.offset = off;
        off += d.size();
        t.push_back(d);
    };
    auto conv = [&](const std::string &n, int kh, int kw, int cin, int cout) {
        add(n + "/w", {kh, kw, cin, cout});
        add(n + "/b", {cout});
    };
    auto ln = [&](const std::string &n, int ch) {
        add(n + "/g", {ch});
        add(n + "/b", {ch});
    };
    const int *L = kLstmSizes;
    const int a = c.adim + c.sdim, K = c.num_masks;
    const int fc_in = (c.height / 8) * (c.width / 8) * L[4];
    conv("enc0", 5, 5, 3, 32);                 ln("ln1", 32);
    conv("lstm1", 5, 5, 32 + L[0], 4 * L[0]);  ln("ln2", L[0]);
    conv("lstm2", 5, 5, L[0] + L[1], 4 * L[1]); ln("ln3", L[1]);
    conv("enc1", 3, 3, L[1], L[1]);
    conv("lstm3", 5, 5, L[1] + L[2], 4 * L[2]); ln("ln4", L[2]);
    conv("lstm4", 5, 5, L[2] + L[3], 4 * L[3]); ln("ln5", L[3]);
    conv("enc2", 3, 3, L[3], L[3]);
    conv("enc3", 1, 1, L[3] + a, L[3]);
    conv("lstm5", 5, 5, L[3] + L[4], 4 * L[4]); ln("ln6", L[4]);
    conv("convt1", 3, 3, L[4], L[4]);
    conv("lstm6", 5, 5, L[4] + L[5], 4 * L[5]); ln("ln7", L[5]);
    conv("convt2", 3, 3, L[5] + L[1], L[5]);
    conv("lstm7", 5, 5, L[5] + L[6], 4 * L[6]); ln("ln8", L[6]);
    conv("convt3", 3, 3, L[6] + 32, 32);        ln("ln9", 32);
    conv("rgb", 1, 1, 32, 3);
    conv("masks", 1, 1, 32, K + 1);
    add("cdna/w", {fc_in, kTaps * K});
    add("cdna/b", {kTaps * K});
    add("state/w", {a, c.sdim});
    add("state/b", {c.sdim});
    return t;
}

static const TensorDesc *find_tensor(const std::vector<TensorDesc> &t, const std::string &name) {
    for (const auto &d : t)
        if (d.name == name) return &d;
    return nullptr;
}

// ------------------------------------------------------------------ one dense layer
enum PackMode { PACK_PLAIN, PACK_LSTM, PACK_CONVT };

struct ConvLayer {
    std::string name;
    PackMode mode;
    int G;
    int Hin, Win, Hout, Wout;       // Hout/Wout: GEMM row grid
    int KH, KW, stride, pad;        // kernel geometry as the GEMM sees it
    int segC[2], nseg;
    int KC, nchunk[2];
    int mrep;                       // MFMA row blocks per wave: the workgroup covers 128 * mrep rows
    int prec = 0;                   // 1: split-bf16 tile (conv-LSTM only)
    unsigned short *d_w16 = nullptr;
    int NI, TH, TW, RPI, tilesY, tilesX;
    int ncg, Cout;
    int nsplit, chunks_per_split, n_valid;
    int stats_nparts;               // partial sums this layer's epilogue writes per sample
    size_t lds_bytes;
    float *d_w = nullptr, *d_b = nullptr;
};

static int round_up(int x, int m) { return (x + m - 1) / m * m; }

static size_t conv_lds_bytes(const ConvLayer &l, int KC) {
    const int LH = (l.TH - 1) * l.stride + l.KH, LW = (l.TW - 1) * l.stride + l.KW;
    // A tile + LayerNorm table + reduction scratch (+ for 4-gate layers the double-buffered
    // per-tap B blocks: 2 x KC/8 x [4 gates][64 lanes] float4)
    const size_t b_lds = l.mode == PACK_LSTM ? (size_t)2 * (KC / 8) * 4 * 64 * 16 : 0;
    return ((size_t)l.NI * LH * LW * (KC + 4) + 4 * (size_t)l.NI) * 4 + 64 + b_lds;
}

// choose tile shape and chunk size for a layer whose GEMM row grid is Hout x Wout
static void plan_geometry(ConvLayer &l, bool needs_stats, bool one_pixel_images) {
    const int rows = 128 * l.mrep, wrows = 32 * l.mrep;
    if (one_pixel_images) {         // FC: every sample is a 1x1 image with many channels
        l.TH = l.TW = 1; l.tilesY = l.tilesX = 1; l.RPI = 1; l.NI = rows;
    } else {
        l.TW = std::min(l.Wout, 32);
        l.TH = std::min(l.Hout, rows / l.TW);
        l.tilesX = (l.Wout + l.TW - 1) / l.TW;
        l.tilesY = (l.Hout + l.TH - 1) / l.TH;
        const int px = l.TH * l.TW;
        if (l.tilesX * l.tilesY == 1 && px <= rows / 2) {
            // several whole images per workgroup; with statistics every wave must sit inside one image
            l.RPI = needs_stats ? round_up(px, wrows) : px;
            l.NI = rows / l.RPI;
        } else {
            l.RPI = rows; l.NI = 1;
        }
    }
    const int maxC = std::max(l.segC[0], l.nseg > 1 ? l.segC[1] : 0);
    int KC = 32;
    while (KC > 8 && (KC > round_up(maxC, 8) || conv_lds_bytes(l, KC) > 78 * 1024 ||
                      l.segC[0] % KC || (l.nseg > 1 && l.segC[1] % KC)))
        KC >>= 1;
    if (l.mode == PACK_LSTM)
        if (const char *e = getenv("VF_LSTM_KC")) KC = std::min(KC, std::max(8, atoi(e)));   // tuning knob
    if (l.prec == 1) KC = kBfKC;        // the split-bf16 tile stages 16-channel chunks
    l.KC = KC;
    for (int s = 0; s < 2; ++s) l.nchunk[s] = s < l.nseg ? (l.segC[s] + KC - 1) / KC : 0;
    l.lds_bytes = conv_lds_bytes(l, KC);
    if (l.prec == 1) {
        const int LH = (l.TH - 1) * l.stride + l.KH, LW = (l.TW - 1) * l.stride + l.KW;
        l.lds_bytes = bf16x6_lds_bytes(l.NI, LH, LW);
    }
    l.stats_nparts = (l.NI == 1 ? l.tilesY * l.tilesX : 1) * l.ncg;
}

// canonical [KH][KW][Cin][Ctot] -> packed [chunk][tap][k8][khalf][Ntot][4]
static std::vector<float> pack_weights(const ConvLayer &l, const float *w, int KHc, int KWc, int Cin, int Ctot) {
    const int G = l.G, KC = l.KC, K8 = KC / 8, ntaps = l.KH * l.KW;
    const int Ntot = l.ncg * G * 32;
    const int nchunks = l.nchunk[0] + l.nchunk[1];
    std::vector<float> out((size_t)nchunks * ntaps * K8 * 2 * Ntot * 4, 0.f);
    for (int ci = 0; ci < nchunks; ++ci) {
        const int s = ci < l.nchunk[0] ? 0 : 1;
        const int c0 = (s == 0 ? ci : ci - l.nchunk[0]) * KC;
        const int seg_off = s == 0 ? 0 : l.segC[0];
        for (int ty = 0; ty < l.KH; ++ty)
            for (int tx = 0; tx < l.KW; ++tx)
                for (int k8 = 0; k8 < K8; ++k8)
                    for (int kh = 0; kh < 2; ++kh)
                        for (int col = 0; col < Ntot; ++col) {
                            const int cgi = col / (G * 32), g = (col / 32) % G, nn = col % 32;
                            const int co = cgi * 32 + nn;
                            if (co >= l.Cout) continue;
                            int ky = ty, kx = tx, ocol;
                            if (l.mode == PACK_LSTM) {
                                ocol = g * l.Cout + co;
                            } else if (l.mode == PACK_CONVT) {
                                // output (2y+py, 2x+px) gathers input (y-1+ty, x-1+tx) through
                                // canonical tap k = parity + 2*(1 - t)
                                ky = (g >> 1) + 2 * (1 - ty);
                                kx = (g & 1) + 2 * (1 - tx);
                                if (ky >= KHc || kx >= KWc) continue;
                                ocol = co;
                            } else {
                                ocol = co;
                            }
                            for (int j = 0; j < 4; ++j) {
                                const int c = c0 + k8 * 8 + kh * 4 + j;
                                if (c >= l.segC[s]) continue;
                                const int cin = seg_off + c;
                                const size_t dst = ((((((size_t)ci * ntaps + (ty * l.KW + tx)) * K8 + k8) * 2 + kh) * Ntot + col) * 4) + j;
                                out[dst] = w[(((size_t)ky * KWc + kx) * Cin + cin) * Ctot + ocol];
                            }
                        }
    }
    return out;
}

static unsigned short bf16_rne(float x) {        // round-to-nearest-even, as v_cvt_pk_bf16_f32
    unsigned u;
    memcpy(&u, &x, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return (unsigned short)(u >> 16);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float bf16_to_f32(unsigned short h) {
    unsigned u = (unsigned)h << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}

// canonical LSTM weights [5][5][Cin][4C] -> three exact bf16 pieces, packed
// [chunk16][tap][cg][gate][plane][k-half][32 columns][8 channels]   (vf_conv_bf16x6.h)
static std::vector<unsigned short> pack_weights_bf16x3(const ConvLayer &l, const float *w, int Cin, int Ctot) {
    const int ntaps = l.KH * l.KW, nchunks = l.nchunk[0] + l.nchunk[1];
    std::vector<unsigned short> out((size_t)nchunks * ntaps * l.ncg * 4 * 3 * 64 * 8, 0);
    for (int ci = 0; ci < nchunks; ++ci) {
        const int s = ci < l.nchunk[0] ? 0 : 1;
        const int c0 = (s == 0 ? ci : ci - l.nchunk[0]) * kBfKC;
        const int seg_off = s == 0 ? 0 : l.segC[0];
        for (int tap = 0; tap < ntaps; ++tap)
            for (int cgi = 0; cgi < l.ncg; ++cgi)
                for (int g = 0; g < 4; ++g)
                    for (int kh = 0; kh < 2; ++kh)
                        for (int nn = 0; nn < 32; ++nn)
                            for (int j = 0; j < 8; ++j) {
                                const int c = c0 + kh * 8 + j, co = cgi * 32 + nn;
                                if (c >= l.segC[s] || co >= l.Cout) continue;
                                const float x = w[((size_t)tap * Cin + seg_off + c) * Ctot + g * l.Cout + co];
                                const unsigned short p0 = bf16_rne(x);
                                const float r1 = x - bf16_to_f32(p0);
                                const unsigned short p1 = bf16_rne(r1);
                                const unsigned short p2 = bf16_rne(r1 - bf16_to_f32(p1));
                                const unsigned short pc[3] = {p0, p1, p2};
                                for (int pl = 0; pl < 3; ++pl) {
                                    const size_t unit = ((((size_t)(ci * ntaps + tap) * l.ncg + cgi) * 4 + g) * 3 + pl) * 64 +
                                                        kh * 32 + nn;
                                    out[unit * 8 + j] = pc[pl];
                                }
                            }
    }
    return out;
}

static std::vector<float> pack_bias(const ConvLayer &l, const float *b) {
    const int G = l.G;
    std::vector<float> out((size_t)l.ncg * G * 32, 0.f);
    for (int cgi = 0; cgi < l.ncg; ++cgi)
        for (int g = 0; g < G; ++g)
            for (int nn = 0; nn < 32; ++nn) {
                const int co = cgi * 32 + nn;
                if (co >= l.Cout) continue;
                const int src = l.mode == PACK_LSTM ? g * l.Cout + co : co;
                out[((size_t)cgi * G + g) * 32 + nn] = b[src];
            }
    return out;
}

}  // namespace vf

using namespace vf;

// Views of the engine's working buffers: one per sub-batch (offset to its first sample) and
// one "shared" set of batch-1 buffers per sub-batch for tensors that are identical for every
// sample (see run_steps).
struct BatchView {
    float *enc0_o, *enc1_o, *enc2_o, *enc3_o, *enc4_o, *enc5_o, *enc6_o;
    float *c_state[7], *h_state[7][2];
    double *st_enc0, *st_h[7], *st_enc6;
    float *sbias, *fc_part, *kern;
    float *frames_all, *distrib_all, *states_all;
    double *sums;
    const float *actions;
};

// ------------------------------------------------------------------ the engine
struct vf_handle {
    vf_config cfg;
    int H, W, T, S, ND, K;              // S = steps per rollout = T + n_context - 1
    int ntiles;                         // composite tiles per image
    std::vector<TensorDesc> table;
    bool have_weights = false, have_context = false;
    int last_B = 0;

    // layers
    ConvLayer enc0, lstm[7], enc1, enc2, enc3, convt1, convt2, convt3, fc;

    // small dense parameters on the device
    float *d_ln_g[9] = {nullptr}, *d_ln_b[9] = {nullptr};
    float *d_w_rgb = nullptr, *d_b_rgb = nullptr, *d_w_mask = nullptr, *d_b_mask = nullptr;
    float *d_w_state = nullptr, *d_b_state = nullptr, *d_w_sa = nullptr, *d_b_fc = nullptr;

    // context
    float *ctx_frames = nullptr, *ctx_distrib = nullptr, *ctx_states = nullptr, *ctx_actions = nullptr;

    // activations
    float *enc0_o = nullptr, *enc1_o = nullptr, *enc2_o = nullptr, *enc3_o = nullptr;
    float *enc4_o = nullptr, *enc5_o = nullptr, *enc6_o = nullptr;
    float *c_state[7] = {nullptr}, *h_state[7][2] = {{nullptr}};
    double *st_enc0 = nullptr, *st_h[7] = {nullptr}, *st_enc6 = nullptr;
    float *sbias = nullptr, *fc_part = nullptr, *kern = nullptr;
    size_t lstm_elems[7] = {0};

    // predictions of the last rollout
    float *frames_all = nullptr, *distrib_all = nullptr, *states_all = nullptr;
    double *sums = nullptr;
    long long sums_step_stride = 0;

    std::vector<void *> allocs;

    // batch-1 buffers for tensors shared by all samples of a sub-batch (context de-duplication)
    bool dedup = true;
    std::vector<BatchView> shared_views;

    // persistent single-launch rollout (vf_persistent.h)
    bool persistent = false;
    int n_cu = 256;
    float *actions_buf = nullptr;
    struct SchedCache {                 // device copy of one schedule + the key it was built for
        PhaseDesc *d_phases = nullptr;
        int B = -1, groups = 0, offset = 0, items = 0, counters = 0, phases = 0;
        bool dedup = true;
        int32_t goal[2 * kMaxDesig] = {0};
        double flops = 0.0;
        size_t lds = 0;
        std::vector<int> types, nitems;
    } sched[2];                         // [0]: full rollout, [1]: shared units skipped (cached)
    int last_sched = 0;
    size_t sched_capacity = 0, counter_capacity = 0;
    int *d_sync = nullptr;              // [ticket, status, counters...]
    unsigned long long *d_stats = nullptr;  // debugging aid (VF_PERSIST_STATS): per-phase wait/run ticks
    int n_groups = 1, group_offset = 9;
    int persist_wgs_per_cu = 2;

    // cross-rollout cache of the shared (batch-1, context-only) units
    bool cache_shared = true, shared_valid = false;
    int shared_cfg = -1;

    // sub-batch streams (forked from / joined to the caller's stream inside vf_rollout)
    int n_sub = 1;
    std::vector<hipStream_t> sub_streams;
    std::vector<hipEvent_t> ev_join;
    hipEvent_t ev_fork = nullptr;

    // optional per-launch timing of the conv-LSTM kernel (HIP events on the launch stream)
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    double prof_flops = 0.0;        // algorithmic FLOPs of the launches bracketed so far
};

namespace vf {

template <typename T>
static int dev_alloc(vf_handle *h, T **p, size_t n) {
    void *q = nullptr;
    if (hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess)
        return fail(VF_ERR_NOMEM, "hipMalloc of " + std::to_string(n * sizeof(T)) + " bytes failed");
    h->allocs.push_back(q);
    *p = reinterpret_cast<T *>(q);
    return VF_OK;
}

static int validate(const vf_config *c) {
    if (!c) return fail(VF_ERR_INVALID, "null config");
    if (c->height <= 0 || c->width <= 0 || c->height % 8 || c->width % 8)
        return fail(VF_ERR_INVALID, "height/width must be positive multiples of 8");
    if (c->ndesig < 1 || c->ndesig > kMaxDesig) return fail(VF_ERR_INVALID, "ndesig must be 1..4");
    if (c->n_context < 1 || c->sequence_length <= c->n_context)
        return fail(VF_ERR_INVALID, "need n_context >= 1 and sequence_length > n_context");
    if (c->adim < 1 || c->sdim < 1 || c->adim + c->sdim > 32)
        return fail(VF_ERR_INVALID, "need adim, sdim >= 1 and adim + sdim <= 32");
    if (c->num_masks != 10) return fail(VF_ERR_INVALID, "num_masks must be 10 in this build");
    if (c->max_batch < 1) return fail(VF_ERR_INVALID, "max_batch must be >= 1");
    if (c->precision != 0 && c->precision != 1) return fail(VF_ERR_INVALID, "precision must be 0 (fp32) or 1 (split bf16)");
    return VF_OK;
}

static void init_layer(ConvLayer &l, const char *name, PackMode mode, int Hin, int Win, int Hout, int Wout,
                       int KH, int KW, int stride, int pad, int c0, int c1, int Cout, bool stats,
                       bool fc = false, int mrep = 1, int prec = 0) {
    l.name = name; l.mode = mode; l.G = (mode == PACK_PLAIN) ? 1 : 4; l.mrep = prec == 1 ? 1 : mrep; l.prec = prec;
    l.Hin = Hin; l.Win = Win; l.Hout = Hout; l.Wout = Wout;
    l.KH = KH; l.KW = KW; l.stride = stride; l.pad = pad;
    l.segC[0] = c0; l.segC[1] = c1; l.nseg = c1 > 0 ? 2 : 1;
    l.Cout = Cout; l.ncg = (Cout + 31) / 32;
    l.nsplit = 1; l.n_valid = Cout;
    plan_geometry(l, stats, fc);
    l.chunks_per_split = l.nchunk[0] + l.nchunk[1];
}

static int upload(vf_handle *h, float **dst, const float *src, size_t n) {
    int rc = dev_alloc(h, dst, n);
    if (rc) return rc;
    VF_HIP_CHECK(hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice));
    return VF_OK;
}

template <int G, int EPI, int MREP>
static int launch_conv_m(const ConvLayer &l, const ConvParams &p, hipStream_t st) {
    static size_t configured = 0;
    if (l.lds_bytes > configured) {
        VF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma_kernel<G, EPI, MREP>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds_bytes));
        configured = l.lds_bytes;
    }
    const int tiles = l.NI == 1 ? p.B * l.tilesY * l.tilesX : (p.B + l.NI - 1) / l.NI;
    dim3 grid(tiles, l.ncg, l.nsplit);
    hipLaunchKernelGGL((conv_mfma_kernel<G, EPI, MREP>), grid, dim3(kConvThreads), l.lds_bytes, st, p);
    VF_HIP_CHECK(hipGetLastError());
    return VF_OK;
}

// which (G, EPI, MREP) instances exist: LSTM in both tile heights, the FC with 256 rows (it has
// few rows and a long K), every other layer with 128-row tiles
template <int MREP>
static int launch_lstm_bf16x6(const ConvLayer &l, const ConvParams &p, hipStream_t st) {
    static size_t configured = 0;
    if (l.lds_bytes > configured) {
        VF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_lstm_bf16x6_kernel<MREP>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds_bytes));
        configured = l.lds_bytes;
    }
    const int tiles = l.NI == 1 ? p.B * l.tilesY * l.tilesX : (p.B + l.NI - 1) / l.NI;
    hipLaunchKernelGGL((conv_lstm_bf16x6_kernel<MREP>), dim3(tiles, l.ncg), dim3(kConvThreads), l.lds_bytes, st, p);
    VF_HIP_CHECK(hipGetLastError());
    return VF_OK;
}

template <int G, int EPI>
static int launch_conv_t(const ConvLayer &l, const ConvParams &p, hipStream_t st) {
    if constexpr (EPI == EPI_LSTM) {
        if (l.prec == 1) return launch_lstm_bf16x6<1>(l, p, st);        // 128-row tiles only
        return l.mrep == 1 ? launch_conv_m<G, EPI, 1>(l, p, st) : launch_conv_m<G, EPI, 2>(l, p, st);
    } else if constexpr (EPI == EPI_PARTIAL) {
        return launch_conv_m<G, EPI, 2>(l, p, st);
    } else {
        return launch_conv_m<G, EPI, 1>(l, p, st);
    }
}

struct SegArg {
    const float *ptr; long long bstride; const double *ln_part; long long ln_bstride; int ln_nparts; float ln_inv_n;
    const float *gamma, *beta; int gamma_mod; int relu;
};

static ConvParams make_params(const ConvLayer &l, int B, const SegArg &s0, const SegArg *s1) {
    ConvParams p;
    memset(&p, 0, sizeof(p));
    const SegArg *sa[2] = {&s0, s1};
    for (int s = 0; s < l.nseg; ++s) {
        p.seg[s].ptr = sa[s]->ptr; p.seg[s].bstride = sa[s]->bstride; p.seg[s].C = l.segC[s];
        p.seg[s].nchunk = l.nchunk[s];
        p.seg[s].ln_part = sa[s]->ln_part; p.seg[s].ln_nparts = sa[s]->ln_nparts;
        p.seg[s].ln_bstride = sa[s]->ln_bstride;
        p.seg[s].ln_inv_n = sa[s]->ln_inv_n;
        p.seg[s].gamma = sa[s]->gamma; p.seg[s].beta = sa[s]->beta;
        p.seg[s].gamma_mod = sa[s]->gamma_mod > 0 ? sa[s]->gamma_mod : 1;
        p.seg[s].relu = sa[s]->relu;
    }
    p.nseg = l.nseg; p.B = B;
    p.Hin = l.Hin; p.Win = l.Win; p.Hout = l.Hout; p.Wout = l.Wout;
    p.KH = l.KH; p.KW = l.KW; p.stride = l.stride; p.pad = l.pad; p.KC = l.KC;
    p.NI = l.NI; p.TH = l.TH; p.TW = l.TW; p.RPI = l.RPI; p.tilesY = l.tilesY; p.tilesX = l.tilesX;
    p.ncg = l.ncg; p.Cout = l.Cout; p.Wp = l.d_w; p.Wp16 = l.d_w16; p.bias = l.d_b;
    p.chunks_per_split = l.chunks_per_split; p.n_valid = l.n_valid;
    p.stats_nparts = l.stats_nparts;
    return p;
}

}  // namespace vf

// ================================================================== C ABI
extern "C" {

int vf_abi_version(void) { return VF_ABI_VERSION; }

const char *vf_last_error(void) { return g_last_error.c_str(); }

size_t vf_weight_count(const vf_config *cfg) {
    if (validate(cfg)) return 0;
    auto t = tensor_table(*cfg);
    return t.back().offset + t.back().size();
}

double vf_macs_per_sample_step(const vf_config *cfg) {
    if (validate(cfg)) return 0.0;
    const int H = cfg->height, W = cfg->width;
    auto t = tensor_table(*cfg);
    struct { const char *n; int h, w; } res[] = {
        {"enc0", H / 2, W / 2}, {"lstm1", H / 2, W / 2}, {"lstm2", H / 2, W / 2}, {"enc1", H / 4, W / 4},
        {"lstm3", H / 4, W / 4}, {"lstm4", H / 4, W / 4}, {"enc2", H / 8, W / 8}, {"enc3", H / 8, W / 8},
        {"lstm5", H / 8, W / 8}, {"convt1", H / 8, W / 8}, {"lstm6", H / 4, W / 4}, {"convt2", H / 4, W / 4},
        {"lstm7", H / 2, W / 2}, {"convt3", H / 2, W / 2}, {"rgb", H, W}, {"masks", H, W}};
    double macs = 0;
    for (auto &r : res) {
        const TensorDesc *d = find_tensor(t, std::string(r.n) + "/w");
        macs += (double)r.h * r.w * d->shape[0] * d->shape[1] * d->shape[2] * d->shape[3];
    }
    const TensorDesc *fc = find_tensor(t, "cdna/w"), *sw = find_tensor(t, "state/w");
    macs += (double)fc->shape[0] * fc->shape[1] + (double)sw->shape[0] * sw->shape[1];
    macs += (double)H * W * kTaps * (3 + cfg->ndesig) * cfg->num_masks;
    return macs;
}

int vf_create(const vf_config *cfg, vf_handle **out) {
    if (!out) return fail(VF_ERR_INVALID, "null out pointer");
    *out = nullptr;
    int rc = validate(cfg);
    if (rc) return rc;
    VF_HIP_CHECK(hipSetDevice(cfg->device));
    vf_handle *h = new vf_handle();
    h->cfg = *cfg;
    h->H = cfg->height; h->W = cfg->width; h->ND = cfg->ndesig; h->K = cfg->num_masks;
    h->T = cfg->sequence_length - cfg->n_context;
    h->S = h->T + cfg->n_context - 1;
    h->table = tensor_table(*cfg);
    const int H = h->H, W = h->W, Bc = cfg->max_batch, ND = h->ND;
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8;
    const int *L = kLstmSizes;
    h->ntiles = ((H + kCompTile - 1) / kCompTile) * ((W + kCompTile - 1) / kCompTile);

    // rows per LSTM workgroup: 128 (mrep 1) keeps items short - the per-sample dependency chain,
    // not the MFMA rate, bounds a 200-sample rollout; VF_LSTM_MREP=2222222 selects 256-row tiles
    int lstm_mrep[7] = {1, 1, 1, 1, 1, 1, 1};
    if (const char *e = getenv("VF_LSTM_MREP"))
        for (int k = 0; k < 7 && e[k]; ++k) lstm_mrep[k] = e[k] == '2' ? 2 : 1;
    init_layer(h->enc0, "enc0", PACK_PLAIN, H, W, H2, W2, 5, 5, 2, 1, 3, 0, 32, true);
    init_layer(h->lstm[0], "lstm1", PACK_LSTM, H2, W2, H2, W2, 5, 5, 1, 2, 32, L[0], L[0], true, false, lstm_mrep[0], cfg->precision);
    init_layer(h->lstm[1], "lstm2", PACK_LSTM, H2, W2, H2, W2, 5, 5, 1, 2, L[0], L[1], L[1], true, false, lstm_mrep[1], cfg->precision);
    init_layer(h->enc1, "enc1", PACK_PLAIN, H2, W2, H4, W4, 3, 3, 2, 0, L[1], 0, L[1], false);
    init_layer(h->lstm[2], "lstm3", PACK_LSTM, H4, W4, H4, W4, 5, 5, 1, 2, L[1], L[2], L[2], true, false, lstm_mrep[2], cfg->precision);
    init_layer(h->lstm[3], "lstm4", PACK_LSTM, H4, W4, H4, W4, 5, 5, 1, 2, L[2], L[3], L[3], true, false, lstm_mrep[3], cfg->precision);
    init_layer(h->enc2, "enc2", PACK_PLAIN, H4, W4, H8, W8, 3, 3, 2, 0, L[3], 0, L[3], false);
    init_layer(h->enc3, "enc3", PACK_PLAIN, H8, W8, H8, W8, 1, 1, 1, 0, L[3], 0, L[3], false);
    init_layer(h->lstm[4], "lstm5", PACK_LSTM, H8, W8, H8, W8, 5, 5, 1, 2, L[3], L[4], L[4], true, false, lstm_mrep[4], cfg->precision);
    init_layer(h->convt1, "convt1", PACK_CONVT, H8, W8, H8, W8, 2, 2, 1, 1, L[4], 0, L[4], false);
    init_layer(h->lstm[5], "lstm6", PACK_LSTM, H4, W4, H4, W4, 5, 5, 1, 2, L[4], L[5], L[5], true, false, lstm_mrep[5], cfg->precision);
    init_layer(h->convt2, "convt2", PACK_CONVT, H4, W4, H4, W4, 2, 2, 1, 1, L[5], L[1], L[5], false);
    init_layer(h->lstm[6], "lstm7", PACK_LSTM, H2, W2, H2, W2, 5, 5, 1, 2, L[5], L[6], L[6], true, false, lstm_mrep[6], cfg->precision);
    init_layer(h->convt3, "convt3", PACK_CONVT, H2, W2, H2, W2, 2, 2, 1, 1, L[6], 32, 32, true);
    // CDNA FC as a K-split GEMM over 1x1 "images"
    init_layer(h->fc, "cdna", PACK_PLAIN, 1, 1, 1, 1, 1, 1, 1, 0, H8 * W8 * L[4], 0, kTaps * h->K, false, true, 2);
    {
        ConvLayer &f = h->fc;
        const int total = f.nchunk[0];
        f.nsplit = std::min(32, total);
        f.chunks_per_split = (total + f.nsplit - 1) / f.nsplit;
        f.nsplit = (total + f.chunks_per_split - 1) / f.chunks_per_split;
        f.n_valid = kTaps * h->K;
    }

#define VF_ALLOC(ptr, n)                           \
    do {                                           \
        rc = dev_alloc(h, &(ptr), (size_t)(n));    \
        if (rc) { vf_destroy(h); return rc; }      \
    } while (0)

    const int nc = cfg->n_context;
    VF_ALLOC(h->ctx_frames, (size_t)nc * H * W * 3);
    VF_ALLOC(h->ctx_distrib, (size_t)nc * H * W * ND);
    VF_ALLOC(h->ctx_states, (size_t)nc * cfg->sdim);
    VF_ALLOC(h->ctx_actions, (size_t)std::max(nc - 1, 1) * cfg->adim);

    VF_ALLOC(h->enc0_o, (size_t)Bc * H2 * W2 * 32);
    VF_ALLOC(h->enc1_o, (size_t)Bc * H4 * W4 * L[1]);
    VF_ALLOC(h->enc2_o, (size_t)Bc * H8 * W8 * L[3]);
    VF_ALLOC(h->enc3_o, (size_t)Bc * H8 * W8 * L[3]);
    VF_ALLOC(h->enc4_o, (size_t)Bc * H4 * W4 * L[4]);
    VF_ALLOC(h->enc5_o, (size_t)Bc * H2 * W2 * L[5]);
    VF_ALLOC(h->enc6_o, (size_t)Bc * H * W * 32);
    const int lh[7] = {H2, H2, H4, H4, H8, H4, H2}, lw[7] = {W2, W2, W4, W4, W8, W4, W2};
    for (int k = 0; k < 7; ++k) {
        h->lstm_elems[k] = (size_t)Bc * lh[k] * lw[k] * L[k];
        VF_ALLOC(h->c_state[k], h->lstm_elems[k]);
        VF_ALLOC(h->h_state[k][0], h->lstm_elems[k]);
        VF_ALLOC(h->h_state[k][1], h->lstm_elems[k]);
        VF_ALLOC(h->st_h[k], (size_t)Bc * h->lstm[k].stats_nparts * 2);
    }
    VF_ALLOC(h->st_enc0, (size_t)Bc * h->enc0.stats_nparts * 2);
    VF_ALLOC(h->st_enc6, (size_t)Bc * h->convt3.stats_nparts * 2);
    VF_ALLOC(h->sbias, (size_t)Bc * L[3]);
    VF_ALLOC(h->fc_part, (size_t)h->fc.nsplit * Bc * kTaps * h->K);
    VF_ALLOC(h->kern, (size_t)Bc * kTaps * h->K);
    VF_ALLOC(h->frames_all, (size_t)Bc * h->T * H * W * 3);
    VF_ALLOC(h->distrib_all, (size_t)Bc * h->T * H * W * ND);
    VF_ALLOC(h->states_all, (size_t)Bc * h->T * cfg->sdim);
    h->sums_step_stride = (long long)Bc * ND * h->ntiles * 2;
    VF_ALLOC(h->sums, (size_t)h->T * h->sums_step_stride);
    VF_ALLOC(h->actions_buf, (size_t)Bc * h->T * cfg->adim);
    h->sched_capacity = ((size_t)h->S * 20 + 8) * kMaxSubBatches;
    h->counter_capacity = ((size_t)h->S * 20 + 8) * ((size_t)Bc + kMaxSubBatches);
    VF_ALLOC(h->sched[0].d_phases, h->sched_capacity);
    VF_ALLOC(h->sched[1].d_phases, h->sched_capacity);
    VF_ALLOC(h->d_sync, 2 + h->counter_capacity);
    if (hipMemset(h->d_sync, 0, 2 * sizeof(int)) != hipSuccess) {
        vf_destroy(h);
        return fail(VF_ERR_HIP, "hipMemset of the scheduler words failed");
    }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0)
            h->n_cu = prop.multiProcessorCount;
    }
    for (int i = 0; i < kMaxSubBatches; ++i) {
        BatchView sv;
        memset(&sv, 0, sizeof(sv));
        VF_ALLOC(sv.enc0_o, (size_t)H2 * W2 * 32);
        VF_ALLOC(sv.enc1_o, (size_t)H4 * W4 * L[1]);
        VF_ALLOC(sv.enc2_o, (size_t)H8 * W8 * L[3]);
        VF_ALLOC(sv.enc3_o, (size_t)H8 * W8 * L[3]);
        VF_ALLOC(sv.enc4_o, (size_t)H4 * W4 * L[4]);
        VF_ALLOC(sv.enc5_o, (size_t)H2 * W2 * L[5]);
        for (int k = 0; k < 7; ++k) {
            const size_t per = (size_t)lh[k] * lw[k] * L[k];
            VF_ALLOC(sv.c_state[k], per);
            VF_ALLOC(sv.h_state[k][0], per);
            VF_ALLOC(sv.h_state[k][1], per);
            VF_ALLOC(sv.st_h[k], (size_t)h->lstm[k].stats_nparts * 2);
        }
        VF_ALLOC(sv.st_enc0, (size_t)h->enc0.stats_nparts * 2);
        VF_ALLOC(sv.sbias, (size_t)L[3]);
        h->shared_views.push_back(sv);
    }
#undef VF_ALLOC
    *out = h;
    return VF_OK;
}

int vf_destroy(vf_handle *h) {
    if (!h) return VF_OK;
    for (void *p : h->allocs) (void)hipFree(p);
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->ev_join) (void)hipEventDestroy(e);
    for (hipStream_t s : h->sub_streams) (void)hipStreamDestroy(s);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    delete h;
    return VF_OK;
}

int vf_load_weights(vf_handle *h, const float *blob, size_t n_floats) {
    if (!h || !blob) return fail(VF_ERR_INVALID, "null handle or blob");
    const size_t want = h->table.back().offset + h->table.back().size();
    if (n_floats != want)
        return fail(VF_ERR_INVALID, "weight blob has " + std::to_string(n_floats) + " floats, expected " +
                                        std::to_string(want));
    VF_HIP_CHECK(hipSetDevice(h->cfg.device));
    int rc;
    auto T = [&](const std::string &name) { return find_tensor(h->table, name); };
    auto pack_layer = [&](ConvLayer &l) -> int {
        const TensorDesc *w = T(l.name + "/w"), *b = T(l.name + "/b");
        std::vector<float> wp, bp;
        if (l.name == "cdna") {
            wp = pack_weights(l, blob + w->offset, 1, 1, w->shape[0], w->shape[1]);
            bp.assign((size_t)l.ncg * 32, 0.f);     // bias is added by cdna_finalize
        } else if (l.name == "enc3") {
            // only the enc2 rows go through the GEMM; the action/state rows become a per-sample bias
            wp = pack_weights(l, blob + w->offset, 1, 1, w->shape[2], w->shape[3]);
            bp = pack_bias(l, blob + b->offset);
        } else {
            wp = pack_weights(l, blob + w->offset, w->shape[0], w->shape[1], w->shape[2], w->shape[3]);
            bp = pack_bias(l, blob + b->offset);
        }
        int r = upload(h, &l.d_w, wp.data(), wp.size());
        if (r) return r;
        if (l.prec == 1) {
            std::vector<unsigned short> w16 = pack_weights_bf16x3(l, blob + w->offset, w->shape[2], w->shape[3]);
            if ((r = dev_alloc(h, &l.d_w16, w16.size()))) return r;
            VF_HIP_CHECK(hipMemcpy(l.d_w16, w16.data(), w16.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
        }
        return upload(h, &l.d_b, bp.data(), bp.size());
    };
    ConvLayer *layers[] = {&h->enc0, &h->lstm[0], &h->lstm[1], &h->enc1, &h->lstm[2], &h->lstm[3], &h->enc2,
                           &h->enc3, &h->lstm[4], &h->convt1, &h->lstm[5], &h->convt2, &h->lstm[6],
                           &h->convt3, &h->fc};
    for (ConvLayer *l : layers)
        if ((rc = pack_layer(*l))) return rc;

    for (int i = 0; i < 9; ++i) {
        const std::string n = "ln" + std::to_string(i + 1);
        const TensorDesc *g = T(n + "/g"), *b = T(n + "/b");
        if ((rc = upload(h, &h->d_ln_g[i], blob + g->offset, g->size()))) return rc;
        if ((rc = upload(h, &h->d_ln_b[i], blob + b->offset, b->size()))) return rc;
    }
    const TensorDesc *d;
    d = T("rgb/w");   if ((rc = upload(h, &h->d_w_rgb, blob + d->offset, d->size()))) return rc;
    d = T("rgb/b");   if ((rc = upload(h, &h->d_b_rgb, blob + d->offset, d->size()))) return rc;
    d = T("masks/w"); if ((rc = upload(h, &h->d_w_mask, blob + d->offset, d->size()))) return rc;
    d = T("masks/b"); if ((rc = upload(h, &h->d_b_mask, blob + d->offset, d->size()))) return rc;
    d = T("state/w"); if ((rc = upload(h, &h->d_w_state, blob + d->offset, d->size()))) return rc;
    d = T("state/b"); if ((rc = upload(h, &h->d_b_state, blob + d->offset, d->size()))) return rc;
    d = T("cdna/b");  if ((rc = upload(h, &h->d_b_fc, blob + d->offset, d->size()))) return rc;
    // enc3 rows [L3 .. L3+adim+sdim) x 64: the smeared action/state inputs
    d = T("enc3/w");
    const int L3 = kLstmSizes[3];
    if ((rc = upload(h, &h->d_w_sa, blob + d->offset + (size_t)L3 * d->shape[3],
                     (size_t)(h->cfg.adim + h->cfg.sdim) * d->shape[3])))
        return rc;
    VF_HIP_CHECK(hipDeviceSynchronize());
    h->have_weights = true;
    h->shared_valid = false;
    return VF_OK;
}

int vf_set_context(vf_handle *h, const uint8_t *d_frames, const float *d_states, const float *d_ctx_actions,
                   const float *d_ctx_distrib, void *stream) {
    if (!h || !d_frames || !d_states || !d_ctx_distrib) return fail(VF_ERR_INVALID, "null argument");
    const int nc = h->cfg.n_context;
    if (nc > 1 && !d_ctx_actions) return fail(VF_ERR_INVALID, "context actions required when n_context > 1");
    VF_HIP_CHECK(hipSetDevice(h->cfg.device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int n_frames = nc * h->H * h->W * 3, n_d = nc * h->H * h->W * h->ND;
    const int n_s = nc * h->cfg.sdim, n_a = (nc - 1) * h->cfg.adim;
    const int n = std::max(std::max(n_frames, n_d), std::max(n_s, n_a));
    hipLaunchKernelGGL(set_context_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_frames, h->ctx_frames,
                       n_frames, d_states, h->ctx_states, n_s, d_ctx_actions, h->ctx_actions, n_a,
                       d_ctx_distrib, h->ctx_distrib, n_d);
    VF_HIP_CHECK(hipGetLastError());
    h->have_context = true;
    h->shared_valid = false;        // the shared units are functions of the context
    return VF_OK;
}

}  // extern "C"

static BatchView make_view(vf_handle *h, const float *d_actions, int b0) {
    const vf_config &c = h->cfg;
    const int H = h->H, W = h->W, T = h->T, ND = h->ND;
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8;
    const int *L = kLstmSizes;
    const int lh[7] = {H2, H2, H4, H4, H8, H4, H2}, lw[7] = {W2, W2, W4, W4, W8, W4, W2};
    const size_t b = (size_t)b0;
    BatchView v;
    v.enc0_o = h->enc0_o + b * H2 * W2 * 32;
    v.enc1_o = h->enc1_o + b * H4 * W4 * L[1];
    v.enc2_o = h->enc2_o + b * H8 * W8 * L[3];
    v.enc3_o = h->enc3_o + b * H8 * W8 * L[3];
    v.enc4_o = h->enc4_o + b * H4 * W4 * L[4];
    v.enc5_o = h->enc5_o + b * H2 * W2 * L[5];
    v.enc6_o = h->enc6_o + b * H * W * 32;
    for (int k = 0; k < 7; ++k) {
        const size_t per = (size_t)lh[k] * lw[k] * L[k];
        v.c_state[k] = h->c_state[k] + b * per;
        v.h_state[k][0] = h->h_state[k][0] + b * per;
        v.h_state[k][1] = h->h_state[k][1] + b * per;
        v.st_h[k] = h->st_h[k] + b * h->lstm[k].stats_nparts * 2;
    }
    v.st_enc0 = h->st_enc0 + b * h->enc0.stats_nparts * 2;
    v.st_enc6 = h->st_enc6 + b * h->convt3.stats_nparts * 2;
    v.sbias = h->sbias + b * L[3];
    v.fc_part = h->fc_part + b * h->fc.nsplit * kTaps * h->K;
    v.kern = h->kern + b * kTaps * h->K;
    v.frames_all = h->frames_all + b * T * H * W * 3;
    v.distrib_all = h->distrib_all + b * T * H * W * ND;
    v.states_all = h->states_all + b * T * c.sdim;
    v.sums = h->sums + b * ND * h->ntiles * 2;
    v.actions = d_actions + b * T * c.adim;
    return v;
}

// ------------------------------------------------------------------ rollout emission
// emit_rollout() walks the S steps of the predictor once and hands every unit of device work
// to a sink, together with the units it depends on.  LaunchSink enqueues one kernel per unit
// (stream order makes the dependencies implicit); ScheduleSink records the units as phases of
// the persistent launch (vf_persistent.h) with explicit per-sample dependencies.
static const int kSkipped = 1 << 30;      // id of a unit whose cached result is reused

struct LaunchSink {
    vf_handle *h;
    hipStream_t st;
    static int skipped() { return VF_OK; }

    int conv(int type, const ConvLayer &l, const ConvParams &p, std::initializer_list<int>) {
        switch (type) {
            case PH_LSTM: {
                if (!h->profiling) return launch_conv_t<4, EPI_LSTM>(l, p, st);
                while (h->ev_pool.size() < h->ev_used + 2) {
                    hipEvent_t e;
                    VF_HIP_CHECK(hipEventCreate(&e));
                    h->ev_pool.push_back(e);
                }
                VF_HIP_CHECK(hipEventRecord(h->ev_pool[h->ev_used], st));
                int r = launch_conv_t<4, EPI_LSTM>(l, p, st);
                VF_HIP_CHECK(hipEventRecord(h->ev_pool[h->ev_used + 1], st));
                h->ev_used += 2;
                h->prof_flops += 2.0 * p.B * l.Hout * l.Wout * 25.0 * (l.segC[0] + l.segC[1]) * 4.0 * l.Cout;
                return r;
            }
            case PH_CONV_RELU: return launch_conv_t<1, EPI_BIAS_RELU>(l, p, st);
            case PH_CONV_RAW: return launch_conv_t<1, EPI_RAW_STATS>(l, p, st);
            case PH_CONVT_RELU: return launch_conv_t<4, EPI_CONVT_RELU>(l, p, st);
            case PH_CONVT_RAW: return launch_conv_t<4, EPI_CONVT_RAW_STATS>(l, p, st);
            default: return launch_conv_t<1, EPI_PARTIAL>(l, p, st);
        }
    }
    int sa(const SaParams &p, std::initializer_list<int>) {
        hipLaunchKernelGGL(sa_kernel, dim3(p.B), dim3(64), 0, st, p);
        return VF_OK;
    }
    int fin(const FinParams &p, std::initializer_list<int>) {
        hipLaunchKernelGGL(cdna_finalize_kernel, dim3(p.B), dim3(256), 0, st, p);
        return VF_OK;
    }
    int composite(const CompositeParams &p, int ntiles, std::initializer_list<int>) {
        dim3 grid(ntiles, p.B);
        switch (p.ND) {
            case 1: hipLaunchKernelGGL((composite_kernel<1, 10>), grid, dim3(256), 0, st, p); break;
            case 2: hipLaunchKernelGGL((composite_kernel<2, 10>), grid, dim3(256), 0, st, p); break;
            case 3: hipLaunchKernelGGL((composite_kernel<3, 10>), grid, dim3(256), 0, st, p); break;
            default: hipLaunchKernelGGL((composite_kernel<4, 10>), grid, dim3(256), 0, st, p); break;
        }
        VF_HIP_CHECK(hipGetLastError());
        return VF_OK;
    }
    static bool failed(int rc) { return rc != VF_OK; }
};

struct ScheduleSink {
    std::vector<PhaseDesc> phases;
    int next_ticket = 0, next_counter = 0;      // tickets are re-assigned when groups are merged
    double flops = 0.0;         // algorithmic FLOPs of all MFMA (conv / FC) phases
    size_t max_lds = 0;

    int add(PhaseDesc &P, int n_items, int counters, std::initializer_list<int> deps) {
        P.first_ticket = next_ticket; P.n_items = n_items;
        P.cnt_base = next_counter;
        next_ticket += n_items; next_counter += counters;
        P.ndep = 0;
        for (int d : deps) {
            if (d < 0 || d == kSkipped) continue;
            const PhaseDesc &Q = phases[d];
            PhaseDep &dp = P.dep[P.ndep++];
            dp.cnt_base = Q.cnt_base;
            if (Q.whole) { dp.mode = 1; dp.expect = Q.n_items; }
            else {
                dp.mode = (Q.B == 1) ? 1 : 0;       // a batch-1 producer is shared by every sample
                switch (Q.type) {
                    case PH_SA: case PH_CDNA_FIN: dp.expect = 1; break;
                    case PH_COMPOSITE: dp.expect = Q.gx; break;
                    default: dp.expect = (Q.NI == 1 ? Q.tiles_per_img : 1) * Q.gy;
                }
            }
        }
        phases.push_back(P);
        return (int)phases.size() - 1;
    }
    int conv(int type, const ConvLayer &l, const ConvParams &p, std::initializer_list<int> deps) {
        PhaseDesc P;
        memset(&P, 0, sizeof(P));
        P.type = type; P.conv = p; P.B = p.B;
        P.NI = l.NI; P.tiles_per_img = l.tilesY * l.tilesX;
        P.gx = l.NI == 1 ? p.B * P.tiles_per_img : (p.B + l.NI - 1) / l.NI;
        P.gy = l.ncg;
        P.whole = type == PH_FC_PARTIAL;
        P.mrep = l.mrep;
        P.prec = l.prec;
        max_lds = std::max(max_lds, l.lds_bytes);
        const double rows = (double)p.B * l.Hout * l.Wout;
        const double taps = l.mode == PACK_CONVT ? 9.0 / 4.0 * 4.0 : (double)l.KH * l.KW;   // real taps
        flops += 2.0 * rows * taps * (l.segC[0] + (l.nseg > 1 ? l.segC[1] : 0)) *
                 (l.mode == PACK_LSTM ? 4.0 : 1.0) * l.Cout;
        return add(P, P.gx * P.gy * l.nsplit, P.whole ? 1 : p.B, deps);
    }
    int sa(const SaParams &p, std::initializer_list<int> deps) {
        PhaseDesc P;
        memset(&P, 0, sizeof(P));
        P.type = PH_SA; P.sa = p; P.B = p.B;
        return add(P, (p.B + kSaPerItem - 1) / kSaPerItem, p.B, deps);
    }
    int fin(const FinParams &p, std::initializer_list<int> deps) {
        PhaseDesc P;
        memset(&P, 0, sizeof(P));
        P.type = PH_CDNA_FIN; P.fin = p; P.B = p.B;
        return add(P, p.B, p.B, deps);
    }
    int composite(const CompositeParams &p, int ntiles, std::initializer_list<int> deps) {
        PhaseDesc P;
        memset(&P, 0, sizeof(P));
        P.type = PH_COMPOSITE; P.comp = p; P.B = p.B; P.gx = ntiles;
        return add(P, ntiles * p.B, p.B, deps);
    }
    static bool failed(int rc) { return rc < 0; }
    static int skipped() { return kSkipped; }
};

// Context de-duplication: while a step's inputs are the context, part of the network sees the
// same input for every sample - at steps s < n_context-1 everything (frame, action and state
// all come from the context), and at steps s < n_context the encoder up to enc2 (enc0, lstm1-4,
// enc1, enc2: the per-sample action only enters at enc3).  Those units run once with batch 1
// into the "shared" buffers and their consumers read them with batch stride 0; the arithmetic
// per sample is unchanged, so results are bit-identical to the redundant evaluation.
//
// The shared units depend only on the context and the weights, so they are also cached ACROSS
// rollouts: the CEM iterations of one planning call keep the same context, and every rollout
// after the first skips them (`skip_shared`) and reads the shared buffers of the first.
template <class Sink>
static int emit_rollout(vf_handle *h, const BatchView &v, const BatchView &sh, int B, const int32_t *goal_pix,
                        Sink &sink, bool skip_shared) {
    const vf_config &c = h->cfg;
    const int H = h->H, W = h->W, T = h->T, ND = h->ND, nc = c.n_context;
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8;
    const int *L = kLstmSizes;
    const int lh[7] = {H2, H2, H4, H4, H8, H4, H2}, lw[7] = {W2, W2, W4, W4, W8, W4, W2};

    auto all_shared = [&](int s) { return h->dedup && s < nc - 1; };
    auto enc_shared = [&](int s) { return h->dedup && s < nc; };
    // is the output of lstm k at step s one shared image?  (s < 0: the shared zero state)
    auto lstm_shared = [&](int k, int s) { return s < 0 || all_shared(s) || (k < 4 && enc_shared(s)); };

    auto plain = [](const float *ptr, long long bs) {
        SegArg s; memset(&s, 0, sizeof(s)); s.ptr = ptr; s.bstride = bs; s.gamma_mod = 1; return s;
    };
    auto normed = [](const float *ptr, long long bs, const double *part, int nparts, bool shared,
                     long long count, const float *g, const float *b, int gmod, int relu) {
        SegArg s; s.ptr = ptr; s.bstride = bs; s.ln_part = part; s.ln_nparts = nparts;
        s.ln_bstride = shared ? 0 : (long long)nparts * 2;
        s.ln_inv_n = (float)(1.0 / (double)count); s.gamma = g; s.beta = b; s.gamma_mod = gmod; s.relu = relu;
        return s;
    };
#define VF_EMIT(var, expr)                 \
    const int var = (expr);                \
    if (Sink::failed(var)) return var;
// a shared unit whose result is still valid from an earlier rollout is not emitted again
#define VF_EMIT_SH(var, shared, expr) VF_EMIT(var, ((shared) && skip_shared) ? Sink::skipped() : (expr))

    int last = -1;      // terminal unit of the previous step
    for (int s = 0; s < h->S; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        const bool produce = s >= nc - 1;
        const int t_out = s - (nc - 1);
        const bool enc_sh = enc_shared(s), all_sh = all_shared(s);
        const BatchView &E = enc_sh ? sh : v;       // encoder tensors of this step
        const BatchView &D = all_sh ? sh : v;       // tensors from enc3 on
        const int BE = enc_sh ? 1 : B, BD = all_sh ? 1 : B;
        auto bs = [](bool shared, long long per) { return shared ? 0LL : per; };

        // ---- state FC + action/state bias of enc3
        SaParams sp; memset(&sp, 0, sizeof(sp));
        if (s < nc - 1) { sp.action = h->ctx_actions + (size_t)s * c.adim; sp.action_bstride = 0; }
        else { sp.action = v.actions + (size_t)(s - (nc - 1)) * c.adim; sp.action_bstride = (long long)T * c.adim; }
        if (s < nc) { sp.state = h->ctx_states + (size_t)s * c.sdim; sp.state_bstride = 0; }
        else { sp.state = v.states_all + (size_t)(s - nc) * c.sdim; sp.state_bstride = (long long)T * c.sdim; }
        sp.adim = c.adim; sp.sdim = c.sdim; sp.B = BD;
        sp.w_state = h->d_w_state; sp.b_state = h->d_b_state; sp.w_sa = h->d_w_sa; sp.n_out = L[3];
        sp.state_out = produce ? v.states_all + (size_t)t_out * c.sdim : nullptr;
        sp.state_out_bstride = (long long)T * c.sdim;
        sp.sbias = D.sbias;
        VF_EMIT_SH(u_sa, all_sh, sink.sa(sp, {last}))

        // ---- encoder
        const float *frame_in; long long frame_bs;
        if (s < nc) { frame_in = h->ctx_frames + (size_t)s * H * W * 3; frame_bs = 0; }
        else { frame_in = v.frames_all + (size_t)(s - nc) * H * W * 3; frame_bs = (long long)T * H * W * 3; }

        ConvParams p = make_params(h->enc0, BE, plain(frame_in, frame_bs), nullptr);
        p.out = E.enc0_o; p.stats = E.st_enc0;
        VF_EMIT_SH(u_enc0, enc_sh, sink.conv(PH_CONV_RAW, h->enc0, p, {last}))

        SegArg enc0_n = normed(E.enc0_o, bs(enc_sh, (long long)H2 * W2 * 32), E.st_enc0, h->enc0.stats_nparts,
                               enc_sh, (long long)H2 * W2 * 32, h->d_ln_g[0], h->d_ln_b[0], 32, 1);
        // LayerNorm index: ln1 = enc0, ln2..ln8 = lstm1..7, ln9 = convt3
        auto h_normed = [&](int k) {        // normalised new hidden state of lstm k at this step
            const bool shd = lstm_shared(k, s);
            const BatchView &O = shd ? sh : v;
            const long long per = (long long)lh[k] * lw[k] * L[k];
            return normed(O.h_state[k][nxt], bs(shd, per), O.st_h[k], h->lstm[k].stats_nparts, shd, per,
                          h->d_ln_g[k + 1], h->d_ln_b[k + 1], L[k], 0);
        };
        auto lstm_params = [&](int k, const SegArg &x) {
            const bool out_sh = lstm_shared(k, s), in_sh = lstm_shared(k, s - 1);
            const BatchView &O = out_sh ? sh : v, &I = in_sh ? sh : v;
            const long long per = (long long)lh[k] * lw[k] * L[k];
            SegArg hs = plain(I.h_state[k][cur], bs(in_sh, per));
            ConvParams q = make_params(h->lstm[k], out_sh ? 1 : B, x, &hs);
            q.out = O.h_state[k][nxt]; q.cstate = O.c_state[k]; q.stats = O.st_h[k];
            q.cstate_in = I.c_state[k]; q.cin_bstride = bs(in_sh, per);
            return q;
        };
        VF_EMIT_SH(u_l1, lstm_shared(0, s), sink.conv(PH_LSTM, h->lstm[0], lstm_params(0, enc0_n), {u_enc0}))
        VF_EMIT_SH(u_l2, lstm_shared(1, s), sink.conv(PH_LSTM, h->lstm[1], lstm_params(1, h_normed(0)), {u_l1}))

        p = make_params(h->enc1, BE, h_normed(1), nullptr);
        p.out = E.enc1_o;
        VF_EMIT_SH(u_enc1, enc_sh, sink.conv(PH_CONV_RELU, h->enc1, p, {u_l2}))

        VF_EMIT_SH(u_l3, lstm_shared(2, s), sink.conv(PH_LSTM, h->lstm[2],
                                lstm_params(2, plain(E.enc1_o, bs(enc_sh, (long long)H4 * W4 * L[1]))), {u_enc1}))
        VF_EMIT_SH(u_l4, lstm_shared(3, s), sink.conv(PH_LSTM, h->lstm[3], lstm_params(3, h_normed(2)), {u_l3}))

        p = make_params(h->enc2, BE, h_normed(3), nullptr);
        p.out = E.enc2_o;
        VF_EMIT_SH(u_enc2, enc_sh, sink.conv(PH_CONV_RELU, h->enc2, p, {u_l4}))

        p = make_params(h->enc3, BD, plain(E.enc2_o, bs(enc_sh, (long long)H8 * W8 * L[3])), nullptr);
        p.out = D.enc3_o; p.sbias = D.sbias; p.sbias_ld = L[3];
        VF_EMIT_SH(u_enc3, all_sh, sink.conv(PH_CONV_RELU, h->enc3, p, {u_enc2, u_sa}))

        VF_EMIT_SH(u_l5, lstm_shared(4, s), sink.conv(PH_LSTM, h->lstm[4],
                                lstm_params(4, plain(D.enc3_o, bs(all_sh, (long long)H8 * W8 * L[3]))), {u_enc3}))
        SegArg h5n = h_normed(4);

        // ---- decoder
        p = make_params(h->convt1, BD, h5n, nullptr);
        p.out = D.enc4_o;
        VF_EMIT_SH(u_t1, all_sh, sink.conv(PH_CONVT_RELU, h->convt1, p, {u_l5}))
        VF_EMIT_SH(u_l6, lstm_shared(5, s), sink.conv(PH_LSTM, h->lstm[5],
                                lstm_params(5, plain(D.enc4_o, bs(all_sh, (long long)H4 * W4 * L[4]))), {u_t1}))

        SegArg enc1_s = plain(E.enc1_o, bs(enc_sh, (long long)H4 * W4 * L[1]));
        p = make_params(h->convt2, BD, h_normed(5), &enc1_s);
        p.out = D.enc5_o;
        VF_EMIT_SH(u_t2, all_sh, sink.conv(PH_CONVT_RELU, h->convt2, p, {u_l6}))
        VF_EMIT_SH(u_l7, lstm_shared(6, s), sink.conv(PH_LSTM, h->lstm[6],
                                lstm_params(6, plain(D.enc5_o, bs(all_sh, (long long)H2 * W2 * L[5]))), {u_t2}))
        last = u_l7;

        // ---- CDNA kernels (only needed when this step's prediction is used).  Emitted late: the FC
        // needs lstm5 of EVERY sample, and its only consumer is the compositing at the end of the step.
        int u_fin = -1;
        if (produce) {
            SegArg flat = h5n;      // same LayerNorm, viewed as [B][1][1][H8*W8*128]
            p = make_params(h->fc, B, flat, nullptr);
            p.out = v.fc_part;
            VF_EMIT(u_fc, sink.conv(PH_FC_PARTIAL, h->fc, p, {u_l5}))
            FinParams fp;
            fp.partial = v.fc_part; fp.nsplit = h->fc.nsplit; fp.B = B; fp.K = h->K;
            fp.bias = h->d_b_fc; fp.kern = v.kern;
            u_fin = sink.fin(fp, {u_fc});
            if (Sink::failed(u_fin)) return u_fin;
        }

        if (produce) {      // never an all-shared step
            p = make_params(h->convt3, B, h_normed(6), &enc0_n);
            p.out = v.enc6_o; p.stats = v.st_enc6;
            VF_EMIT(u_t3, sink.conv(PH_CONVT_RAW, h->convt3, p, {u_l7}))

            CompositeParams cp; memset(&cp, 0, sizeof(cp));
            cp.B = B; cp.H = H; cp.W = W; cp.ND = ND; cp.K = h->K;
            cp.enc6 = v.enc6_o; cp.ln_part = v.st_enc6; cp.ln_nparts = h->convt3.stats_nparts;
            cp.ln_inv_n = (float)(1.0 / ((double)H * W * 32));
            cp.gamma = h->d_ln_g[8]; cp.beta = h->d_ln_b[8];
            cp.w_rgb = h->d_w_rgb; cp.b_rgb = h->d_b_rgb; cp.w_mask = h->d_w_mask; cp.b_mask = h->d_b_mask;
            cp.kern = v.kern;
            cp.prev_frame = frame_in; cp.prev_frame_bstride = frame_bs;
            if (s < nc) {
                cp.prev_distrib = h->ctx_distrib + (size_t)s * H * W * ND; cp.prev_distrib_bstride = 0;
                cp.prev_sums = nullptr;
            } else {
                cp.prev_distrib = v.distrib_all + (size_t)(s - nc) * H * W * ND;
                cp.prev_distrib_bstride = (long long)T * H * W * ND;
                cp.prev_sums = v.sums + (long long)(s - nc) * h->sums_step_stride;
            }
            cp.out_frame = v.frames_all + (size_t)t_out * H * W * 3; cp.out_frame_bstride = (long long)T * H * W * 3;
            cp.out_distrib = v.distrib_all + (size_t)t_out * H * W * ND;
            cp.out_distrib_bstride = (long long)T * H * W * ND;
            cp.out_sums = v.sums + (long long)t_out * h->sums_step_stride;
            for (int d = 0; d < ND; ++d) { cp.goal[d][0] = goal_pix[2 * d]; cp.goal[d][1] = goal_pix[2 * d + 1]; }
            VF_EMIT(u_comp, sink.composite(cp, h->ntiles, {u_t3, u_fin}))
            last = u_comp;
        }
    }
#undef VF_EMIT_SH
#undef VF_EMIT
    return VF_OK;
}

// the zero initial LSTM state is one shared image per layer
static int zero_shared_state(vf_handle *h, const BatchView &sh, hipStream_t st) {
    const int H = h->H, W = h->W;
    const int *L = kLstmSizes;
    const int lh[7] = {H / 2, H / 2, H / 4, H / 4, H / 8, H / 4, H / 2};
    const int lw[7] = {W / 2, W / 2, W / 4, W / 4, W / 8, W / 4, W / 2};
    for (int k = 0; k < 7; ++k) {
        const size_t bytes = (size_t)lh[k] * lw[k] * L[k] * sizeof(float);
        VF_HIP_CHECK(hipMemsetAsync(sh.c_state[k], 0, bytes, st));
        VF_HIP_CHECK(hipMemsetAsync(sh.h_state[k][0], 0, bytes, st));
    }
    return VF_OK;
}

static int run_steps(vf_handle *h, const BatchView &v, const BatchView &sh, int B, const int32_t *goal_pix,
                     hipStream_t st, bool skip_shared) {
    if (!skip_shared) {
        int rc = zero_shared_state(h, sh, st);
        if (rc) return rc;
    }
    LaunchSink sink{h, st};
    return emit_rollout(h, v, sh, B, goal_pix, sink, skip_shared);
}

// Are the shared buffers of configuration `cfg` (launch mode and split) still valid?
static bool shared_cache_hit(vf_handle *h, int cfg) {
    return h->dedup && h->cache_shared && h->shared_valid && h->shared_cfg == cfg;
}

template <int ND, int WPS>
static int launch_persistent_w(vf_handle *h, const Schedule &sc, int grid, size_t lds, hipStream_t st) {
    static size_t configured = 0;
    if (lds > configured) {
        VF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&rollout_persistent_kernel<ND, WPS>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = lds;
    }
    hipLaunchKernelGGL((rollout_persistent_kernel<ND, WPS>), dim3(grid), dim3(kConvThreads), lds, st, sc.phases,
                       sc);
    VF_HIP_CHECK(hipGetLastError());
    return VF_OK;
}

template <int ND>
static int launch_persistent_t(vf_handle *h, const Schedule &sc, int grid, size_t lds, hipStream_t st) {
    // 3 resident workgroups per CU need the 168-VGPR build, which has no 256-row LSTM tile
    bool any_mrep2 = false;
    for (int k = 0; k < 7; ++k) any_mrep2 = any_mrep2 || h->lstm[k].mrep == 2;
    if (h->persist_wgs_per_cu >= 3 && !any_mrep2) return launch_persistent_w<ND, 3>(h, sc, grid, lds, st);
    return launch_persistent_w<ND, 2>(h, sc, grid, lds, st);
}

// the whole rollout as one persistent launch (vf_persistent.h)
static int run_persistent(vf_handle *h, const float *d_actions, int B, const int32_t *goal_pix, hipStream_t st) {
    int rc;
    // the schedule holds pointers into the handle's own action buffer, so it only depends on
    // (B, goal pixels, options) and is rebuilt when those change
    const size_t act_bytes = (size_t)B * h->T * h->cfg.adim * sizeof(float);
    VF_HIP_CHECK(hipMemcpyAsync(h->actions_buf, d_actions, act_bytes, hipMemcpyDeviceToDevice, st));
    const int ngroups = std::max(1, std::min({h->n_groups, kMaxSubBatches, B / 8}));
    const int cfg = 1000 + ngroups;
    const bool skip_shared = shared_cache_hit(h, cfg);
    vf_handle::SchedCache &sc_host = h->sched[skip_shared ? 1 : 0];
    bool rebuild = sc_host.B != B || sc_host.dedup != h->dedup || sc_host.groups != ngroups ||
                   sc_host.offset != h->group_offset;
    for (int d = 0; d < 2 * h->ND; ++d) rebuild = rebuild || sc_host.goal[d] != goal_pix[d];
    if (rebuild) {
        // Sample groups: each group of samples gets its own phase list (own shared buffers, own
        // counters); the lists are merged with a phase offset (tuning knob; 1 group by default).
        std::vector<ScheduleSink> sinks(ngroups);
        int counters = 0;
        double flops = 0.0;
        size_t max_lds = 0;
        for (int g = 0; g < ngroups; ++g) {
            const int b0 = (int)((long long)B * g / ngroups), b1 = (int)((long long)B * (g + 1) / ngroups);
            sinks[g].next_counter = counters;
            if ((rc = emit_rollout(h, make_view(h, h->actions_buf, b0), h->shared_views[g], b1 - b0, goal_pix,
                                   sinks[g], skip_shared)) < 0)
                return rc;
            counters = sinks[g].next_counter;
            flops += sinks[g].flops;
            max_lds = std::max(max_lds, sinks[g].max_lds);
        }
        std::vector<PhaseDesc> merged;
        std::vector<size_t> pos(ngroups, 0);
        for (long long vt = 0;; ++vt) {     // virtual time: group g runs its phase i at vt = i + g * offset
            bool any_left = false;
            for (int g = 0; g < ngroups; ++g) {
                const long long i = vt - (long long)g * h->group_offset;
                if (pos[g] < sinks[g].phases.size()) any_left = true;
                if (i >= 0 && (size_t)i == pos[g] && pos[g] < sinks[g].phases.size())
                    merged.push_back(sinks[g].phases[pos[g]++]);
            }
            if (!any_left) break;
        }
        int ticket = 0;
        for (PhaseDesc &P : merged) { P.first_ticket = ticket; ticket += P.n_items; }
        if (merged.size() > h->sched_capacity || (size_t)counters > h->counter_capacity)
            return fail(VF_ERR_INVALID, "persistent schedule exceeds its preallocated capacity");
        // an earlier rollout may still be reading the device copy
        VF_HIP_CHECK(hipStreamSynchronize(st));
        VF_HIP_CHECK(hipMemcpy(sc_host.d_phases, merged.data(), merged.size() * sizeof(PhaseDesc),
                               hipMemcpyHostToDevice));
        sc_host.B = B; sc_host.dedup = h->dedup; sc_host.groups = ngroups; sc_host.offset = h->group_offset;
        for (int d = 0; d < 2 * h->ND; ++d) sc_host.goal[d] = goal_pix[d];
        sc_host.items = ticket; sc_host.counters = counters;
        sc_host.phases = (int)merged.size();
        sc_host.types.clear(); sc_host.nitems.clear();
        for (const PhaseDesc &P : merged) { sc_host.types.push_back(P.type); sc_host.nitems.push_back(P.n_items); }
        sc_host.flops = flops;
        sc_host.lds = std::max(max_lds, (size_t)composite_lds_floats<kMaxDesig, 10>() * 4) + 16;
    }
    h->last_sched = skip_shared ? 1 : 0;
    if (!skip_shared)
        for (int g = 0; g < ngroups; ++g)
            if ((rc = zero_shared_state(h, h->shared_views[g], st))) return rc;
    VF_HIP_CHECK(hipMemsetAsync(h->d_sync, 0, (2 + (size_t)sc_host.counters) * sizeof(int), st));
    Schedule sc;
    sc.phases = sc_host.d_phases; sc.n_phases = sc_host.phases; sc.total_items = sc_host.items;
    sc.ticket = h->d_sync; sc.status = h->d_sync + 1; sc.counters = h->d_sync + 2;
    sc.stats = nullptr;
    sc.debug_no_fence = getenv("VF_DEBUG_NO_FENCE") ? 1 : 0;
    if (getenv("VF_PERSIST_STATS")) {
        if (!h->d_stats) {
            void *q = nullptr;
            VF_HIP_CHECK(hipMalloc(&q, h->sched_capacity * 2 * sizeof(unsigned long long)));
            h->allocs.push_back(q);
            h->d_stats = reinterpret_cast<unsigned long long *>(q);
        }
        VF_HIP_CHECK(hipMemsetAsync(h->d_stats, 0, h->sched_capacity * 2 * sizeof(unsigned long long), st));
        sc.stats = h->d_stats;
    }
    // resident workgroups per CU: bounded by the LDS a workgroup needs (160 KiB per CU)
    const int by_lds = (int)std::max<size_t>(1, (160 * 1024) / sc_host.lds);
    const int grid = std::min(sc_host.items, h->n_cu * std::min(h->persist_wgs_per_cu, by_lds));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->profiling) {
        while (h->ev_pool.size() < h->ev_used + 2) {
            hipEvent_t e;
            VF_HIP_CHECK(hipEventCreate(&e));
            h->ev_pool.push_back(e);
        }
        e0 = h->ev_pool[h->ev_used]; e1 = h->ev_pool[h->ev_used + 1];
        VF_HIP_CHECK(hipEventRecord(e0, st));
    }
    switch (h->ND) {
        case 1: rc = launch_persistent_t<1>(h, sc, grid, sc_host.lds, st); break;
        case 2: rc = launch_persistent_t<2>(h, sc, grid, sc_host.lds, st); break;
        case 3: rc = launch_persistent_t<3>(h, sc, grid, sc_host.lds, st); break;
        default: rc = launch_persistent_t<4>(h, sc, grid, sc_host.lds, st); break;
    }
    if (rc) return rc;
    if (h->profiling) {
        VF_HIP_CHECK(hipEventRecord(e1, st));
        h->ev_used += 2;
        h->prof_flops += sc_host.flops;
    }
    h->shared_valid = h->dedup;
    h->shared_cfg = cfg;
    return VF_OK;
}

extern "C" {

int vf_rollout(vf_handle *h, const float *d_actions, int32_t B, const int32_t *goal_pix, float finalweight,
               float *d_scores, float *d_scores_per_task, void *stream) {
    if (!h || !d_actions || !goal_pix || !d_scores) return fail(VF_ERR_INVALID, "null argument");
    if (!h->have_weights) return fail(VF_ERR_NOWEIGHTS, "vf_load_weights has not been called");
    if (!h->have_context) return fail(VF_ERR_NOCONTEXT, "vf_set_context has not been called");
    if (B < 1 || B > h->cfg.max_batch)
        return fail(VF_ERR_INVALID, "batch " + std::to_string(B) + " outside 1.." + std::to_string(h->cfg.max_batch));
    VF_HIP_CHECK(hipSetDevice(h->cfg.device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rc;

    // Samples never interact before the scores are compared, so the batch is cut into
    // sub-batches that advance on forked streams: while one sub-batch drains the tail of a
    // layer, the other's workgroups fill the idle CUs.  Results are bit-identical for any split.
    const int nsub = std::max(1, std::min({h->n_sub, B / 16, (int)h->sub_streams.size() + 1}));
    if (h->persistent) {
        if ((rc = run_persistent(h, d_actions, B, goal_pix, st))) return rc;
    } else if (nsub == 1) {
        const bool skip = shared_cache_hit(h, 1);
        if ((rc = run_steps(h, make_view(h, d_actions, 0), h->shared_views[0], B, goal_pix, st, skip))) return rc;
        h->shared_valid = h->dedup; h->shared_cfg = 1;
    } else {
        const bool skip = shared_cache_hit(h, 100 + nsub);
        VF_HIP_CHECK(hipEventRecord(h->ev_fork, st));
        for (int i = 0; i < nsub; ++i) {
            const int b0 = (int)((long long)B * i / nsub), b1 = (int)((long long)B * (i + 1) / nsub);
            hipStream_t ss = i == 0 ? st : h->sub_streams[i - 1];
            if (i > 0) VF_HIP_CHECK(hipStreamWaitEvent(ss, h->ev_fork, 0));
            if ((rc = run_steps(h, make_view(h, d_actions, b0), h->shared_views[i], b1 - b0, goal_pix, ss, skip)))
                return rc;
            if (i > 0) {
                VF_HIP_CHECK(hipEventRecord(h->ev_join[i - 1], ss));
                VF_HIP_CHECK(hipStreamWaitEvent(st, h->ev_join[i - 1], 0));
            }
        }
        h->shared_valid = h->dedup; h->shared_cfg = 100 + nsub;
    }
    hipLaunchKernelGGL(scores_kernel, dim3((B + 63) / 64), dim3(64), 0, st, h->sums, h->sums_step_stride, B,
                       h->T, h->ND, h->ntiles, finalweight, d_scores, d_scores_per_task);
    VF_HIP_CHECK(hipGetLastError());
    h->last_B = B;
    return VF_OK;
}

int vf_set_persistent(vf_handle *h, int32_t enable) {
    if (!h) return fail(VF_ERR_INVALID, "null handle");
    h->persistent = enable != 0;
    // tuning knobs of the persistent schedule (sample groups, their phase offset, resident
    // workgroups per CU); read once per switch so A/B runs need no rebuild
    if (const char *e = getenv("VF_GROUPS")) h->n_groups = std::max(1, atoi(e));
    if (const char *e = getenv("VF_GROUP_OFFSET")) h->group_offset = std::max(0, atoi(e));
    if (const char *e = getenv("VF_PERSIST_WGS_PER_CU")) h->persist_wgs_per_cu = std::max(1, std::min(4, atoi(e)));
    return VF_OK;
}

int vf_device_status(vf_handle *h, int32_t *status) {
    if (!h || !status) return fail(VF_ERR_INVALID, "null argument");
    VF_HIP_CHECK(hipSetDevice(h->cfg.device));
    VF_HIP_CHECK(hipMemcpy(status, h->d_sync + 1, sizeof(int), hipMemcpyDeviceToHost));
    return VF_OK;
}

// debugging aid: per-phase (type, items, wait ticks, run ticks) of the last persistent rollout
int vf_debug_phase_stats(vf_handle *h, int32_t max_phases, int32_t *types, int32_t *items, uint64_t *wait_run) {
    if (!h || !h->d_stats) return fail(VF_ERR_INVALID, "no phase statistics (set VF_PERSIST_STATS)");
    VF_HIP_CHECK(hipSetDevice(h->cfg.device));
    VF_HIP_CHECK(hipDeviceSynchronize());
    const vf_handle::SchedCache &sc = h->sched[h->last_sched];
    const int n = std::min<int>(max_phases, sc.phases);
    for (int i = 0; i < n; ++i) { types[i] = sc.types[i]; items[i] = sc.nitems[i]; }
    VF_HIP_CHECK(hipMemcpy(wait_run, h->d_stats, (size_t)n * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return n;
}

#ifdef VF_TILE_STATS
int vf_debug_tile_clocks(uint64_t *out /*[16][8]*/, int32_t reset) {
    if (hipDeviceSynchronize() != hipSuccess) return fail(VF_ERR_HIP, "sync failed");
    VF_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(vf::g_tile_clk), sizeof(uint64_t) * 128));
    if (reset) {
        static const uint64_t zeros[128] = {0};
        VF_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(vf::g_tile_clk), zeros, sizeof(zeros)));
    }
    return VF_OK;
}
#endif

int vf_set_dedup(vf_handle *h, int32_t enable) {
    if (!h) return fail(VF_ERR_INVALID, "null handle");
    h->dedup = enable != 0;
    h->shared_valid = false;
    if (const char *e = getenv("VF_CACHE_SHARED")) h->cache_shared = atoi(e) != 0;
    return VF_OK;
}

int vf_set_substreams(vf_handle *h, int32_t n) {
    if (!h || n < 1 || n > kMaxSubBatches) return fail(VF_ERR_INVALID, "sub-stream count must be 1..8");
    VF_HIP_CHECK(hipSetDevice(h->cfg.device));
    while ((int)h->sub_streams.size() < n - 1) {
        hipStream_t s;
        hipEvent_t e;
        VF_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        VF_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->sub_streams.push_back(s);
        h->ev_join.push_back(e);
    }
    if (!h->ev_fork) VF_HIP_CHECK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    h->n_sub = n;
    return VF_OK;
}

int vf_export(vf_handle *h, int32_t first, int32_t count, float *d_frames, float *d_distrib, float *d_states,
              void *stream) {
    if (!h) return fail(VF_ERR_INVALID, "null handle");
    if (first < 0 || count < 1 || first + count > h->last_B)
        return fail(VF_ERR_INVALID, "sample range outside the last rollout");
    VF_HIP_CHECK(hipSetDevice(h->cfg.device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t HW = (size_t)h->H * h->W;
    if (d_frames)
        VF_HIP_CHECK(hipMemcpyAsync(d_frames, h->frames_all + (size_t)first * h->T * HW * 3,
                                    (size_t)count * h->T * HW * 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (d_states)
        VF_HIP_CHECK(hipMemcpyAsync(d_states, h->states_all + (size_t)first * h->T * h->cfg.sdim,
                                    (size_t)count * h->T * h->cfg.sdim * sizeof(float), hipMemcpyDeviceToDevice,
                                    st));
    if (d_distrib) {
        const long long n = (long long)count * h->T * HW * h->ND;
        hipLaunchKernelGGL(export_distrib_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                           h->distrib_all, h->sums, h->sums_step_stride, first, count, h->T, (int)HW, h->ND,
                           h->ntiles, d_distrib);
        VF_HIP_CHECK(hipGetLastError());
    }
    return VF_OK;
}

int vf_set_profiling(vf_handle *h, int32_t enable) {
    if (!h) return fail(VF_ERR_INVALID, "null handle");
    h->profiling = enable != 0;
    h->ev_used = 0;
    h->prof_flops = 0.0;
    return VF_OK;
}

int vf_get_profile(vf_handle *h, double *kernel_ms, int64_t *launches, double *flops, double *busy_ms) {
    if (!h || !kernel_ms || !launches || !flops || !busy_ms) return fail(VF_ERR_INVALID, "null argument");
    VF_HIP_CHECK(hipSetDevice(h->cfg.device));
    double ms = 0.0;
    std::vector<std::pair<float, float>> spans;
    for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
        VF_HIP_CHECK(hipEventSynchronize(h->ev_pool[i + 1]));
        float dt = 0.f, t0 = 0.f;
        VF_HIP_CHECK(hipEventElapsedTime(&dt, h->ev_pool[i], h->ev_pool[i + 1]));
        if (i > 0) VF_HIP_CHECK(hipEventElapsedTime(&t0, h->ev_pool[0], h->ev_pool[i]));
        ms += dt;
        spans.emplace_back(t0, t0 + dt);
    }
    // time during which at least one bracketed launch was in flight (== kernel_ms on one stream)
    std::sort(spans.begin(), spans.end());
    double busy = 0.0, lo = 0.0, hi = -1.0;
    for (const auto &sp : spans) {
        if (hi < 0.0 || sp.first > hi) {
            if (hi >= 0.0) busy += hi - lo;
            lo = sp.first; hi = sp.second;
        } else if (sp.second > hi) {
            hi = sp.second;
        }
    }
    if (hi >= 0.0) busy += hi - lo;
    *kernel_ms = ms;
    *busy_ms = busy;
    *launches = (int64_t)(h->ev_used / 2);
    *flops = h->prof_flops;
    h->ev_used = 0;
    h->prof_flops = 0.0;
    return VF_OK;
}

}  // extern "C"
