// vf_engine.hip - host side of libvf_hip.so: device buffers, weight re-packing for the MFMA
// kernels, the per-step launch sequence of the CDNA predictor and the C ABI of include/vf_hip.h.
//
// Layer table and semantics: visual_foresight_amd/video_prediction/cdna_arch.py (the reference
// repo holds no network code).  Boundary semantics replaced here:
// visual_mpc/video_prediction/setup_predictor.py:98-114,164-200 and pred_util.py:4-48 of the
// reference (context slicing, /255, batch-1 context broadcast, per-sample action batch), the
// per-view networks of vpred_model_interface.py:60-88 (one weight set per camera, outputs stacked
// on a camera axis) and the registration arithmetic of
// visual_mpc/policy/cem_controllers/register_gtruth_controller.py:54-173.
//
// Memory contract (include/vf_hip.h): every device buffer - packed weights included - is sized
// from the config and allocated in vf_create(); later calls only fill them.  vf_rollout() never
// synchronises the caller's stream: a changed schedule is staged in a ring of pinned host
// buffers and uploaded with hipMemcpyAsync on that stream.
//
// -DVF_HOST_SELFTEST builds the same file for the host only (tools/host_selftest.cc, ASan/UBSan):
// device allocations become address reservations and uploads become checksums, so the packer
// and the schedule builder run - and are bounds-checked - without a GPU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

#include <dlfcn.h>
#ifdef VF_HOST_SELFTEST
#include <sys/mman.h>
#endif

#include "../../include/vf_hip.h"
#include "vf_conv_mfma.h"
#include "vf_small_kernels.h"
#include "vf_conv_bf16x6.h"
#include "vf_persistent.h"

#ifndef VF_WT_DEFAULT
#define VF_WT_DEFAULT 3             // write-through publish: bit 0 the conv-LSTM tiles, bit 1 the light layers (A/B: -DVF_WT_DEFAULT=n)
#endif

namespace vf {

static thread_local std::string g_last_error;

static int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

// "No exception crosses this boundary" (include/vf_hip.h): every extern "C" body runs inside VF_API_TRY / VF_API_CATCH.  The
// out-of-memory handler must not allocate: its message fits the small-string buffer of g_last_error.
template <class R> static R api_error(int code);
template <> int api_error<int>(int code) { return code; }
template <> size_t api_error<size_t>(int) { return 0; }
template <> double api_error<double>(int) { return 0.0; }
#define VF_API_TRY try {
#define VF_API_CATCH_CLEANUP(RET_, CLEANUP_)                                                               \
    } catch (const std::bad_alloc &) {                                                                     \
        CLEANUP_                                                                                           \
        g_last_error.assign("out of memory");                                                              \
        return api_error<RET_>(VF_ERR_NOMEM);                                                              \
    } catch (const std::exception &e_) {                                                                   \
        CLEANUP_                                                                                           \
        try { g_last_error = std::string("exception: ") + e_.what(); } catch (...) { g_last_error.assign("exception"); } \
        return api_error<RET_>(VF_ERR_INVALID);                                                            \
    } catch (...) {                                                                                        \
        CLEANUP_                                                                                           \
        g_last_error.assign("unknown exception");                                                          \
        return api_error<RET_>(VF_ERR_INVALID);                                                            \
    }
#define VF_API_CATCH(RET_) VF_API_CATCH_CLEANUP(RET_, {})

// Failure injection of the ASan / UBSan host build (tools/sanitize/host_selftest.cc): the next pass through injection point
// `where` throws - 0: std::bad_alloc, 1: std::runtime_error, 2: an int - so the tests can show that the status code and
// vf_last_error() come back, nothing leaks and the handle stays usable.
#ifdef VF_HOST_SELFTEST
#include <stdexcept>
static int g_inject_where = 0, g_inject_kind = 0;
static void inject_point(int where) {
    if (g_inject_where != where) return;
    g_inject_where = 0;
    if (g_inject_kind == 0) throw std::bad_alloc();
    if (g_inject_kind == 1) throw std::runtime_error("injected failure");
    throw 42;
}
#define VF_INJECT(W_) inject_point(W_)
#else
#define VF_INJECT(W_) do { } while (0)
#endif

#define VF_HIP_CHECK(expr)                                                                  \
    do {                                                                                    \
        hipError_t err_ = (expr);                                                           \
        if (err_ != hipSuccess)                                                             \
            return fail(VF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(err_));   \
    } while (0)

static const int kLstmSizes[7] = {32, 32, 64, 64, 128, 64, 32};
static const int kEnc00Ch = 16;             // channels of the extra encoder scale of arch 1 (savp_arch.py)
static const int kNumLn = 11;               // ln1..ln9, lna, lnb
static const int kSchedRing = 4;            // pinned staging buffers for schedule uploads
// ints in front of the completion counters: the ticket heads ...
static const int kTicketInts = kQueues * kTicketStride;
// ... and the per-CU state table of the cooperative priority scheme (vf_persistent.h: kCuKeys x kCuWords)
static const int kSyncHead = kTicketInts + kCuKeys * kCuWords;

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

static int a_of(const vf_config &c) { return c.adim + c.sdim; }

// arch 3 (the published SAVP generator): vf_engine_savp3.inc
struct Savp3;
static std::vector<TensorDesc> s3_tensor_table(const vf_config &c);
static double s3_macs(const vf_config &c);
static int s3_validate(const vf_config *c);

static std::vector<TensorDesc> tensor_table(const vf_config &c) {
    if (c.arch == 3) return s3_tensor_table(c);
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
    // arch 1 / 2 (savp_arch.py): one more encoder / decoder scale around the three-scale core; arch 2: the conditioning
    // vector [action, latent, state] is an input of every conv-LSTM (canonical channel order [x | cond | h])
    const bool savp = c.arch >= 1;
    const int cc = c.arch == 2 ? a_of(c) : 0;
    const int Hc = savp ? c.height / 2 : c.height, Wc = savp ? c.width / 2 : c.width;
    const int fc_in = (Hc / 8) * (Wc / 8) * L[4];
    // arch 0 with layer_spec = 1: the decoder widths of the PUBLIC CDNA prediction_model (arXiv:1605.07157's code keeps the
    // concatenated width through its transposed convs: convt2 96 -> 96, convt3 64 -> 64, so lstm7 reads 96 + 32 channels and
    // the heads 64); layer_spec = 0: the widths of SURVEY row a14 (convt2 96 -> 64, convt3 64 -> 32)
    const bool pub = c.arch == 0 && c.layer_spec == 1;
    const int c_t2 = pub ? L[5] + L[1] : L[5], c_top = pub ? L[6] + 32 : 32;
    if (savp) { conv("enc00", 5, 5, 3, kEnc00Ch); ln("lna", kEnc00Ch); }
    conv("enc0", 5, 5, savp ? kEnc00Ch : 3, 32); ln("ln1", 32);
    conv("lstm1", 5, 5, 32 + cc + L[0], 4 * L[0]);  ln("ln2", L[0]);
    conv("lstm2", 5, 5, L[0] + cc + L[1], 4 * L[1]); ln("ln3", L[1]);
    conv("enc1", 3, 3, L[1], L[1]);
    conv("lstm3", 5, 5, L[1] + cc + L[2], 4 * L[2]); ln("ln4", L[2]);
    conv("lstm4", 5, 5, L[2] + cc + L[3], 4 * L[3]); ln("ln5", L[3]);
    conv("enc2", 3, 3, L[3], L[3]);
    conv("enc3", 1, 1, L[3] + a, L[3]);
    conv("lstm5", 5, 5, L[3] + cc + L[4], 4 * L[4]); ln("ln6", L[4]);
    conv("convt1", 3, 3, L[4], L[4]);
    conv("lstm6", 5, 5, L[4] + cc + L[5], 4 * L[5]); ln("ln7", L[5]);
    conv("convt2", 3, 3, L[5] + L[1], c_t2);
    conv("lstm7", 5, 5, c_t2 + cc + L[6], 4 * L[6]); ln("ln8", L[6]);
    conv("convt3", 3, 3, L[6] + 32, c_top);     ln("ln9", c_top);
    if (savp) { conv("convt4", 3, 3, 32 + kEnc00Ch, 32); ln("lnb", 32); }
    conv("rgb", 1, 1, c_top, 3);
    conv("masks", 1, 1, c_top, K + 1);
    // arch 2: the FOUR CDNA kernels of the published generator (the engine pads them to its num_masks = 6 slots at load)
    const int KF = c.arch == 2 ? K - 2 : K;
    add("cdna/w", {fc_in, kTaps * KF});
    add("cdna/b", {kTaps * KF});
    add("state/w", {a, c.sdim});
    add("state/b", {c.sdim});
    return t;
}

// LayerNorm parameter slot i of ViewData: ln1..ln9, then lna (enc00) and lnb (convt4) of arch 1
static std::string ln_name(int i) { return i < 9 ? "ln" + std::to_string(i + 1) : (i == 9 ? "lna" : "lnb"); }

static const TensorDesc *find_tensor(const std::vector<TensorDesc> &t, const std::string &name) {
    for (const auto &d : t)
        if (d.name == name) return &d;
    return nullptr;
}

// ------------------------------------------------------------------ one dense layer
enum PackMode { PACK_PLAIN, PACK_LSTM, PACK_CONVT };
static const int kNumConvLayers = 24;

struct ConvLayer {          // geometry only: shared by every view; the packed weights are per view
    std::string name;
    int id = -1;                    // slot in ViewData::lw
    PackMode mode;
    int G;
    int Hin, Win, Hout, Wout;       // Hout/Wout: GEMM row grid
    int KH, KW, stride, pad;        // kernel geometry as the GEMM sees it
    int segC[2], nseg;
    int seg_off[2];                 // first canonical input channel of each segment (conv-LSTM: seg 0 is the
                                    // recurrent input h, which sits BEHIND the layer input x in the canonical
                                    // [x | h] channel order - the recurrent chunks lead the K order of every plan)
    int KC, nchunk[2];
    int mrep;                       // MFMA row blocks per wave: the workgroup covers 128 * mrep rows;
                                    // 0 / -1 = the 64- / 32-row conv-LSTM tiles (waves split rows x gates)
    int prec = 0;                   // 1: split-bf16 tile (conv-LSTM only)
    bool first_valu = false;        // the 5 x 5 / 2 conv on the 3-channel frame as a vector-ALU tile (vf_conv_first.h; mrep 8):
                                    // 16 x 16 output pixels per item, canonical [tap][channel][Cout] weights
    bool gs_v2 = false;              // ... in its final form (vf_conv_gsplit.h: one rolling weight register set)
    bool gsplit = false;            // 128-row fp32 conv-LSTM tile with 32-channel chunks: the gate-split tile (wave w =
                                    // gate w of all four row blocks, weights from L2 into registers, no barrier per tap)
    int NI, TH, TW, RPI, tilesY, tilesX;
    int ni_cap = 0;                 // > 0: at most this many whole images per workgroup (plans for narrow phases)
    int kc_cap = 0;                 // > 0: chunk size at most this (a narrow-phase plan that shares the regular plan's packed weights)
    int ncg, Cout;
    int nsplit, chunks_per_split, n_valid;
    int stats_nparts;               // partial sums this layer's epilogue writes per sample
    size_t lds_bytes;
    size_t packed_w() const {       // floats of the packed fp32 weights (pack_weights)
        if (first_valu) return (size_t)KH * KW * segC[0] * Cout;
        return (size_t)(nchunk[0] + nchunk[1]) * KH * KW * (KC / 8) * 2 * ((size_t)ncg * G * 32) * 4;
    }
    size_t packed_w16() const {     // bf16 values of the 3-plane split weights (pack_weights_bf16x3)
        return prec == 1 ? (size_t)(nchunk[0] + nchunk[1]) * KH * KW * ncg * 4 * 3 * 64 * 8 : 0;
    }
    size_t packed_b() const { return (size_t)ncg * G * 32; }
};

// can the top transposed conv `l` be fused with the compositing (vf_fused_top.h)?  One image and one channel
// group per tile, an output region of whole 4 x 16 cost-sum blocks, and the LDS of the fused epilogue
static bool top_fusable(const ConvLayer &l, int ND) {
    // (whole 4 x 16 cost-sum blocks in the image too: the fused epilogue stores a block row as 16-byte pieces)
    return l.NI == 1 && l.ncg == 1 && l.Cout == 32 && l.nsplit == 1 && l.TH * l.TW <= 128 &&
           (2 * l.TH) % kSumBlockH == 0 && (2 * l.TW) % kSumBlockW == 0 &&
           (2 * l.Hout) % kSumBlockH == 0 && (2 * l.Wout) % kSumBlockW == 0 &&
           fused_top_lds_floats(l.TH, l.TW, ND) * 4 <= 78 * 1024;
}

static int round_up(int x, int m) { return (x + m - 1) / m * m; }

static size_t conv_lds_bytes(const ConvLayer &l, int KC) {
    const int LH = (l.TH - 1) * l.stride + l.KH, LW = (l.TW - 1) * l.stride + l.KW;
    // A tile + LayerNorm table + reduction scratch (+ for 4-gate layers the double-buffered
    // per-tap B blocks: 2 x KC/8 x [4 gates][64 lanes] float4)
    const size_t b_lds = (l.mode == PACK_LSTM && l.mrep <= 1) ? (size_t)2 * (KC / 8) * 4 * 64 * 16 : 0;
    // conv-LSTM tiles: LayerNorm gain / offset of every input channel (conv_tile's gbTab)
    const size_t gb_lds = l.mode == PACK_LSTM ? (size_t)2 * round_up(l.segC[0] + (l.nseg > 1 ? l.segC[1] : 0), 4) * 4 : 0;
    const size_t need = ((size_t)l.NI * LH * LW * (KC + 4) + 4 * (size_t)l.NI) * 4 + 64 + gb_lds + b_lds;
    // (the light layers' epilogue turns the output tile over in four wave-private LDS slabs: conv_epilogue)
    return std::max(need, (size_t)vf::kEpiVecFloats * 4 + 64);
}

// choose tile shape and chunk size for a layer whose GEMM row grid is Hout x Wout
static void plan_geometry(ConvLayer &l, bool needs_stats, bool one_pixel_images) {
    const int rows = l.mrep == 0 ? 64 : (l.mrep < 0 ? 32 : 128 * l.mrep), wrows = l.mrep <= 0 ? 32 : 32 * l.mrep;
    if (one_pixel_images) {         // FC: every sample is a 1x1 image with many channels
        l.TH = l.TW = 1; l.tilesY = l.tilesX = 1; l.RPI = 1; l.NI = rows;
    } else {
        l.TW = std::min(l.Wout, 32);
        // the gate-split 128-row conv-LSTM tile (vf_conv_gsplit.h): 8 x 16 output pixels have a smaller halo than 4 x 32
        // (240 instead of 288 staged pixels per chunk)
        if (l.mode == PACK_LSTM && l.prec == 0 && l.mrep == 1 && l.KH == 5 && l.KW == 5) l.TW = std::min(l.Wout, 16);
#ifdef VF_DEBUG_KNOBS
        if (const char *e = getenv("VF_TILE_W")) l.TW = std::min(l.Wout, std::max(8, atoi(e)));
#endif
        l.TH = std::min(l.Hout, rows / l.TW);
        l.tilesX = (l.Wout + l.TW - 1) / l.TW;
        l.tilesY = (l.Hout + l.TH - 1) / l.TH;
        const int px = l.TH * l.TW;
        if (l.tilesX * l.tilesY == 1 && px <= rows / 2) {
            // several whole images per workgroup; with statistics every wave must sit inside one image
            l.RPI = needs_stats ? round_up(px, wrows) : px;
            l.NI = rows / l.RPI;
            if (l.ni_cap > 0) l.NI = std::min(l.NI, l.ni_cap);
        } else {
            l.RPI = rows; l.NI = 1;
        }
    }
    const int maxC = std::max(l.segC[0], l.nseg > 1 ? l.segC[1] : 0);
    int KC = 32;
    while (KC > 8 && (KC > round_up(maxC, 8) || conv_lds_bytes(l, KC) > 78 * 1024 || (l.kc_cap > 0 && KC > l.kc_cap) ||
                      l.segC[0] % KC || (l.nseg > 1 && l.segC[1] % KC)))
        KC >>= 1;
    if (l.prec == 1) KC = kBfKC;        // the split-bf16 tile stages 16-channel chunks
    l.KC = KC;
    for (int s = 0; s < 2; ++s) l.nchunk[s] = s < l.nseg ? (l.segC[s] + KC - 1) / KC : 0;
    l.lds_bytes = conv_lds_bytes(l, KC);
    // (the device code also has the 256-row variant - conv_lstm_gsplit_kernel<2>, eight row blocks per wave - but its
    // 128 accumulator registers + two weight sets + two operand sets spill inside the K loop: 75.1 vs 67.1 ms at C2,
    // so 256-row plans keep the weights-from-L2 tile; VF_GSPLIT256=1 in a -DVF_DEBUG_KNOBS build selects it)
    bool gs256 = false;
#ifdef VF_DEBUG_KNOBS
    if (const char *e = getenv("VF_GSPLIT256")) gs256 = atoi(e) != 0;
#endif
    bool gs64 = true;
#ifdef VF_DEBUG_KNOBS
    if (const char *e = getenv("VF_GSPLIT64")) gs64 = atoi(e) != 0;
#endif
    l.gsplit = l.mode == PACK_LSTM && (l.mrep == 1 || (gs64 && l.mrep == 0) || (gs256 && l.mrep == 2 && l.NI == 1)) &&
               l.prec == 0 && KC == 32 && l.KH == 5 && l.KW == 5;
    l.gs_v2 = false;
    if (l.gsplit) {
        // no weight buffers in LDS, but the epilogue's gate exchange (64 KiB over the dead operand tile; 32 KiB for the
        // 64-row tile) + its scratch
        const size_t b_lds = l.mrep <= 1 ? (size_t)2 * (KC / 8) * 4 * 64 * 16 : 0;
        const int LH = (l.TH - 1) * l.stride + l.KH, LW = (l.TW - 1) * l.stride + l.KW;
        // the final form of the 128-row tile (vf_conv_gsplit.h) stages ten elements per thread at most
        l.gs_v2 = l.mrep == 1 && l.stride == 1 && (size_t)l.NI * LH * LW <= 320;
#ifdef VF_DEBUG_KNOBS
        if (const char *e = getenv("VF_GSPLIT_V2")) l.gs_v2 = l.gs_v2 && atoi(e) != 0;
#else
        // production builds carry ONE 128-row gate-split tile (vf_conv_gsplit.h); a geometry it cannot stage (none of
        // the shipped networks has one) falls back to the weights-through-LDS tile instead of the first-generation
        // gate-split tile, which only exists in -DVF_DEBUG_KNOBS builds
        if (l.mrep == 1 && !l.gs_v2) l.gsplit = false;
#endif
        if (l.gsplit) l.lds_bytes = std::max(l.lds_bytes - b_lds, (size_t)vf::kGsXchFloats * 4 + 64);
    }
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
        const int seg_off = l.seg_off[s];
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
        const int seg_off = l.seg_off[s];
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

// Views of the engine's working buffers: one per camera view and one "shared" set of batch-1
// buffers per view for tensors that are identical for every sample (see emit_rollout).
struct BatchView {
    float *enc0_o, *enc1_o, *enc2_o, *enc3_o, *enc4_o, *enc5_o, *enc6_o;
    float *enc00_o, *enc7_o;            // arch 1: extra encoder / decoder scale
    float *c_state[7], *h_state[7][2];
    long long *st_enc0, *st_h[7], *st_enc6, *st_enc00, *st_enc7;
    float *sbias, *fc_part, *kern;
    float *cond_bias[7][2];             // (two buffers, by step parity: the biases of step s + 1 are computed while the
                                        //  epilogues of step s still read theirs)
    float *frames_all, *distrib_all, *states_all;
    double *sums;
    const float *actions;
};

// Device-resident parameters and context of one camera view (the reference's multi-view models
// are one network per view sharing actions and states, vpred_model_interface.py:60-88).
struct LayerW { float *w = nullptr, *b = nullptr; unsigned short *w16 = nullptr; };
struct ViewData {
    LayerW lw[kNumConvLayers];
    float *ln_g[kNumLn] = {nullptr}, *ln_b[kNumLn] = {nullptr};    // ln1..ln9, lna (enc00), lnb (convt4)
    float *w_rgb = nullptr, *b_rgb = nullptr, *w_mask = nullptr, *b_mask = nullptr;
    float *w_state = nullptr, *b_state = nullptr, *w_sa = nullptr, *b_fc = nullptr;
    float *w_cond[7] = {nullptr};       // arch 2: conditioning rows of every conv-LSTM's weights, [25][adim + sdim][4C]
    float *ctx_frames = nullptr, *ctx_distrib = nullptr;
};

struct AllocRec { void *p; size_t bytes; };

// ------------------------------------------------------------------ the engine
struct vf_handle {
    vf_config cfg;
    int H, W, T, S, ND, K;              // S = steps per rollout = T + n_context - 1
    bool savp = false;                  // vf_config.arch >= 1: four-scale SAVP-class generator (savp_arch.py)
    vf::Savp3 *s3 = nullptr;            // vf_config.arch == 3: the published SAVP generator (vf_engine_savp3.inc)
    bool cond = false;                  // vf_config.arch == 2: [action, latent, state] conditions every conv-LSTM
    float *cond_bias[7] = {nullptr};    // ... through per-sample border-class biases [2 step parities][ncam][max_batch][25][4C]
    int Hc, Wc;                         // input size of the three-scale conv-LSTM core (H, W; arch 1: H/2, W/2)
    int c_t2 = 64, c_top = 32;          // output channels of convt2 / convt3 (arch 0, layer_spec 1 - the public table: 96 / 64)
    int ncam = 1, n_draws = 1;
    int ntiles;                         // composite tiles per image
    int nblocks;                        // cost-sum blocks per image (4 x 16 pixels, vf_small_kernels.h)
    std::vector<TensorDesc> table;
    size_t blob_floats = 0;             // canonical floats per view
    bool have_weights = false, have_context = false;
    int last_B = 0;

    // layers (geometry)
    ConvLayer enc0, lstm[7], enc1, enc2, enc3, convt1, convt2, convt3, fc;
    ConvLayer fc_wide;                  // the FC as one item per (128-row tile, K split) with all column groups (vf_fc_tile.h)
    bool fc_wide_ok = false;
    ConvLayer enc00, convt4;            // arch 1 only
    // One-image-per-workgroup plans of the bottleneck's light layers (8x8 images: two fit a 128-row tile): twice the
    // items, each shorter - for batches whose phases do not fill the workgroup slots anyway (same packed weights,
    // same arithmetic per output)
    ConvLayer enc2_one, enc3_one, convt1_one;
    // Second tile plan of every conv-LSTM (256 GEMM rows per workgroup, weights read straight from L2 so that
    // the larger input tile fits the LDS with the SAME 32-channel chunks): fewer, longer items - better per
    // FLOP once a phase has far more items than workgroup slots, worse for the per-sample dependency chain of
    // a small batch.  Both plans accumulate every output in the same K order and LayerNorm statistics are
    // exact integers (vf_conv_mfma.h), so the choice is invisible in the results and may follow the batch size.
    ConvLayer lstm_big[7];
    bool have_big = false, big_ok[7] = {false};
    // Third and fourth plan: 64 and 32 GEMM rows per workgroup (conv_tile<..., RB = 2 / 1>), for batches so
    // small that a rollout is bound by the per-sample dependency chain: more items, each shorter.
    ConvLayer lstm_half[7], lstm_quarter[7];
    bool half_ok[7] = {false}, quarter_ok[7] = {false};
    int st_rows[7] = {0};               // LayerNorm partial-sum slots per sample of lstm k (max over its plans)
    int mrep_override[7] = {0};         // VF_DEBUG_KNOBS: 1 / 2 / 3 (64 rows) forces a plan, 0 = automatic
    std::vector<ConvLayer *> layers;    // in slot order
    std::vector<ViewData> views;

    // context: frames / distributions [ncam][n_context][..] (views[v] points into them), states and
    // executed actions shared by the views
    float *ctx_frames_all = nullptr, *ctx_distrib_all = nullptr;
    float *ctx_states = nullptr, *ctx_actions = nullptr;

    // activations, [ncam][max_batch] samples each
    float *enc0_o = nullptr, *enc1_o = nullptr, *enc2_o = nullptr, *enc3_o = nullptr;
    float *enc4_o = nullptr, *enc5_o = nullptr, *enc6_o = nullptr, *enc00_o = nullptr, *enc7_o = nullptr;
    float *c_state[7] = {nullptr}, *h_state[7][2] = {{nullptr}};
    long long *st_enc0 = nullptr, *st_h[7] = {nullptr}, *st_enc6 = nullptr, *st_enc00 = nullptr, *st_enc7 = nullptr;
    float *sbias = nullptr, *fc_part = nullptr, *kern = nullptr;

    // predictions of the last rollout
    float *frames_all = nullptr, *distrib_all = nullptr, *states_all = nullptr;
    double *sums = nullptr;
    long long sums_step_stride = 0, sums_view_stride = 0;

    std::vector<AllocRec> allocs;

    // batch-1 buffers for tensors shared by all samples (context de-duplication): [view]
    bool dedup = true;
    std::vector<BatchView> shared_views;

    // persistent single-launch rollout (vf_persistent.h)
    bool persistent = false;
    int n_cu = 256;
    float *actions_buf = nullptr;
    struct SchedCache {                 // device copy of one schedule + the key it was built for
        PhaseDesc *d_phases = nullptr;
        int B = -1, items = 0, counters = 0, phases = 0;
        bool dedup = true;
        int xcd_queues = 0, nq = 1, total_q[kQueues] = {0};
        int options = -1;               // sched_options() the schedule was built with (every toggle build_schedule reads)
        double flops = 0.0;
        size_t lds = 0;
        std::vector<int> types, nitems;
    } sched[2];                         // [0]: full rollout, [1]: shared units skipped (cached)
    int last_sched = 0;
    size_t sched_capacity = 0, counter_capacity = 0;
    PhaseDesc *stage[kSchedRing] = {nullptr};   // pinned host staging of schedule uploads
    hipEvent_t stage_done[kSchedRing] = {nullptr};
    bool stage_used[kSchedRing] = {false};
    int stage_next = 0;
    int *d_sync = nullptr;              // [kQueues ticket heads, one cache line each | counters...]
    int xcd_queues = kQueues;           // ticket queues of the persistent launch (vf_set_xcd_queues): kQueues or 1
    bool early_start = true;            // conv-LSTM items start on h(s-1) alone and wait for x(s) mid-item (ConvParams::late_cnt)
    bool fuse_top = true;               // vf_set_fuse_top: top transposed conv + compositing as one item (vf_fused_top.h)
    bool fuse_pair = true;              // ... and enc2 + enc3 as one item (conv_pair_epilogue); follows vf_set_fuse_top
    bool wt_publish = VF_WT_DEFAULT != 0;   // conv-LSTM tiles publish write-through (ConvParams::wt_out, no release fence)
    int yield_budget = -1;              // cooperative CU priority ("yielding", vf_conv_mfma.h): polls an early-started conv-LSTM
                                        // item may spend yielding to its CU partner; 0 = off, -1 = by batch size (yield_for)
    bool pair_allowed = true;           // (-DVF_DEBUG_KNOBS: VF_FUSE_PAIR=0, read once in vf_create)
    int *d_status = nullptr;            // sticky failure word of the persistent kernel
    unsigned long long *d_stats = nullptr;  // per-phase wait/run ticks (vf_set_phase_stats)
    bool phase_stats = false;
    int persist_wgs_per_cu = 2;
    size_t max_lds = 0;                 // largest dynamic LDS any tile of this engine needs

    // cross-rollout cache of the shared (batch-1, context-only) units
    bool cache_shared = true, shared_valid = false;
    int shared_cfg = -1;

    // optional per-launch timing of the dominant kernel (HIP events on the launch stream)
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    double prof_flops = 0.0;        // algorithmic FLOPs of the launches bracketed so far

#ifdef VF_HOST_SELFTEST
    char *fake_base = nullptr;
    size_t fake_used = 0, fake_size = 0;
    unsigned long long upload_checksum = 0;
#endif
};

namespace vf {

template <typename T>
static int dev_alloc(vf_handle *h, T **p, size_t n) {
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    void *q = nullptr;
#ifdef VF_HOST_SELFTEST
    // address reservation only: the self-test never dereferences a "device" pointer
    const size_t aligned = (bytes + 255) & ~(size_t)255;
    if (h->fake_used + aligned > h->fake_size) return fail(VF_ERR_NOMEM, "self-test address space exhausted");
    q = h->fake_base + h->fake_used;
    h->fake_used += aligned;
#else
    if (hipMalloc(&q, bytes) != hipSuccess)
        return fail(VF_ERR_NOMEM, "hipMalloc of " + std::to_string(bytes) + " bytes failed");
#endif
    h->allocs.push_back({q, bytes});
    *p = reinterpret_cast<T *>(q);
    return VF_OK;
}

// host -> device copy of freshly packed parameters (blocking; vf_load_weights only)
static int dev_write(vf_handle *h, void *dst, const void *src, size_t bytes) {
#ifdef VF_HOST_SELFTEST
    const unsigned char *s = static_cast<const unsigned char *>(src);
    unsigned long long acc = h->upload_checksum;
    for (size_t i = 0; i < bytes; ++i) acc = acc * 1099511628211ull + s[i];
    h->upload_checksum = acc;
    (void)dst;
#else
    (void)h;
    VF_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
#endif
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
    if (c->arch == 3) {
        if (c->max_batch < 1) return fail(VF_ERR_INVALID, "max_batch must be >= 1");
        if (c->ncam < 0 || c->ncam > kMaxCam) return fail(VF_ERR_INVALID, "ncam must be 1..4 (0 = 1)");
        if (c->n_draws < 0) return fail(VF_ERR_INVALID, "n_draws must be >= 1 (0 = 1)");
        if (c->n_draws > 1 && c->max_batch % c->n_draws) return fail(VF_ERR_INVALID, "max_batch must be a multiple of n_draws");
        return s3_validate(c);
    }
    if (c->zdim != 0) return fail(VF_ERR_INVALID, "zdim belongs to arch 3 (must be 0 otherwise)");
    if (c->layer_spec != 0 && !(c->arch == 0 && c->layer_spec == 1))
        return fail(VF_ERR_INVALID, "layer_spec: arch 3's layer table, or 1 with arch 0 (the public CDNA decoder widths); 0 otherwise");
    if (c->arch == 0 && c->layer_spec == 1 && c->precision != 0)
        return fail(VF_ERR_INVALID, "the public decoder table (arch 0, layer_spec 1) is built for precision 0 (exact fp32) only");
    if (c->arch == 2 ? c->num_masks != 6 : c->num_masks != 10)
        return fail(VF_ERR_INVALID, "num_masks must be 10 (arch 0 / 1), 6 (arch 2: four CDNA warps + previous + first + scratch) or 4 (arch 3)");
    if (c->max_batch < 1) return fail(VF_ERR_INVALID, "max_batch must be >= 1");
    if (c->precision != 0 && c->precision != 1) return fail(VF_ERR_INVALID, "precision must be 0 (fp32) or 1 (split bf16)");
    if (c->ncam < 0 || c->ncam > kMaxCam) return fail(VF_ERR_INVALID, "ncam must be 1..4 (0 = 1)");
    if (c->n_draws < 0) return fail(VF_ERR_INVALID, "n_draws must be >= 1 (0 = 1)");
    if (c->n_draws > 1 && c->max_batch % c->n_draws)
        return fail(VF_ERR_INVALID, "max_batch must be a multiple of n_draws");
    if (c->arch < 0 || c->arch > 3)
        return fail(VF_ERR_INVALID, "arch must be 0 (CDNA), 1 (SAVP-class, four scales), 2 (1 + per-layer conditioning, "
                                    "published compositing) or 3 (the published SAVP generator)");
    if (c->arch >= 1 && (c->height % 16 || c->width % 16))
        return fail(VF_ERR_INVALID, "arch 1 / 2 need height/width that are multiples of 16");
    if (c->arch == 2 && c->precision != 0)
        return fail(VF_ERR_INVALID, "arch 2 is built for precision 0 (exact fp32) only");
    if (c->arch == 2 && (c->height < 64 || c->width < 64))
        return fail(VF_ERR_INVALID, "arch 2 needs images of at least 64 x 64 (border classes of the 8 x 8 bottleneck)");
    return VF_OK;
}

static void init_layer(ConvLayer &l, const char *name, PackMode mode, int Hin, int Win, int Hout, int Wout,
                       int KH, int KW, int stride, int pad, int c0, int c1, int Cout, bool stats,
                       bool fc = false, int mrep = 1, int prec = 0, bool second_first = false, int cond_ch = 0) {
    l.name = name; l.mode = mode; l.G = (mode == PACK_PLAIN) ? 1 : 4; l.mrep = prec == 1 ? 1 : mrep; l.prec = prec;
    l.Hin = Hin; l.Win = Win; l.Hout = Hout; l.Wout = Wout;
    l.KH = KH; l.KW = KW; l.stride = stride; l.pad = pad;
    l.segC[0] = c0; l.segC[1] = c1; l.nseg = c1 > 0 ? 2 : 1;
    l.seg_off[0] = 0; l.seg_off[1] = c0;
    // callers pass the channel counts in canonical (concatenation) order; the recurrent input of a conv-LSTM and the
    // encoder skip tensor of a decoder conv (second_first) become segment 0: they exist early (ConvParams::late_cnt)
    if (mode == PACK_LSTM || (second_first && c1 > 0)) {
        l.segC[0] = c1; l.segC[1] = c0;
        l.seg_off[0] = c0 + cond_ch; l.seg_off[1] = 0;     // (arch 2: the conditioning rows sit between x and h and are NOT
    }                                                       //  part of the GEMM: CondParams, vf_small_kernels.h)
    l.Cout = Cout; l.ncg = (Cout + 31) / 32;
    l.nsplit = 1; l.n_valid = Cout;
    plan_geometry(l, stats, fc);
    l.chunks_per_split = l.nchunk[0] + l.nchunk[1];
    // (arch 2: the 128-row gate-split tile adds the conditioning biases through two LDS tables behind its exchange buffer)
    if (cond_ch > 0 && l.gs_v2) l.lds_bytes = std::max(l.lds_bytes, (size_t)vf::kGsCondFloats * 4 + 64);
    // the first conv of the encoder (3-channel frame in, exact statistics out): one thread per output pixel on the vector
    // ALUs instead of a K = 75 GEMM padded to 200 on the matrix pipe (vf_conv_first.h)
    bool first = mode == PACK_PLAIN && !fc && stats && l.nseg == 1 && c0 == vf::kFirstCin && KH == vf::kFirstK &&
                 KW == vf::kFirstK && stride == vf::kFirstStride && prec == 0 && (Cout == 16 || Cout == 32);
#ifdef VF_DEBUG_KNOBS
    if (const char *e = getenv("VF_FIRST_VALU")) first = first && atoi(e) != 0;
#endif
    if (first) {
        l.first_valu = true; l.mrep = 8;
        l.TW = std::min(Wout, 16); l.TH = std::min(Hout, vf::kConvThreads / l.TW);
        l.tilesX = (Wout + l.TW - 1) / l.TW; l.tilesY = (Hout + l.TH - 1) / l.TH;
        l.NI = 1; l.RPI = l.TH * l.TW;
        l.stats_nparts = l.tilesY * l.tilesX;
        l.lds_bytes = vf::first_lds_floats(l.TH, l.TW) * 4 + 64;
    }
}

// every kernel instance may be asked for up to h->max_lds bytes of dynamic LDS; the attribute is
// per device, so it is (re)applied by every vf_create on that handle's device
template <class K>
static int allow_lds(K kernel, size_t bytes) {
#ifndef VF_HOST_SELFTEST
    VF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
#else
    (void)kernel; (void)bytes;
#endif
    return VF_OK;
}

#ifndef VF_HOST_SELFTEST
static int configure_kernels(vf_handle *h) {
    const size_t n = h->max_lds;
    int rc;
    if ((rc = allow_lds(&conv_mfma_kernel<4, EPI_LSTM, 1>, n))) return rc;
#ifdef VF_DEBUG_KNOBS
    if ((rc = allow_lds(&conv_mfma_kernel<4, EPI_LSTM, 2>, n))) return rc;
    if ((rc = allow_lds(&conv_lstm_gsplit_kernel<1>, n))) return rc;
    if ((rc = allow_lds(&conv_lstm_gsplit_kernel<2>, n))) return rc;
    if ((rc = allow_lds(&conv_lstm_split_kernel<2>, n))) return rc;
#endif
    if ((rc = allow_lds(&conv_mfma_kernel<1, EPI_BIAS_RELU, 1>, n))) return rc;
    if ((rc = allow_lds(&conv_mfma_kernel<1, EPI_RAW_STATS, 1>, n))) return rc;
    if ((rc = allow_lds(&conv_mfma_kernel<4, EPI_CONVT_RELU, 1>, n))) return rc;
    if ((rc = allow_lds(&conv_mfma_kernel<4, EPI_CONVT_RAW_STATS, 1>, n))) return rc;
    if ((rc = allow_lds(&conv_mfma_kernel<1, EPI_PARTIAL, 2>, n))) return rc;
    if ((rc = allow_lds(&conv_lstm_bf16x6_kernel<1>, n))) return rc;
    if ((rc = allow_lds(&conv_lstm_gsplit64_kernel, n))) return rc;
    if ((rc = allow_lds(&conv_lstm_gsplit2_kernel<4>, n))) return rc;
    if ((rc = allow_lds(&conv_lstm_split_kernel<1>, n))) return rc;
    if ((rc = allow_lds(&conv_mfma_kernel<1, EPI_RAW, 1>, n))) return rc;
    if ((rc = allow_lds(&conv_gates_raw_kernel, n))) return rc;
    if ((rc = allow_lds(&conv_mfma_kernel<2, EPI_RAW, 1>, n))) return rc;
    if ((rc = allow_lds(&conv_mfma_kernel<1, EPI_RAW, 2>, n))) return rc;
    if ((rc = allow_lds(&conv_mfma_kernel<2, EPI_RAW, 2>, n))) return rc;
    if ((rc = allow_lds(&conv_mfma_kernel<4, EPI_RAW, 1>, n))) return rc;
    if ((rc = allow_lds(&ew_kernel<1>, ew_lds_bytes(EW_TOP3, 1)))) return rc;
    if ((rc = allow_lds(&ew_kernel<2>, ew_lds_bytes(EW_TOP3, 2)))) return rc;
    if ((rc = allow_lds(&ew_kernel<3>, ew_lds_bytes(EW_TOP3, 3)))) return rc;
    if ((rc = allow_lds(&ew_kernel<4>, ew_lds_bytes(EW_TOP3, 4)))) return rc;
    const size_t np = n + kCtlWords * sizeof(int);
    if ((rc = allow_lds(&rollout_persistent_kernel<1>, np))) return rc;
    if ((rc = allow_lds(&rollout_persistent_kernel<2>, np))) return rc;
    if ((rc = allow_lds(&rollout_persistent_kernel<3>, np))) return rc;
    if ((rc = allow_lds(&rollout_persistent_kernel<4>, np))) return rc;
    return VF_OK;
}

template <int G, int EPI, int MREP>
static int launch_conv_m(const ConvLayer &l, const ConvParams &p, hipStream_t st) {
    const int tiles = l.NI == 1 ? p.B * l.tilesY * l.tilesX : (p.B + l.NI - 1) / l.NI;
    dim3 grid(tiles, l.ncg, l.nsplit);
    hipLaunchKernelGGL((conv_mfma_kernel<G, EPI, MREP>), grid, dim3(kConvThreads), l.lds_bytes, st, p);
    VF_HIP_CHECK(hipGetLastError());
    return VF_OK;
}

static int launch_lstm_split(const ConvLayer &l, const ConvParams &p, hipStream_t st) {
    const int tiles = l.NI == 1 ? p.B * l.tilesY * l.tilesX : (p.B + l.NI - 1) / l.NI;
    // (tiles no production plan selects - the first-generation gate-split tiles and the 64-row tile with its weights
    // through LDS - are compiled into -DVF_DEBUG_KNOBS builds only; plan_geometry / vf_create never plan them otherwise)
    if (l.gs_v2)
        hipLaunchKernelGGL(conv_lstm_gsplit2_kernel<4>, dim3(tiles, l.ncg), dim3(kConvThreads), l.lds_bytes, st, p);
    else if (l.gsplit && l.mrep == 0)
        hipLaunchKernelGGL(conv_lstm_gsplit64_kernel, dim3(tiles, l.ncg), dim3(kConvThreads), l.lds_bytes, st, p);
#ifdef VF_DEBUG_KNOBS
    else if (l.gsplit && l.mrep == 2)
        hipLaunchKernelGGL(conv_lstm_gsplit_kernel<2>, dim3(tiles, l.ncg), dim3(kConvThreads), l.lds_bytes, st, p);
    else if (l.gsplit)
        hipLaunchKernelGGL(conv_lstm_gsplit_kernel<1>, dim3(tiles, l.ncg), dim3(kConvThreads), l.lds_bytes, st, p);
    else if (l.mrep == 0)
        hipLaunchKernelGGL(conv_lstm_split_kernel<2>, dim3(tiles, l.ncg), dim3(kConvThreads), l.lds_bytes, st, p);
#else
    else if (l.gsplit || l.mrep == 0)
        return fail(VF_ERR_INVALID, "conv-LSTM tile plan not compiled into this build (needs -DVF_DEBUG_KNOBS)");
#endif
    else
        hipLaunchKernelGGL(conv_lstm_split_kernel<1>, dim3(tiles, l.ncg), dim3(kConvThreads), l.lds_bytes, st, p);
    VF_HIP_CHECK(hipGetLastError());
    return VF_OK;
}

template <int MREP>
static int launch_lstm_bf16x6(const ConvLayer &l, const ConvParams &p, hipStream_t st) {
    const int tiles = l.NI == 1 ? p.B * l.tilesY * l.tilesX : (p.B + l.NI - 1) / l.NI;
    hipLaunchKernelGGL((conv_lstm_bf16x6_kernel<MREP>), dim3(tiles, l.ncg), dim3(kConvThreads), l.lds_bytes, st, p);
    VF_HIP_CHECK(hipGetLastError());
    return VF_OK;
}

// which (G, EPI, MREP) instances exist: LSTM in both tile heights, the FC with 256 rows (it has
// few rows and a long K), every other layer with 128-row tiles
template <int G, int EPI>
static int launch_conv_t(const ConvLayer &l, const ConvParams &p, hipStream_t st) {
    if constexpr (EPI == EPI_LSTM) {
        if (l.prec == 1) return launch_lstm_bf16x6<1>(l, p, st);        // 128-row tiles only
        if (l.mrep <= 0 || l.gsplit) return launch_lstm_split(l, p, st);
#ifdef VF_DEBUG_KNOBS
        if (l.mrep == 2) return launch_conv_m<G, EPI, 2>(l, p, st);
#else
        if (l.mrep == 2) return fail(VF_ERR_INVALID, "256-row conv-LSTM plan not compiled into this build");
#endif
        return launch_conv_m<G, EPI, 1>(l, p, st);
    } else if constexpr (EPI == EPI_PARTIAL) {
        return launch_conv_m<G, EPI, 2>(l, p, st);
    } else {
        return launch_conv_m<G, EPI, 1>(l, p, st);
    }
}

#endif  // VF_HOST_SELFTEST

struct SegArg {
    const float *ptr; long long bstride; const long long *ln_part; long long ln_bstride; int ln_nparts; float ln_inv_n;
    const float *gamma, *beta; int gamma_mod; int relu;
};

static ConvParams make_params(const ConvLayer &l, const LayerW &w, int B, const SegArg &s0, const SegArg *s1) {
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
    p.ncg = l.ncg; p.Cout = l.Cout; p.Wp = w.w; p.Wp16 = w.w16; p.bias = w.b;
    p.chunks_per_split = l.chunks_per_split; p.n_valid = l.n_valid;
    p.stats_nparts = l.stats_nparts;    // row stride of p.stats; the conv-LSTM plans overwrite it with st_rows[k]
    p.tile_variant = l.prec == 1 ? 1 : 0;
    return p;
}

}  // namespace vf

#include "vf_engine_savp3.inc"

// ================================================================== C ABI
extern "C" {

int vf_abi_version(void) { return VF_ABI_VERSION; }

const char *vf_last_error(void) { return g_last_error.c_str(); }

size_t vf_weight_count(const vf_config *cfg) {
    VF_API_TRY
    if (validate(cfg)) return 0;
    auto t = tensor_table(*cfg);
    return t.back().offset + t.back().size();
    VF_API_CATCH(size_t)
}

double vf_macs_per_sample_step(const vf_config *cfg) {
    VF_API_TRY
    if (validate(cfg)) return 0.0;
    if (cfg->arch == 3) return s3_macs(*cfg);
    const bool savp = cfg->arch >= 1;
    const int HF = cfg->height, WF = cfg->width;            // full resolution: heads and warps
    const int H = savp ? HF / 2 : HF, W = savp ? WF / 2 : WF;   // core
    auto t = tensor_table(*cfg);
    struct { const char *n; int h, w; } res[] = {
        {"enc0", H / 2, W / 2}, {"lstm1", H / 2, W / 2}, {"lstm2", H / 2, W / 2}, {"enc1", H / 4, W / 4},
        {"lstm3", H / 4, W / 4}, {"lstm4", H / 4, W / 4}, {"enc2", H / 8, W / 8}, {"enc3", H / 8, W / 8},
        {"lstm5", H / 8, W / 8}, {"convt1", H / 8, W / 8}, {"lstm6", H / 4, W / 4}, {"convt2", H / 4, W / 4},
        {"lstm7", H / 2, W / 2}, {"convt3", H / 2, W / 2}, {"rgb", HF, WF}, {"masks", HF, WF},
        {"enc00", HF / 2, WF / 2}, {"convt4", HF / 2, WF / 2}};
    double macs = 0;
    for (auto &r : res) {
        const TensorDesc *d = find_tensor(t, std::string(r.n) + "/w");
        if (!d) continue;               // arch 0 has no enc00 / convt4
        macs += (double)r.h * r.w * d->shape[0] * d->shape[1] * d->shape[2] * d->shape[3];
    }
    const TensorDesc *fc = find_tensor(t, "cdna/w"), *sw = find_tensor(t, "state/w");
    macs += (double)fc->shape[0] * fc->shape[1] + (double)sw->shape[0] * sw->shape[1];
    macs += (double)HF * WF * kTaps * (3 + cfg->ndesig) * (cfg->arch == 2 ? cfg->num_masks - 2 : cfg->num_masks);
    return macs;
    VF_API_CATCH(double)
}

}  // extern "C"

// layers, plans and buffers of the CDNA-core networks (arch 0 - 2)
static int cdna_create(vf_handle *h) {
    const vf_config *cfg = &h->cfg;
    int rc;
    const int H = h->H, W = h->W, Bc = cfg->max_batch, ND = h->ND, NV = h->ncam;
    const size_t BV = (size_t)Bc * NV;          // samples x views: rows of every per-sample buffer
    const int Hc = h->Hc, Wc = h->Wc;           // the three-scale core works on Hc x Wc
    const int H2 = Hc / 2, W2 = Wc / 2, H4 = Hc / 4, W4 = Wc / 4, H8 = Hc / 8, W8 = Wc / 8;
    const int *L = kLstmSizes;
    // rows per LSTM workgroup: 128 (mrep 1) keeps items short - the per-sample dependency chain,
    // not the MFMA rate, bounds a 200-sample rollout
    const int lstm_mrep[7] = {1, 1, 1, 1, 1, 1, 1};
#ifdef VF_DEBUG_KNOBS
    if (const char *e = getenv("VF_LSTM_MREP"))
        for (int k = 0; k < 7 && e[k]; ++k)         // per layer: h = 64 rows, 1 = 128, 2 = 256, anything else automatic
            h->mrep_override[k] = e[k] == '2' ? 2 : (e[k] == '1' ? 1 : (e[k] == 'h' ? 3 : (e[k] == 'q' ? 4 : 0)));
#endif
#ifdef VF_DEBUG_KNOBS
    if (const char *e = getenv("VF_YIELD")) h->yield_budget = atoi(e);
    if (const char *e = getenv("VF_WT")) h->wt_publish = atoi(e) != 0;
    // A/B knob of debug builds, read ONCE per handle (not inside a setter the caller may never invoke)
    static const bool knob_no_pair = getenv("VF_FUSE_PAIR") && atoi(getenv("VF_FUSE_PAIR")) == 0;
    h->pair_allowed = !knob_no_pair;
    h->fuse_pair = h->fuse_pair && h->pair_allowed;
#endif
    const int ccond = h->cond ? cfg->adim + cfg->sdim : 0;     // conditioning rows in every conv-LSTM's canonical weights
    if (h->savp) {
        init_layer(h->enc00, "enc00", PACK_PLAIN, H, W, Hc, Wc, 5, 5, 2, 1, 3, 0, kEnc00Ch, true);
        init_layer(h->convt4, "convt4", PACK_CONVT, Hc, Wc, Hc, Wc, 2, 2, 1, 1, 32, kEnc00Ch, 32, true, false, 1, 0, true);
    }
    init_layer(h->enc0, "enc0", PACK_PLAIN, Hc, Wc, H2, W2, 5, 5, 2, 1, h->savp ? kEnc00Ch : 3, 0, 32, true);
    init_layer(h->lstm[0], "lstm1", PACK_LSTM, H2, W2, H2, W2, 5, 5, 1, 2, 32, L[0], L[0], true, false, lstm_mrep[0], cfg->precision, false, ccond);
    init_layer(h->lstm[1], "lstm2", PACK_LSTM, H2, W2, H2, W2, 5, 5, 1, 2, L[0], L[1], L[1], true, false, lstm_mrep[1], cfg->precision, false, ccond);
    init_layer(h->enc1, "enc1", PACK_PLAIN, H2, W2, H4, W4, 3, 3, 2, 0, L[1], 0, L[1], false);
    init_layer(h->lstm[2], "lstm3", PACK_LSTM, H4, W4, H4, W4, 5, 5, 1, 2, L[1], L[2], L[2], true, false, lstm_mrep[2], cfg->precision, false, ccond);
    init_layer(h->lstm[3], "lstm4", PACK_LSTM, H4, W4, H4, W4, 5, 5, 1, 2, L[2], L[3], L[3], true, false, lstm_mrep[3], cfg->precision, false, ccond);
    init_layer(h->enc2, "enc2", PACK_PLAIN, H4, W4, H8, W8, 3, 3, 2, 0, L[3], 0, L[3], false);
    init_layer(h->enc3, "enc3", PACK_PLAIN, H8, W8, H8, W8, 1, 1, 1, 0, L[3], 0, L[3], false);
    init_layer(h->lstm[4], "lstm5", PACK_LSTM, H8, W8, H8, W8, 5, 5, 1, 2, L[3], L[4], L[4], true, false, lstm_mrep[4], cfg->precision, false, ccond);
    init_layer(h->convt1, "convt1", PACK_CONVT, H8, W8, H8, W8, 2, 2, 1, 1, L[4], 0, L[4], false);
    h->enc2_one.ni_cap = h->enc3_one.ni_cap = h->convt1_one.ni_cap = 1;
    h->enc2_one.kc_cap = h->enc2.KC;   // the one-image plan of enc2 chunks like the regular one: it shares its packed weights,
                                       // and enc2 + enc3 can be one item per IMAGE where a phase is narrow (conv_pair)
    init_layer(h->enc2_one, "enc2", PACK_PLAIN, H4, W4, H8, W8, 3, 3, 2, 0, L[3], 0, L[3], false);
    h->enc2_one.lds_bytes = std::max(h->enc2_one.lds_bytes, (size_t)2 * 128 * 36 * 4);     // (room for the pair's hand-over tile)
    init_layer(h->enc3_one, "enc3", PACK_PLAIN, H8, W8, H8, W8, 1, 1, 1, 0, L[3], 0, L[3], false);
    init_layer(h->convt1_one, "convt1", PACK_CONVT, H8, W8, H8, W8, 2, 2, 1, 1, L[4], 0, L[4], false);
    init_layer(h->lstm[5], "lstm6", PACK_LSTM, H4, W4, H4, W4, 5, 5, 1, 2, L[4], L[5], L[5], true, false, lstm_mrep[5], cfg->precision, false, ccond);
    const bool pub = cfg->arch == 0 && cfg->layer_spec == 1;
    h->c_t2 = pub ? L[5] + L[1] : L[5];
    h->c_top = pub ? L[6] + 32 : 32;
    init_layer(h->convt2, "convt2", PACK_CONVT, H4, W4, H4, W4, 2, 2, 1, 1, L[5], L[1], h->c_t2, false, false, 1, 0, true);
    init_layer(h->lstm[6], "lstm7", PACK_LSTM, H2, W2, H2, W2, 5, 5, 1, 2, h->c_t2, L[6], L[6], true, false, lstm_mrep[6], cfg->precision, false, ccond);
    init_layer(h->convt3, "convt3", PACK_CONVT, H2, W2, H2, W2, 2, 2, 1, 1, L[6], 32, h->c_top, true, false, 1, 0, true);
    // CDNA FC as a K-split GEMM over 1x1 "images"
    init_layer(h->fc, "cdna", PACK_PLAIN, 1, 1, 1, 1, 1, 1, 1, 0, H8 * W8 * L[4], 0, kTaps * h->K, false, true, 2);
    {
        ConvLayer &f = h->fc;
        const int total = f.nchunk[0];
        f.nsplit = std::min(32, total);
        f.chunks_per_split = (total + f.nsplit - 1) / f.nsplit;
        f.nsplit = (total + f.chunks_per_split - 1) / f.chunks_per_split;
        f.n_valid = kTaps * h->K;
        // the persistent schedule's plan: same packed weights, same chunks and splits, 128 rows x all column groups
        h->fc_wide = f;
        h->fc_wide.mrep = 7; h->fc_wide.NI = kFcRows; h->fc_wide.lds_bytes = fc_wide_lds_bytes();
        h->fc_wide_ok = f.KC == 32 && f.ncg <= kFcGroups && f.nseg == 1 && f.segC[0] % 32 == 0;
    }
    h->layers = {&h->enc0, &h->lstm[0], &h->lstm[1], &h->enc1, &h->lstm[2], &h->lstm[3], &h->enc2, &h->enc3,
                 &h->lstm[4], &h->convt1, &h->lstm[5], &h->convt2, &h->lstm[6], &h->convt3, &h->fc};
    if (h->savp) { h->layers.push_back(&h->enc00); h->layers.push_back(&h->convt4); }
    h->have_big = cfg->precision == 0;      // the split-bf16 tile has 128 rows only
    for (int k = 0; k < 7; ++k) {
        h->st_rows[k] = h->lstm[k].stats_nparts;
        if (!h->have_big) continue;
        const ConvLayer &sm = h->lstm[k];
        init_layer(h->lstm_big[k], sm.name.c_str(), PACK_LSTM, sm.Hin, sm.Win, sm.Hout, sm.Wout, 5, 5, 1, 2, sm.segC[1],
                   sm.segC[0], sm.Cout, true, false, 2, 0, false, ccond);
        // same chunking = same K order per output (bit-identical results) and the same packed weights
#ifdef VF_DEBUG_KNOBS
        h->big_ok[k] = h->lstm_big[k].KC == sm.KC;
#else
        h->big_ok[k] = false;       // the 256-row tile lost to the 128-row gate-split tile at every batch size (round 3): debug builds only
#endif
        if (h->big_ok[k]) h->st_rows[k] = std::max(h->st_rows[k], h->lstm_big[k].stats_nparts);
        init_layer(h->lstm_half[k], sm.name.c_str(), PACK_LSTM, sm.Hin, sm.Win, sm.Hout, sm.Wout, 5, 5, 1, 2, sm.segC[1],
                   sm.segC[0], sm.Cout, true, false, 0, 0, false, ccond);
        h->half_ok[k] = h->lstm_half[k].KC == sm.KC && sm.KC == 32;
#ifndef VF_DEBUG_KNOBS
        h->half_ok[k] = h->half_ok[k] && h->lstm_half[k].gsplit;     // the only 64-row tile of a production build
#endif
        if (h->half_ok[k]) h->st_rows[k] = std::max(h->st_rows[k], h->lstm_half[k].stats_nparts);
        init_layer(h->lstm_quarter[k], sm.name.c_str(), PACK_LSTM, sm.Hin, sm.Win, sm.Hout, sm.Wout, 5, 5, 1, 2,
                   sm.segC[1], sm.segC[0], sm.Cout, true, false, -1, 0, false, ccond);
        h->quarter_ok[k] = h->lstm_quarter[k].KC == sm.KC && sm.KC == 32;
        if (h->quarter_ok[k]) h->st_rows[k] = std::max(h->st_rows[k], h->lstm_quarter[k].stats_nparts);
    }
    h->max_lds = (size_t)composite_lds_floats<kMaxDesig, 10>() * 4;
    for (size_t i = 0; i < h->layers.size(); ++i) {
        h->layers[i]->id = (int)i;
        h->max_lds = std::max(h->max_lds, h->layers[i]->lds_bytes);
    }
    h->enc2_one.id = h->enc2.id; h->enc3_one.id = h->enc3.id; h->convt1_one.id = h->convt1.id;   // shared packed weights
    h->fc_wide.id = h->fc.id;
    for (int k = 0; k < 7; ++k)
        if (h->have_big) {
            h->lstm_big[k].id = h->lstm_half[k].id = h->lstm_quarter[k].id = h->lstm[k].id;    // shared packed weights
            if (h->big_ok[k]) h->max_lds = std::max(h->max_lds, h->lstm_big[k].lds_bytes);
            if (h->half_ok[k]) h->max_lds = std::max(h->max_lds, h->lstm_half[k].lds_bytes);
            if (h->quarter_ok[k]) h->max_lds = std::max(h->max_lds, h->lstm_quarter[k].lds_bytes);
        }
    {       // the fused decoder top (vf_fused_top.h) may need more than any stand-alone tile of a small image
        const ConvLayer &top = h->savp ? h->convt4 : h->convt3;
        if (top_fusable(top, h->ND))
            h->max_lds = std::max(h->max_lds, fused_top_lds_floats(top.TH, top.TW, h->ND) * 4);
    }
    h->max_lds += 16;

#define VF_ALLOC(ptr, n)                           \
    do {                                           \
        rc = dev_alloc(h, &(ptr), (size_t)(n));    \
        if (rc) return rc;      \
    } while (0)

    // ---- per-view parameters and context: sized from the layer plans, filled by vf_load_weights
    const int nc = cfg->n_context;
    const int nsa = cfg->adim + cfg->sdim;
    h->views.resize(NV);
    for (ViewData &vd : h->views) {
        for (const ConvLayer *l : h->layers) {
            VF_ALLOC(vd.lw[l->id].w, l->packed_w());
            VF_ALLOC(vd.lw[l->id].b, l->packed_b());
            if (l->prec == 1) VF_ALLOC(vd.lw[l->id].w16, l->packed_w16());
        }
        for (int i = 0; i < kNumLn; ++i) {
            const TensorDesc *g = find_tensor(h->table, ln_name(i) + "/g");
            if (!g) continue;           // lna / lnb exist in arch 1 only
            VF_ALLOC(vd.ln_g[i], g->size());
            VF_ALLOC(vd.ln_b[i], g->size());
        }
        VF_ALLOC(vd.w_rgb, h->c_top * 3); VF_ALLOC(vd.b_rgb, 3);
        VF_ALLOC(vd.w_mask, h->c_top * (h->K + 1)); VF_ALLOC(vd.b_mask, h->K + 1);
        VF_ALLOC(vd.w_state, (size_t)nsa * cfg->sdim); VF_ALLOC(vd.b_state, cfg->sdim);
        VF_ALLOC(vd.w_sa, (size_t)nsa * L[3]); VF_ALLOC(vd.b_fc, kTaps * h->K);
        if (h->cond)
            for (int k = 0; k < 7; ++k) VF_ALLOC(vd.w_cond[k], (size_t)kTaps * nsa * 4 * L[k]);
    }
    VF_ALLOC(h->ctx_frames_all, (size_t)NV * nc * H * W * 3);
    VF_ALLOC(h->ctx_distrib_all, (size_t)NV * nc * H * W * ND);
    for (int v = 0; v < NV; ++v) {
        h->views[v].ctx_frames = h->ctx_frames_all + (size_t)v * nc * H * W * 3;
        h->views[v].ctx_distrib = h->ctx_distrib_all + (size_t)v * nc * H * W * ND;
    }
    VF_ALLOC(h->ctx_states, (size_t)nc * cfg->sdim);
    VF_ALLOC(h->ctx_actions, (size_t)std::max(nc - 1, 1) * cfg->adim);

    VF_ALLOC(h->enc0_o, BV * H2 * W2 * 32);
    VF_ALLOC(h->enc1_o, BV * H4 * W4 * L[1]);
    VF_ALLOC(h->enc2_o, BV * H8 * W8 * L[3]);
    VF_ALLOC(h->enc3_o, BV * H8 * W8 * L[3]);
    VF_ALLOC(h->enc4_o, BV * H4 * W4 * L[4]);
    VF_ALLOC(h->enc5_o, BV * H2 * W2 * h->c_t2);
    VF_ALLOC(h->enc6_o, BV * Hc * Wc * h->c_top);
    if (h->savp) {
        VF_ALLOC(h->enc00_o, BV * Hc * Wc * kEnc00Ch);
        VF_ALLOC(h->enc7_o, BV * H * W * 32);
        VF_ALLOC(h->st_enc00, BV * h->enc00.stats_nparts * 2);
        VF_ALLOC(h->st_enc7, BV * h->convt4.stats_nparts * 2);
    }
    const int lh[7] = {H2, H2, H4, H4, H8, H4, H2}, lw[7] = {W2, W2, W4, W4, W8, W4, W2};
    for (int k = 0; k < 7; ++k) {
        const size_t elems = BV * lh[k] * lw[k] * L[k];
        VF_ALLOC(h->c_state[k], elems);
        VF_ALLOC(h->h_state[k][0], elems);
        VF_ALLOC(h->h_state[k][1], elems);
        VF_ALLOC(h->st_h[k], BV * h->st_rows[k] * 2);
    }
    VF_ALLOC(h->st_enc0, BV * h->enc0.stats_nparts * 2);
    VF_ALLOC(h->st_enc6, BV * h->convt3.stats_nparts * 2);
    VF_ALLOC(h->sbias, BV * L[3]);
    if (h->cond)
        for (int k = 0; k < 7; ++k) VF_ALLOC(h->cond_bias[k], 2 * BV * kCondClasses * 4 * L[k]);
    VF_ALLOC(h->fc_part, BV * h->fc.nsplit * kTaps * h->K);
    VF_ALLOC(h->kern, BV * kTaps * h->K);
    VF_ALLOC(h->frames_all, BV * h->T * H * W * 3);
    VF_ALLOC(h->distrib_all, BV * h->T * H * W * ND);
    VF_ALLOC(h->states_all, BV * h->T * cfg->sdim);
    h->sums_view_stride = (long long)Bc * ND * h->nblocks * 2;
    h->sums_step_stride = h->sums_view_stride * NV;
    VF_ALLOC(h->sums, (size_t)h->T * h->sums_step_stride);
    VF_ALLOC(h->actions_buf, (size_t)Bc * h->T * cfg->adim);
    h->sched_capacity = ((size_t)h->S * 32 + 8) * NV;       // (arch 2: up to 29 phases per step)
    h->counter_capacity = ((size_t)h->S * 32 + 8) * ((size_t)Bc + 1) * NV;
    VF_ALLOC(h->sched[0].d_phases, h->sched_capacity);
    VF_ALLOC(h->sched[1].d_phases, h->sched_capacity);
    VF_ALLOC(h->d_sync, kSyncHead + h->counter_capacity);
    VF_ALLOC(h->d_status, 1);
    VF_ALLOC(h->d_stats, h->sched_capacity * 2);
    for (int i = 0; i < NV; ++i) {
        BatchView sv;
        memset(&sv, 0, sizeof(sv));
        VF_ALLOC(sv.enc0_o, (size_t)H2 * W2 * 32);
        VF_ALLOC(sv.enc1_o, (size_t)H4 * W4 * L[1]);
        VF_ALLOC(sv.enc2_o, (size_t)H8 * W8 * L[3]);
        VF_ALLOC(sv.enc3_o, (size_t)H8 * W8 * L[3]);
        VF_ALLOC(sv.enc4_o, (size_t)H4 * W4 * L[4]);
        VF_ALLOC(sv.enc5_o, (size_t)H2 * W2 * h->c_t2);
        for (int k = 0; k < 7; ++k) {
            const size_t per = (size_t)lh[k] * lw[k] * L[k];
            VF_ALLOC(sv.c_state[k], per);
            VF_ALLOC(sv.h_state[k][0], per);
            VF_ALLOC(sv.h_state[k][1], per);
            VF_ALLOC(sv.st_h[k], (size_t)h->st_rows[k] * 2);
        }
        VF_ALLOC(sv.st_enc0, (size_t)h->enc0.stats_nparts * 2);
        if (h->savp) {
            VF_ALLOC(sv.enc00_o, (size_t)Hc * Wc * kEnc00Ch);
            VF_ALLOC(sv.st_enc00, (size_t)h->enc00.stats_nparts * 2);
        }
        VF_ALLOC(sv.sbias, (size_t)L[3]);
        if (h->cond)
            for (int k = 0; k < 7; ++k) {
                VF_ALLOC(sv.cond_bias[k][0], (size_t)2 * kCondClasses * 4 * L[k]);
                sv.cond_bias[k][1] = sv.cond_bias[k][0] + (size_t)kCondClasses * 4 * L[k];
            }
        h->shared_views.push_back(sv);
    }
#undef VF_ALLOC
    return VF_OK;
}

extern "C" {

int vf_create(const vf_config *cfg, vf_handle **out) {
    vf_handle *made = nullptr;     // (released if anything below throws)
    VF_API_TRY
    if (!out) return fail(VF_ERR_INVALID, "null out pointer");
    *out = nullptr;
    int rc = validate(cfg);
    if (rc) return rc;
#ifndef VF_HOST_SELFTEST
    VF_HIP_CHECK(hipSetDevice(cfg->device));
#endif
    vf_handle *h = new vf_handle();
    made = h;
    h->cfg = *cfg;
    h->ncam = std::max(1, cfg->ncam);
    h->n_draws = std::max(1, cfg->n_draws);
    h->cfg.ncam = h->ncam; h->cfg.n_draws = h->n_draws;
    h->H = cfg->height; h->W = cfg->width; h->ND = cfg->ndesig; h->K = cfg->num_masks;
    h->T = cfg->sequence_length - cfg->n_context;
    h->S = h->T + cfg->n_context - 1;
    h->table = tensor_table(*cfg);
    h->blob_floats = h->table.back().offset + h->table.back().size();
    h->savp = cfg->arch == 1 || cfg->arch == 2;
    h->cond = cfg->arch == 2;
    h->Hc = h->savp ? h->H / 2 : h->H; h->Wc = h->savp ? h->W / 2 : h->W;
    const int H = h->H, W = h->W;
    h->ntiles = ((H + kCompTile - 1) / kCompTile) * ((W + kCompTile - 1) / kCompTile);
    h->nblocks = sum_blocks(H, W);
#ifdef VF_HOST_SELFTEST
    h->fake_size = (size_t)1 << 40;
    void *base = mmap(nullptr, h->fake_size, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (base == MAP_FAILED) { delete h; return fail(VF_ERR_NOMEM, "self-test address reservation failed"); }
    h->fake_base = static_cast<char *>(base);
#endif

    rc = cfg->arch == 3 ? s3_create(h) : cdna_create(h);
    if (rc) { vf_destroy(h); return rc; }
    VF_INJECT(3);
#ifndef VF_HOST_SELFTEST
    if (hipMemset(h->d_sync, 0, kSyncHead * sizeof(int)) != hipSuccess || hipMemset(h->d_status, 0, sizeof(int)) != hipSuccess) {
        vf_destroy(h);
        return fail(VF_ERR_HIP, "hipMemset of the scheduler words failed");
    }
    for (int i = 0; i < kSchedRing; ++i) {
        void *q = nullptr;
        if (hipHostMalloc(&q, h->sched_capacity * sizeof(PhaseDesc), hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&h->stage_done[i], hipEventDisableTiming) != hipSuccess) {
            vf_destroy(h);
            return fail(VF_ERR_NOMEM, "pinned staging buffer for the schedule failed");
        }
        h->stage[i] = static_cast<PhaseDesc *>(q);
    }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0)
            h->n_cu = prop.multiProcessorCount;
    }
    if ((rc = configure_kernels(h))) { vf_destroy(h); return rc; }
#endif
    *out = h;
    return VF_OK;
    VF_API_CATCH_CLEANUP(int, { if (made) vf_destroy(made); if (out) *out = nullptr; })
}

int vf_destroy(vf_handle *h) {
    VF_API_TRY
    if (!h) return VF_OK;
#ifdef VF_HOST_SELFTEST
    if (h->fake_base) munmap(h->fake_base, h->fake_size);
#else
    (void)hipSetDevice(h->cfg.device);
    (void)hipDeviceSynchronize();
    for (const AllocRec &a : h->allocs) (void)hipFree(a.p);
    for (int i = 0; i < kSchedRing; ++i) {
        if (h->stage[i]) (void)hipHostFree(h->stage[i]);
        if (h->stage_done[i]) (void)hipEventDestroy(h->stage_done[i]);
    }
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
#endif
    s3_free(h);
    delete h;
    return VF_OK;
    VF_API_CATCH(int)
}

int vf_load_weights(vf_handle *h, const float *blob_all, size_t n_floats) {
    VF_API_TRY
    if (!h || !blob_all) return fail(VF_ERR_INVALID, "null handle or blob");
    VF_INJECT(2);
    const size_t want = h->blob_floats * h->ncam;
    if (n_floats != want)
        return fail(VF_ERR_INVALID, "weight blob has " + std::to_string(n_floats) + " floats, expected " +
                                        std::to_string(want) + " (" + std::to_string(h->ncam) + " view(s))");
#ifndef VF_HOST_SELFTEST
    VF_HIP_CHECK(hipSetDevice(h->cfg.device));
    // the previous weights may still be in use by a rollout in flight
    VF_HIP_CHECK(hipDeviceSynchronize());
#endif
    int rc;
    if (h->cfg.arch == 3) {
        for (int view = 0; view < h->ncam; ++view)
            if ((rc = s3_load_weights(h, blob_all + (size_t)view * h->blob_floats, view))) return rc;
#ifndef VF_HOST_SELFTEST
        VF_HIP_CHECK(hipDeviceSynchronize());
#endif
        h->have_weights = true;
        h->shared_valid = false;
        return VF_OK;
    }
    auto T = [&](const std::string &name) { return find_tensor(h->table, name); };
    for (int view = 0; view < h->ncam; ++view) {
        const float *blob = blob_all + (size_t)view * h->blob_floats;
        ViewData &vd = h->views[view];
        for (const ConvLayer *lp : h->layers) {
            const ConvLayer &l = *lp;
            const TensorDesc *w = T(l.name + "/w"), *b = T(l.name + "/b");
            std::vector<float> wp, bp;
            if (l.name == "cdna" && h->cond) {
                // canonical [fc_in][25 x 4] -> the engine's [fc_in][25 x 6]: two dead kernels of zero weights (they meet a
                // zero mask in the compositing), so every kernel-count-dependent stride stays num_masks
                const int KF = h->K - 2, fc_in = w->shape[0];
                std::vector<float> wide((size_t)fc_in * kTaps * h->K, 0.f);
                for (int r = 0; r < fc_in; ++r)
                    for (int tap = 0; tap < kTaps; ++tap)
                        for (int k = 0; k < KF; ++k)
                            wide[((size_t)r * kTaps + tap) * h->K + k] = (blob + w->offset)[((size_t)r * kTaps + tap) * KF + k];
                wp = pack_weights(l, wide.data(), 1, 1, fc_in, kTaps * h->K);
                bp.assign(l.packed_b(), 0.f);
            } else if (l.name == "cdna") {
                wp = pack_weights(l, blob + w->offset, 1, 1, w->shape[0], w->shape[1]);
                bp.assign(l.packed_b(), 0.f);       // bias is added by cdna_finalize
            } else if (l.name == "enc3") {
                // only the enc2 rows go through the GEMM; the action/state rows become a per-sample bias
                wp = pack_weights(l, blob + w->offset, 1, 1, w->shape[2], w->shape[3]);
                bp = pack_bias(l, blob + b->offset);
            } else if (l.first_valu) {      // canonical [5][5][3][Cout] as it is
                wp.assign(blob + w->offset, blob + w->offset + w->size());
                bp = pack_bias(l, blob + b->offset);
            } else {
                wp = pack_weights(l, blob + w->offset, w->shape[0], w->shape[1], w->shape[2], w->shape[3]);
                bp = pack_bias(l, blob + b->offset);
            }
            if (wp.size() != l.packed_w() || bp.size() != l.packed_b())
                return fail(VF_ERR_INVALID, "internal: packed size of " + l.name + " differs from its plan");
            if ((rc = dev_write(h, vd.lw[l.id].w, wp.data(), wp.size() * sizeof(float)))) return rc;
            if ((rc = dev_write(h, vd.lw[l.id].b, bp.data(), bp.size() * sizeof(float)))) return rc;
            if (l.prec == 1) {
                std::vector<unsigned short> w16 = pack_weights_bf16x3(l, blob + w->offset, w->shape[2], w->shape[3]);
                if (w16.size() != l.packed_w16())
                    return fail(VF_ERR_INVALID, "internal: split-bf16 size of " + l.name + " differs from its plan");
                if ((rc = dev_write(h, vd.lw[l.id].w16, w16.data(), w16.size() * sizeof(unsigned short)))) return rc;
            }
        }
        for (int i = 0; i < kNumLn; ++i) {
            const std::string n = ln_name(i);
            const TensorDesc *g = T(n + "/g"), *b = T(n + "/b");
            if (!g) continue;
            if ((rc = dev_write(h, vd.ln_g[i], blob + g->offset, g->size() * sizeof(float)))) return rc;
            if ((rc = dev_write(h, vd.ln_b[i], blob + b->offset, b->size() * sizeof(float)))) return rc;
        }
        const TensorDesc *d;
        d = T("rgb/w");   if ((rc = dev_write(h, vd.w_rgb, blob + d->offset, d->size() * sizeof(float)))) return rc;
        d = T("rgb/b");   if ((rc = dev_write(h, vd.b_rgb, blob + d->offset, d->size() * sizeof(float)))) return rc;
        {   // mask head.  arch 2 keeps the PUBLISHED order of the compositing layers in the checkpoint - [four CDNA warps,
            // previous frame, first frame, scratch] - and the kernels' order on the device: [previous, scratch, first, warps]
            const TensorDesc *mw = T("masks/w"), *mb = T("masks/b");
            const int NM = h->K + 1;
            std::vector<float> w(mw->size()), bb(mb->size());
            for (int j = 0; j < NM; ++j) {
                const int src = !h->cond ? j : (j == 0 ? NM - 3 : (j == 1 ? NM - 1 : (j == 2 ? NM - 2 : j - 3)));
                for (int c = 0; c < h->c_top; ++c) w[(size_t)c * NM + j] = (blob + mw->offset)[(size_t)c * NM + src];
                bb[j] = (blob + mb->offset)[src];
            }
            if ((rc = dev_write(h, vd.w_mask, w.data(), w.size() * sizeof(float)))) return rc;
            if ((rc = dev_write(h, vd.b_mask, bb.data(), bb.size() * sizeof(float)))) return rc;
        }
        if (h->cond) {      // conditioning rows [tap][Cx + c][4C] of every conv-LSTM's canonical [5][5][Cx + nsa + Ch][4C]
            const int nsa = h->cfg.adim + h->cfg.sdim;
            for (int k = 0; k < 7; ++k) {
                const ConvLayer &l = h->lstm[k];
                const TensorDesc *w = T(l.name + "/w");
                const int Cin = w->shape[2], C4 = w->shape[3], Cx = l.segC[1];
                std::vector<float> wc((size_t)kTaps * nsa * C4);
                for (int tap = 0; tap < kTaps; ++tap)
                    for (int c = 0; c < nsa; ++c)
                        memcpy(&wc[((size_t)tap * nsa + c) * C4], blob + w->offset + ((size_t)tap * Cin + Cx + c) * C4,
                               (size_t)C4 * sizeof(float));
                if ((rc = dev_write(h, vd.w_cond[k], wc.data(), wc.size() * sizeof(float)))) return rc;
            }
        }
        d = T("state/w"); if ((rc = dev_write(h, vd.w_state, blob + d->offset, d->size() * sizeof(float)))) return rc;
        d = T("state/b"); if ((rc = dev_write(h, vd.b_state, blob + d->offset, d->size() * sizeof(float)))) return rc;
        d = T("cdna/b");
        if (h->cond) {
            std::vector<float> bw((size_t)kTaps * h->K, 0.f);
            for (int tap = 0; tap < kTaps; ++tap)
                for (int k = 0; k < h->K - 2; ++k) bw[(size_t)tap * h->K + k] = (blob + d->offset)[(size_t)tap * (h->K - 2) + k];
            if ((rc = dev_write(h, vd.b_fc, bw.data(), bw.size() * sizeof(float)))) return rc;
        } else if ((rc = dev_write(h, vd.b_fc, blob + d->offset, d->size() * sizeof(float)))) return rc;
        // enc3 rows [L3 .. L3+adim+sdim) x 64: the smeared action/state inputs
        d = T("enc3/w");
        const int L3 = kLstmSizes[3];
        if ((rc = dev_write(h, vd.w_sa, blob + d->offset + (size_t)L3 * d->shape[3],
                            (size_t)(h->cfg.adim + h->cfg.sdim) * d->shape[3] * sizeof(float))))
            return rc;
    }
#ifndef VF_HOST_SELFTEST
    VF_HIP_CHECK(hipDeviceSynchronize());
#endif
    h->have_weights = true;
    h->shared_valid = false;
    return VF_OK;
    VF_API_CATCH(int)
}

#ifndef VF_HOST_SELFTEST
int vf_set_context(vf_handle *h, const uint8_t *d_frames, const float *d_states, const float *d_ctx_actions,
                   const float *d_ctx_distrib, void *stream) {
    VF_API_TRY
    if (!h || !d_frames || !d_states || !d_ctx_distrib) return fail(VF_ERR_INVALID, "null argument");
    const int nc = h->cfg.n_context;
    if (nc > 1 && !d_ctx_actions) return fail(VF_ERR_INVALID, "context actions required when n_context > 1");
    VF_HIP_CHECK(hipSetDevice(h->cfg.device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int hw3 = h->H * h->W * 3, hwd = h->H * h->W * h->ND;
    const int n_s = nc * h->cfg.sdim, n_a = (nc - 1) * h->cfg.adim;
    const int n = std::max(std::max(nc * h->ncam * hw3, nc * h->ncam * hwd), std::max(n_s, n_a));
    // [nc][ncam][..] of the caller -> [ncam][nc][..] (the views' context buffers are one allocation)
    hipLaunchKernelGGL(set_context_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_frames, h->ctx_frames_all, nc,
                       h->ncam, hw3, d_ctx_distrib, h->ctx_distrib_all, hwd, d_states, h->ctx_states, n_s,
                       d_ctx_actions, h->ctx_actions, n_a);
    VF_HIP_CHECK(hipGetLastError());
    h->have_context = true;
    h->shared_valid = false;        // the shared units are functions of the context
    return VF_OK;
    VF_API_CATCH(int)
}
#endif

}  // extern "C"

static BatchView make_view(vf_handle *h, int view, const float *d_actions, int b0) {
    const vf_config &c = h->cfg;
    const int H = h->H, W = h->W, T = h->T, ND = h->ND, Hc = h->Hc, Wc = h->Wc;
    const int H2 = Hc / 2, W2 = Wc / 2, H4 = Hc / 4, W4 = Wc / 4, H8 = Hc / 8, W8 = Wc / 8;
    const int *L = kLstmSizes;
    const int lh[7] = {H2, H2, H4, H4, H8, H4, H2}, lw[7] = {W2, W2, W4, W4, W8, W4, W2};
    const size_t b = (size_t)view * c.max_batch + (size_t)b0;      // row of every [ncam][max_batch] buffer
    BatchView v;
    memset(&v, 0, sizeof(v));
    if (h->savp) {
        v.enc00_o = h->enc00_o + b * Hc * Wc * kEnc00Ch;
        v.enc7_o = h->enc7_o + b * H * W * 32;
        v.st_enc00 = h->st_enc00 + b * h->enc00.stats_nparts * 2;
        v.st_enc7 = h->st_enc7 + b * h->convt4.stats_nparts * 2;
    }
    v.enc0_o = h->enc0_o + b * H2 * W2 * 32;
    v.enc1_o = h->enc1_o + b * H4 * W4 * L[1];
    v.enc2_o = h->enc2_o + b * H8 * W8 * L[3];
    v.enc3_o = h->enc3_o + b * H8 * W8 * L[3];
    v.enc4_o = h->enc4_o + b * H4 * W4 * L[4];
    v.enc5_o = h->enc5_o + b * H2 * W2 * h->c_t2;
    v.enc6_o = h->enc6_o + b * Hc * Wc * h->c_top;
    for (int k = 0; k < 7; ++k) {
        const size_t per = (size_t)lh[k] * lw[k] * L[k];
        v.c_state[k] = h->c_state[k] + b * per;
        v.h_state[k][0] = h->h_state[k][0] + b * per;
        v.h_state[k][1] = h->h_state[k][1] + b * per;
        v.st_h[k] = h->st_h[k] + b * h->st_rows[k] * 2;
    }
    v.st_enc0 = h->st_enc0 + b * h->enc0.stats_nparts * 2;
    v.st_enc6 = h->st_enc6 + b * h->convt3.stats_nparts * 2;
    v.sbias = h->sbias + b * L[3];
    if (h->cond)
        for (int k = 0; k < 7; ++k)
            for (int par = 0; par < 2; ++par)
                v.cond_bias[k][par] = h->cond_bias[k] + ((size_t)par * c.max_batch * h->ncam + b) * kCondClasses * 4 * L[k];
    v.fc_part = h->fc_part + b * h->fc.nsplit * kTaps * h->K;
    v.kern = h->kern + b * kTaps * h->K;
    v.frames_all = h->frames_all + b * T * H * W * 3;
    v.distrib_all = h->distrib_all + b * T * H * W * ND;
    v.states_all = h->states_all + b * T * c.sdim;
    v.sums = h->sums + (long long)view * h->sums_view_stride + (size_t)b0 * ND * h->nblocks * 2;
    v.actions = d_actions + (size_t)b0 * T * c.adim;       // the views share the action sequences
    return v;
}

// ------------------------------------------------------------------ rollout emission
// emit_rollout() walks the S steps of the predictor once for one view and hands every unit of
// device work to a sink, together with the units it depends on.  LaunchSink enqueues one kernel
// per unit (stream order makes the dependencies implicit); ScheduleSink records the units as
// phases of the persistent launch (vf_persistent.h) with explicit per-sample dependencies.
static const int kSkipped = 1 << 30;      // id of a unit whose cached result is reused

#ifndef VF_HOST_SELFTEST
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
            case PH_CONV_RAW:
                if (l.first_valu) {
                    const dim3 grid(p.B * l.tilesY * l.tilesX);
                    if (l.Cout == 16) hipLaunchKernelGGL(conv_first_kernel<16>, grid, dim3(kConvThreads), l.lds_bytes, st, p);
                    else hipLaunchKernelGGL(conv_first_kernel<32>, grid, dim3(kConvThreads), l.lds_bytes, st, p);
                    VF_HIP_CHECK(hipGetLastError());
                    return VF_OK;
                }
                return launch_conv_t<1, EPI_RAW_STATS>(l, p, st);
            case PH_CONVT_RELU: return launch_conv_t<4, EPI_CONVT_RELU>(l, p, st);
            case PH_CONVT_RAW: return launch_conv_t<4, EPI_CONVT_RAW_STATS>(l, p, st);
            case PH_CONV_RAW3: return l.mrep == 2 ? launch_conv_m<1, EPI_RAW, 2>(l, p, st) : launch_conv_m<1, EPI_RAW, 1>(l, p, st);
            case PH_CONV_RAW3G2: return l.mrep == 2 ? launch_conv_m<2, EPI_RAW, 2>(l, p, st) : launch_conv_m<2, EPI_RAW, 1>(l, p, st);
            case PH_CONV_RAW3G4: return launch_conv_m<4, EPI_RAW, 1>(l, p, st);
            case PH_GATES_RAW: {
                const int tiles = l.NI == 1 ? p.B * l.tilesY * l.tilesX : (p.B + l.NI - 1) / l.NI;
                hipLaunchKernelGGL(conv_gates_raw_kernel, dim3(tiles, l.ncg), dim3(kConvThreads), l.lds_bytes, st, p);
                VF_HIP_CHECK(hipGetLastError());
                return VF_OK;
            }
            default: return launch_conv_t<1, EPI_PARTIAL>(l, p, st);
        }
    }
    // element-wise items of arch 3 (vf_savp3.h): one workgroup per item
    int ew(const EwParams &p, int /*view*/, std::initializer_list<int>) {
        const int items = p.spi > 0 ? (p.B + p.spi - 1) / p.spi : p.B * p.gx;
        const size_t lds = ew_lds_bytes(p.op, h->ND);
        switch (h->ND) {
            case 1: hipLaunchKernelGGL(ew_kernel<1>, dim3(items), dim3(kConvThreads), lds, st, p); break;
            case 2: hipLaunchKernelGGL(ew_kernel<2>, dim3(items), dim3(kConvThreads), lds, st, p); break;
            case 3: hipLaunchKernelGGL(ew_kernel<3>, dim3(items), dim3(kConvThreads), lds, st, p); break;
            default: hipLaunchKernelGGL(ew_kernel<4>, dim3(items), dim3(kConvThreads), lds, st, p); break;
        }
        VF_HIP_CHECK(hipGetLastError());
        return VF_OK;
    }
    int lstm(const ConvLayer &l, const ConvParams &p, int /*u_prev*/, int /*u_x*/, int /*u_cond*/ = -1) { return conv(PH_LSTM, l, p, {}); }
    int cond(const CondParams &p, std::initializer_list<int>) {
        hipLaunchKernelGGL(cond_bias_kernel, dim3((p.B + kCondPerItem - 1) / kCondPerItem), dim3(256), 0, st, p);
        VF_HIP_CHECK(hipGetLastError());
        return VF_OK;
    }
    int conv_late(int type, const ConvLayer &l, const ConvParams &p, int /*u_early*/, int /*u_late*/, int /*u_extra*/ = -1) { return conv(type, l, p, {}); }
    static bool pair_capable() { return false; }    // one launch per layer: enc2 and enc3 stay two kernels
    static const ConvLayer &fc_plan(const vf_handle *h) { return h->fc; }
    int conv_pair(const ConvLayer &, const ConvParams &, const ConvLayer &, const ConvParams &, std::initializer_list<int>) {
        return VF_ERR_INVALID;
    }
    int sa(const SaParams &p, std::initializer_list<int>) {
        hipLaunchKernelGGL(sa_kernel, dim3(p.B), dim3(64), 0, st, p);
        return VF_OK;
    }
    int fin(const FinParams &p, std::initializer_list<int>) {
        hipLaunchKernelGGL(cdna_finalize_kernel, dim3(p.B), dim3(256), 0, st, p);
        return VF_OK;
    }
    int composite(const CompositeParams &p, int ntiles, int /*view*/, std::initializer_list<int>) {
        dim3 grid(ntiles, p.B);
        if (p.K == 6) {         // arch 2: four CDNA warps + previous + first frame + scratch
            switch (p.ND) {
                case 1: hipLaunchKernelGGL((composite_kernel<1, 6>), grid, dim3(256), 0, st, p); break;
                case 2: hipLaunchKernelGGL((composite_kernel<2, 6>), grid, dim3(256), 0, st, p); break;
                case 3: hipLaunchKernelGGL((composite_kernel<3, 6>), grid, dim3(256), 0, st, p); break;
                default: hipLaunchKernelGGL((composite_kernel<4, 6>), grid, dim3(256), 0, st, p); break;
            }
        } else {
            switch (p.ND) {
                case 1: hipLaunchKernelGGL((composite_kernel<1, 10>), grid, dim3(256), 0, st, p); break;
                case 2: hipLaunchKernelGGL((composite_kernel<2, 10>), grid, dim3(256), 0, st, p); break;
                case 3: hipLaunchKernelGGL((composite_kernel<3, 10>), grid, dim3(256), 0, st, p); break;
                default: hipLaunchKernelGGL((composite_kernel<4, 10>), grid, dim3(256), 0, st, p); break;
            }
        }
        VF_HIP_CHECK(hipGetLastError());
        return VF_OK;
    }
    int top(const ConvLayer &l, const ConvParams &p, const CompositeParams &cp, int ntiles, int view, int /*d_early*/,
            int /*d_late*/, int dfin, bool /*fuse*/) {      // one launch per layer: never fused
        int rc = conv(PH_CONVT_RAW, l, p, {});
        if (rc) return rc;
        return composite(cp, ntiles, view, {dfin});
    }
    static bool failed(int rc) { return rc != VF_OK; }
};
#endif

struct ScheduleSink {
    std::vector<PhaseDesc> phases;
    int next_ticket = 0, next_counter = 0;      // tickets are re-assigned when the views are merged
    double flops = 0.0;         // algorithmic FLOPs of all MFMA (conv / FC) phases
    size_t max_lds = 0;

    // how a consumer recognises that producer phase Q is done with a sample
    static PhaseDep dep_on(const PhaseDesc &Q) {
        PhaseDep dp;
        dp.cnt_base = Q.cnt_base;
        if (Q.whole) { dp.mode = 1; dp.expect = Q.n_items; }
        else {
            dp.mode = (Q.B == 1) ? 1 : 0;       // a batch-1 producer is shared by every sample
            switch (Q.type) {
                case PH_SA: case PH_CDNA_FIN: case PH_COND: dp.expect = 1; break;
                case PH_COMPOSITE: dp.expect = Q.gx; break;
                case PH_EW: dp.expect = Q.ew.spi > 0 ? 1 : Q.gx; break;
                default: dp.expect = (Q.NI == 1 ? Q.tiles_per_img : 1) * Q.gy;
            }
        }
        return dp;
    }
    int add(PhaseDesc &P, int n_items, int counters, std::initializer_list<int> deps) {
        P.first_ticket = next_ticket; P.n_items = n_items;
        P.cnt_base = next_counter;
        next_ticket += n_items; next_counter += counters;
        P.ndep = 0;
        for (int d : deps) {
            if (d < 0 || d == kSkipped) continue;
            P.dep[P.ndep++] = dep_on(phases[d]);
        }
        phases.push_back(P);
        return (int)phases.size() - 1;
    }
    bool early_start = true;
    // conv-LSTM of one step: u_prev = the same cell at the previous step (producer of the recurrent input and of the
    // cell state), u_x = the producer of the layer input.  Early start: the item is released by u_prev alone and
    // waits for u_x after its recurrent chunks (ConvParams::late_cnt; the counters' address is patched in at upload).
    int lstm(const ConvLayer &l, const ConvParams &p, int u_prev, int u_x, int u_cond = -1) {
        return conv_late(PH_LSTM, l, p, u_prev, u_x, u_cond);
    }
    // arch 2: the border-class biases of the tiled conditioning vector for one conv-LSTM, kCondPerItem samples per item
    // (the layer's conditioning weights are read once per item)
    int cond(const CondParams &p, std::initializer_list<int> deps) {
        PhaseDesc P;
        memset(&P, 0, sizeof(P));
        P.type = PH_COND; P.cond = p; P.B = p.B;
        return add(P, (p.B + kCondPerItem - 1) / kCondPerItem, p.B, deps);
    }
    // element-wise items of arch 3 (vf_savp3.h): gx items per sample, or one item per spi samples
    int ew(const EwParams &p, int view, std::initializer_list<int> deps) {
        PhaseDesc P;
        memset(&P, 0, sizeof(P));
        P.type = PH_EW; P.ew = p; P.B = p.B; P.view = view;
        P.gx = p.spi > 0 ? 1 : p.gx; P.gy = 1;
        const int items = p.spi > 0 ? (p.B + p.spi - 1) / p.spi : p.B * p.gx;
        max_lds = std::max(max_lds, ew_lds_bytes(p.op, p.op == EW_TOP3 ? p.top.ND : 1));
        return add(P, items, p.B, deps);
    }
    // a two-input tile whose segment 0 comes from u_early and whose segment 1 from u_late (decoder convs: the encoder
    // skip tensor first, the previous layer's output late)
    int conv_late(int type, const ConvLayer &l, const ConvParams &p, int u_early, int u_late, int u_extra = -1) {
        const bool late = early_start && u_late >= 0 && u_late != kSkipped;
        if (!late) return conv(type, l, p, {u_early, u_late, u_extra});
        const PhaseDep ld = dep_on(phases[u_late]);
        const int u = conv(type, l, p, {u_early, u_extra});
        if (u >= 0) { phases[u].has_late = 1; phases[u].late = ld; }
        return u;
    }
    int conv(int type, const ConvLayer &l, const ConvParams &p, std::initializer_list<int> deps) {
        PhaseDesc P;
        memset(&P, 0, sizeof(P));
        P.type = type; P.conv = p; P.B = p.B;
        P.NI = l.NI; P.tiles_per_img = l.tilesY * l.tilesX;
        P.gx = l.NI == 1 ? p.B * P.tiles_per_img : (p.B + l.NI - 1) / l.NI;
        P.gy = l.ncg;
        if (type == PH_FC_PARTIAL && l.mrep == 7) P.gy = 1;     // all column groups in one item (vf_fc_tile.h)
        P.whole = type == PH_FC_PARTIAL;
        P.mrep = l.gsplit ? (l.mrep == 2 ? 4 : (l.mrep == 0 ? 5 : (l.gs_v2 ? 6 : 3))) : l.mrep;
        P.prec = p.tile_variant;
        max_lds = std::max(max_lds, l.lds_bytes);
        const double rows = (double)p.B * l.Hout * l.Wout;
        const double taps = l.mode == PACK_CONVT ? 9.0 / 4.0 * 4.0 : (double)l.KH * l.KW;   // real taps
        if (!l.first_valu)      // (the first conv runs on the vector ALUs: not matrix work, not counted)
            flops += 2.0 * rows * taps * (l.segC[0] + (l.nseg > 1 ? l.segC[1] : 0) - p.chunk_begin * l.KC) *
                     (l.mode == PACK_LSTM ? 4.0 : (l.mode == PACK_PLAIN ? (double)l.G : 1.0)) * l.Cout;
        return add(P, P.gx * P.gy * l.nsplit, P.whole ? 1 : p.B, deps);
    }
    int sa(const SaParams &p, std::initializer_list<int> deps) {
        PhaseDesc P;
        memset(&P, 0, sizeof(P));
        P.type = PH_SA; P.sa = p; P.B = p.B;
        return add(P, (p.B + kSaPerItem - 1) / kSaPerItem, p.B, deps);
    }
    // a conv whose tiles hold whole images + the 1x1 conv that consumes it, as one item per row tile (conv_pair_epilogue):
    // the first conv's two channel groups become the two "gates" of the workgroup (ncg 1, G 2: the same packed weights)
    static bool pair_capable() { return true; }
    static const ConvLayer &fc_plan(const vf_handle *h) { return h->fc_wide_ok ? h->fc_wide : h->fc; }
    static bool pairable(const ConvLayer &a, const ConvLayer &b) {
        return a.mode == PACK_PLAIN && b.mode == PACK_PLAIN && a.nseg == 1 && b.nseg == 1 && a.ncg == 2 && b.ncg == 2 &&
               a.Cout == 64 && b.segC[0] == 64 && b.Cout <= 64 && b.KH == 1 && b.KW == 1 && b.stride == 1 && b.pad == 0 &&
               b.KC == 32 && a.tilesY * a.tilesX == 1 && b.tilesY * b.tilesX == 1 && a.NI == b.NI && a.RPI == b.RPI &&
               a.TH == b.TH && a.TW == b.TW && a.Hout == b.Hout && a.Wout == b.Wout && a.nsplit == 1 && b.nsplit == 1 &&
               (a.KH * a.KW * (a.KC / 8)) % 2 == 0 &&     // the G = 2 K loop of conv_tile runs whole rings of 4 or 2 steps
               a.lds_bytes >= (size_t)2 * 128 * 36 * 4;
    }
    int conv_pair(const ConvLayer &a, const ConvParams &pa, const ConvLayer &b, const ConvParams &pb,
                  std::initializer_list<int> deps) {
        PhaseDesc P;
        memset(&P, 0, sizeof(P));
        P.type = PH_CONV_PAIR; P.conv = pa; P.conv2 = pb; P.B = pa.B;
        P.conv.ncg = 1;         // both channel groups in this item
        P.NI = a.NI; P.tiles_per_img = 1;
        P.gx = (pa.B + a.NI - 1) / a.NI; P.gy = 1;
        P.mrep = 1;
        max_lds = std::max(max_lds, a.lds_bytes);
        const double rows = (double)pa.B * a.Hout * a.Wout;
        flops += 2.0 * rows * a.KH * a.KW * a.segC[0] * a.Cout + 2.0 * rows * b.segC[0] * b.Cout;
        return add(P, P.gx, pa.B, deps);
    }
    int fin(const FinParams &p, std::initializer_list<int> deps) {
        PhaseDesc P;
        memset(&P, 0, sizeof(P));
        P.type = PH_CDNA_FIN; P.fin = p; P.B = p.B;
        return add(P, p.B, p.B, deps);
    }
    int composite(const CompositeParams &p, int ntiles, int view, std::initializer_list<int> deps) {
        PhaseDesc P;
        memset(&P, 0, sizeof(P));
        P.type = PH_COMPOSITE; P.comp = p; P.B = p.B; P.gx = ntiles; P.view = view;
        return add(P, ntiles * p.B, p.B, deps);
    }
    // top transposed conv + compositing: one fused item per conv tile where the tile geometry allows it (a region
    // of whole 4 x 16 cost-sum blocks, one image and one channel group per tile, LDS), two phases otherwise
    static bool fusable(const ConvLayer &l, int ND) { return top_fusable(l, ND); }
    int top(const ConvLayer &l, const ConvParams &p, const CompositeParams &cp, int ntiles, int view, int d_early,
            int d_late, int dfin, bool fuse) {
        if (!fuse || !fusable(l, cp.ND)) {
            const int u = conv_late(PH_CONVT_RAW, l, p, d_early, d_late);
            if (u < 0) return u;
            return composite(cp, ntiles, view, {u, dfin});
        }
        const bool late = early_start && d_late >= 0 && d_late != kSkipped;
        PhaseDesc P;
        memset(&P, 0, sizeof(P));
        P.type = PH_TOP_FUSED; P.conv = p; P.comp = cp; P.B = p.B; P.view = view;
        P.NI = 1; P.tiles_per_img = l.tilesY * l.tilesX;
        P.gx = p.B * P.tiles_per_img; P.gy = 1;
        P.mrep = 1;
        P.aux_base = next_counter + p.B;        // behind the completion counters of this phase
        max_lds = std::max(max_lds, std::max(l.lds_bytes, fused_top_lds_floats(l.TH, l.TW, cp.ND) * 4));
        flops += 2.0 * (double)p.B * l.Hout * l.Wout * 9.0 * (l.segC[0] + (l.nseg > 1 ? l.segC[1] : 0)) * l.Cout;
        if (!late) return add(P, P.gx, 2 * p.B, {d_early, d_late, dfin});
        P.has_late = 1; P.late = dep_on(phases[d_late]);
        return add(P, P.gx, 2 * p.B, {d_early, dfin});
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
//
// goal_pix: this view's [ND][2] goal pixels (only the per-layer composite kernel reads them from
// its parameters; the persistent kernel receives all goals as launch arguments).
template <class Sink>
static int emit_rollout(vf_handle *h, int view, const BatchView &v, const BatchView &sh, int B,
                        const int32_t *goal_pix, Sink &sink, bool skip_shared) {
    const vf_config &c = h->cfg;
    const ViewData &vd = h->views[view];
    const int H = h->H, W = h->W, T = h->T, ND = h->ND, nc = c.n_context, Hc = h->Hc, Wc = h->Wc;
    const int H2 = Hc / 2, W2 = Wc / 2, H4 = Hc / 4, W4 = Wc / 4, H8 = Hc / 8, W8 = Wc / 8;
    const int *L = kLstmSizes;
    const int lh[7] = {H2, H2, H4, H4, H8, H4, H2}, lw[7] = {W2, W2, W4, W4, W8, W4, W2};
    auto params = [&](const ConvLayer &l, int Bp, const SegArg &s0, const SegArg *s1) {
        return make_params(l, vd.lw[l.id], Bp, s0, s1);
    };

    // tile plan of conv-LSTM k for a phase of Bp samples: the 256-row plan once the phase has many more items than
    // the chip has workgroup slots (everything from ~500 64x64-samples on; the two widest layers from ~150)
    auto lstm_plan = [&](int k, int Bp) -> const ConvLayer & {
        if (!h->have_big) return h->lstm[k];
        // Cost model of one phase on S = 2 x n_cu workgroup slots, fitted to the per-item times of the persistent
        // launch (profiles/r02_phase_stats_*.txt, r03_tile_plan_sweep.txt): an item of a plan with `rows` GEMM rows
        // costs d = fixed + slope * K microseconds (K = taps x input channels), a phase of n items takes about
        // max(d, n * d / S) - the per-sample chain, or the slot time.  The plan with the smallest estimate wins; within
        // 15 % the larger tile is kept (fewer items: less scheduling, staging and epilogue work per FLOP).  With the
        // early start of the conv-LSTM items (recurrent half under the previous layer) the larger tiles pay off at
        // smaller batches than they used to: at 125 samples lstm1/2/7 take 256 rows (500 items, one round of slots)
        // instead of 128 (1000 items), 55.5 -> 52.6 ms per rollout.
        const ConvLayer &mid = h->lstm[k];
        const double K = 25.0 * (mid.segC[0] + mid.segC[1]);
        const double S = 2.0 * h->n_cu;
        struct Cand { int want; const ConvLayer *l; double fixed, slope; };
        // (round 3: the 128-row tile is the gate-split one now - 0.91 of the matrix pipe with the CU to itself against
        // 0.84 for the 256-row tile, tools/trace_cu.py - and wins at every batch size from 100 to 1000 samples,
        // profiles/r03_plan_sweep_gsplit.txt; the 256-row entry is priced so that the 15 % preference for the larger tile below
        // no longer selects it)
        const Cand cands[4] = {{2, h->big_ok[k] ? &h->lstm_big[k] : nullptr, 50.0, 0.300},
                               {1, &mid, 33.0, 0.114},
                               {3, h->half_ok[k] ? &h->lstm_half[k] : nullptr, 30.0, 0.057},
                               {4, h->quarter_ok[k] ? &h->lstm_quarter[k] : nullptr, 28.0, 0.030}};
        int want = 1;
        double best = 0.0;
        for (const Cand &c : cands) {           // largest tile first
            if (!c.l) continue;
            const ConvLayer &l = *c.l;
            const double n = (double)(l.NI == 1 ? (long long)Bp * l.tilesY * l.tilesX : (Bp + l.NI - 1) / l.NI) * l.ncg;
            const double d = c.fixed + c.slope * K;
            const double t = std::max(d, n * d / S);
            if (best == 0.0 || t < 0.85 * best) { best = t; want = c.want; }
        }
        if (h->mrep_override[k]) want = h->mrep_override[k];
        if (want == 2 && h->big_ok[k]) return h->lstm_big[k];
        if (want == 4 && h->quarter_ok[k]) return h->lstm_quarter[k];
        if (want >= 3 && h->half_ok[k]) return h->lstm_half[k];
        return h->lstm[k];
    };
    // light layers of the bottleneck: one image per workgroup for small batches (measured: -2.1 % at 25 samples, 0 at 50, +0.3 % at 100, +1.1 % at 200)
    auto light_plan = [&](const ConvLayer &reg, const ConvLayer &one, int Bp) -> const ConvLayer & {
        if (reg.NI <= 1 || one.NI != 1 || one.KC != reg.KC) return reg;
        const long long items = (long long)((Bp + reg.NI - 1) / reg.NI) * reg.ncg;
        return items <= h->n_cu / 8 ? one : reg;
    };
    auto all_shared = [&](int s) { return h->dedup && s < nc - 1; };
    auto enc_shared = [&](int s) { return h->dedup && s < nc; };
    // (arch 2: the action conditions every conv-LSTM, so at step n_context - 1 only the convs in front of lstm1 still see
    // the same input for every sample)
    auto core_shared = [&](int s) { return h->cond ? all_shared(s) : enc_shared(s); };
    // is the output of lstm k at step s one shared image?  (s < 0: the shared zero state)
    auto lstm_shared = [&](int k, int s) { return s < 0 || all_shared(s) || (k < 4 && core_shared(s)); };

    auto plain = [](const float *ptr, long long bs) {
        SegArg s; memset(&s, 0, sizeof(s)); s.ptr = ptr; s.bstride = bs; s.gamma_mod = 1; return s;
    };
    auto normed = [](const float *ptr, long long bs, const long long *part, int nparts, int row_slots, bool shared,
                     long long count, const float *g, const float *b, int gmod, int relu) {
        SegArg s; s.ptr = ptr; s.bstride = bs; s.ln_part = part; s.ln_nparts = nparts;
        s.ln_bstride = shared ? 0 : (long long)row_slots * 2;
        s.ln_inv_n = (float)(1.0 / (double)count); s.gamma = g; s.beta = b; s.gamma_mod = gmod; s.relu = relu;
        return s;
    };
#define VF_EMIT(var, expr)                 \
    const int var = (expr);                \
    if (Sink::failed(var)) return var;
// a shared unit whose result is still valid from an earlier rollout is not emitted again
#define VF_EMIT_SH(var, shared, expr) VF_EMIT(var, ((shared) && skip_shared) ? Sink::skipped() : (expr))

    int last = -1;      // terminal unit of the previous step
    int u_cond_next[7] = {-1, -1, -1, -1, -1, -1, -1};     // arch 2: the conditioning-bias units of the coming step (emitted a step ahead)
    int u_prev[7] = {-1, -1, -1, -1, -1, -1, -1};   // conv-LSTM k of the previous step
    for (int s = 0; s < h->S; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        const bool produce = s >= nc - 1;
        const int t_out = s - (nc - 1);
        const bool enc0_sh = enc_shared(s), enc_sh = core_shared(s), all_sh = all_shared(s);
        const BatchView &E0 = enc0_sh ? sh : v;     // enc00 / enc0: the convs in front of lstm1
        const BatchView &E = enc_sh ? sh : v;       // the other encoder tensors of this step
        const BatchView &D = all_sh ? sh : v;       // tensors from enc3 on
        const int BE0 = enc0_sh ? 1 : B, BE = enc_sh ? 1 : B, BD = all_sh ? 1 : B;
        auto bs = [](bool shared, long long per) { return shared ? 0LL : per; };

        // ---- state FC + action/state bias of enc3
        SaParams sp; memset(&sp, 0, sizeof(sp));
        if (s < nc - 1) { sp.action = h->ctx_actions + (size_t)s * c.adim; sp.action_bstride = 0; }
        else { sp.action = v.actions + (size_t)(s - (nc - 1)) * c.adim; sp.action_bstride = (long long)T * c.adim; }
        if (s < nc) { sp.state = h->ctx_states + (size_t)s * c.sdim; sp.state_bstride = 0; }
        else { sp.state = v.states_all + (size_t)(s - nc) * c.sdim; sp.state_bstride = (long long)T * c.sdim; }
        sp.adim = c.adim; sp.sdim = c.sdim; sp.B = BD;
        sp.w_state = vd.w_state; sp.b_state = vd.b_state; sp.w_sa = vd.w_sa; sp.n_out = L[3];
        sp.state_out = produce ? v.states_all + (size_t)t_out * c.sdim : nullptr;
        sp.state_out_bstride = (long long)T * c.sdim;
        sp.sbias = D.sbias;
        VF_EMIT_SH(u_sa, all_sh, sink.sa(sp, {last}))

        // ---- arch 2: the conditioning biases of the seven conv-LSTMs of step `sc` (inputs: that step's action / latent and
        // the state the state FC of step sc - 1 produced; two buffers by step parity).  Step 0's are emitted here; those of
        // step s + 1 are emitted in the MIDDLE of step s, behind lstm5 (below): their only producer, this step's state FC, is
        // long done there, the phases around the 8 x 8 bottleneck are the narrow ones of a step (slots idle), and the head of
        // step s + 1 - state FC, first convs, lstm1: every sample's chain starts there - no longer queues behind 1100 bias items
        int u_cond[7];
        for (int k = 0; k < 7; ++k) u_cond[k] = u_cond_next[k];
        auto emit_cond = [&](const int sc, const int u_state, int (&out)[7]) -> int {
            for (int k = 0; k < 7; ++k) out[k] = -1;
            if (!h->cond || sc >= h->S) return VF_OK;
            const float *act, *sta; long long act_bs, sta_bs;
            if (sc < nc - 1) { act = h->ctx_actions + (size_t)sc * c.adim; act_bs = 0; }
            else { act = v.actions + (size_t)(sc - (nc - 1)) * c.adim; act_bs = (long long)T * c.adim; }
            if (sc < nc) { sta = h->ctx_states + (size_t)sc * c.sdim; sta_bs = 0; }
            else { sta = v.states_all + (size_t)(sc - nc) * c.sdim; sta_bs = (long long)T * c.sdim; }
            for (int k = 0; k < 7; ++k) {
                const bool shd = lstm_shared(k, sc);
                CondParams cq; memset(&cq, 0, sizeof(cq));
                cq.action = act; cq.action_bstride = act_bs;
                cq.state = sta; cq.state_bstride = sta_bs;
                cq.adim = c.adim; cq.sdim = c.sdim; cq.B = shd ? 1 : B;
                cq.w = vd.w_cond[k]; cq.C4 = 4 * L[k];
                cq.out = (shd ? sh : v).cond_bias[k][sc & 1];
                VF_EMIT_SH(u_ck, shd, sink.cond(cq, {u_state}))
                out[k] = u_ck;
            }
            return VF_OK;
        };
        if (s == 0) {
            const int rc0 = emit_cond(0, -1, u_cond);
            if (Sink::failed(rc0)) return rc0;
        }

        // ---- encoder
        const float *frame_in; long long frame_bs;
        if (s < nc) { frame_in = vd.ctx_frames + (size_t)s * H * W * 3; frame_bs = 0; }
        else { frame_in = v.frames_all + (size_t)(s - nc) * H * W * 3; frame_bs = (long long)T * H * W * 3; }

        ConvParams p;
        int u_enc00 = last;
        SegArg enc00_n = plain(nullptr, 0);
        if (h->savp) {      // extra encoder scale: enc00 = relu(LNa(conv5x5/2(frame))), applied on staging
            p = params(h->enc00, BE0, plain(frame_in, frame_bs), nullptr);
            p.out = E0.enc00_o; p.stats = E0.st_enc00;
            VF_EMIT_SH(u_e00, enc0_sh, sink.conv(PH_CONV_RAW, h->enc00, p, {last}))
            u_enc00 = u_e00;
            enc00_n = normed(E0.enc00_o, bs(enc0_sh, (long long)Hc * Wc * kEnc00Ch), E0.st_enc00, h->enc00.stats_nparts,
                             h->enc00.stats_nparts, enc0_sh, (long long)Hc * Wc * kEnc00Ch, vd.ln_g[9], vd.ln_b[9],
                             kEnc00Ch, 1);
        }
        p = params(h->enc0, BE0, h->savp ? enc00_n : plain(frame_in, frame_bs), nullptr);
        p.out = E0.enc0_o; p.stats = E0.st_enc0;
        VF_EMIT_SH(u_enc0, enc0_sh, sink.conv(PH_CONV_RAW, h->enc0, p, {u_enc00}))

        SegArg enc0_n = normed(E0.enc0_o, bs(enc0_sh, (long long)H2 * W2 * 32), E0.st_enc0, h->enc0.stats_nparts,
                               h->enc0.stats_nparts, enc0_sh, (long long)H2 * W2 * 32, vd.ln_g[0], vd.ln_b[0], 32, 1);
        // LayerNorm index: ln1 = enc0, ln2..ln8 = lstm1..7, ln9 = convt3
        auto h_normed = [&](int k) {        // normalised new hidden state of lstm k at this step
            const bool shd = lstm_shared(k, s);
            const BatchView &O = shd ? sh : v;
            const long long per = (long long)lh[k] * lw[k] * L[k];
            return normed(O.h_state[k][nxt], bs(shd, per), O.st_h[k], lstm_plan(k, shd ? 1 : B).stats_nparts,
                          h->st_rows[k], shd, per, vd.ln_g[k + 1], vd.ln_b[k + 1], L[k], 0);
        };
        auto lstm_params = [&](int k, const SegArg &x) {
            const bool out_sh = lstm_shared(k, s), in_sh = lstm_shared(k, s - 1);
            const BatchView &O = out_sh ? sh : v, &I = in_sh ? sh : v;
            const long long per = (long long)lh[k] * lw[k] * L[k];
            SegArg hs = plain(I.h_state[k][cur], bs(in_sh, per));
            ConvParams q = params(lstm_plan(k, out_sh ? 1 : B), out_sh ? 1 : B, hs, &x);     // recurrent input first
            q.out = O.h_state[k][nxt]; q.cstate = O.c_state[k]; q.stats = O.st_h[k];
            q.stats_nparts = h->st_rows[k];
            q.cstate_in = I.c_state[k]; q.cin_bstride = bs(in_sh, per);
            if (h->cond) q.cond_bias = O.cond_bias[k][s & 1];
#ifdef VF_DEBUG_KNOBS
            // TIMING ONLY (wrong results): the recurrent half of every conv-LSTM item vanishes - an upper bound on what
            // taking it off the per-sample dependency chain could give a small shard (round 4, EXPERIMENTS.md)
            static const bool no_rec = getenv("VF_TIMING_NO_RECURRENT") && atoi(getenv("VF_TIMING_NO_RECURRENT"));
            if (no_rec) { q.seg[0].nchunk = 0; q.chunks_per_split = q.seg[1].nchunk; }
#endif
            return q;
        };
        VF_EMIT_SH(u_l1, lstm_shared(0, s), sink.lstm(lstm_plan(0, lstm_shared(0, s) ? 1 : B), lstm_params(0, enc0_n), u_prev[0], u_enc0, u_cond[0]))
        VF_EMIT_SH(u_l2, lstm_shared(1, s), sink.lstm(lstm_plan(1, lstm_shared(1, s) ? 1 : B), lstm_params(1, h_normed(0)), u_prev[1], u_l1, u_cond[1]))

        p = params(h->enc1, BE, h_normed(1), nullptr);
        p.out = E.enc1_o;
        VF_EMIT_SH(u_enc1, enc_sh, sink.conv(PH_CONV_RELU, h->enc1, p, {u_l2}))

        VF_EMIT_SH(u_l3, lstm_shared(2, s), sink.lstm(lstm_plan(2, lstm_shared(2, s) ? 1 : B),
                                lstm_params(2, plain(E.enc1_o, bs(enc_sh, (long long)H4 * W4 * L[1]))), u_prev[2], u_enc1, u_cond[2]))
        VF_EMIT_SH(u_l4, lstm_shared(3, s), sink.lstm(lstm_plan(3, lstm_shared(3, s) ? 1 : B), lstm_params(3, h_normed(2)), u_prev[3], u_l3, u_cond[3]))

        const ConvLayer &enc2_l = light_plan(h->enc2, h->enc2_one, BE);
        const ConvLayer &enc3_l = light_plan(h->enc3, h->enc3_one, BD);
        p = params(enc2_l, BE, h_normed(3), nullptr);
        p.out = E.enc2_o;
        ConvParams p3 = params(enc3_l, BD, plain(E.enc2_o, bs(enc_sh, (long long)H8 * W8 * L[3])), nullptr);
        p3.out = D.enc3_o; p3.sbias = D.sbias; p3.sbias_ld = L[3];
        int u_enc3 = -1;
        // enc2 + enc3 as ONE item per row tile where both run on the same samples (the persistent schedule, from the first
        // step without shared encoder units on): one dependency hop and one memory round trip less on every sample's chain
        // (small batches: the one-image plan of enc3 does not match enc2's row tiles - enc2 keeps its regular plan because
        //  its weights are packed for that plan's 16-channel chunks - so the pair takes enc3's regular plan)
        const bool pair_ok = Sink::pair_capable() && h->fuse_pair && enc_sh == all_sh && BE == BD;
        if (pair_ok && !ScheduleSink::pairable(enc2_l, enc3_l) && ScheduleSink::pairable(enc2_l, h->enc3)) {
            p3 = params(h->enc3, BD, plain(E.enc2_o, bs(enc_sh, (long long)H8 * W8 * L[3])), nullptr);
            p3.out = D.enc3_o; p3.sbias = D.sbias; p3.sbias_ld = L[3];
            VF_EMIT_SH(u_pair, enc_sh, sink.conv_pair(enc2_l, p, h->enc3, p3, {u_l4, u_sa}))
            u_enc3 = u_pair;
        } else if (pair_ok && ScheduleSink::pairable(enc2_l, enc3_l)) {
            VF_EMIT_SH(u_pair, enc_sh, sink.conv_pair(enc2_l, p, enc3_l, p3, {u_l4, u_sa}))
            u_enc3 = u_pair;
        } else {
            VF_EMIT_SH(u_enc2, enc_sh, sink.conv(PH_CONV_RELU, enc2_l, p, {u_l4}))
            VF_EMIT_SH(u_enc3_, all_sh, sink.conv(PH_CONV_RELU, enc3_l, p3, {u_enc2, u_sa}))
            u_enc3 = u_enc3_;
        }

        VF_EMIT_SH(u_l5, lstm_shared(4, s), sink.lstm(lstm_plan(4, lstm_shared(4, s) ? 1 : B),
                                lstm_params(4, plain(D.enc3_o, bs(all_sh, (long long)H8 * W8 * L[3]))), u_prev[4], u_enc3, u_cond[4]))
        SegArg h5n = h_normed(4);

        // ---- decoder
        const ConvLayer &convt1_l = light_plan(h->convt1, h->convt1_one, BD);
        p = params(convt1_l, BD, h5n, nullptr);
        p.out = D.enc4_o;
        {   // (arch 2) the conditioning biases of the NEXT step: see the head of the step
            const int rcn = emit_cond(s + 1, u_sa, u_cond_next);
            if (Sink::failed(rcn)) return rcn;
        }
        VF_EMIT_SH(u_t1, all_sh, sink.conv(PH_CONVT_RELU, convt1_l, p, {u_l5}))
        VF_EMIT_SH(u_l6, lstm_shared(5, s), sink.lstm(lstm_plan(5, lstm_shared(5, s) ? 1 : B),
                                lstm_params(5, plain(D.enc4_o, bs(all_sh, (long long)H4 * W4 * L[4]))), u_prev[5], u_t1, u_cond[5]))

        // ---- CDNA kernels (only needed when this step's prediction is used).  The FC needs lstm5 of EVERY sample
        // and its only consumer is the compositing at the end of the step: it is emitted here, behind lstm6, where
        // its ready-to-run items fill the slots that would otherwise draw transposed-conv items still waiting for
        // the second round of lstm6 tiles.
        int u_fin = -1, u_fc = -1;
        if (produce) {
            SegArg flat = h5n;      // same LayerNorm, viewed as [B][1][1][H8*W8*128]
            const ConvLayer &fc_l = Sink::fc_plan(h);
            p = params(fc_l, B, flat, nullptr);
            p.out = v.fc_part;
            VF_EMIT(u_fc_, sink.conv(PH_FC_PARTIAL, fc_l, p, {u_l5}))
            u_fc = u_fc_;
        }

        SegArg enc1_s = plain(E.enc1_o, bs(enc_sh, (long long)H4 * W4 * L[1])), h6n = h_normed(5);
        p = params(h->convt2, BD, enc1_s, &h6n);        // the skip tensor first: it exists since the encoder (early start)
        p.out = D.enc5_o;
        VF_EMIT_SH(u_t2, all_sh, sink.conv_late(PH_CONVT_RELU, h->convt2, p, u_enc1, u_l6))
        VF_EMIT_SH(u_l7, lstm_shared(6, s), sink.lstm(lstm_plan(6, lstm_shared(6, s) ? 1 : B),
                                lstm_params(6, plain(D.enc5_o, bs(all_sh, (long long)H2 * W2 * h->c_t2))), u_prev[6], u_t2, u_cond[6]))
        last = u_l7;
        {
            const int now[7] = {u_l1, u_l2, u_l3, u_l4, u_l5, u_l6, u_l7};
            for (int k = 0; k < 7; ++k) u_prev[k] = now[k];
        }
        if (produce) {      // the (tiny) finalise step of the CDNA kernels: behind lstm7, by when the FC has long finished
            FinParams fp;
            fp.partial = v.fc_part; fp.nsplit = h->fc.nsplit; fp.B = B; fp.K = h->K;
            fp.bias = vd.b_fc; fp.kern = v.kern;
            u_fin = sink.fin(fp, {u_fc});
            if (Sink::failed(u_fin)) return u_fin;
        }

        if (produce) {      // never an all-shared step
            // the top transposed conv and the compositing go to the sink together: the persistent schedule may
            // fuse them into one item per tile (vf_fused_top.h)
            SegArg h7n = h_normed(6);
            p = params(h->convt3, B, enc0_n, &h7n);     // skip tensor first (early start)
            p.out = v.enc6_o; p.stats = v.st_enc6;
            const ConvLayer *top_l = &h->convt3;
            int top_early = u_enc0, top_late = u_l7;
            if (h->savp) {  // extra decoder scale: enc7 = convT(concat[relu(LN9(enc6)), relu(LNa(enc00))]), LNb on use
                // (an encoder-shared enc00 of a context step is read with batch stride 0)
                SegArg enc6_n = normed(v.enc6_o, (long long)Hc * Wc * 32, v.st_enc6, h->convt3.stats_nparts,
                                       h->convt3.stats_nparts, false, (long long)Hc * Wc * 32, vd.ln_g[8], vd.ln_b[8], 32, 1);   // (arch 1 / 2: c_top = 32)
                VF_EMIT(u_t3, sink.conv_late(PH_CONVT_RAW, h->convt3, p, u_enc0, u_l7))
                p = params(h->convt4, B, enc00_n, &enc6_n);
                p.out = v.enc7_o; p.stats = v.st_enc7;
                top_l = &h->convt4; top_early = u_enc00; top_late = u_t3;
            }

            CompositeParams cp; memset(&cp, 0, sizeof(cp));
            cp.B = B; cp.H = H; cp.W = W; cp.ND = ND; cp.K = h->K;
            if (h->savp) {
                cp.enc6 = v.enc7_o; cp.ln_part = v.st_enc7; cp.ln_nparts = h->convt4.stats_nparts;
                cp.gamma = vd.ln_g[10]; cp.beta = vd.ln_b[10];
                cp.first_frame = vd.ctx_frames; cp.first_distrib = vd.ctx_distrib;     // context frame 0
            } else {
                cp.enc6 = v.enc6_o; cp.ln_part = v.st_enc6; cp.ln_nparts = h->convt3.stats_nparts;
                cp.gamma = vd.ln_g[8]; cp.beta = vd.ln_b[8];
            }
            cp.ln_inv_n = (float)(1.0 / ((double)H * W * h->c_top));
            cp.CF = h->c_top;
            cp.w_rgb = vd.w_rgb; cp.b_rgb = vd.b_rgb; cp.w_mask = vd.w_mask; cp.b_mask = vd.b_mask;
            cp.kern = v.kern;
            cp.prev_frame = frame_in; cp.prev_frame_bstride = frame_bs;
            if (s < nc) {
                cp.prev_distrib = vd.ctx_distrib + (size_t)s * H * W * ND; cp.prev_distrib_bstride = 0;
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
            if (goal_pix)
                for (int d = 0; d < ND; ++d) { cp.goal[d][0] = goal_pix[2 * d]; cp.goal[d][1] = goal_pix[2 * d + 1]; }
            VF_EMIT(u_comp, sink.top(*top_l, p, cp, h->ntiles, view, top_early, top_late, u_fin, h->fuse_top))
            last = u_comp;
        }
    }
#undef VF_EMIT_SH
#undef VF_EMIT
    return VF_OK;
}

// Write-through publish (ConvParams::wt_out: the item bumps its completion counters WITHOUT a release fence) is only sound
// for a tile ALL of whose global stores are sc1 / agent-scope atomic stores.  This is the one place that knows which
// (phase type, tile plan, output width) combinations dispatch such an epilogue on the device (vf_persistent.h's switch,
// conv_epilogue's kVec predicate, lstm_gsplit_epilogue / gates_raw_epilogue): build_schedule derives wt_out from it - never
// from the phase type alone - and vf_selftest_schedule re-checks every phase against the restated conditions.
static bool wt_epilogue(const PhaseDesc &P) {
    switch (P.type) {
        case PH_EW:             // vf_savp3.h: the items whose every output is a 16-byte store (OutBuf / the block turn-over of the
                                // fused heads, which needs whole 4 x 16 blocks) or an atomic store
            if ((VF_WT_DEFAULT & 2) == 0) return false;
            if (P.ew.op == EW_INORM || P.ew.op == EW_INCELL || P.ew.op == EW_UPSAMPLE) return true;
            return P.ew.op == EW_TOP3 && P.ew.top.H % kSumBlockH == 0 && P.ew.top.W % kSumBlockW == 0;
        case PH_LSTM:           // gate-split 128- / 64-row tiles and the 32-row tile (lstm_gsplit_epilogue), exact fp32
            return P.prec == 0 && (P.mrep == 6 || P.mrep == 5 || P.mrep == -1);
        case PH_CONV_RELU: case PH_CONV_RAW: case PH_CONVT_RELU: case PH_CONVT_RAW:
            // conv_epilogue's vectorised form: one row block per wave, whole channel quads; the vector-ALU first conv
            // (mrep 8, vf_conv_first.h: Cout / 4 16-byte stores per pixel + an atomic-store partial)
            return (VF_WT_DEFAULT & 2) != 0 && (P.mrep == 1 || (P.type == PH_CONV_RAW && P.mrep == 8)) && P.conv.Cout % 4 == 0;
        case PH_CONV_RAW3: case PH_CONV_RAW3G2: case PH_CONV_RAW3G4:     // EPI_RAW is vectorised for one and two row blocks per wave
            return (VF_WT_DEFAULT & 2) != 0 && (P.mrep == 1 || P.mrep == 2) && P.conv.Cout % 4 == 0;
        case PH_GATES_RAW:      // gates_raw_epilogue: sixteen 16-byte stores per lane, nothing else
            return true;
        case PH_TOP_FUSED:      // fused_top_body: blocks turned over in LDS into 16-byte stores, partials and sums as atomic stores
            return (VF_WT_DEFAULT & 2) != 0;
        default:
            return false;
    }
}

// ------------------------------------------------------------------ persistent schedule (host part)
struct BuiltSchedule {
    std::vector<PhaseDesc> phases;
    int items = 0, counters = 0;
    int nq = 1, total_q[kQueues] = {0};
    double flops = 0.0;
    size_t lds = 0;
};

// One phase list per view (own weights, own buffers, own counters), merged phase by phase so that
// the views advance together and a phase's items of both views are neighbours in ticket order.
static int build_schedule(vf_handle *h, int B, bool skip_shared, BuiltSchedule &out) {
    std::vector<ScheduleSink> sinks(h->ncam);
    VF_INJECT(1);
    int counters = 0, rc;
    out.flops = 0.0;
    size_t max_lds = 0;
    for (int v = 0; v < h->ncam; ++v) {
        sinks[v].next_counter = counters;
        sinks[v].early_start = h->early_start;
        if (h->cfg.arch == 3)
            rc = emit_rollout_s3(h, v, B, h->actions_buf, nullptr, sinks[v]);
        else
            rc = emit_rollout(h, v, make_view(h, v, h->actions_buf, 0), h->shared_views[(size_t)v], B, nullptr, sinks[v],
                              skip_shared);
        if (rc < 0) return rc;
        counters = sinks[v].next_counter;
        out.flops += sinks[v].flops;
        max_lds = std::max(max_lds, sinks[v].max_lds);
    }
    out.phases.clear();
    const size_t n = sinks[0].phases.size();
    for (size_t i = 0; i < n; ++i)
        for (int v = 0; v < h->ncam; ++v) out.phases.push_back(sinks[v].phases[i]);
    int ticket = 0;
    for (PhaseDesc &P : out.phases) { P.first_ticket = ticket; ticket += P.n_items; }
    out.items = ticket;
#ifndef VF_DEBUG_KNOBS
    // the persistent kernel of a production build carries the tiles 6 / 5 / -1 / 1 only (vf_persistent.h)
    for (const PhaseDesc &P : out.phases)
        if (P.type == PH_LSTM && P.prec == 0 && !(P.mrep == 6 || P.mrep == 5 || P.mrep == -1 || P.mrep == 1))
            return fail(VF_ERR_INVALID, "conv-LSTM tile plan " + std::to_string(P.mrep) +
                                            " is not compiled into this build (needs -DVF_DEBUG_KNOBS)");
#endif
    // Deal every phase's items to the XCD queues (vf_persistent.h): item = (unit * q_inner + inner) * q_gy + cg
    // goes to queue (unit % (nq / q_gy)) * q_gy + cg.  unit = sample (or sample group) of the item, inner = its
    // tile within the sample, cg = output-channel group.  Launches too small to occupy every XCD keep one queue.
    // one queue per XCD the device exposes: 8 on the whole chip (256 CUs), fewer in a partitioned mode (a queue
    // that no workgroup calls its own would only be served by thieves - too late for the items that wait on it)
    int n_xcd = 1;
    while (n_xcd < kQueues && n_xcd * 2 * 32 <= h->n_cu) n_xcd *= 2;
    const int nq = (h->xcd_queues > 1 && ticket >= 4 * h->n_cu) ? n_xcd : 1;
    out.nq = nq;
    for (int q = 0; q < kQueues; ++q) out.total_q[q] = 0;
    for (PhaseDesc &P : out.phases) {
        P.q_gy = 1; P.q_inner = 1;
        if (nq > 1) {
            if (ph_is_conv(P.type) || P.type == PH_TOP_FUSED) {     // (fused: a sample's tiles MUST share a queue)
                if (P.gy <= nq && nq % P.gy == 0) P.q_gy = P.gy;
                if (P.NI == 1 && P.n_items == P.gx * P.gy) P.q_inner = P.tiles_per_img;
                if (P.n_items % (P.q_gy * P.q_inner)) { P.q_gy = 1; P.q_inner = 1; }
            } else if (P.type == PH_COMPOSITE || (P.type == PH_EW && P.ew.spi == 0)) {
                P.q_inner = P.gx;      // (a sample's items stay in one queue - the one its GEMM tiles' outputs are warm in)
            }
        }
        const int units = P.n_items / (P.q_gy * P.q_inner), per = nq / P.q_gy;
        // The units that do not fill a whole round of the queue groups are dealt tile by tile (PhaseDesc::q_full) -
        // except where a sample's tiles must meet in one queue (fused top: its tiles wait for each other; compositing).
        const bool split_tail = nq > 1 && P.q_inner > 1 && ph_is_conv(P.type);
        const int full_units = split_tail ? (units / per) * per : units;
        const int n_tail = (units - full_units) * P.q_inner;       // (unit, tile) pairs dealt one by one
        P.q_full = split_tail ? (full_units / per) * P.q_inner : 0x7FFFFFFF;
        for (int q = 0; q < kQueues; ++q) {
            const int qb = q / P.q_gy;
            P.first_q[q] = out.total_q[q < nq ? q : 0];
            if (q >= nq) P.n_q[q] = 0;
            else if (split_tail) P.n_q[q] = P.q_full + (qb < n_tail ? (n_tail - qb + per - 1) / per : 0);
            else P.n_q[q] = qb < units ? ((units - qb + per - 1) / per) * P.q_inner : 0;
            if (q < nq) out.total_q[q] += P.n_q[q];
        }
    }
    for (PhaseDesc &P : out.phases) {
        if (ph_is_conv(P.type) || P.type == PH_TOP_FUSED) P.conv.wt_out = (h->wt_publish && wt_epilogue(P)) ? 1 : 0;
        if (P.type == PH_EW) P.ew.wt = (h->wt_publish && wt_epilogue(P)) ? 1 : 0;
    }
    out.counters = counters;
    const size_t comp_lds[kMaxDesig] = {(size_t)composite_lds_floats<1, 10>() * 4, (size_t)composite_lds_floats<2, 10>() * 4,
                                        (size_t)composite_lds_floats<3, 10>() * 4, (size_t)composite_lds_floats<4, 10>() * 4};
    out.lds = std::max(max_lds, comp_lds[h->ND - 1]) + 16;
    if (out.phases.size() > h->sched_capacity || (size_t)counters > h->counter_capacity)
        return fail(VF_ERR_INVALID, "persistent schedule exceeds its preallocated capacity");
    return VF_OK;
}

#ifdef VF_HOST_SELFTEST
// ------------------------------------------------------------------ host self-test hooks
// (tools/host_selftest.cc; ASan/UBSan build).  Checks the invariants the device relies on.
static bool in_allocs(const vf_handle *h, const void *p, size_t bytes) {
    if (!p) return true;
    const char *q = static_cast<const char *>(p);
    for (const AllocRec &a : h->allocs) {
        const char *lo = static_cast<const char *>(a.p);
        if (q >= lo && q + bytes <= lo + a.bytes) return true;
    }
    return false;
}

extern "C" int vf_selftest_schedule(vf_handle *h, int32_t B, int32_t skip_shared, int64_t *out_items,
                                    uint64_t *out_upload_checksum) {
    VF_API_TRY
    if (!h) return fail(VF_ERR_INVALID, "null handle");
    BuiltSchedule bs;
    int rc = build_schedule(h, B, skip_shared != 0, bs);
    if (rc) return rc;
    int ticket = 0;
    for (size_t i = 0; i < bs.phases.size(); ++i) {
        const PhaseDesc &P = bs.phases[i];
        if (P.first_ticket != ticket || P.n_items <= 0) return fail(VF_ERR_INVALID, "tickets are not contiguous");
        ticket += P.n_items;
        if (P.ndep < 0 || P.ndep > kMaxDeps) return fail(VF_ERR_INVALID, "bad dependency count");
        for (int d = 0; d < P.ndep; ++d) {
            // a dependency must point at the counters of a phase with smaller tickets
            bool found = false;
            for (size_t j = 0; j < i && !found; ++j) found = bs.phases[j].cnt_base == P.dep[d].cnt_base;
            if (!found) return fail(VF_ERR_INVALID, "phase " + std::to_string(i) + " depends on a later phase");
            if (P.dep[d].expect <= 0) return fail(VF_ERR_INVALID, "non-positive expected count");
        }
        if (P.has_late) {
            bool found = false;
            for (size_t j = 0; j < i && !found; ++j) found = bs.phases[j].cnt_base == P.late.cnt_base;
            if (!found || !(ph_is_conv(P.type) || P.type == PH_TOP_FUSED) || P.late.expect <= 0)
                return fail(VF_ERR_INVALID, "phase " + std::to_string(i) + ": bad late dependency");
        }
        const int ncnt = P.whole ? 1 : (P.type == PH_TOP_FUSED ? 2 * P.B : P.B);
        if (P.cnt_base < 0 || P.cnt_base + ncnt > bs.counters) return fail(VF_ERR_INVALID, "counter out of range");
        // every pointer a tile dereferences must lie inside an allocation of this handle
        bool ok = true;
        if (P.type == PH_CONV_PAIR) {      // the 1x1 conv behind the first one: weights, biases, output, per-sample bias
            const ConvParams &c2 = P.conv2;
            ok = ok && P.conv.ncg == 1 && c2.ncg == 2 && c2.KC == 32 && c2.nseg == 1 && c2.seg[0].nchunk == 2;
            ok = ok && in_allocs(h, c2.Wp, (size_t)2 * 4 * 2 * 64 * 4 * 4) && in_allocs(h, c2.bias, 64 * 4);
            ok = ok && in_allocs(h, c2.out, (size_t)P.B * c2.Hout * c2.Wout * c2.Cout * 4);
            ok = ok && in_allocs(h, c2.sbias, c2.sbias ? (size_t)((P.B - 1) * c2.sbias_ld + c2.Cout) * 4 : 0);
        }
        if (P.type <= PH_FC_PARTIAL || ph_is_conv(P.type) || P.type == PH_TOP_FUSED || P.type == PH_CONV_PAIR) {
            const ConvParams &c = P.conv;
            for (int s = 0; s < c.nseg; ++s) {
                const long long span = (long long)(P.B - 1) * c.seg[s].bstride + (long long)c.Hin * c.Win * c.seg[s].C;
                ok = ok && in_allocs(h, c.seg[s].ptr, (size_t)span * 4);
                ok = ok && in_allocs(h, c.seg[s].ln_part, c.seg[s].ln_part ? (size_t)((P.B - 1) * c.seg[s].ln_bstride + c.seg[s].ln_nparts * 2) * 8 : 0);
            }
            ok = ok && in_allocs(h, c.Wp, 16) && in_allocs(h, c.bias, 4) && in_allocs(h, c.out, 4);
            ok = ok && in_allocs(h, c.cstate, 4) && in_allocs(h, c.cstate_in, 4) && in_allocs(h, c.stats, 8);
        }
        // a write-through item must be one whose every store is a 16-byte sc1 store (restated here, not read from wt_epilogue)
        if (P.conv.wt_out != 0) {
            const bool lstm_vec = P.type == PH_LSTM && P.prec == 0 && (P.mrep == 6 || P.mrep == 5 || P.mrep == -1);
            const bool light_vec = P.type >= PH_CONV_RELU && P.type <= PH_CONVT_RAW && P.conv.Cout % 4 == 0 &&
                                   (P.mrep == 1 || (P.mrep == 8 && P.type == PH_CONV_RAW && (P.conv.Cout == 16 || P.conv.Cout == 32)));
            const bool raw3_vec = (P.type == PH_CONV_RAW3 || P.type == PH_CONV_RAW3G2 || P.type == PH_CONV_RAW3G4) &&
                                  (P.mrep == 1 || P.mrep == 2) && P.conv.Cout % 4 == 0;
            if (!(lstm_vec || light_vec || raw3_vec || P.type == PH_GATES_RAW || P.type == PH_TOP_FUSED))
                return fail(VF_ERR_INVALID, "phase " + std::to_string(i) + ": write-through publish on a tile with plain stores");
        }
        if (P.type == PH_EW && P.ew.wt != 0) {
            const int op = P.ew.op;
            const bool vec = op == EW_INORM || op == EW_INCELL || op == EW_UPSAMPLE ||
                             (op == EW_TOP3 && P.ew.top.H % 4 == 0 && P.ew.top.W % 16 == 0);
            if (!vec) return fail(VF_ERR_INVALID, "phase " + std::to_string(i) + ": write-through publish on an item with plain stores");
        }
        if (P.type == PH_TOP_FUSED && (P.aux_base != P.cnt_base + P.B || P.aux_base + P.B > bs.counters))
            return fail(VF_ERR_INVALID, "fused phase: bad auxiliary counters");
        if (P.type <= PH_FC_PARTIAL || ph_is_conv(P.type)) {
        } else if (P.type == PH_EW) {
            const EwParams &e = P.ew;
            if (e.op == EW_INORM || e.op == EW_INCELL) {
                const NormParams &q = e.norm;
                const size_t hw = (size_t)q.H * q.W, ctot = (size_t)(e.op == EW_INCELL ? 4 : 1) * q.C;
                ok = ok && in_allocs(h, q.in, ((size_t)(P.B - 1) * q.in_bs + hw * ctot) * 4);
                const size_t co = q.split > 0 ? (size_t)q.split : (size_t)q.C;
                ok = ok && in_allocs(h, q.out, ((size_t)(P.B - 1) * q.out_bs + hw * co) * 4);
                ok = ok && (q.split == 0 || (in_allocs(h, q.out2, ((size_t)(P.B - 1) * q.out_bs + hw * co) * 4) && q.C == 2 * q.split));
                ok = ok && in_allocs(h, q.cond, q.cond ? ((size_t)(P.B - 1) * q.cond_bs + 25 * ctot) * 4 : 0);
                ok = ok && in_allocs(h, q.g0, ctot * 4) && in_allocs(h, q.b0, ctot * 4) && q.C % q.cpi == 0 && e.gx == q.C / q.cpi;
                ok = ok && in_allocs(h, q.tab, q.tab ? (size_t)P.B * q.C * 2 * 4 : 0);
                if (e.op == EW_INCELL)
                    ok = ok && in_allocs(h, q.cout, ((size_t)(P.B - 1) * q.cout_bs + hw * q.C) * 4) &&
                         in_allocs(h, q.cprev, q.cprev ? ((size_t)(P.B - 1) * q.cprev_bs + hw * q.C) * 4 : 0) &&
                         in_allocs(h, q.g1, (size_t)q.C * 4) && in_allocs(h, q.b1, (size_t)q.C * 4);
            } else if (e.op == EW_UPSAMPLE) {
                const UpParams &q = e.up;
                const size_t hw = (size_t)q.h * q.w;
                ok = ok && in_allocs(h, q.in0, ((size_t)(P.B - 1) * q.in0_bs + hw * q.C0) * 4);
                ok = ok && in_allocs(h, q.in1, q.C1 ? ((size_t)(P.B - 1) * q.in1_bs + hw * q.C1) * 4 : 0);
                ok = ok && in_allocs(h, q.out, ((size_t)(P.B - 1) * q.out_bs + 4 * hw * (q.C0 + q.C1)) * 4);
                ok = ok && e.gx * q.rows >= 2 * q.h && (size_t)q.rows * 2 * q.w * ((q.C0 + q.C1) / 4) < 65536;
            } else if (e.op == EW_SA3) {
                const Sa3Params &q = e.sa;
                const int ncond = q.adim + q.sdim;
                ok = ok && in_allocs(h, q.condvec, (size_t)P.B * ncond * 4) && in_allocs(h, q.rnn_state, (size_t)P.B * 2 * q.zdim * 4) &&
                     in_allocs(h, q.action, 4) && in_allocs(h, q.state, 4);
            } else if (e.op == EW_COND3) {
                const Cond3Params &q = e.cond;
                ok = ok && in_allocs(h, q.condvec, (size_t)P.B * q.ncond * 4) && in_allocs(h, q.out, (size_t)P.B * 25 * q.Ctot * 4) &&
                     in_allocs(h, q.w, (size_t)q.KH * q.KH * q.ncond * q.Ctot * 4) && q.KH <= 6;
            } else {
                const TopParams &q = e.top;
                const size_t hw = (size_t)q.H * q.W;
                if (e.op == EW_TOP3) {
                    ok = ok && in_allocs(h, q.hmhs_raw, (size_t)P.B * hw * 64 * 4) && in_allocs(h, q.norm_tab, (size_t)P.B * 128 * 4);
                    ok = ok && in_allocs(h, q.w_scr, (size_t)9 * 32 * 4 * 4) && in_allocs(h, q.b_scr, 16);
                    ok = ok && in_allocs(h, q.w_msk, (size_t)9 * kT3MaskIn * 8 * 4) && in_allocs(h, q.b_msk, 32);
                    ok = ok && e.gx == ((q.H + kT3H - 1) / kT3H) * ((q.W + kT3W - 1) / kT3W);
                } else {
                    ok = ok && in_allocs(h, q.trans, (size_t)P.B * hw * kTransCh * 4) && in_allocs(h, q.transd, (size_t)P.B * hw * kNumWarp3 * q.ND * 4);
                    ok = ok && in_allocs(h, q.mlog, (size_t)P.B * hw * kMaskCh * 4) && in_allocs(h, q.scr_raw, (size_t)P.B * hw * kScrCh * 4);
                }
                ok = ok && in_allocs(h, q.prev_frame, ((size_t)(P.B - 1) * q.prev_frame_bs + hw * 3) * 4);
                ok = ok && in_allocs(h, q.prev_distrib, ((size_t)(P.B - 1) * q.prev_distrib_bs + hw * q.ND) * 4);
                ok = ok && in_allocs(h, q.out_frame, ((size_t)(P.B - 1) * q.out_frame_bs + hw * 3) * 4);
                ok = ok && in_allocs(h, q.out_distrib, ((size_t)(P.B - 1) * q.out_distrib_bs + hw * q.ND) * 4);
                ok = ok && in_allocs(h, q.out_sums, (size_t)P.B * q.ND * h->nblocks * 2 * 8) && in_allocs(h, q.kern, (size_t)P.B * kTaps * kNumWarp3 * 4);
            }
        } else if (P.type == PH_COMPOSITE || P.type == PH_TOP_FUSED) {
            const CompositeParams &c = P.comp;
            const size_t hw = (size_t)c.H * c.W;
            if (P.type == PH_COMPOSITE) ok = ok && in_allocs(h, c.enc6, (size_t)P.B * hw * (c.CF > 32 ? c.CF : 32) * 4);
            ok = ok && in_allocs(h, c.prev_frame, ((size_t)(P.B - 1) * c.prev_frame_bstride + hw * 3) * 4);
            ok = ok && in_allocs(h, c.prev_distrib, ((size_t)(P.B - 1) * c.prev_distrib_bstride + hw * c.ND) * 4);
            ok = ok && in_allocs(h, c.out_frame, ((size_t)(P.B - 1) * c.out_frame_bstride + hw * 3) * 4);
            ok = ok && in_allocs(h, c.out_distrib, ((size_t)(P.B - 1) * c.out_distrib_bstride + hw * c.ND) * 4);
            ok = ok && in_allocs(h, c.out_sums, (size_t)P.B * c.ND * h->nblocks * 2 * 8);
            ok = ok && in_allocs(h, c.kern, (size_t)P.B * kTaps * c.K * 4);
            ok = ok && in_allocs(h, c.first_frame, hw * 3 * 4) && in_allocs(h, c.first_distrib, hw * c.ND * 4);
        } else if (P.type == PH_SA) {
            ok = ok && in_allocs(h, P.sa.sbias, (size_t)P.B * P.sa.n_out * 4) && in_allocs(h, P.sa.action, 4) &&
                 in_allocs(h, P.sa.state, 4);
        } else if (P.type == PH_COND) {
            ok = ok && in_allocs(h, P.cond.out, (size_t)P.B * kCondClasses * P.cond.C4 * 4) &&
                 in_allocs(h, P.cond.w, (size_t)kTaps * (P.cond.adim + P.cond.sdim) * P.cond.C4 * 4) &&
                 in_allocs(h, P.cond.action, 4) && in_allocs(h, P.cond.state, 4);
        } else if (P.type == PH_CONV_PAIR) {
        } else {
            ok = ok && in_allocs(h, P.fin.partial, (size_t)P.fin.nsplit * P.B * kTaps * P.fin.K * 4) &&
                 in_allocs(h, P.fin.kern, (size_t)P.B * kTaps * P.fin.K * 4);
        }
        if (!ok) return fail(VF_ERR_INVALID, "phase " + std::to_string(i) + " points outside the handle's buffers");
    }
    if (ticket != bs.items) return fail(VF_ERR_INVALID, "item count mismatch");
    // the queue dealing must be a bijection: walking every queue position of a phase with the device's formula
    // (rollout_persistent_kernel) visits each item of each phase exactly once, queues in phase order
    {
        int head[kQueues] = {0}, total = 0;
        std::vector<char> seen;
        for (size_t i = 0; i < bs.phases.size(); ++i) {
            const PhaseDesc &P = bs.phases[i];
            if (P.q_gy < 1 || P.q_inner < 1 || bs.nq % P.q_gy) return fail(VF_ERR_INVALID, "bad dealing rule");
            seen.assign((size_t)P.n_items, 0);
            for (int q = 0; q < bs.nq; ++q) {
                if (P.first_q[q] != head[q]) return fail(VF_ERR_INVALID, "queue ranges are not contiguous");
                for (int lq = 0; lq < P.n_q[q]; ++lq) {
                    const int per = bs.nq / P.q_gy, qb = q / P.q_gy, cg = q - qb * P.q_gy;
                    int unit, inner;
                    if (lq < P.q_full) {
                        const int grp = lq / P.q_inner;
                        inner = lq - grp * P.q_inner; unit = grp * per + qb;
                    } else {
                        const int ii = (lq - P.q_full) * per + qb, u = ii / P.q_inner;
                        inner = ii - u * P.q_inner; unit = (P.q_full / P.q_inner) * per + u;
                    }
                    const int local = (unit * P.q_inner + inner) * P.q_gy + cg;
                    if (local < 0 || local >= P.n_items || seen[(size_t)local])
                        return fail(VF_ERR_INVALID, "phase " + std::to_string(i) + ": dealing is not a bijection");
                    seen[(size_t)local] = 1;
                }
                head[q] += P.n_q[q];
                total += P.n_q[q];
            }
            if (bs.nq > 1 && P.q_inner > 1 && ph_is_conv(P.type)) {    // tile-by-tile tail: balanced to one item
                int lo = P.n_q[0], hi = P.n_q[0];
                for (int q = 0; q < bs.nq; ++q) { lo = std::min(lo, P.n_q[q]); hi = std::max(hi, P.n_q[q]); }
                if (hi - lo > 1) return fail(VF_ERR_INVALID, "phase " + std::to_string(i) + ": queues are not balanced");
            }
            for (int q = bs.nq; q < kQueues; ++q)
                if (P.n_q[q]) return fail(VF_ERR_INVALID, "items in an unused queue");
        }
        for (int q = 0; q < bs.nq; ++q)
            if (head[q] != bs.total_q[q]) return fail(VF_ERR_INVALID, "queue totals mismatch");
        if (total != bs.items) return fail(VF_ERR_INVALID, "dealt item count mismatch");
    }
    if (out_items) *out_items = bs.items;
    if (out_upload_checksum) *out_upload_checksum = h->upload_checksum;
    return VF_OK;
    VF_API_CATCH(int)
}
extern "C" int vf_selftest_inject(int32_t where, int32_t kind) {
    g_inject_where = where; g_inject_kind = kind;
    return VF_OK;
}
extern "C" int vf_set_fuse_top(vf_handle *h, int32_t enable) {      // (the device build defines it further down)
    VF_API_TRY
    if (!h) return fail(VF_ERR_INVALID, "null handle");
    h->fuse_top = enable != 0;
    h->fuse_pair = enable != 0;
    return VF_OK;
    VF_API_CATCH(int)
}
#else   // ------------------------------------------------------------------ device execution

// the zero initial LSTM state is one shared image per layer
static int zero_shared_state(vf_handle *h, const BatchView &sh, hipStream_t st) {
    const int H = h->Hc, W = h->Wc;
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

// arch 3: nothing to clear - at step 0 every conv-LSTM skips its recurrent chunks (ConvParams::chunk_begin: h(-1) = 0 would
// only multiply zeros) and its cell item takes c(-1) = 0 from a null pointer
static int s3_zero_state(vf_handle *, int, hipStream_t) { return VF_OK; }

static int run_steps(vf_handle *h, int view, const BatchView &v, const BatchView &sh, int B,
                     const int32_t *goal_pix, hipStream_t st, bool skip_shared) {
    if (h->cfg.arch == 3) {
        if (view == 0) {
            int rc = s3_zero_state(h, B, st);
            if (rc) return rc;
        }
        LaunchSink sink{h, st};
        return emit_rollout_s3(h, view, B, v.actions, goal_pix + (size_t)view * h->ND * 2, sink);
    }
    if (!skip_shared) {
        int rc = zero_shared_state(h, sh, st);
        if (rc) return rc;
    }
    LaunchSink sink{h, st};
    return emit_rollout(h, view, v, sh, B, goal_pix + (size_t)view * h->ND * 2, sink, skip_shared);
}

// every toggle emit_rollout / build_schedule read besides (B, dedup, xcd_queues): part of the schedule cache's key, so an
// option changed between two rollouts can never meet a schedule built for the old value
#ifndef VF_YIELD_DEFAULT
#define VF_YIELD_DEFAULT 120        // polls of ~0.4 us an item may spend yielding (A/B builds: -DVF_YIELD_DEFAULT=n)
#endif
#ifndef VF_YIELD_MAX_ITEMS
#define VF_YIELD_MAX_ITEMS 3        // ... in launches whose widest conv-LSTM phase has fewer items than this many x CUs
#endif
// Yield budget of a launch of B sequences (vf_conv_mfma.h, "yielding").  The scheme pays where a rollout is bound by the
// per-sample dependency chain - the shards of the multi-GPU configs: 25 x T13 11.57 -> 11.12 ms, 50: 18.98 -> 18.56 (same
// box, bit-identical) - is a wash from ~100 samples on (100: 32.87 -> 33.37, 125 x T15: 46.44 -> 46.08, 160: equal) and
// costs a launch that fills the chip 0.5 % (200: 62.75 -> 63.12 ms: there every K loop is somebody's throughput).  So it
// follows the width of the phases: the widest conv-LSTM phase of the 64 x 64 network has 8 items per sample at 128-row
// tiles -> on below 96 samples per view.
static int yield_for(const vf_handle *h, int B) {
    if (!h->early_start || h->cfg.arch == 3) return 0;
    if (h->yield_budget >= 0) return h->yield_budget;
    const long long widest = (long long)B * h->ncam * std::max(1, (h->Hc / 2) * (h->Wc / 2) / 128);
    return widest < (long long)VF_YIELD_MAX_ITEMS * h->n_cu ? VF_YIELD_DEFAULT : 0;
}
static int sched_options(const vf_handle *h, int B) {
    return (h->fuse_top ? 1 : 0) | (h->fuse_pair ? 2 : 0) | (h->early_start ? 4 : 0) | (h->wt_publish ? 8 : 0) |
           (yield_for(h, B) << 4);
}

// Are the shared buffers of configuration `cfg` (launch mode and split) still valid?
static bool shared_cache_hit(vf_handle *h, int cfg) {
    return h->dedup && h->cache_shared && h->shared_valid && h->shared_cfg == cfg;
}

template <int ND>
static int launch_persistent_t(const Schedule &sc, int grid, size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((rollout_persistent_kernel<ND>), dim3(grid), dim3(kConvThreads), lds, st, sc.phases, sc);
    VF_HIP_CHECK(hipGetLastError());
    return VF_OK;
}

// the whole rollout (every view) as one persistent launch (vf_persistent.h)
static int run_persistent(vf_handle *h, const float *d_actions, int B, const int32_t *goal_pix, hipStream_t st) {
    int rc;
    // the schedule holds pointers into the handle's own buffers (the action sequences are copied
    // into actions_buf, the goal pixels travel as launch arguments), so it only depends on
    // (B, options) and is rebuilt when those change - typically for a ragged last chunk
    const size_t act_bytes = (size_t)B * h->T * h->cfg.adim * sizeof(float);
    VF_HIP_CHECK(hipMemcpyAsync(h->actions_buf, d_actions, act_bytes, hipMemcpyDeviceToDevice, st));
    const int cfg = 1000;
    const bool skip_shared = shared_cache_hit(h, cfg);
    vf_handle::SchedCache &sc_host = h->sched[skip_shared ? 1 : 0];
    if (sc_host.B != B || sc_host.dedup != h->dedup || sc_host.xcd_queues != h->xcd_queues ||
        sc_host.options != sched_options(h, B)) {
        BuiltSchedule bs;
        if ((rc = build_schedule(h, B, skip_shared, bs))) return rc;
        // Upload without synchronising the caller's stream: the copy is stream-ordered behind the
        // rollouts still reading the device schedule, and the pinned staging buffer is only reused
        // once its own copy has completed (ring of kSchedRing; waits only if that many rebuilds are
        // in flight at once).
        const int slot = h->stage_next;
        h->stage_next = (slot + 1) % kSchedRing;
        if (h->stage_used[slot]) VF_HIP_CHECK(hipEventSynchronize(h->stage_done[slot]));
        for (size_t i = 0; i < bs.phases.size(); ++i) {     // device addresses the fused / early-started items need
            PhaseDesc &P = bs.phases[i];
            if (P.has_late) {
                P.conv.late_cnt = h->d_sync + kSyncHead + P.late.cnt_base;
                P.conv.late_expect = P.late.expect;
                P.conv.late_mode = P.late.mode;
                P.conv.late_status = h->d_status;
                if (yield_for(h, B) > 0) {      // every early-started item publishes its state around its mid-item wait;
                    P.conv.cu_state = h->d_sync + kTicketInts;      // only the conv-LSTMs' recurrent halves yield
                    P.conv.yield_budget = P.type == PH_LSTM ? yield_for(h, B) : 0;
                }
            }
            // (write-through publish: decided in build_schedule from the tile's epilogue, wt_epilogue())
            if (P.type == PH_CONV_PAIR) P.conv.fuse_next = &sc_host.d_phases[i].conv2;
            if (P.type != PH_TOP_FUSED) continue;
            P.conv.fuse_comp = &sc_host.d_phases[i].comp;
            P.conv.fuse_ready = h->d_sync + kSyncHead + P.aux_base;
            P.conv.fuse_status = h->d_status;
            P.conv.fuse_view = P.view;
            P.conv.fuse_nd = h->ND;
        }
        memcpy(h->stage[slot], bs.phases.data(), bs.phases.size() * sizeof(PhaseDesc));
        VF_HIP_CHECK(hipMemcpyAsync(sc_host.d_phases, h->stage[slot], bs.phases.size() * sizeof(PhaseDesc),
                                    hipMemcpyHostToDevice, st));
        VF_HIP_CHECK(hipEventRecord(h->stage_done[slot], st));
        h->stage_used[slot] = true;
        sc_host.B = B; sc_host.dedup = h->dedup;
        sc_host.xcd_queues = h->xcd_queues;
        sc_host.options = sched_options(h, B);
        sc_host.items = bs.items; sc_host.counters = bs.counters;
        sc_host.nq = bs.nq;
        for (int q = 0; q < kQueues; ++q) sc_host.total_q[q] = bs.total_q[q];
        sc_host.phases = (int)bs.phases.size();
        sc_host.types.clear(); sc_host.nitems.clear();
        for (const PhaseDesc &P : bs.phases) {      // (element-wise phases of arch 3 report 100 + their operation)
            sc_host.types.push_back(P.type == PH_EW ? 100 + P.ew.op : P.type);
            sc_host.nitems.push_back(P.n_items);
        }
        sc_host.flops = bs.flops;
        sc_host.lds = bs.lds;
    }
    h->last_sched = skip_shared ? 1 : 0;
    if (h->cfg.arch == 3) {
        if ((rc = s3_zero_state(h, B, st))) return rc;
    } else if (!skip_shared)
        for (int v = 0; v < h->ncam; ++v)
            if ((rc = zero_shared_state(h, h->shared_views[(size_t)v], st))) return rc;
    VF_HIP_CHECK(hipMemsetAsync(h->d_sync, 0, (kSyncHead + (size_t)sc_host.counters) * sizeof(int), st));
    Schedule sc;
    memset(&sc, 0, sizeof(sc));
    sc.phases = sc_host.d_phases; sc.n_phases = sc_host.phases; sc.total_items = sc_host.items;
    sc.ticket = h->d_sync; sc.counters = h->d_sync + kSyncHead; sc.status = h->d_status;
    sc.nq = sc_host.nq;
    for (int q = 0; q < kQueues; ++q) sc.total_q[q] = sc_host.total_q[q];
    sc.cu_tab = yield_for(h, B) > 0 ? h->d_sync + kTicketInts : nullptr;
    sc.stats = nullptr;
    sc.nd = h->ND;
    for (int i = 0; i < h->ncam * h->ND * 2; ++i) sc.goal[i] = goal_pix[i];
    if (h->phase_stats) {
        VF_HIP_CHECK(hipMemsetAsync(h->d_stats, 0, h->sched_capacity * 2 * sizeof(unsigned long long), st));
        sc.stats = h->d_stats;
    }

    // resident workgroups per CU: bounded by the LDS a workgroup needs (160 KiB per CU)
    const size_t lds = sc_host.lds + kCtlWords * sizeof(int);
    const int by_lds = (int)std::max<size_t>(1, (160 * 1024) / lds);
    const int wgs_per_cu = std::min(h->persist_wgs_per_cu, by_lds);
    const int grid = std::min(sc_host.items, h->n_cu * wgs_per_cu);
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
        case 1: rc = launch_persistent_t<1>(sc, grid, lds, st); break;
        case 2: rc = launch_persistent_t<2>(sc, grid, lds, st); break;
        case 3: rc = launch_persistent_t<3>(sc, grid, lds, st); break;
        default: rc = launch_persistent_t<4>(sc, grid, lds, st); break;
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
               const float *task_weights, double *d_scores, double *d_scores_per_task, void *stream) {
    VF_API_TRY
    if (!h || !d_actions || !goal_pix || !d_scores) return fail(VF_ERR_INVALID, "null argument");
    if (!h->have_weights) return fail(VF_ERR_NOWEIGHTS, "vf_load_weights has not been called");
    if (!h->have_context) return fail(VF_ERR_NOCONTEXT, "vf_set_context has not been called");
    if (B < 1 || B > h->cfg.max_batch)
        return fail(VF_ERR_INVALID, "batch " + std::to_string(B) + " outside 1.." + std::to_string(h->cfg.max_batch));
    if (B % h->n_draws)
        return fail(VF_ERR_INVALID, "batch " + std::to_string(B) + " is not a multiple of n_draws = " +
                                        std::to_string(h->n_draws));
    VF_HIP_CHECK(hipSetDevice(h->cfg.device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rc;

    if (h->persistent) {
        if ((rc = run_persistent(h, d_actions, B, goal_pix, st))) return rc;
    } else {
        const bool skip = shared_cache_hit(h, 1);
        for (int v = 0; v < h->ncam; ++v) {
            BatchView bv;
            if (h->cfg.arch == 3) { memset(&bv, 0, sizeof(bv)); bv.actions = d_actions; }      // (arch 3 keeps its buffers in Savp3)
            else bv = make_view(h, v, d_actions, 0);
            if ((rc = run_steps(h, v, bv, h->shared_views[(size_t)v], B, goal_pix, st, skip))) return rc;
        }
        h->shared_valid = h->dedup; h->shared_cfg = 1;
    }
    TaskWeights tw;
    memset(&tw, 0, sizeof(tw));
    if (task_weights) {
        tw.use = 1;
        for (int i = 0; i < h->ncam * h->ND; ++i) tw.w[i] = task_weights[i];
    }
    const int n_actions = B / h->n_draws;
    hipLaunchKernelGGL(scores_kernel, dim3(n_actions), dim3(64), 0, st, h->sums, h->sums_step_stride,
                       h->sums_view_stride, n_actions, h->n_draws, h->T, h->ND, h->ncam, h->nblocks, finalweight, tw,
                       h->d_status, d_scores, d_scores_per_task);
    VF_HIP_CHECK(hipGetLastError());
    h->last_B = B;
    return VF_OK;
    VF_API_CATCH(int)
}

int vf_set_persistent(vf_handle *h, int32_t enable) {
    VF_API_TRY
    if (!h) return fail(VF_ERR_INVALID, "null handle");
    h->persistent = enable != 0;
#ifdef VF_DEBUG_KNOBS
    if (const char *e = getenv("VF_PERSIST_WGS_PER_CU")) h->persist_wgs_per_cu = std::max(1, std::min(2, atoi(e)));
    if (const char *e = getenv("VF_EARLY_START")) h->early_start = atoi(e) != 0;
#endif
    return VF_OK;
    VF_API_CATCH(int)
}

int vf_set_fuse_top(vf_handle *h, int32_t enable) {
    VF_API_TRY
    if (!h) return fail(VF_ERR_INVALID, "null handle");
    h->fuse_top = enable != 0;
    h->fuse_pair = enable != 0 && h->pair_allowed;
    return VF_OK;
    VF_API_CATCH(int)
}

int vf_set_sched_option(vf_handle *h, int32_t option, int32_t value) {
    VF_API_TRY
    if (!h) return fail(VF_ERR_INVALID, "null handle");
    switch (option) {
        case VF_OPT_YIELD_BUDGET:
            if (value < -1 || value > 100000) return fail(VF_ERR_INVALID, "yield budget must be -1 (automatic) or 0..100000");
            h->yield_budget = value;
            return VF_OK;
        case VF_OPT_WRITE_THROUGH:
            h->wt_publish = value != 0;
            return VF_OK;
        default:
            return fail(VF_ERR_INVALID, "unknown scheduling option " + std::to_string(option));
    }
    VF_API_CATCH(int)
}

int vf_set_xcd_queues(vf_handle *h, int32_t enable) {
    VF_API_TRY
    if (!h) return fail(VF_ERR_INVALID, "null handle");
    h->xcd_queues = enable ? kQueues : 1;
    return VF_OK;
    VF_API_CATCH(int)
}

int vf_device_status(vf_handle *h, int32_t *status) {
    VF_API_TRY
    if (!h || !status) return fail(VF_ERR_INVALID, "null argument");
    VF_HIP_CHECK(hipSetDevice(h->cfg.device));
    VF_HIP_CHECK(hipDeviceSynchronize());
    VF_HIP_CHECK(hipMemcpy(status, h->d_status, sizeof(int), hipMemcpyDeviceToHost));
    if (*status != 0) {     // observed: re-arm; the abandoned launch may have left the context-only buffers half-written
        VF_HIP_CHECK(hipMemset(h->d_status, 0, sizeof(int)));
        h->shared_valid = false;
    }
    return VF_OK;
    VF_API_CATCH(int)
}

// debugging aid: make the next rollouts fail as if a producer never arrived (tests of the in-band
// failure path); the word is cleared again by vf_device_status
int vf_debug_poison_status(vf_handle *h) {
    VF_API_TRY
    if (!h) return fail(VF_ERR_INVALID, "null handle");
    VF_HIP_CHECK(hipSetDevice(h->cfg.device));
    VF_HIP_CHECK(hipDeviceSynchronize());
    const int one = 1;
    VF_HIP_CHECK(hipMemcpy(h->d_status, &one, sizeof(int), hipMemcpyHostToDevice));
    return VF_OK;
    VF_API_CATCH(int)
}

// debugging aid: per-phase (type, items, wait ticks, run ticks) of the last persistent rollout
int vf_set_phase_stats(vf_handle *h, int32_t enable) {
    VF_API_TRY
    if (!h) return fail(VF_ERR_INVALID, "null handle");
    h->phase_stats = enable != 0;
    return VF_OK;
    VF_API_CATCH(int)
}

int vf_debug_phase_stats(vf_handle *h, int32_t max_phases, int32_t *types, int32_t *items, uint64_t *wait_run) {
    VF_API_TRY
    if (!h || !h->phase_stats) return fail(VF_ERR_INVALID, "no phase statistics (vf_set_phase_stats)");
    VF_HIP_CHECK(hipSetDevice(h->cfg.device));
    VF_HIP_CHECK(hipDeviceSynchronize());
    const vf_handle::SchedCache &sc = h->sched[h->last_sched];
    const int n = std::min<int>(max_phases, sc.phases);
    for (int i = 0; i < n; ++i) { types[i] = sc.types[i]; items[i] = sc.nitems[i]; }
    VF_HIP_CHECK(hipMemcpy(wait_run, h->d_stats, (size_t)n * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return n;
    VF_API_CATCH(int)
}

#ifdef VF_TILE_STATS
int vf_debug_tile_clocks(uint64_t *out /*[32][8]*/, int32_t reset) {
    VF_API_TRY
    if (hipDeviceSynchronize() != hipSuccess) return fail(VF_ERR_HIP, "sync failed");
    VF_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(vf::g_tile_clk), sizeof(uint64_t) * 256));
    if (reset) {
        static const uint64_t zeros[256] = {0};
        VF_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(vf::g_tile_clk), zeros, sizeof(zeros)));
    }
    return VF_OK;
    VF_API_CATCH(int)
}
#endif

#ifdef VF_TRACE
// diagnostic build: the event log of the last persistent launch, [512 workgroups][8192] words + the event counts
extern "C" int vf_debug_trace(uint64_t *events, uint32_t *counts) {
    VF_API_TRY
    if (hipDeviceSynchronize() != hipSuccess) return fail(VF_ERR_HIP, "sync failed");
    VF_HIP_CHECK(hipMemcpyFromSymbol(events, HIP_SYMBOL(vf::g_trace), sizeof(uint64_t) * vf::kTraceWgs * vf::kTraceMax));
    VF_HIP_CHECK(hipMemcpyFromSymbol(counts, HIP_SYMBOL(vf::g_trace_n), sizeof(uint32_t) * vf::kTraceWgs));
    return VF_OK;
    VF_API_CATCH(int)
}
#endif

int vf_set_dedup(vf_handle *h, int32_t enable) {
    VF_API_TRY
    if (!h) return fail(VF_ERR_INVALID, "null handle");
    h->dedup = enable != 0 && h->cfg.arch != 3;     // (arch 3 has no context de-duplication)
    h->shared_valid = false;
    return VF_OK;
    VF_API_CATCH(int)
}

int vf_export(vf_handle *h, int32_t first, int32_t count, float *d_frames, float *d_distrib, float *d_states,
              void *stream) {
    VF_API_TRY
    if (!h) return fail(VF_ERR_INVALID, "null handle");
    if (first < 0 || count < 1 || first + count > h->last_B)
        return fail(VF_ERR_INVALID, "sample range outside the last rollout");
    VF_HIP_CHECK(hipSetDevice(h->cfg.device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t HW = (size_t)h->H * h->W;
    const long long view_rows = (long long)h->cfg.max_batch * h->T;
    if (d_frames) {
        if (h->ncam == 1) {
            VF_HIP_CHECK(hipMemcpyAsync(d_frames, h->frames_all + (size_t)first * h->T * HW * 3,
                                        (size_t)count * h->T * HW * 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
        } else {
            const long long n = (long long)count * h->T * h->ncam * HW * 3;
            hipLaunchKernelGGL(export_frames_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                               h->frames_all, view_rows * (long long)HW * 3, first, count, h->T, h->ncam,
                               (int)(HW * 3), d_frames);
            VF_HIP_CHECK(hipGetLastError());
        }
    }
    if (d_states)       // the views share the state trajectory's inputs; view 0's prediction is reported
        VF_HIP_CHECK(hipMemcpyAsync(d_states, h->states_all + (size_t)first * h->T * h->cfg.sdim,
                                    (size_t)count * h->T * h->cfg.sdim * sizeof(float), hipMemcpyDeviceToDevice,
                                    st));
    if (d_distrib) {
        const long long n = (long long)count * h->T * h->ncam * HW * h->ND;
        hipLaunchKernelGGL(export_distrib_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                           h->distrib_all, view_rows * (long long)HW * h->ND, h->sums, h->sums_step_stride,
                           h->sums_view_stride, first, count, h->T, h->ncam, (int)HW, h->ND, h->nblocks, d_distrib);
        VF_HIP_CHECK(hipGetLastError());
    }
    return VF_OK;
    VF_API_CATCH(int)
}

int vf_register(vf_handle *h, const float *d_current, const float *d_reference, const float *d_flow,
                const int32_t *d_pix, int32_t ntask, int32_t region, int32_t clip_sub, float *d_warped,
                float *d_warp_pts, float *d_desig, float *d_err, void *stream) {
    VF_API_TRY
    if (!h || !d_current || !d_reference || !d_flow || !d_pix || !d_desig || !d_err)
        return fail(VF_ERR_INVALID, "null argument");
    if (ntask < 1 || region < 0 || (2 * region + 1) * (2 * region + 1) > kRegMaxWin || (clip_sub != 0 && clip_sub != 1))
        return fail(VF_ERR_INVALID, "need ntask >= 1, 0 <= region <= 5 and clip_sub in {0, 1}");
    VF_HIP_CHECK(hipSetDevice(h->cfg.device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d_warped || d_warp_pts) {
        const int n = h->ncam * h->H * h->W;
        hipLaunchKernelGGL(warp_image_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_current, d_flow, h->ncam,
                           h->H, h->W, d_warped, d_warp_pts);
        VF_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(register_kernel, dim3(h->ncam * ntask), dim3(128), 0, st, d_current, d_reference, d_flow,
                       d_pix, h->ncam, ntask, h->H, h->W, region, clip_sub, d_desig, d_err);
    VF_HIP_CHECK(hipGetLastError());
    return VF_OK;
    VF_API_CATCH(int)
}

// RCCL is bound at first use (dlopen of the library the process already carries - PyTorch ships
// its own librccl.so - or the system one), once per process, so libvf_hip.so itself has no link
// dependency on it and communicators created through vf_comm_init_all are served by the same
// library instance as the collectives.
struct RcclApi {
    void *lib = nullptr;
    int (*all_gather)(const void *, void *, size_t, int, void *, void *) = nullptr;
    int (*group_start)() = nullptr;
    int (*group_end)() = nullptr;
    int (*comm_init_all)(void **, int, const int *) = nullptr;
    int (*comm_destroy)(void *) = nullptr;
};
static RcclApi g_rccl;

static int bind_rccl() {
    if (g_rccl.all_gather) return VF_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *lib = nullptr;
    for (const char *n : names)
        if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) return fail(VF_ERR_HIP, std::string("cannot load RCCL: ") + dlerror());
    RcclApi api;
    api.lib = lib;
    api.all_gather = reinterpret_cast<int (*)(const void *, void *, size_t, int, void *, void *)>(dlsym(lib, "ncclAllGather"));
    api.group_start = reinterpret_cast<int (*)()>(dlsym(lib, "ncclGroupStart"));
    api.group_end = reinterpret_cast<int (*)()>(dlsym(lib, "ncclGroupEnd"));
    api.comm_init_all = reinterpret_cast<int (*)(void **, int, const int *)>(dlsym(lib, "ncclCommInitAll"));
    api.comm_destroy = reinterpret_cast<int (*)(void *)>(dlsym(lib, "ncclCommDestroy"));
    if (!api.all_gather || !api.group_start || !api.group_end || !api.comm_init_all || !api.comm_destroy)
        return fail(VF_ERR_HIP, "the RCCL library lacks ncclAllGather / ncclGroup* / ncclCommInitAll / ncclCommDestroy");
    g_rccl = api;       // (the library stays loaded for the life of the process)
    return VF_OK;
}

static const int kNcclFloat64 = 8;      // ncclFloat64 in nccl.h

int vf_allgather_scores(vf_handle *h, void *nccl_comm, const double *d_local, int32_t n_local, double *d_all,
                        void *stream) {
    VF_API_TRY
    if (!h || !nccl_comm || !d_local || !d_all || n_local < 1) return fail(VF_ERR_INVALID, "null or empty argument");
    int rc = bind_rccl();
    if (rc) return rc;
    VF_HIP_CHECK(hipSetDevice(h->cfg.device));
    rc = g_rccl.all_gather(d_local, d_all, (size_t)n_local, kNcclFloat64, nccl_comm, stream);
    if (rc != 0) return fail(VF_ERR_HIP, "ncclAllGather failed with ncclResult_t " + std::to_string(rc));
    return VF_OK;
    VF_API_CATCH(int)
}

int vf_comm_init_all(int32_t n, const int32_t *devices, void **comms) {
    VF_API_TRY
    if (n < 1 || !devices || !comms) return fail(VF_ERR_INVALID, "null or empty argument");
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j)
            if (devices[i] == devices[j])
                return fail(VF_ERR_INVALID, "vf_comm_init_all: device " + std::to_string(devices[i]) +
                                                " listed twice (one communicator rank per GPU)");
    int rc = bind_rccl();
    if (rc) return rc;
    std::vector<int> devs(devices, devices + n);
    rc = g_rccl.comm_init_all(comms, n, devs.data());
    if (rc != 0) return fail(VF_ERR_HIP, "ncclCommInitAll failed with ncclResult_t " + std::to_string(rc));
    return VF_OK;
    VF_API_CATCH(int)
}

int vf_comm_destroy(void *comm) {
    VF_API_TRY
    if (!comm) return VF_OK;
    int rc = bind_rccl();
    if (rc) return rc;
    rc = g_rccl.comm_destroy(comm);
    if (rc != 0) return fail(VF_ERR_HIP, "ncclCommDestroy failed with ncclResult_t " + std::to_string(rc));
    return VF_OK;
    VF_API_CATCH(int)
}

int vf_allgather_scores_group(int32_t n, vf_handle *const *hs, void *const *comms, const double *const *d_local,
                              int32_t n_local, double *const *d_all, void *const *streams) {
    VF_API_TRY
    if (n < 1 || !hs || !comms || !d_local || !d_all || n_local < 1) return fail(VF_ERR_INVALID, "null or empty argument");
    for (int i = 0; i < n; ++i)
        if (!hs[i] || !comms[i] || !d_local[i] || !d_all[i]) return fail(VF_ERR_INVALID, "null entry " + std::to_string(i));
    int rc = bind_rccl();
    if (rc) return rc;
    int caller_dev = -1;        // the loop below walks the lanes' devices: hand the calling thread's device back
    if (hipGetDevice(&caller_dev) != hipSuccess) caller_dev = -1;
    if ((rc = g_rccl.group_start()) != 0) return fail(VF_ERR_HIP, "ncclGroupStart failed with ncclResult_t " + std::to_string(rc));
    int first_bad = 0;
    for (int i = 0; i < n && !first_bad; ++i) {
        if (hipSetDevice(hs[i]->cfg.device) != hipSuccess) { first_bad = -1; break; }
        first_bad = g_rccl.all_gather(d_local[i], d_all[i], (size_t)n_local, kNcclFloat64, comms[i],
                                      streams ? streams[i] : nullptr);
    }
    rc = g_rccl.group_end();        // always closes the group
    if (caller_dev >= 0) (void)hipSetDevice(caller_dev);
    if (first_bad) return fail(VF_ERR_HIP, "grouped ncclAllGather failed (" + std::to_string(first_bad) + ")");
    if (rc != 0) return fail(VF_ERR_HIP, "ncclGroupEnd failed with ncclResult_t " + std::to_string(rc));
    return VF_OK;
    VF_API_CATCH(int)
}

int vf_set_profiling(vf_handle *h, int32_t enable) {
    VF_API_TRY
    if (!h) return fail(VF_ERR_INVALID, "null handle");
    h->profiling = enable != 0;
    h->ev_used = 0;
    h->prof_flops = 0.0;
    return VF_OK;
    VF_API_CATCH(int)
}

int vf_get_profile(vf_handle *h, double *kernel_ms, int64_t *launches, double *flops, double *busy_ms) {
    VF_API_TRY
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
    VF_API_CATCH(int)
}

}  // extern "C"
#endif  // VF_HOST_SELFTEST
