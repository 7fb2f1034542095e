// host_selftest.cc - driver of the ASan/UBSan CPU build of the engine's host-side code
// ("build the C-ABI lib with an ASan/UBSan CPU variant").
//
// vf_engine.hip is compiled for the HOST ONLY with -DVF_HOST_SELFTEST: device allocations become
// address reservations, uploads become checksums.  What runs under the sanitizers is exactly the
// product's host code: the tensor table, the layer planner, the weight packers (fp32 and
// split-bf16), the rollout emitter and the persistent-schedule builder - for several shapes,
// view counts, batch sizes and both schedule variants - and vf_selftest_schedule() checks the
// invariants the device relies on (contiguous tickets, dependencies on earlier phases only,
// counters in range, every pointer of every phase inside a buffer of the handle).
// Never built for or run on a GPU (GPU sanitizers are not available on this pool).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/vf_hip.h"

extern "C" int vf_set_fuse_top(vf_handle *h, int32_t enable);
extern "C" int vf_selftest_inject(int32_t where, int32_t kind);
extern "C" int vf_selftest_schedule(vf_handle *h, int32_t B, int32_t skip_shared, int64_t *out_items,
                                    uint64_t *out_upload_checksum);

static int run_case(int H, int W, int adim, int sdim, int nd, int nctx, int T, int max_batch, int precision, int ncam,
                    int n_draws, const int *batches, int n_batches, int arch = 0, int zdim = 0, int layer_spec = 0) {
    vf_config cfg = {H, W, adim, sdim, nd, nctx, nctx + T, arch == 3 ? 4 : (arch == 2 ? 6 : 10), max_batch, 0, precision, ncam, n_draws, arch,
                     zdim, layer_spec};
    const size_t n = vf_weight_count(&cfg);
    if (n == 0) { std::fprintf(stderr, "weight count failed: %s\n", vf_last_error()); return 1; }
    std::vector<float> blob(n * (size_t)ncam);
    uint32_t s = 12345u + (uint32_t)(H * 7 + W * 3 + adim + ncam);
    for (float &x : blob) { s = s * 1664525u + 1013904223u; x = ((float)(s >> 8) / 16777216.0f - 0.5f) * 0.2f; }
    vf_handle *h = nullptr;
    if (vf_create(&cfg, &h)) { std::fprintf(stderr, "vf_create failed: %s\n", vf_last_error()); return 1; }
    if (vf_load_weights(h, blob.data(), blob.size()) != 0 ||
        vf_load_weights(h, blob.data(), blob.size()) != 0) {            // a reload reuses the same buffers
        std::fprintf(stderr, "vf_load_weights failed: %s\n", vf_last_error());
        return 1;
    }
    if (vf_load_weights(h, blob.data(), blob.size() - 1) == 0) { std::fprintf(stderr, "short blob accepted\n"); return 1; }
    uint64_t sum = 0;
    for (int fuse = 1; fuse >= 0; --fuse) {
      vf_set_fuse_top(h, fuse);        // fused decoder top, then the two-phase top
      for (int i = 0; i < n_batches; ++i)
        for (int skip = 0; skip < 2; ++skip) {
            int64_t items = 0;
            if (vf_selftest_schedule(h, batches[i], skip, &items, &sum)) {
                std::fprintf(stderr, "schedule B=%d skip=%d: %s\n", batches[i], skip, vf_last_error());
                return 1;
            }
            std::printf("  %dx%d adim %d nd %d ncam %d prec %d  B=%-4d %s%s: %lld items\n", H, W, adim, nd, ncam,
                        precision, batches[i], skip ? "cached-context" : "full", fuse ? "" : " (unfused top)",
                        (long long)items);
        }
    }
    std::printf("  packed-weight checksum %016llx\n", (unsigned long long)sum);
    vf_destroy(h);
    return 0;
}

int main() {
    const int b_small[] = {1, 7, 16, 37};
    const int b_c2[] = {200, 125, 25};
    const int b_c3[] = {600, 88};
    const int b_c5[] = {50, 5};
    int rc = 0;
    rc |= run_case(32, 32, 4, 5, 1, 2, 3, 37, 0, 1, 1, b_small, 4);
    rc |= run_case(48, 64, 3, 3, 2, 2, 2, 37, 1, 1, 1, b_small, 4);
    rc |= run_case(64, 64, 4, 5, 1, 1, 2, 16, 0, 1, 1, b_small, 3);
    rc |= run_case(64, 64, 4, 5, 1, 2, 13, 200, 0, 1, 1, b_c2, 3);
    rc |= run_case(64, 64, 4, 5, 2, 2, 13, 600, 0, 2, 1, b_c3, 2);
    rc |= run_case(128, 128, 12, 5, 1, 2, 15, 50, 1, 1, 5, b_c5, 2);
    rc |= run_case(40, 56, 5, 5, 4, 2, 2, 16, 0, 3, 1, b_small, 3);
    // arch 1: the SAVP-class four-scale generator (config 5 shard and a small odd shape)
    rc |= run_case(128, 128, 12, 5, 1, 2, 15, 125, 0, 1, 5, b_c2 + 1, 2, 1);
    rc |= run_case(48, 80, 6, 3, 2, 2, 2, 37, 1, 2, 1, b_small, 4, 1);
    // arch 2: arch 1 + the conditioning vector in every conv-LSTM (PH_COND items) + the seven-layer compositing
    rc |= run_case(128, 128, 12, 5, 1, 2, 15, 125, 0, 1, 5, b_c2 + 1, 2, 2);
    rc |= run_case(64, 80, 6, 3, 2, 2, 2, 37, 0, 2, 1, b_small, 4, 2);
    rc |= run_case(64, 64, 12, 5, 3, 1, 2, 16, 0, 1, 1, b_small, 3, 2);
    // arch 0 on the decoder widths of the public CDNA code (layer_spec 1: convt2 96 -> 96, convt3 64 -> 64, unfused top)
    rc |= run_case(64, 64, 4, 5, 2, 2, 3, 37, 0, 1, 1, b_small, 4, 0, 0, 1);
    rc |= run_case(48, 64, 3, 3, 1, 2, 13, 200, 0, 2, 1, b_c2, 3, 0, 0, 1);
    // arch 3: the published SAVP generator - every layer table (32 / 64 / 128 pixels, the paper's table forced on 128 x 128),
    // a config-5 shard, two views, an odd shape
    rc |= run_case(32, 32, 12, 5, 1, 2, 3, 37, 0, 1, 1, b_small, 4, 3, 8);
    rc |= run_case(64, 64, 12, 5, 2, 2, 3, 37, 0, 2, 1, b_small, 4, 3, 8);
    rc |= run_case(48, 64, 6, 3, 4, 1, 2, 16, 0, 1, 1, b_small, 3, 3, 2);
    rc |= run_case(128, 128, 12, 5, 1, 2, 15, 125, 0, 1, 5, b_c2 + 1, 2, 3, 8);
    rc |= run_case(128, 128, 12, 5, 1, 2, 4, 10, 0, 1, 5, b_c5 + 1, 1, 3, 8, 64);
    // invalid configurations are refused, not crashed on
    vf_config bad = {60, 64, 4, 5, 1, 2, 15, 10, 8, 0, 0, 1, 1, 0};
    vf_handle *h = nullptr;
    if (vf_create(&bad, &h) == 0) { std::fprintf(stderr, "invalid config accepted\n"); rc = 1; }
    vf_config bad2 = {64, 64, 4, 5, 1, 2, 15, 10, 8, 0, 0, 5, 1, 0};
    if (vf_create(&bad2, &h) == 0) { std::fprintf(stderr, "ncam 5 accepted\n"); rc = 1; }
    vf_config bad3 = {72, 64, 4, 5, 1, 2, 15, 10, 8, 0, 0, 1, 1, 1};       // arch 1 needs multiples of 16
    if (vf_create(&bad3, &h) == 0) { std::fprintf(stderr, "arch 1 at 72x64 accepted\n"); rc = 1; }
    vf_config bad4 = {64, 64, 12, 5, 1, 2, 15, 10, 8, 0, 0, 1, 1, 2};      // arch 2 composes four warps: num_masks 6
    if (vf_create(&bad4, &h) == 0) { std::fprintf(stderr, "arch 2 with num_masks 10 accepted\n"); rc = 1; }
    vf_config bad5 = {64, 64, 12, 5, 1, 2, 15, 6, 8, 0, 1, 1, 1, 2};       // arch 2 is fp32 only
    if (vf_create(&bad5, &h) == 0) { std::fprintf(stderr, "arch 2 in the split-bf16 mode accepted\n"); rc = 1; }
    vf_config bad6 = {64, 64, 12, 5, 1, 2, 15, 4, 8, 0, 0, 1, 1, 3, 0, 0};     // arch 3 needs zdim
    if (vf_create(&bad6, &h) == 0) { std::fprintf(stderr, "arch 3 without latent channels accepted\n"); rc = 1; }
    vf_config bad7 = {64, 64, 12, 5, 1, 2, 15, 10, 8, 0, 0, 1, 1, 0, 8, 0};    // zdim belongs to arch 3
    if (vf_create(&bad7, &h) == 0) { std::fprintf(stderr, "zdim with arch 0 accepted\n"); rc = 1; }
    vf_config bad8 = {72, 64, 12, 5, 1, 2, 15, 4, 8, 0, 0, 1, 1, 3, 8, 128};   // four scales need multiples of 16
    if (vf_create(&bad8, &h) == 0) { std::fprintf(stderr, "arch 3 / four scales at 72x64 accepted\n"); rc = 1; }
    // "No exception crosses this boundary" (include/vf_hip.h): a std::bad_alloc / std::exception / foreign throw inside the
    // schedule builder, the weight packer or vf_create comes back as a status code with vf_last_error() set, leaks nothing
    // (ASan's leak check runs at exit) and leaves the handle usable and destroyable
    {
        vf_config cfg = {64, 64, 4, 5, 1, 2, 5, 10, 16, 0, 0, 1, 1, 0, 0, 0};
        const size_t n = vf_weight_count(&cfg);
        std::vector<float> blob(n, 0.01f);
        vf_handle *hh = nullptr;
        const int want_code[3] = {VF_ERR_NOMEM, VF_ERR_INVALID, VF_ERR_INVALID};
        const char *want_msg[3] = {"out of memory", "exception: injected failure", "unknown exception"};
        for (int kind = 0; kind < 3; ++kind) {
            vf_selftest_inject(3, kind);
            hh = reinterpret_cast<vf_handle *>(1);
            const int r = vf_create(&cfg, &hh);
            if (r != want_code[kind] || hh != nullptr || std::string(vf_last_error()) != want_msg[kind]) {
                std::fprintf(stderr, "vf_create under injected failure %d: rc %d, '%s'\n", kind, r, vf_last_error()); rc = 1;
            }
        }
        if (vf_create(&cfg, &hh)) { std::fprintf(stderr, "vf_create after the injected failures: %s\n", vf_last_error()); rc = 1; }
        for (int kind = 0; kind < 3 && hh; ++kind) {
            vf_selftest_inject(2, kind);
            int r = vf_load_weights(hh, blob.data(), blob.size());
            if (r != want_code[kind] || std::string(vf_last_error()) != want_msg[kind]) {
                std::fprintf(stderr, "vf_load_weights under injected failure %d: rc %d, '%s'\n", kind, r, vf_last_error()); rc = 1;
            }
            if (vf_load_weights(hh, blob.data(), blob.size())) { std::fprintf(stderr, "reload after failure: %s\n", vf_last_error()); rc = 1; }
            vf_selftest_inject(1, kind);
            int64_t items = 0; uint64_t sum = 0;
            r = vf_selftest_schedule(hh, 7, 0, &items, &sum);
            if (r != want_code[kind] || std::string(vf_last_error()) != want_msg[kind]) {
                std::fprintf(stderr, "build_schedule under injected failure %d: rc %d, '%s'\n", kind, r, vf_last_error()); rc = 1;
            }
            if (vf_selftest_schedule(hh, 7, 0, &items, &sum) || items <= 0) {
                std::fprintf(stderr, "schedule after failure: %s\n", vf_last_error()); rc = 1;
            }
        }
        if (hh && vf_destroy(hh)) { std::fprintf(stderr, "vf_destroy after the injected failures failed\n"); rc = 1; }
        std::printf("  injected failures (bad_alloc, std::exception, foreign) in vf_create / vf_load_weights / build_schedule: %s\n",
                    rc ? "FAILED" : "status codes returned, handle reusable");
    }
    std::printf(rc ? "HOST SELFTEST FAILED\n" : "HOST SELFTEST OK\n");
    return rc;
}
