/* vf_c_host.c - a plain C host of libvf_hip.so: the drop-in boundary driven through include/vf_hip.h alone.
 *
 * No Python, no PyTorch: device memory comes from the HIP runtime's C API, everything else from the ten-odd entry
 * points a non-Python maintainer of the reference's planner would bind:
 *
 *     vf_create -> vf_load_weights -> vf_set_persistent -> vf_set_context -> vf_rollout -> vf_device_status -> vf_export
 *
 * i.e. what `self.predictor = predictor_class(...)`, `.restore()` and `self.predictor(context, {'actions'})` +
 * `_eval_pixel_cost` do in visual_mpc/policy/cem_controllers/pixel_cost_controller.py:29-36,83-84,135-166 of the
 * reference.  tests/test_gpu_c_host.py builds this file with gcc, feeds it the inputs of a Python-side planning call and
 * requires bit-identical scores and predictions.
 *
 *     gcc -O2 -std=c11 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tools/c_host/vf_c_host.c \
 *         visual_foresight_amd/libvf_hip.so -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -o build/vf_c_host
 *     build/vf_c_host <input.bin> <output.bin>
 *
 * input.bin : int32 header {H, W, adim, sdim, ndesig, n_context, sequence_length, B, n_export, n_weights}, then
 *             float32 weights[n_weights], uint8 frames[nc][1][H][W][3], float32 states[nc][sdim],
 *             float32 ctx_actions[max(nc-1,1)][adim], float32 distrib[nc][1][H][W][nd], float32 actions[B][T][adim],
 *             int32 goal[nd][2], float32 finalweight
 * output.bin: float64 scores[B], float64 per_task[B][nd], float32 frames[n_export][T][1][H][W][3],
 *             float32 distrib[n_export][T][1][H][W][nd], float32 states[n_export][T][sdim]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <hip/hip_runtime_api.h>

#include "vf_hip.h"

#define HIP_OK(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_)); return 2; } \
    } while (0)
#define VF_CALL(expr)                                                                            \
    do {                                                                                         \
        int rc_ = (expr);                                                                        \
        if (rc_ != VF_OK) { fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, vf_last_error()); return 3; } \
    } while (0)

static void *slurp(FILE *f, size_t bytes) {
    void *p = malloc(bytes ? bytes : 1);
    if (!p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read (%zu bytes)\n", bytes); exit(4); }
    return p;
}

static int to_device(void **d, const void *h, size_t bytes) {
    HIP_OK(hipMalloc(d, bytes ? bytes : 1));
    HIP_OK(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
    return 0;
}

int main(int argc, char **argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s input.bin output.bin\n", argv[0]); return 1; }
    if (vf_abi_version() != VF_ABI_VERSION) { fprintf(stderr, "ABI %d, header %d\n", vf_abi_version(), VF_ABI_VERSION); return 1; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int32_t hd[10];
    if (fread(hd, sizeof(int32_t), 10, f) != 10) { fprintf(stderr, "bad header\n"); return 4; }
    const int H = hd[0], W = hd[1], adim = hd[2], sdim = hd[3], nd = hd[4], nc = hd[5], seq = hd[6], B = hd[7], nex = hd[8];
    const size_t nw = (size_t)hd[9];
    const int T = seq - nc, nca = nc > 1 ? nc - 1 : 1;

    vf_config cfg = {0};
    cfg.height = H; cfg.width = W; cfg.adim = adim; cfg.sdim = sdim; cfg.ndesig = nd; cfg.n_context = nc;
    cfg.sequence_length = seq; cfg.num_masks = 10; cfg.max_batch = B; cfg.device = 0; cfg.precision = 0;
    cfg.ncam = 1; cfg.n_draws = 1; cfg.arch = 0;
    if (vf_weight_count(&cfg) != nw) { fprintf(stderr, "weights: file has %zu, library wants %zu\n", nw, vf_weight_count(&cfg)); return 4; }

    float *weights = slurp(f, nw * sizeof(float));
    const size_t n_fr = (size_t)nc * H * W * 3, n_st = (size_t)nc * sdim, n_ca = (size_t)nca * adim;
    const size_t n_di = (size_t)nc * H * W * nd, n_ac = (size_t)B * T * adim;
    uint8_t *frames = slurp(f, n_fr);
    float *states = slurp(f, n_st * 4), *ctx_act = slurp(f, n_ca * 4), *distrib = slurp(f, n_di * 4), *actions = slurp(f, n_ac * 4);
    int32_t *goal = slurp(f, (size_t)nd * 2 * 4);
    float *fw = slurp(f, 4);
    fclose(f);

    HIP_OK(hipSetDevice(0));
    vf_handle *h = NULL;
    VF_CALL(vf_create(&cfg, &h));
    VF_CALL(vf_load_weights(h, weights, nw));
    VF_CALL(vf_set_persistent(h, 1));

    void *d_fr, *d_st, *d_ca, *d_di, *d_ac, *d_sc, *d_pt, *d_of, *d_od, *d_os;
    if (to_device(&d_fr, frames, n_fr) || to_device(&d_st, states, n_st * 4) || to_device(&d_ca, ctx_act, n_ca * 4) ||
        to_device(&d_di, distrib, n_di * 4) || to_device(&d_ac, actions, n_ac * 4)) return 2;
    const size_t n_of = (size_t)nex * T * H * W * 3, n_od = (size_t)nex * T * H * W * nd, n_os = (size_t)nex * T * sdim;
    HIP_OK(hipMalloc(&d_sc, (size_t)B * 8));
    HIP_OK(hipMalloc(&d_pt, (size_t)B * nd * 8));
    HIP_OK(hipMalloc(&d_of, n_of * 4 + 4));
    HIP_OK(hipMalloc(&d_od, n_od * 4 + 4));
    HIP_OK(hipMalloc(&d_os, n_os * 4 + 4));

    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    VF_CALL(vf_set_context(h, d_fr, d_st, d_ca, d_di, st));
    VF_CALL(vf_rollout(h, d_ac, B, goal, *fw, NULL, d_sc, d_pt, st));
    int32_t status = -1;
    VF_CALL(vf_device_status(h, &status));
    if (status != 0) { fprintf(stderr, "device status %d\n", status); return 5; }
    if (nex > 0) VF_CALL(vf_export(h, 0, nex, d_of, d_od, d_os, st));
    HIP_OK(hipStreamSynchronize(st));

    double *scores = malloc((size_t)B * 8), *per_task = malloc((size_t)B * nd * 8);
    float *of = malloc(n_of * 4 + 4), *od = malloc(n_od * 4 + 4), *os = malloc(n_os * 4 + 4);
    HIP_OK(hipMemcpy(scores, d_sc, (size_t)B * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(per_task, d_pt, (size_t)B * nd * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(of, d_of, n_of * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(od, d_od, n_od * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(os, d_os, n_os * 4, hipMemcpyDeviceToHost));

    FILE *o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 1; }
    fwrite(scores, 8, (size_t)B, o); fwrite(per_task, 8, (size_t)B * nd, o);
    fwrite(of, 4, n_of, o); fwrite(od, 4, n_od, o); fwrite(os, 4, n_os, o);
    fclose(o);
    printf("vf_c_host: %d sequences x %d steps rolled through the C ABI (version %d); score[0] = %.17g\n", B, T,
           vf_abi_version(), scores[0]);
    VF_CALL(vf_destroy(h));
    return 0;
}
