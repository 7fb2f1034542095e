/* vf_hip.h - C ABI of libvf_hip.so, the MI355X (gfx950) video-prediction + cost engine.
 *
 * The reference (SudeepDasari/visual_foresight) has no FFI: its planner crosses to the device
 * through one Python call into a TensorFlow-1 session.  Each entry point below names the
 * reference interface it replaces (paths relative to the reference root).  All bulk pointers
 * are DEVICE pointers owned by the caller (PyTorch-ROCm tensors are the usual carrier); small
 * control data (config, goal pixels) is passed by value / host pointer.  Every function
 * returns 0 on success or a negative vf_status; vf_last_error() describes the last failure of
 * the calling thread.  No exception crosses this boundary.  Every device buffer - packed
 * weights, activations, predictions, schedules - is sized from the config and allocated in
 * vf_create(); no later call allocates device memory (vf_load_weights refills the same
 * buffers).  vf_set_context / vf_rollout / vf_export / vf_register / vf_allgather_scores only
 * enqueue work on the caller's HIP stream (hipStream_t passed as void*; NULL = the default
 * stream) and never synchronise it; a handle is driven from one stream at a time.
 * vf_load_weights, vf_device_status and the debug hooks are the blocking calls.
 *
 * Tensor layouts (C order, float32 unless noted; ncam = views, the reference stacks views on
 * a camera axis, visual_mpc/video_prediction/vpred_model_interface.py:78,88):
 *   context frames   uint8  [n_context][ncam][H][W][3]
 *   context distrib         [n_context][ncam][H][W][ndesig]
 *   context states          [n_context][sdim]
 *   context actions         [n_context-1][adim]
 *   actions                 [B][T][adim]                    T = sequence_length - n_context
 *   predicted frames        [B][T][ncam][H][W][3]           in [0,1]
 *   predicted distrib       [B][T][ncam][H][W][ndesig]      each (b,t,c,.,.,p) plane sums to 1
 *   predicted states        [B][T][sdim]
 *   scores        float64   [B / n_draws], scores_per_task [B / n_draws][ncam * ndesig]
 *                           (camera-major, as pixel_cost_controller.py:138-149 stacks them; float64
 *                           like the reference's host cost under NumPy >= 2, so device rounding
 *                           cannot create ties in front of the elite argsort)
 */
#ifndef VF_HIP_H
#define VF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VF_ABI_VERSION 7

typedef enum vf_status {
    VF_OK = 0,
    VF_ERR_INVALID = -1,   /* bad argument / shape / state */
    VF_ERR_HIP = -2,       /* a HIP runtime call failed */
    VF_ERR_NOMEM = -3,     /* device allocation failed */
    VF_ERR_NOWEIGHTS = -4, /* rollout before vf_load_weights */
    VF_ERR_NOCONTEXT = -5  /* rollout before vf_set_context */
} vf_status;

/* Static shape of one predictor instance.  Replaces the `conf` dict + placeholders of
 * visual_mpc/video_prediction/setup_predictor.py:98-114 (orig_size, adim, sdim, ndesig,
 * context_frames, sequence_length, batch_size) and the predictor hparams of
 * visual_mpc/policy/cem_controllers/pixel_cost_controller.py:29-33
 * (designated_pixel_count, run_batch_size). */
typedef struct vf_config {
    int32_t height, width;      /* multiples of 8 */
    int32_t adim, sdim;
    int32_t ndesig;             /* designated pixels per view (1..4) */
    int32_t n_context;          /* context frames (>= 1) */
    int32_t sequence_length;    /* n_context + T */
    int32_t num_masks;          /* CDNA kernel slots K (masks = K + 1): 10 for arch 0 / 1; 6 for arch 2 (four kernels in
                                 * the checkpoint, seven compositing layers); 4 for arch 3 (four kernels, seven layers) */
    int32_t max_batch;          /* run_batch_size: most samples per vf_rollout call */
    int32_t device;             /* HIP device ordinal */
    int32_t precision;          /* arithmetic of the conv-LSTM gate GEMMs (96 % of the work):
                                 * 0 = exact fp32 MFMA (default);
                                 * 1 = fp32 emulated with six bf16 MFMA products per multiply
                                 *     (3-way exact operand split, fp32 accumulate; fp32-class
                                 *     accuracy, see csrc/vf_conv_bf16x6.h).  Everything else is
                                 *     fp32 in both modes. */
    int32_t ncam;               /* camera views (1..4; 0 = 1): one network (own weights) per view,
                                 * all rolled by the same launch on the same action sequences
                                 * (reference: IndepMultiSAVP..., vpred_model_interface.py:60-88) */
    int32_t n_draws;            /* latent draws per action (>= 1; 0 = 1): every n_draws consecutive
                                 * sequences of a rollout are draws of ONE action and their costs
                                 * are averaged on the device (the reference's hook repeats each
                                 * action stochastic_planning[0] times,
                                 * samplers/gaussian_sampler.py:140-141) */
    int32_t arch;               /* network: 0 = action-conditioned CDNA conv-LSTM (cdna_arch.py; height and
                                 * width multiples of 8); 1 = SAVP-class stochastic generator (savp_arch.py:
                                 * the same conv-LSTM core between one more encoder and decoder scale, the
                                 * first context frame as an extra compositing layer, the per-step latent as
                                 * extra action channels; multiples of 16); 2 = arch 1 moved closer to the
                                 * published SAVP generator (savp_arch.py, Savp2Config: the vector [action, latent,
                                 * state] conditions EVERY conv-LSTM - lstm weights [5][5][Cx + adim + sdim + Ch][4Ch] -
                                 * and the compositing is the published one: four CDNA kernels, layers [warps,
                                 * previous, first, scratch]; exact fp32 only, at least 64 x 64); 3 = the PUBLISHED SAVP
                                 * generator (savp3_arch.py; arXiv:1804.01523 appendix A: instance norm after every conv and
                                 * inside the conv-LSTM cells, conv + average pool / bilinear up-sampling + conv, the layer
                                 * table by image size, [action, state, rnn_z(latent)] tile-concatenated in front of every
                                 * conv and conv-LSTM, 3x3 heads, dependent masks, symmetric-padded CDNA warps; exact fp32
                                 * only).  The reference selects the class through conf['model'],
                                 * vpred_model_interface.py:52-58 */
    int32_t zdim;               /* arch 3: how many of the adim action channels are the per-step latent z_t (they go
                                 * through the rnn_z cell and do not reach the state predictor); 0 for arch 0 - 2 */
    int32_t layer_spec;         /* arch 3: 0 = encoder / decoder table by min(height, width) as the public code selects it;
                                 * 32 / 64 / 128 force a table (64 = the paper's five-cell network); 0 for arch 0 - 2 */
} vf_config;

typedef struct vf_handle vf_handle;

int vf_abi_version(void);
const char *vf_last_error(void);

/* Number of float32 values of ONE view's weights for `cfg` (the canonical tensor table of
 * visual_foresight_amd/video_prediction/cdna_arch.py, concatenated in table order);
 * vf_load_weights expects ncam such blobs back to back. */
size_t vf_weight_count(const vf_config *cfg);

/* Build one engine: allocates every device buffer (weights, recurrent state, activations,
 * predictions for max_batch samples).  Replaces setup_predictor()'s graph/session build,
 * visual_mpc/video_prediction/setup_predictor.py:61-128. */
int vf_create(const vf_config *cfg, vf_handle **out);
int vf_destroy(vf_handle *h);

/* Upload network weights from a HOST blob in canonical layout (views back to back) and re-pack
 * them for the MFMA kernels into the buffers vf_create allocated; may be called again at any
 * time (hot swap; blocks until rollouts in flight have finished).  Replaces saver.restore /
 * model.restore, setup_predictor.py:130-145, and predictor.restore(),
 * pixel_cost_controller.py:34. */
int vf_load_weights(vf_handle *h, const float *host_blob, size_t n_floats);

/* Install the planning context (device pointers).  Replaces get_context(),
 * visual_mpc/video_prediction/pred_util.py:4-13 (last n_context frames, uint8 -> float/255) and
 * the batch-1 placeholders tiled per tower, setup_predictor.py:40-44: the context is kept once
 * and broadcast to every sample by the kernels, never materialised per sample. */
int vf_set_context(vf_handle *h, const uint8_t *d_frames, const float *d_states,
                   const float *d_ctx_actions, const float *d_ctx_distrib, void *stream);

/* Roll B (<= max_batch, a multiple of n_draws) action sequences through every view's predictor
 * for T steps and reduce the predicted designated-pixel distributions to costs on the device.
 * Replaces predictor_func()/sess.run, setup_predictor.py:164-200, the call at
 * pixel_cost_controller.py:83, and the host cost of pixel_cost_controller.py:135-197:
 *   e[b][c*ndesig+p] = sum_t w_t * E_{distrib[b,t,c,p]}[ || pix - goal_cp || ] / sum_t w_t,
 *   w = (1, ..., 1, finalweight);   score_b = mean over tasks of e[b][.]        (:153)
 * or, with task_weights (HOST float [ncam*ndesig], NULL = plain mean), the trade-off weighted
 * sum  score_b = sum_i task_weights[i] * e[b][i]  (register_gtruth_controller.py:88-94).  With
 * n_draws > 1 both are means over each action's draws.  goal_pix: HOST int32
 * [ncam][ndesig][2] (row, col).  d_scores_per_task may be NULL.  Predictions stay resident in
 * the handle until the next vf_rollout (see vf_export).  If a tile of the launch gave up
 * waiting for its producers (see vf_device_status) every score of this and of later rollouts
 * is NaN until the status has been read, and the predictions vf_export would copy out are
 * undefined (the launch was abandoned): check the scores or vf_device_status first. */
int vf_rollout(vf_handle *h, const float *d_actions, int32_t B, const int32_t *goal_pix,
               float finalweight, const float *task_weights, double *d_scores,
               double *d_scores_per_task, void *stream);

/* Copy the predictions of the last vf_rollout out in the reference's layout (camera axis,
 * normalised distributions).  Any destination may be NULL.  first/count select a range of rolled
 * sequences.  Replaces the gen_images/gen_distrib/gen_states fetch of
 * setup_predictor.py:155-200. */
int vf_export(vf_handle *h, int32_t first, int32_t count, float *d_frames, float *d_distrib,
              float *d_states, void *stream);

/* Designated-pixel registration (reference
 * visual_mpc/policy/cem_controllers/register_gtruth_controller.py:54-173, get_warp_err).  The
 * registration NETWORK is not part of the reference snapshot; this entry point takes its output,
 * a flow field d_flow [ncam][H][W][2] = (dx, dy) that maps reference pixel (r, c) to the point
 * (x, y) = (c + dx, r + dy) of the current frame, and does the rest on the device:
 *   d_warp_pts [ncam][H][W][2] = (x, y)                        (optional, may be NULL)
 *   d_warped   [ncam][H][W][3] = bilinear sample of d_current at warp_pts, border-clamped (opt.)
 *   per (camera, task) with designated/goal pixel d_pix [ncam][ntask][2] (row, col, int32,
 *   DEVICE) in the reference image:
 *     region == 0: d_desig = flipped warp_pts at the pixel (:129-135), d_err = L2 photometric
 *                  error between d_reference and the warped frame at the pixel (:163-170);
 *     region  > 0: d_desig = per-coordinate MEDIAN of warp_pts over the (2*region+1)^2 window,
 *                  d_err = mean squared error over the window (:139-161); the window is clipped
 *                  to [0, size - clip_sub] (the reference clips the start window with 1, the goal
 *                  window with 0).
 * d_desig float [ncam][ntask][2] (row, col), d_err float [ncam][ntask].  ncam/H/W are the
 * handle's.  The trade-off weights (1/err normalised over cameras and registrations, :88-91)
 * are a handful of numbers and stay on the host. */
int vf_register(vf_handle *h, const float *d_current, const float *d_reference, const float *d_flow,
                const int32_t *d_pix, int32_t ntask, int32_t region, int32_t clip_sub,
                float *d_warped, float *d_warp_pts, float *d_desig, float *d_err, void *stream);

/* The one collective of multi-GPU planning: all-gather every rank's n_local score values (float64)
 * (n_local equal on all ranks: pad ragged shards) into d_all [world * n_local] with RCCL over
 * xGMI, on the caller's stream.  nccl_comm is the caller's ncclComm_t (one rank per GPU).  The
 * reference instead concatenates whole predicted videos of its towers on the host,
 * visual_mpc/video_prediction/setup_predictor.py:155-162.  RCCL is bound with dlopen at first use,
 * so the library has no link-time dependency on it. */
int vf_allgather_scores(vf_handle *h, void *nccl_comm, const double *d_local, int32_t n_local,
                        double *d_all, void *stream);

/* One host thread driving several GPUs (the reference's in-process towers: `ngpu` GPUs behind one
 * policy object, visual_mpc/video_prediction/setup_predictor.py:70,117-123).  vf_comm_init_all
 * creates one RCCL communicator per listed device (ncclCommInitAll; devices distinct) with the same
 * library instance vf_allgather_scores uses; vf_allgather_scores_group issues the all-gather of
 * every local rank - handle hs[i] on device hs[i]'s, communicator comms[i], stream streams[i]
 * (streams NULL = default streams) - inside one ncclGroupStart / ncclGroupEnd, which is how a single
 * thread must drive several ranks.  vf_comm_destroy releases one communicator. */
int vf_comm_init_all(int32_t n, const int32_t *devices, void **comms);
int vf_comm_destroy(void *comm);
int vf_allgather_scores_group(int32_t n, vf_handle *const *hs, void *const *comms,
                              const double *const *d_local, int32_t n_local, double *const *d_all,
                              void *const *streams);

/* Persistent rollout (no reference counterpart).  When enabled vf_rollout runs ALL steps, layers
 * and samples as one persistent launch whose workgroups draw tiles from a ticket queue and
 * honour per-sample dependencies, so the tail of one layer overlaps the head of the next
 * (visual_foresight_amd/csrc/vf_persistent.h).  Results are bit-identical to the per-layer
 * launches.  A tile that waits too long for its producers (a bounded spin, ~seconds) raises a
 * STICKY device status word: from then on every score is NaN.  vf_device_status synchronises
 * the device, returns the word (0 = healthy) and, if it was raised, re-arms it and drops the
 * cached context-only part of the network (the abandoned launch may have left it half-written),
 * so a retry recomputes it. */
int vf_set_persistent(vf_handle *h, int32_t enable);
int vf_device_status(vf_handle *h, int32_t *status);

/* XCD-aware ticket queues of the persistent rollout (default on; no reference counterpart).  The items
 * of every phase are dealt to one queue per XCD so that an output-channel group's weight slice and a
 * sample's neighbouring tiles are served from ONE XCD's L2; a workgroup draws from the queue of the XCD
 * it runs on and steals from the others once its own is empty.  Placement only affects speed: results
 * are bit-identical with one queue (enable = 0), which is the plain phase order. */
int vf_set_xcd_queues(vf_handle *h, int32_t enable);

/* Fused items of the persistent rollout (default on; no reference counterpart).  (1) The last transposed convolution and the
 * compositing of the next frame become ONE item per tile: the tile stays in registers / LDS, its LayerNorm partial
 * is published, the item waits for the sample's other tiles and composes its pixels itself - the full-resolution
 * decoder tensor is never written to memory (visual_foresight_amd/csrc/vf_fused_top.h).  Same arithmetic on the same
 * values: results are bit-identical to the two-phase schedule and to the per-layer launches.  (2) The two convolutions of
 * the 8 x 8 bottleneck (3x3 / 2, then 1x1 with the tiled action / state entering as a per-sample bias) become one item per
 * row tile: the first conv's tile holds whole images and all 64 channels, goes through LDS instead of memory and feeds the
 * second GEMM in its stand-alone K order (conv_pair_epilogue, visual_foresight_amd/csrc/vf_conv_mfma.h) - one dependency
 * hop per sample-step less, bit-identical as well.  enable = 0 switches both off (A/B measurements, tests). */
int vf_set_fuse_top(vf_handle *h, int32_t enable);

/* Scheduling options of the persistent rollout that change timing only, never results (no reference counterpart; round 5).
 *   VF_OPT_YIELD_BUDGET (0): cooperative CU-level priority - the recurrent half of an early-started conv-LSTM item sleeps
 *       while the other workgroup of its CU runs chain-critical work, at most `value` polls of ~0.4 us per item
 *       (visual_foresight_amd/csrc/vf_conv_mfma.h, "yielding"); 0 = off, -1 = automatic (on for small shards).
 *   VF_OPT_WRITE_THROUGH (1): tiles with 16-byte epilogue stores publish their outputs as sc1 (write-through) stores and
 *       skip the release fence in front of their completion counters; 0 = plain stores + release, 1 = on (default).
 * Exists for A/B measurements and for the tests that hold every setting to the same bits. */
#define VF_OPT_YIELD_BUDGET 0
#define VF_OPT_WRITE_THROUGH 1
int vf_set_sched_option(vf_handle *h, int32_t option, int32_t value);

/* Context de-duplication (default on).  While a step's inputs are context, part of the network
 * sees identical inputs for every sample (step < n_context-1: everything; step < n_context: the
 * encoder up to enc2); those launches then run once with batch 1 and are broadcast.  The
 * per-sample arithmetic is unchanged, results are bit-identical either way; the switch exists
 * for A/B measurements. */
int vf_set_dedup(vf_handle *h, int32_t enable);

/* Measurement hooks (no reference counterpart).  While enabled, every launch of the dominant
 * kernel - the persistent rollout, or the fused conv-LSTM gate GEMM of the per-layer path - is
 * bracketed by HIP events on its launch stream (host-side event objects are created on demand).
 * vf_get_profile waits for them and returns: kernel_ms = sum of the per-launch durations,
 * busy_ms = time during which at least one such launch was in flight (== kernel_ms when
 * launches do not overlap), the number of launches and their algorithmic FLOPs
 * (2 * B*H*W * 25*(Cx+Ch) * 4C each); then it resets the counters. */
int vf_set_profiling(vf_handle *h, int32_t enable);
int vf_get_profile(vf_handle *h, double *kernel_ms, int64_t *launches, double *flops,
                   double *busy_ms);

/* Introspection for tests/benchmarks: algorithmic multiply-accumulates of one sample-step. */
double vf_macs_per_sample_step(const vf_config *cfg);

/* Debug hooks: per-phase wait/run ticks of the persistent launch (tools/persist_stats.py), and a
 * switch that raises the device status word by hand (tests of the in-band failure path). */
int vf_set_phase_stats(vf_handle *h, int32_t enable);
int vf_debug_phase_stats(vf_handle *h, int32_t max_phases, int32_t *types, int32_t *items,
                         uint64_t *wait_run);
int vf_debug_poison_status(vf_handle *h);

#ifdef __cplusplus
}
#endif
#endif /* VF_HIP_H */
