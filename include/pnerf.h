/*
 * pnerf.h -- C ABI of libpnerf_hip.so, the MI355X (gfx950) implementation of the Point-NeRF
 * render/optimise hot path.
 *
 * Conventions
 *   - every pointer named d_* is DEVICE memory owned by the caller (contiguous, 16-byte aligned);
 *     the library never allocates or frees device memory: sizes of scratch areas are returned by
 *     the *_bytes() queries and the caller (torch's caching allocator in the Python host) provides
 *     them;
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it (no host
 *     synchronisation) unless its comment says "synchronous";
 *   - return value: 0 = ok, <0 = PNERF_E_* (nothing was enqueued);
 *   - not thread-safe per stream, re-entrant across streams.
 *
 * Each entry point cites the reference interface it replaces (paths under /root/reference).
 */
#ifndef PNERF_H
#define PNERF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PNERF_E_INVAL   (-1)   /* bad argument (null pointer, K/SR/P out of range, ...) */
#define PNERF_E_WS      (-2)   /* workspace too small */
#define PNERF_E_LAUNCH  (-3)   /* hipGetLastError() != hipSuccess after a launch */
#define PNERF_E_UNSUP   (-4)   /* configuration not supported by this build */

#define PNERF_MAX_K 16

/* Grid description: what lighting_fast_querier.get_hyperparameters computes per call
 * (models/neural_points/point_query.py:47-71) plus the querier constants of :35-42. */
typedef struct pnerf_grid_params {
    float ranges[6];       /* padded min xyz [0..2] / max xyz [3..5]  (ranges_tensor) */
    float vsize[3];        /* scaled voxel size = vsize * vscale       (scaled_vsize)  */
    int32_t vdim[3];       /* grid dimensions                          (scaled_vdim)   */
    int32_t kernel_size[3];/* neighbor search window (layers = (ks[0]+1)/2) */
    int32_t query_size[3]; /* occupancy dilation window */
    int32_t P;             /* max points per voxel */
    int32_t max_o;         /* max occupied voxels (overflow is reported, see pnerf_grid_info) */
    float radius;          /* radius_limit (0 disables the radius test) */
} pnerf_grid_params;

/* info words written by pnerf_grid_build at the start of the grid workspace */
enum { PNERF_GI_N_IN_GRID = 0, PNERF_GI_N_OCC = 1, PNERF_GI_MAX_CNT = 2, PNERF_GI_CELL0 = 3,
       PNERF_GI_FIRST_IDX = 4, PNERF_GI_OSTART_OFF = 5 /* internal: where the cell offsets sit in the workspace */, PNERF_GI_LEN = 8 };

/* ---- library ---------------------------------------------------------------------------- */
int pnerf_version(void);                 /* 1000*major + minor */
const char *pnerf_arch(void);            /* "gfx950" */

/* ---- voxel grid (replaces claim_occ / map_coor2occ / fill_occ2pnts and their ~140 MB of
 * per-call tables: models/neural_points/cuda/query_worldcoords.cu:18-162, host :308-365).
 * Deterministic by construction: points of a cell are stored in ascending point index (the
 * reference's canonical serial order), the dilated occupancy is a bit field.  The grid is a
 * function of (xyz, params) only and is meant to be cached across calls by the host. */
size_t pnerf_grid_workspace_bytes(const pnerf_grid_params *gp, int n_points);
int pnerf_grid_build(const pnerf_grid_params *gp, const float *d_xyz, int n_points,
                     void *d_grid_ws, size_t ws_bytes, void *stream);
/* synchronous: copies the PNERF_GI_* words to host_info[PNERF_GI_LEN] (stream is synchronised) */
int pnerf_grid_info(const void *d_grid_ws, int32_t *host_info, void *stream);
/* Elementwise minimum and maximum over the points: d_out6 = {min x, min y, min z, max x, max y, max z} (device, 6 floats), what
 * lighting_fast_querier.get_hyperparameters (models/neural_points/point_query.py:51-52) takes with torch.min / torch.max before it clamps to
 * opt.ranges and pads: one HBM pass, asynchronous on `stream`.  Comparisons are exact (no arithmetic on the values); NaN coordinates are not
 * ordered (the reference's reductions would propagate them). */
int pnerf_points_minmax(const float *d_xyz, int64_t n_points, float *d_out6, void *stream);

/* ---- query (replaces mask_raypos / get_shadingloc / query_neigh_along_ray_layered and the ATen
 * glue between them: query_worldcoords.cu:165-302, host :367-431; entry point
 * woord_query_grid_point_index, query_worldcoords.cpp:34-82).
 *
 * Ray samples come either from d_raypos [R,D,3] (the native op's own input) or -- fused, never
 * materialised -- from campos + raydir * mid[d] with the D mid-point depths d_mid computed by the
 * host exactly as near_far_linear_ray_generation does (models/rendering/diff_ray_marching.py:369-392).
 * If jitter > 0 (training: point_query.py:81 uses 0.3) d_mid holds the D un-jittered segment lengths and every segment is
 * scaled by 1 + jitter (U - 0.5) in-kernel, U = pnerf_debug_uniform(seed, ray * D + d); the end points are the SEQUENTIAL fp32
 * running sum (torch.cumsum of the reference's CPU path), so the samples are bit-defined; near/far are then required.
 *
 * Outputs are DENSE OVER ALL R RAYS (no host sync, no compaction):
 *   d_sample_loc  [R,SR,3] f32   world position of the first <=SR occupied samples, 0 elsewhere
 *   d_sample_pidx [R,SR,K] i32   neighbor point indices in the reference's slot order, -1 padded
 *   d_sample_nn   [R,SR]   i32   number of valid neighbors of each sample (0 for empty slots)
 *   d_ray_hit     [R]      i32   1 if the ray has at least one sample with a neighbor
 *   d_valid_list  [R*SR]   i32   ascending list of r*SR+s with nn>0 (the aggregator's work list)
 *   d_counters    [8]      i32   [0]=#valid samples (len of d_valid_list) [1]=#rays hit
 *                                [2]=#selected samples [3]=#valid neighbor slots
 * The reference's [R'',...] outputs are row gathers of these by d_ray_hit (done by the host). */
size_t pnerf_query_workspace_bytes(int R, int SR);
int pnerf_query(const pnerf_grid_params *gp, const void *d_grid_ws,
                const float *d_raypos,                       /* [R,D,3] or NULL */
                const float *campos3_host, const float *d_raydir, const float *d_mid,   /* used if d_raypos==NULL */
                float near_depth, float far_depth, float jitter, uint64_t seed,
                int R, int D, int SR, int K,
                float *d_sample_loc, int32_t *d_sample_pidx, int32_t *d_sample_nn,
                int32_t *d_ray_hit, int32_t *d_valid_list, int32_t *d_counters,
                void *d_query_ws, size_t ws_bytes, void *stream);

/* ---- per-neighbor gather (NeuralPoints.forward's index_select block,
 * models/neural_points/neural_points.py:706-717) and its backward scatter-add.  Row i of each
 * output holds point max(idx[i],0): -1 slots read point 0 exactly as the reference does. */
int pnerf_gather_rows(const float *d_src, int n_src, int width, const int32_t *d_idx, int64_t n_idx,
                      float *d_dst, void *stream);
int pnerf_scatter_add_rows(const float *d_grad_rows, const int32_t *d_idx, int64_t n_idx, int width,
                           float *d_grad_src, int n_src, void *stream);

/* ---- the zero-one regulariser on the neighbor slots' confidences, fused (gather + gradient_clamp of
 * models/aggregators/point_aggregators.py:722-724,812 + loss_zero_one of models/base_rendering_model.py:630-641 + their backward).
 * forward: d_partial[b], b < pnerf_zero_one_blocks(n_idx), = per-block sums of log(v) + log(1 - v) over the n_idx slots
 * (v = clamp(clamp(conf[max(idx, 0)], 1e-4, 1), eps, 1 - eps)); the caller adds them and divides by its (global) element count.
 * backward: d_grad_conf[max(idx, 0)] += d_gscale[0] * (1 / v - 1 / (1 - v)) where the clamp to [eps, 1 - eps] was inactive. */
int pnerf_zero_one_blocks(int64_t n_idx);
int pnerf_zero_one_forward(const float *d_conf, int n_points, const int32_t *d_idx, int64_t n_idx, float eps,
                           float *d_partial, void *stream);
int pnerf_zero_one_backward(const float *d_conf, int n_points, const int32_t *d_idx, int64_t n_idx, float eps,
                            const float *d_gscale, float *d_grad_conf, void *stream);
/* the same over the DENSE neighbor table d_idx [R][slots_per_ray] of a query, restricted to the rays with d_ray_hit[r] > 0 (the reference's
 * conf_coefficient exists for the hit rays only): no [R'', SR, K] copy of the table is made.  d_partial holds pnerf_zero_one_blocks(R * 256)
 * floats; the caller divides by (hit rays) x slots_per_ray. */
int pnerf_zero_one_forward_rays(const float *d_conf, int n_points, const int32_t *d_idx, const int32_t *d_ray_hit, int R, int slots_per_ray, float eps,
                                float *d_partial, void *stream);
int pnerf_zero_one_backward_rays(const float *d_conf, int n_points, const int32_t *d_idx, const int32_t *d_ray_hit, int R, int slots_per_ray, float eps,
                                 const float *d_gscale, float *d_grad_conf, void *stream);
/* the colour term of the training loss (models/base_rendering_model.py:543-551 "ray_masked_coarse_raycolor": sum over the rays that hit of
 * (colour - gt)^2; the caller divides by its global element count) over the DENSE ray colours d_ray_color [R,3] / d_gt [R,3] with the rays' hit
 * flags -- the hit rays are not compacted on the way to a scalar.  forward: d_partial[b], b < pnerf_color_loss_blocks(R), block sums.
 * backward: d_grad_ray_color [R,3] = 2 d_gscale[0] (colour - gt) for a hit ray, 0 for a miss (every element is written). */
int pnerf_color_loss_blocks(int R);
int pnerf_color_loss_forward_rays(const float *d_ray_color, const float *d_gt, const int32_t *d_ray_hit, int R, float *d_partial, void *stream);
int pnerf_color_loss_backward_rays(const float *d_ray_color, const float *d_gt, const int32_t *d_ray_hit, int R, const float *d_gscale,
                                   float *d_grad_ray_color, void *stream);

/* ---- aggregator MLP + renderer (PointAggregator.forward/viewmlp,
 * models/aggregators/point_aggregators.py:488-644,727-814; ray-dist,
 * models/neural_points_volumetric_model.py:271-279; ray_march,
 * models/rendering/diff_ray_marching.py:508-554).  Lego-script architecture only:
 * 284->256->256, (+7)->256->256, alpha 256->1, colour 280->128->128->128->3, LeakyReLU(0.01),
 * linear distance kernel, agg_dist_pers=20, agg_intrp_order=2.                                   */

/* Offsets (in floats) of each tensor inside the flat MLP parameter / gradient vector, in the
 * reference's state_dict order: block1.0.{weight,bias}, block1.2.*, block3.0.*, block3.2.*,
 * alpha_branch.0.*, color_branch.{0,2,4,6}.* ; every weight is [out,in] row-major (torch layout). */
#define PNERF_MLP_NTENSORS 18
int pnerf_mlp_layout(int feat_dim, int64_t *offsets /*[PNERF_MLP_NTENSORS+1]*/);
size_t pnerf_mlp_packed_bytes(void);
/* repack the flat parameter vector into MFMA fragment order: 14 two-plane f16 images (forward W and dgrad W^T of the four aggregator and three colour
 * layers, csrc/f16x3.h) and the 8 mixed-format images of the aggregator layers (f16 h fragments + e4m3 cross-term fragments with block exponents,
 * csrc/mixq.h).  Called once per optimisation step (the weights change); ~20 us. */
int pnerf_mlp_pack(const float *d_params, void *d_packed, void *stream);

typedef struct pnerf_camera {
    float campos[3];
    float camrot[9];       /* c2w rotation, row-major */
    float rw2c[9];         /* NeuralPoints.Rw2c, row-major (identity unless normview) */
    float vsize_z;         /* unscaled opt.vsize[2] (ray-dist clamp) */
    int32_t raydist_mode_unit;
    float bg[3];           /* background colour */
    int32_t has_bg;
} pnerf_camera;

typedef struct pnerf_points {        /* the neural point cloud, all [N,*] row-major f32 */
    const float *xyz;                /* [N,3]  */
    const float *embedding;          /* [N,F]  */
    const float *conf;               /* [N,1]  */
    const float *dir;                /* [N,3]  */
    const float *color;              /* [N,3]  */
    int32_t n, feat_dim;
} pnerf_points;

typedef struct pnerf_point_grads {   /* gradient accumulators (added to, never zeroed) */
    float *embedding, *conf, *dir, *color;
    void *ready_event;               /* optional hipEvent_t (NULL: none): recorded on the call's stream as soon as the four point
                                      * gradients are complete, i.e. BEFORE the weight-gradient GEMMs of the same call are
                                      * enqueued -- a data-parallel caller starts the all-reduce of the (large) point
                                      * gradients on another stream behind this event and overlaps it with those GEMMs */
    const float *zero_one_gscale;    /* optional (NULL: none; pnerf_render_backward only): the conf gradient of the zero-one regulariser on the
                                      * hit rays' conf_coefficient (pnerf_zero_one_backward_rays: += d_gscale[0] (1 / v - 1 / (1 - v)) per neighbor slot,
                                      * empty slots on point 0) is added BY THIS CALL -- the term of a slot rides on the conf atomic the backward issues
                                      * for that neighbor row anyway, the empty slots of the hit rays are one closed-form addition to point 0
                                      * ((#rays hit x SR x K - #valid neighbor slots) identical terms): the regulariser's own pass of ~7 M atomics on
                                      * the same addresses is not run */
    float zero_one_eps;              /* its clamp bound (opt.zero_epsilon) */
} pnerf_point_grads;

/* bytes of saved activations per valid neighbor row / per valid sample (training forward) */
size_t pnerf_agg_saved_bytes(int64_t n_valid_samples, int K);
size_t pnerf_agg_workspace_bytes(int64_t n_valid_max, int K);   /* inference scratch for n_valid_max valid samples */
size_t pnerf_render_backward_workspace_bytes(int R, int SR);

/* Forward: for every valid sample in d_valid_list computes (sigma, r, g, b) into
 * d_decoded [R,SR,4] (zero elsewhere), d_weight [R,SR,K] (normalised distance weights, before
 * confidence), then ray-dist + alpha compositing into d_ray_color [R,3], d_opacity [R,SR],
 * d_bg_trans [R], d_blend_w [R,SR].  n_valid_max = capacity in valid samples (>= d_counters[0], which
 * the host reads once per call to size its buffers).  Inference: d_saved == NULL and d_ws holds
 * pnerf_agg_workspace_bytes(n_valid_max, K).  Training: d_saved holds pnerf_agg_saved_bytes(n_valid_max, K)
 * and keeps the activations needed by pnerf_render_backward (which must get the same n_valid_max). */
int pnerf_render_forward(const pnerf_camera *cam, const pnerf_points *pts, const void *d_packed_mlp,
                         const float *d_params, const float *d_raydir, const float *d_sample_loc, const int32_t *d_sample_pidx,
                         const int32_t *d_sample_nn, const int32_t *d_valid_list, const int32_t *d_counters,
                         int R, int SR, int K,
                         float *d_decoded, float *d_weight, float *d_ray_color, float *d_opacity,
                         float *d_bg_trans, float *d_blend_w,
                         void *d_saved, int64_t n_valid_max, void *d_ws, size_t ws_bytes, void *stream);

/* Arithmetic of the INFERENCE forward (d_saved == NULL; pnerf_render_forward and pnerf_agg_forward): every fp32 GEMM operand is two f16
 * planes and a multiply-add is 3 MFMA products (default: fp32-class accuracy, sigma / RGB ~1e-6 from an fp32 evaluation) or 2 (the
 * weights' residual plane dropped: a third of the matrix work and half of the weight stream less; measured: rendered ray colour 1e-6 ..
 * 2e-5 from fp32, inside the 1e-4 bar, but per-sample sigma / RGB 2e-4 .. 4e-4, outside it -- which is why it is an OPTION for previews
 * and evaluation loops and not the default; nothing differentiates through it).  Training forwards always run 3.  Returns the
 * previous setting, or PNERF_E_INVAL for n not in {2, 3}.  Process-wide. */
int pnerf_set_inference_products(int n);

/* Arithmetic of the WEIGHT-GRADIENT GEMMs of the training backward (dW = dY^T X summed over all neighbor rows / samples of the step; the
 * reference: cuBLAS SGEMMs under loss.backward(), models/mvs_points_volumetric_model.py:98-118).  1 (default): each operand is streamed as
 * ONE f16 plane rounded to nearest, one MFMA product (11-bit significands, fp32 accumulation; the rounding errors are unbiased and add up
 * like a random walk over the rows: DESIGN.md 4.1).  2: both operands as TWO f16 planes (22 bits), three products -- the arithmetic of the
 * forward and of the input-gradient chain, i.e. fp32-class weight gradients; the forward then saves, and the weight-gradient GEMMs stream,
 * twice the bytes (pnerf_agg_saved_bytes grows accordingly: size the arena AFTER choosing the mode, and keep the mode fixed between a
 * training forward and its backward).  Returns the previous setting, or PNERF_E_INVAL for n not in {1, 2}.  Process-wide. */
int pnerf_set_wgrad_planes(int n);
/* Arithmetic of the CROSS TERMS of the aggregator's tile GEMMs (the four 256-wide nn.Linear layers of
 * models/aggregators/point_aggregators.py:286-344, forward and input-gradient chain).  Every fp32 operand is h + m (two f16 numbers, 22 bits) and
 * a product is h*h + (h*m + m*h): the leading term always runs on f16 factors; the two cross terms, 2^-11 of the result, run on
 * 8 (default, csrc/mixq.h): e4m3 factors on v_mfma_scale_f32_32x32x64_f8f6f4 (weights with per-lane block scales), 2 instead of 3 matrix-pipe
 *    passes per multiply-add, 1.3e-6 rms of sum |terms| per 256-term dot product (tests/test_gpu_mix.py);
 * 16 (csrc/f16x3.h, the arithmetic of rounds 2-5): f16 factors, 3e-8 rms.
 * The two-plane weight-gradient mode (pnerf_set_wgrad_planes(2)) and the two-product inference option always use 16.  Returns the previous
 * setting, or PNERF_E_INVAL.  Process-wide; may change between a training forward and its backward (the saved activations do not depend on it). */
int pnerf_set_cross_terms(int bits);
/* WHICH tile kernels use the e4m3 cross terms while pnerf_set_cross_terms is 8: bit 0 = the inference forward, bit 1 = the training forward,
 * bit 2 = the backward's input-gradient chain.  Default 4 (backward only): gradients stay inside every bar of the oracle comparison (the
 * LeakyReLU masks come from an fp32-class forward), the step is 8 % faster.  With the forward bits set sigma / RGB are 1e-5 .. 6e-5 from the
 * oracle (bar 1e-4; f16 cross terms: 1e-6) and the step is 15 % faster, but pre-activations that close to zero take the other LeakyReLU branch
 * than the fp32 oracle's, which the gradient comparisons against the oracle see (bench.py reports that variant).  Returns the previous mask, or
 * PNERF_E_INVAL.  Process-wide. */
int pnerf_set_cross_terms_where(int mask);

/* Backward of pnerf_render_forward for dL/d(ray_color) = d_grad_ray_color [R,3]:
 * accumulates dL/d(MLP params) into d_grad_params (flat, pnerf_mlp_layout order) and
 * dL/d(point tensors) into pg.  n_valid = the n_valid_max given to the forward call;
 * d_ws holds pnerf_render_backward_workspace_bytes(R, SR). */
int pnerf_render_backward(const pnerf_camera *cam, const pnerf_points *pts, const void *d_packed_mlp,
                          const float *d_params,
                          const float *d_raydir, const float *d_sample_loc, const int32_t *d_sample_pidx,
                          const int32_t *d_sample_nn, const int32_t *d_valid_list, const int32_t *d_counters,
                          int R, int SR, int K, int64_t n_valid,
                          const float *d_decoded, const float *d_weight, const float *d_opacity,
                          const float *d_grad_ray_color,
                          void *d_saved, float *d_grad_params, const pnerf_point_grads *pg,
                          void *d_ws, size_t ws_bytes, void *stream);

/* ---- stand-alone (level-1) forms of the same kernels, for callers that use the reference's modules one by one ---- */

/* work list from a caller-supplied per-sample neighbor count (ray_valid = any_K(sample_pnt_mask),
 * models/aggregators/point_aggregators.py:741): d_list = ascending i with d_nn[i] > 0, d_counters[0] = count */
size_t pnerf_compact_workspace_bytes(int64_t n);
/* d_flags [n_points] int32 := 1 for every point index that occurs in d_pidx [n] (>= 0), 0 elsewhere; d_flags[0] := 1 as well if any entry
 * is negative (an empty neighbor slot reads point 0: models/neural_points/neural_points.py:709).  The rows of the per-point gradients a
 * rank can have touched: the sparse gradient exchange of the data-parallel step (no reference counterpart: its DataParallel is batch 1,
 * models/neural_points_volumetric_model.py:165-168). */
int pnerf_touched_flags(const int32_t *d_pidx, int64_t n, int32_t n_points, int32_t *d_flags, void *stream);
int pnerf_compact_valid(const int32_t *d_nn, int64_t n, int32_t *d_list, int32_t *d_counters, void *d_ws, size_t ws_bytes, void *stream);

/* PointAggregator.forward (point_aggregators.py:727-814) on its own: -> d_decoded [R,SR,4], d_weight [R,SR,K].
 * d_xyz_pers [N,3] / d_loc_pers [R,SR,3]: the caller's perspective coordinates (sampled_xyz_pers, sample_loc); both
 * NULL = project from cam.  Scratch/saved sizing as pnerf_render_forward. */
int pnerf_agg_forward(const pnerf_camera *cam, const pnerf_points *pts, const void *d_packed_mlp, const float *d_params,
                      const float *d_raydir, const float *d_sample_loc, const float *d_xyz_pers, const float *d_loc_pers,
                      const int32_t *d_sample_pidx, const int32_t *d_valid_list, const int32_t *d_counters,
                      int R, int SR, int K, float *d_decoded, float *d_weight,
                      void *d_saved, int64_t n_valid_max, void *d_ws, size_t ws_bytes, void *stream);
/* its backward for dL/d(decoded) = d_grad_decoded [R,SR,4]; d_ws holds pnerf_render_backward_workspace_bytes(0, 1) */
int pnerf_agg_backward(const pnerf_camera *cam, const pnerf_points *pts, const void *d_packed_mlp, const float *d_params,
                       const float *d_raydir, const float *d_sample_loc, const int32_t *d_sample_pidx,
                       const int32_t *d_valid_list, const int32_t *d_counters, int R, int SR, int K, int64_t n_valid,
                       const float *d_decoded, const float *d_weight, const float *d_grad_decoded,
                       void *d_saved, float *d_grad_params, const pnerf_point_grads *pg, void *d_ws, size_t ws_bytes, void *stream);

/* ray_march(ray_dist, ray_valid, ray_features, radiance, alpha, bg) (models/rendering/diff_ray_marching.py:508-554):
 * d_ray_dist [R,SR] f32, d_ray_valid [R,SR] u8, d_features [R,SR,4] (sigma,r,g,b), bg3_host NULL = no background. */
int pnerf_raymarch_forward(const float *d_ray_dist, const uint8_t *d_ray_valid, const float *d_features, const float *bg3_host,
                           int R, int SR, float *d_ray_color, float *d_opacity, float *d_acc_trans, float *d_blend_w,
                           float *d_bg_trans, void *stream);
int pnerf_raymarch_backward(const float *d_ray_dist, const uint8_t *d_ray_valid, const float *d_features, const float *bg3_host,
                            int R, int SR, const float *d_grad_ray_color, float *d_grad_features, void *stream);

/* ---- parameter update: one Adam step of one tensor in one pass (the reference steps two torch.optim.Adam instances,
 * models/mvs_points_volumetric_model.py:80-91 and :98-118; lr / plr, betas (0.9, 0.999), eps 1e-8, no weight decay).
 * All four arrays hold n floats (16-byte aligned arrays take the float4 path); step counts from 1 (the value torch keeps in state['step']). */
int pnerf_adam_step(float *d_param, const float *d_grad, float *d_exp_avg, float *d_exp_avg_sq, int64_t n,
                    double lr, double beta1, double beta2, double eps, int64_t step, void *stream);
/* the same update for a list of tensors in ONE launch (per 24 tensors): each entry carries its own size, learning rate and step count,
 * so both of the reference's optimizers (MLP: 18 tensors at lr, points: 4 tensors at plr) are one call.  `tensors` is a HOST array. */
typedef struct pnerf_adam_tensor {
    float *param; const float *grad; float *exp_avg; float *exp_avg_sq;
    int64_t n; double lr; int64_t step;
} pnerf_adam_tensor;
int pnerf_adam_step_multi(const pnerf_adam_tensor *tensors, int count, double beta1, double beta2, double eps, void *stream);

/* ---- point initialisation: voxel down-sampling of a raw cloud (models/mvs/mvs_utils.py:537-561 construct_vox_points_closest,
 * called at run/train_ft.py:138-139).  Voxel of a point = floor((p - space_min) / vox_size) per axis (fp32), points outside
 * [0, r) are dropped and counted.  Outputs for the M occupied voxels in ascending (x, y, z) order (torch.unique(dim=0) order):
 * d_centroid [M,3] mean of the members (summed in point order), d_grid_idx [M,3], d_min_idx [M] = member closest to the centroid
 * (ties: lowest index).  All outputs are sized for n_points voxels; d_counts[0] = M, d_counts[1] = points outside. */
size_t pnerf_voxel_downsample_workspace_bytes(int64_t n_points, int rx, int ry, int rz);
int pnerf_voxel_downsample(const float *d_xyz, int64_t n_points, const float *space_min3_host, const float *vox_size3_host,
                           int rx, int ry, int rz, float *d_centroid, int32_t *d_grid_idx, int64_t *d_min_idx, int32_t *d_counts,
                           void *d_ws, size_t ws_bytes, void *stream);

/* ---- point initialisation: appearance of candidate points from GIVEN 2-D maps (MvsPointsModel.extract_2d
 * models/mvs/mvs_points_model.py:198-218 = homo_warp_nongrid / homo_warp_nongrid_occ models/mvs/mvs_utils.py:299-315,333-369 +
 * extract_from_2d_grid :411-421; called from query_embedding :233-238).  The maps (the source images and the feature pyramid of the
 * reference's FeatureNet) are INPUTS: the 2-D network itself is outside the hot-path scope.
 * d_cam_xyz [n,3]: the points in the frame of the camera `cam_vid`.  A view = (c2w of cam_vid, w2c of the view, the view's 3x3 intrinsic,
 * all row-major; has_w2c = 0 for the view that IS cam_vid: no transform, :302-305).  Pixel = ((p / p.z) @ K^T).xy; in-image test
 * 0 <= pixel <= (WD-1, HD-1) (depth_occ == 0) or on ceil(pixel) (depth_occ != 0), then for depth_occ != 0 the z-buffer test
 * z <= min(z over the in-image points of the same ceil-pixel) + tolerate.  Every map [C,H,W] of a view is sampled at the pixel scaled to
 * [-1,1] by (WD-1, HD-1) (bilinear, zeros outside, align_corners) into columns [out_col, out_col + C) of d_feats [n,feat_cols] or, with
 * is_color, of d_colors [n,color_cols]; rows of points that fail a view's test are zero in that view's columns.  d_mask [n_views,n]
 * (optional): the views' final masks. */
#define PNERF_EX2D_MAX_VIEWS 8
#define PNERF_EX2D_MAX_MAPS 32
typedef struct pnerf_view_desc {
    float c2w[16];          /* c2ws[:, cam_vid] */
    float w2c[16];          /* w2cs[:, vid] */
    float intrinsic[9];     /* intrinsics[:, vid] */
    int32_t has_w2c;
} pnerf_view_desc;
typedef struct pnerf_map_desc {
    const float *d_map;     /* [C,H,W] device: img_feats[lid][vid] */
    int32_t view;           /* index into the views array */
    int32_t C, H, W;
    int32_t out_col;        /* first column in d_feats (d_colors when is_color) */
    int32_t is_color;       /* layer 0 of the reference's pyramid = the image itself (:209-210) */
    int32_t first_of_view;  /* set by the library */
} pnerf_map_desc;
size_t pnerf_extract_2d_workspace_bytes(int64_t n_points, int n_views, int HD, int WD, int depth_occ);
int pnerf_extract_2d(const float *d_cam_xyz, int64_t n_points, const pnerf_view_desc *views, int n_views, const pnerf_map_desc *maps,
                     int n_maps, int HD, int WD, int depth_occ, float tolerate, float *d_feats, int feat_cols, float *d_colors,
                     int color_cols, uint8_t *d_mask, void *d_ws, size_t ws_bytes, void *stream);
/* the "dir" block of query_embedding (models/mvs/mvs_points_model.py:239-251): d_dirs [n, n_views, 3] = unit(p - cam_pos_cam[v]) (norm + 1e-6)
 * @ rot1^T (@ rot2^T when rot2 != NULL); cam_pos_cam_host [n_views,3] = the views' centres in cam_vid's frame, rot1 = c2ws[cam_vid][:3,:3],
 * rot2 = c2ws[ref_vid][:3,:3] unless pointdir_w. */
int pnerf_point_dirs(const float *d_cam_xyz, int64_t n_points, const float *cam_pos_cam_host, int n_views, const float *rot1_host9,
                     const float *rot2_host9, float *d_dirs, void *stream);

/* ---- diagnostics: ONE v_mfma_f32_32x32x16_f16, D = A B with caller-built fragments: d_a / d_b [64 lanes][8] f16 (lane l holds
 * A[l & 31][8 (l >> 5) .. + 7] resp. B[8 (l >> 5) .. + 7][l & 31]), d_out [64 lanes][16] f32 (register r of lane l =
 * D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31]).  The tests pin with it the fragment layout and the un-flushed handling of f16
 * subnormal inputs that the two-plane GEMMs of the aggregator (csrc/f16x3.h) rely on. */
int pnerf_debug_mfma_f16(const void *d_a, const void *d_b, float *d_out, void *stream);
/* ONE tile GEMM in the mixed format of csrc/mixq.h (f16 h.h + e4m3 cross terms): d_out [64][256] = d_x [64][K] d_w[256][K]^T, K in {256, 272, 288}
 * (columns >= 256 run the classic three f16 products), through the tile's LDS format and a packed weight image written to d_img
 * (>= pnerf_mlp_packed_bytes()).  Tests measure it against float64 on the device and against the numpy restatement of the format on the host
 * emulator.  The arithmetic stands behind the nn.Linear layers of models/aggregators/point_aggregators.py:286-344. */
int pnerf_debug_mix_gemm(const float *d_w, int K, const float *d_x, void *d_img, float *d_out, void *stream);
/* measurement aid of bench.py (roofline.peak_measured of the matrix-pipe entries; no reference counterpart): `iters` x 32 register-resident
 * v_mfma_f32_32x32x16_f16 per wave on 2 x 256-thread workgroups per CU, operands all zero (mode 0), one constant (1) or pseudo-random f16 in
 * +-[0.5, 1) (2: they toggle like a GEMM's fragments).  Asynchronous on `stream`; the caller times it with events.  *flop_out = the flops the
 * launch executes; d_scratch: >= 256 floats of device memory (never written in practice). */
int pnerf_debug_mfma_rate(int mode, int iters, float *d_scratch, double *flop_out, void *stream);
/* the counter-based uniforms of the jittered ray sampling (pnerf_query with jitter > 0 draws U[r * D + d] = uniform(seed, r * D + d)):
 * d_out[i] = uniform(seed, first + i) in [0, 1).  With these the jittered samples are a deterministic function of the inputs and
 * equal the reference's near_far_linear_ray_generation (diff_ray_marching.py:369-385, CPU) fed the same numbers, bit for bit. */
int pnerf_debug_uniform(uint64_t seed, uint64_t first, int64_t n, float *d_out, void *stream);
/* the two-plane split of csrc/f16x3.h on n floats (n even): d_h / d_m [n] f16 (high plane: round toward zero; residual plane: round to
 * nearest of x - h); sat != 0 clamps to the f16 range first (the gradient form).  Tests compare it bit for bit with the numpy restatement. */
int pnerf_debug_split(const float *d_x, int64_t n, void *d_h, void *d_m, int sat, void *stream);
/* the positional encoding the aggregator kernels evaluate (csrc/f16x3.h pn_pe_octaves; reference models/helpers/networks.py:175-190):
 * d_out[i][f] = {sin(x_i 2^f), cos(x_i 2^f)}, f < nfreq <= 5, d_out [n][nfreq][2] f32.  Tests measure it against float64 for |x| up to
 * thousands (trained embeddings are not confined to the initialisation's (-0.5, 0.5)). */
int pnerf_debug_pe(const float *d_x, int64_t n, int nfreq, float *d_out, void *stream);

/* ---- per-kernel timing (HIP events recorded on the launch stream; off by default) --------------- */
int pnerf_prof_enable(int on);
int pnerf_prof_kernel_count(void);
const char *pnerf_prof_kernel_name(int id);
/* synchronous: device is synchronised; totals of every launch recorded since the last collect */
int pnerf_prof_collect(double *total_ms, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* PNERF_H */
