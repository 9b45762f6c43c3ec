/* gsrast.h — C ABI of libgsrast.so, the MI355X (gfx950) differentiable Gaussian rasterizer.
 *
 * This is the drop-in boundary of the hot path.  In the reference the boundary is the pybind11
 * module `diff_gaussian_rasterization._C` of the un-vendored submodule (call site:
 * gaussian_renderer/__init__.py:14,36-51,85-93; contract visible at train.py:106,129 and
 * scene/gaussian_model.py:415-417).  Each entry point below names the `_C` function it replaces.
 *
 * Conventions
 *   - plain C types only; every pointer is a DEVICE pointer unless the name ends in `_host`
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued on it
 *   - optional inputs/outputs: NULL <=> not provided (the reference passes empty tensors)
 *   - return value: 0 = ok, negative = gsr_status; message via gsr_last_error() (thread-local)
 *   - the library holds no global mutable state; it is re-entrant (forward on the Python main
 *     thread, backward on the autograd worker thread)
 *   - all float tensors are contiguous fp32, layouts as at the reference call site:
 *       means3D[P,3] shs[P,M,3] colors_precomp[P,3] opacities[P] scales[P,3] rotations[P,4]
 *       cov3D_precomp[P,6] viewmatrix[4,4] projmatrix[4,4] (both already transposed, row-vector
 *       convention: scene/cameras.py:54-56) campos[3] bg[3] out_color[3,H,W] radii[P] (int32)
 */
#ifndef GSRAST_H
#define GSRAST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_VERSION 12
#define GSR_SCREEN_GRAD_STRIDE 12   /* floats per Gaussian in `screen_grads`: (dmean2D.x, dmean2D.y,
                                       dconic A, B, C, dopacity, drgb[3], 3 pad) */

typedef enum gsr_status {
    GSR_OK = 0,
    GSR_ERR_INVALID_ARGUMENT = -1,
    GSR_ERR_HIP = -2,            /* a HIP runtime call or kernel launch failed */
    GSR_ERR_PREFILTERED = -3,    /* prefiltered=1 but a Gaussian failed the frustum test (A.1) */
    GSR_ERR_WORKSPACE = -4       /* workspace too small for this frame */
} gsr_status;

/* Per-call configuration = the scalar fields of GaussianRasterizationSettings
 * (gaussian_renderer/__init__.py:36-49) plus the tile-row slab of multi-GPU sharding. */
typedef struct gsr_frame_desc {
    int32_t P;               /* number of Gaussians (< 2^28)                                     */
    int32_t sh_degree;       /* active SH degree D, 0..3                                          */
    int32_t sh_coeffs;       /* M = stored coefficients per channel of shs[P,M,3]; 0 with colors */
    int32_t width, height;   /* image_width, image_height                                        */
    float tanfovx, tanfovy, scale_modifier;
    int32_t prefiltered, debug;
    int32_t tile_row_begin;  /* slab [tile_row_begin, tile_row_end) in 16-px tile rows;           */
    int32_t tile_row_end;    /* tile_row_end <= 0 means "to the last row" (whole image: 0, 0)     */
} gsr_frame_desc;

/* Host-side plan of one frame, written by gsr_forward_preprocess and handed to the later stages (the
 * library itself keeps no per-frame state).  The forward is PROGRESSIVE: visible Gaussians are sorted by
 * depth once, split into up to GSR_MAX_CHUNKS depth chunks, and each chunk is binned and blended only
 * into tiles that still have an unsaturated pixel; a tile whose 256 pixels have all hit the
 * transmittance cut-off (A.8) takes no further instances.  Pixels are identical to binning everything. */
#define GSR_MAX_CHUNKS 8
typedef struct gsr_frame_plan {
    int64_t num_rendered;                         /* R: sum over Gaussians of the tiles of their binning
                                                     rectangle (the reference's rectangle clipped to the slab and to
                                                     the alpha >= 1/255 bounding box): upper bound of the instances
                                                     emitted; <= the reference's num_rendered                */
    int32_t num_visible;                          /* V: Gaussians with radius > 0                           */
    int32_t num_chunks;                           /* depth chunks planned (1..GSR_MAX_CHUNKS)               */
    int32_t chunk_rank_begin[GSR_MAX_CHUNKS + 1]; /* chunk c = positions [begin[c], begin[c+1]) of the depth order */
    int64_t chunk_instances_max[GSR_MAX_CHUNKS];  /* instances of chunk c if every tile were open           */
    int32_t chunks_run;                           /* out of gsr_forward_render: chunks actually processed   */
    int32_t sort_result;                          /* out of gsr_forward_render: radix buffer holding lists  */
    int64_t instances_emitted;                    /* out of gsr_forward_render: instances actually binned;
                                                     -1 when the last chunk ran (its count is not read back)  */
    int32_t binning_initialised;                  /* gsr_forward_preprocess was given the image workspace and has
                                                     already reset the tile ranges / open flags in it        */
    int32_t screen_prezeroed;                     /* 1 (set by gsr_backward_prepare): screen_grads is already all zero; 2 (set by
                                                     the caller): only the rows of the binned depth prefix will be read (by the
                                                     sparse gsr_backward_geom of this frame) - no clearing at all            */
    int64_t binning_capacity;                     /* IN to gsr_forward_render (and the backward): instances the binning
                                                     workspace was sized for (gsr_binning_size of that number); 0 = R   */
    uint32_t chunk_key_end[GSR_MAX_CHUNKS];       /* chunk c = visible Gaussians whose depth bits lie in (key_end[c-1], key_end[c]]:
                                                     the chunks are SELECTED by depth; each is sorted when it is binned  */
    int32_t chunks_sorted;                        /* chunks [0, chunks_sorted) of the depth order are already sorted: a re-run of
                                                     gsr_forward_render (after GSR_ERR_WORKSPACE) does not sort them again   */
    int32_t chunks_filtered;                      /* bit c: chunk c was put through the live filter (only the Gaussians that can still
                                                     reach an open tile sit at the front of its range, sorted and binned)        */
    int32_t tile_order_ready;                     /* set by gsr_forward when its zero fill also cleared the blend backward's row-valid
                                                     flags (gsr_backward_render then skips that memset); cleared by
                                                     gsr_forward_render                                                         */
    uint32_t key_max;                             /* no visible Gaussian's depth key exceeds it: the last chunk's sort skips the bits above */
} gsr_frame_plan;

typedef struct gsr_camera {      /* tensor fields of GaussianRasterizationSettings (device) */
    const float *bg, *viewmatrix, *projmatrix, *campos;
} gsr_camera;

typedef struct gsr_gaussians {   /* arguments of GaussianRasterizer.forward (device) */
    const float *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
    /* SURVEY 8a row a14, optional: raw != 0 hands over the optimizer's RAW parameters of the reference's
     * GaussianModel and the activations of scene/gaussian_model.py:47-60,108-127 run inside the per-Gaussian
     * kernels instead of as ~30 torch kernels around them:
     *   opacities = _opacity logits        -> sigmoid        scales    = _scaling log-scales -> exp
     *   rotations = _rotation quaternions  -> normalize      shs       = _features_dc   [P,1,3]
     *   shs_rest  = _features_rest [P, sh_coeffs - 1, 3] (the torch.cat of get_features); NULL iff sh_coeffs == 1
     * raw == 2: the same activations, but the SH coefficients are ONE interleaved table, shs = [P, sh_coeffs, 3] exactly as in the
     * activated case (shs_rest NULL; grads.shs is then [P, sh_coeffs, 3] too): a parameter store that keeps get_features as a
     * single leaf needs no cat and no split.
     * colors_precomp and cov3D_precomp must be NULL.  The backward then returns gradients w.r.t. the raw
     * tensors (chain rule through the activations fused into gsr_backward_geom). */
    const float *shs_rest;
    int32_t raw;
} gsr_gaussians;

typedef struct gsr_grads {       /* outputs of the backward; any may be NULL (not wanted) */
    float *means3D;        /* [P,3]   */
    float *means2D;        /* [P,3]   (x, y) = dL/d(NDC position) incl. W/2, H/2; z = 0            */
    float *shs;            /* [P,M,3] */
    float *colors_precomp; /* [P,3]   */
    float *opacities;      /* [P]     */
    float *scales;         /* [P,3]   */
    float *rotations;      /* [P,4]   */
    float *cov3D_precomp;  /* [P,6]   */
    float *shs_rest;       /* [P,M-1,3] raw == 1 only (then shs is [P,1,3]) */
    int32_t prezeroed;     /* != 0: the caller has already zero-filled every non-NULL tensor above (gsr_backward_prepare);
                              the sparse path of gsr_backward_geom then skips its own fill */
} gsr_grads;

int gsr_version(void);
const char *gsr_last_error(void);

/* Sizes of the two frame-sized opaque workspaces (the reference's geomBuffer / imgBuffer, which
 * `_C.rasterize_gaussians` allocates through its resize callback). */
int gsr_workspace_sizes(const gsr_frame_desc *desc, size_t *geom_bytes, size_t *image_bytes);

/* Size of the per-duplicate workspace (the reference's binningBuffer) for `num_rendered` instances: 24 bytes each.
 * The reference sizes it for R = plan->num_rendered.  Here only what the depth chunks actually emit is touched (2-8 % of R
 * on frames whose tiles saturate), so a caller may size it for gsr_binning_first_chunk_capacity() instances, set
 * plan->binning_capacity to that number, and fall back to R when gsr_forward_render answers GSR_ERR_WORKSPACE (a later
 * chunk did not fit; nothing of it was written: re-run gsr_forward_render with the larger workspace). */
int gsr_binning_size(const gsr_frame_desc *desc, int64_t num_rendered, size_t *binning_bytes);
int gsr_binning_first_chunk_capacity(const gsr_frame_plan *plan_host, int64_t *instances);

/* Stage 1 of `_C.rasterize_gaussians`: per-Gaussian preprocess (cull, project, EWA covariance,
 * SH colour), depth sort of the Gaussians, prefix sum of tiles touched in depth order, chunk plan.
 * Writes radii[P] and *plan_host (one host synchronisation; plan->num_rendered sizes the binning
 * workspace).  image_ws (optional, may be NULL): when given, the per-frame reset of the tile ranges and open
 * flags that stage 2 needs is enqueued here, BEFORE the host waits for the plan, so that it runs in the
 * shadow of the readback. */
int gsr_forward_preprocess(const gsr_frame_desc *desc, const gsr_camera *cam, const gsr_gaussians *g,
                           void *geom_ws, void *image_ws, int32_t *radii, gsr_frame_plan *plan_host, void *stream);

/* Stage 2 of `_C.rasterize_gaussians`: per depth chunk — emit (tile, instance) pairs into open tiles,
 * stable radix sort by tile, tile ranges, per-tile front-to-back blend continuing each pixel's state.
 * Stops as soon as no tile is open (one 4-byte readback per chunk).  Updates plan_host->chunks_run /
 * instances_emitted.  Rows of out_color outside the slab are left untouched.  `g` = the same tensors as in stage 1:
 * SH colours (A.6) are evaluated HERE, per chunk, only for the Gaussians of the chunks that get binned. */
int gsr_forward_render(const gsr_frame_desc *desc, const gsr_camera *cam, const gsr_gaussians *g, void *geom_ws,
                       void *binning_ws, void *image_ws, gsr_frame_plan *plan_host, float *out_color, void *stream);

/* Both stages in ONE call, for callers that bring a binning workspace sized from a guess (the last frame of the same shape):
 * gsr_forward_preprocess, then — without going back to the caller, whose own code between the two calls (an interpreter, a
 * framework's dispatcher) costs tens of microseconds of idle stream right after the plan readback — gsr_forward_render into
 * `binning_ws` if its `binning_capacity` instances cover what gsr_binning_first_chunk_capacity asks for.  If they do not:
 * GSR_ERR_WORKSPACE with the plan filled in and nothing of stage 2 enqueued; allocate and call gsr_forward_render.
 * early_fill (optional): the backward's outputs, as for gsr_backward_prepare (grads->..., any NULL tensor is skipped; the
 * screen-space tensor is not part of it: see gsr_frame_plan.screen_prezeroed = 2).  When the frame ends sparse
 * (chunk_rank_begin[chunks_run] * 4 < P) their zero fill is enqueued here, right after the last readback of stage 2, instead of
 * tens of microseconds later from the caller: early_fill->prezeroed is set and gsr_backward_geom skips its own fill. */
int gsr_forward(const gsr_frame_desc *desc, const gsr_camera *cam, const gsr_gaussians *g, void *geom_ws, void *image_ws,
                int32_t *radii, gsr_frame_plan *plan_host, void *binning_ws, int64_t binning_capacity, float *out_color,
                gsr_grads *early_fill, void *stream);

/* Size of the backward-only scratch: one 48-byte gradient row and one "row written" byte per instance the forward EMITTED
 * (plan_host->instances_emitted, a few per cent of num_rendered; the emission bound of the chunks that ran
 * when the forward went through its last chunk), so it is allocated when the backward runs and freed right
 * after it. */
int gsr_backward_rows_size(const gsr_frame_desc *desc, const gsr_frame_plan *plan_host, size_t *rows_bytes);

/* Optional: zero-fill the backward's outputs AHEAD of time, in one launch — typically right after
 * gsr_forward_render returns, when the stream is idle while the host walks back through the caller's code to the
 * loss.  screen_grads [P, GSR_SCREEN_GRAD_STRIDE] (may be NULL) and every non-NULL tensor of `grads` are
 * cleared; plan_host->screen_prezeroed and grads->prezeroed are set so that gsr_backward_render /
 * gsr_backward_geom skip their own fills.  Only worth calling when the sparse geometry backward will run
 * (plan->chunk_rank_begin[plan->chunks_run] * 4 < P). */
int gsr_backward_prepare(const gsr_frame_desc *desc, const gsr_gaussians *g, gsr_frame_plan *plan_host, float *screen_grads,
                         gsr_grads *grads, void *stream);

/* First half of `_C.rasterize_gaussians_backward`: the blend's backward over the slab's tiles into rows_ws and the
 * deterministic per-Gaussian reduction -> screen_grads[P, GSR_SCREEN_GRAD_STRIDE].  out_color = the image the forward
 * of this frame wrote ([3,H,W], unchanged since: the backward walks the lists front to back and needs the final pixel). */
int gsr_backward_render(const gsr_frame_desc *desc, const gsr_camera *cam, const void *geom_ws, void *binning_ws,
                        const void *image_ws, void *rows_ws, const gsr_frame_plan *plan_host, const float *out_color,
                        const float *dL_dcolor, float *screen_grads, void *stream);

/* List entries per work unit of the blend backward (a build constant: binning workspaces hold one 4 KB checkpoint per that
 * many instances). */
int gsr_bwd_segment_entries(void);

/* Second half of `_C.rasterize_gaussians_backward`: per-Gaussian backward (2D covariance, projection,
 * SH, 3D covariance) for Gaussians [g_begin, g_end).  Every non-NULL output row in that range is
 * written (zeros for invisible Gaussians and for SH coefficients above the active degree).
 * binned_ranks: a HINT — the number of leading depth ranks outside which `screen_grads` is known to be all
 * zero (plan->chunk_rank_begin[plan->chunks_run] for gradients that come from THIS frame's
 * gsr_backward_render, or the maximum over ranks after a multi-GPU sum), or -1 for "unknown".  When the whole
 * range is requested and the prefix is short the outputs are memset and only the prefix is visited.
 * own_plan (may be NULL): the frame's plan, when screen_grads come from gsr_backward_render of this very frame (and only then):
 * replaces the hint — a rank that emitted no instance is skipped before anything of it is read, and a chunk that went through the
 * live filter counts as nothing when choosing between "memset + visit" and the dense pass (a training frame whose last chunk is
 * most of the scene binned a few per cent of it). */
int gsr_backward_geom(const gsr_frame_desc *desc, const gsr_camera *cam, const gsr_gaussians *g,
                      const int32_t *radii, const void *geom_ws, const float *screen_grads, int32_t g_begin,
                      int32_t g_end, int32_t binned_ranks, const gsr_frame_plan *own_plan, const gsr_grads *out, void *stream);

/* gsr_backward_geom for an explicit list of Gaussians: the sparse geometry backward visits rows[0 .. n_rows) (device,
 * Gaussian indices) instead of the frame's own binned depth prefix; every other row of the outputs must already be zero
 * (gsr_backward_prepare) or is cleared first.  n_rows * 4 >= P runs the dense kernel over all P.  This is what a rank of
 * a tile-row-sharded render calls after the ranks have summed their screen-space gradients: the list is then the union
 * of what the ranks binned (all Gaussians whose depth key is <= the largest chunk end any rank reached). */
int gsr_backward_geom_rows(const gsr_frame_desc *desc, const gsr_camera *cam, const gsr_gaussians *g, const int32_t *radii,
                           const void *geom_ws, const float *screen_grads, const int32_t *rows, int32_t n_rows, const gsr_grads *out,
                           void *stream);

/* Device pointers into a frame's geometry workspace (valid after gsr_forward_preprocess):
 *   depth_keys [P]  by Gaussian: the bits of its view depth (a positive binary32: ordered as an integer), 0xFFFFFFFF = not
 *                   visible.  Identical on every rank of a sharded render (visibility is that of the full image).
 *   depth_order [P] the depth order: plan->chunk_rank_begin[c] .. [c+1] holds chunk c's Gaussians, sorted by (depth, index)
 *                   for the chunks gsr_forward_render has run.
 * Either may be NULL (not wanted). */
int gsr_frame_arrays(const gsr_frame_desc *desc, const void *geom_ws, const uint32_t **depth_keys, const uint32_t **depth_order);

/* Multi-GPU gradient exchange (the "allreduce_screen" mode of sharded.py): after the ranks agreed on the largest
 * chunk end any of them binned (key_max, a depth key; n_rows = how many visible Gaussians have key <= key_max: exact, every
 * rank knows it), gather lists those Gaussians in index order (rows[n_rows]) and packs their 48-byte screen-gradient rows
 * (packed[n_rows, 12]) for ONE all-reduce; scatter writes the summed rows back.  Every rank builds the same list: the keys
 * do not depend on its slab.  Uses the frame's selection scratch inside geom_ws (idle after the forward). */
int gsr_exchange_rows_gather(const gsr_frame_desc *desc, void *geom_ws, uint32_t key_max, const float *screen_grads, int32_t n_rows,
                             int32_t *rows, float *packed, void *stream);
int gsr_exchange_rows_scatter(const gsr_frame_desc *desc, int32_t n_rows, const int32_t *rows, const float *packed, float *screen_grads,
                              void *stream);

/* `_C.mark_visible`: present[i] = 1 iff Gaussian i passes the near-plane test (A.1). */
int gsr_mark_visible(int32_t P, const float *means3D, const float *viewmatrix, const float *projmatrix,
                     uint8_t *present, void *stream);

/* Introspection for tests / profiling: copies of intermediate device arrays' ADDRESSES inside the
 * workspaces (no data is copied).  Any out-pointer may be NULL. */
typedef struct gsr_debug_views {
    const float *splat_records;   /* [P,12]: x, y, conicA, conicB, conicC, opacity, r, g, b, depth, packed tile rect (2) */
    const uint32_t *tiles_touched; /* [P,2] by Gaussian: (tiles touched, optical mass in 1/64 pixel-neper units) */
    const uint32_t *depth_order;   /* [P]  depth rank -> Gaussian (invisible ones last)                   */
    const uint32_t *point_offsets; /* [P]  inclusive scan of tiles touched, in depth order               */
    const uint8_t *clamped;        /* [P]  bit c set <=> channel c clamped */
    const uint32_t *sorted_gaussian; /* [<=R] per binned instance (chunks concatenated): Gaussian index in bits 0..27, in bits
                                        28..31 the mask of the tile's 8x8 quadrants the splat can reach (sub-tile culling) */
    const uint32_t *ranges;        /* [GSR_MAX_CHUNKS, Tn, 2] absolute [start, end) per chunk and tile   */
    const float *final_T;          /* [H*W] (negative sign marks a pixel that hit the cut-off)           */
    const int32_t *n_contrib;      /* [H*W] encoded: (chunk + 1) << 26 | position in that chunk's range  */
    const uint32_t *tile_walk;     /* [GSR_MAX_CHUNKS, Tn] entries of chunk c's range the blend backward walks on a tile (its deepest
                                      contributor there); defined where ranges[c][tile] is not empty      */
    const uint32_t *bwd_units;     /* the blend backward's work units (tile | chunk << 24, segment), appended by the forward's waves as they
                                      finish: per shard (= tile % 8) 17 lists, laid out [8][cap_full + 16 cap_part][2]: full segments, then
                                      the pairs' partial last segments in 16 length classes, longest first */
    const uint32_t *bwd_unit_count; /* [8, 17] units per shard and list */
    uint32_t bwd_unit_cap_full, bwd_unit_cap_part;
} gsr_debug_views;
int gsr_debug_get_views(const gsr_frame_desc *desc, const void *geom_ws, const void *binning_ws,
                        const void *image_ws, const gsr_frame_plan *plan_host, gsr_debug_views *views);

/* ---- the loss of the reference's timed window (train.py:104-105), fused.
 * loss = (1 - lambda) * mean|image - target| + lambda * (1 - mean SSIM(image, target)), SSIM exactly as
 * utils/loss_utils.py:33-63 (11x11 Gaussian window sigma 1.5, depthwise, zero padding, C1 = 1e-4, C2 = 9e-4).
 * image/target: [channels, H, W] fp32.  forward writes out3 = (loss, l1, ssim) (device) and keeps the
 * per-pixel SSIM derivatives in `workspace` for the backward; backward writes grad_image =
 * upstream[0] * dloss/dimage (upstream: device scalar, NULL = 1). */
int gsr_loss_workspace_size(int32_t channels, int32_t height, int32_t width, size_t *bytes);
int gsr_loss_l1_ssim_forward(int32_t channels, int32_t height, int32_t width, float lambda_dssim, const float *image,
                             const float *target, void *workspace, float *out3, void *stream);
int gsr_loss_l1_ssim_backward(int32_t channels, int32_t height, int32_t width, float lambda_dssim, const float *upstream,
                              const float *image, const float *target, const void *workspace, float *grad_image,
                              void *stream);
/* The two terms on their own, for a caller that composes the loss itself exactly as train.py:104-105 does
 * (Ll1 = l1_loss(image, gt); loss = (1 - lambda) * Ll1 + lambda * (1 - ssim(image, gt))): gsr_loss_l1_ssim_forward's
 * out3[1] / out3[2] ARE l1_loss / ssim of utils/loss_utils.py:17-18,33-63; d ssim / d image is gsr_loss_l1_ssim_backward
 * with lambda_dssim = 1 and the upstream negated; d l1_loss / d image = upstream[0] * sign(image - target) / n is this: */
int gsr_loss_l1_backward(int64_t n, const float *upstream, const float *image, const float *target, float *grad_image,
                         void *stream);
/* Multi-GPU slab variants (SURVEY 8e "slab-local loss"): this rank owns image rows [row_begin, row_end) of a
 * full-size image whose rows within 10 of the slab are valid.  forward_rows writes out2 = (sum |image - target|,
 * sum of the SSIM map) over the slab's rows — un-normalised: the caller adds the ranks' pairs and forms
 * loss = (1 - lambda) * l1 / n + lambda * (1 - ssim / n), n = channels * height * width — and the derivative
 * maps of the slab plus 5 rows either side.  backward_rows writes grad_image rows [row_begin, row_end) only:
 * the full d loss / d image of those rows (window contributions from the neighbouring slabs included). */
int gsr_loss_l1_ssim_forward_rows(int32_t channels, int32_t height, int32_t width, const float *image, const float *target,
                                  void *workspace, float *out2, int32_t row_begin, int32_t row_end, void *stream);
int gsr_loss_l1_ssim_backward_rows(int32_t channels, int32_t height, int32_t width, float lambda_dssim, const float *upstream,
                                   const float *image, const float *target, const void *workspace, float *grad_image,
                                   int32_t row_begin, int32_t row_end, void *stream);

/* ---- `simple_knn._C.distCUDA2` (scene/gaussian_model.py:144-145): mean of the squared
 * distances from each point to its 3 nearest other points (exact), used once to initialise the scales.
 * xyz[P,3] -> mean_dist2[P]. */
int gsr_dist2_workspace_size(int32_t P, size_t *bytes);
int gsr_dist2_knn3(int32_t P, const float *xyz, float *mean_dist2, void *workspace, void *stream);

/* ---- one torch.optim.Adam step (no weight decay / amsgrad; the reference uses eps = 1e-15,
 * scene/gaussian_model.py:173) over n contiguous fp32 elements in a single pass.  step = 1 for the first update. */
int gsr_adam_step(int64_t n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float lr, float beta1,
                  float beta2, float eps, int64_t step, void *stream);
/* The same step over `rows` rows of `row_len` >= 4 contiguous elements whose first `split` elements use
 * lr_head and the rest lr_tail: the interleaved SH table [P, M, 3] (row_len = 3 M, split = 3), which stands for the
 * reference's two tensors _features_dc (lr = feature_lr) and _features_rest (lr = feature_lr / 20,
 * scene/gaussian_model.py:166-168).  Element for element the result equals two gsr_adam_step calls on the two tensors. */
int gsr_adam_step_split(int64_t rows, int32_t row_len, int32_t split, float *param, const float *grad, float *exp_avg,
                        float *exp_avg_sq, float lr_head, float lr_tail, float beta1, float beta2, float eps, int64_t step,
                        void *stream);

/* The whole optimizer step in ONE launch: up to GSR_ADAM_MAX_TENSORS tensors, each with its own learning rate(s) and step
 * count (the reference's six groups: five tensors here, the SH table with split rows).  tensors: HOST array.  Element for
 * element what the per-tensor calls compute. */
#define GSR_ADAM_MAX_TENSORS 8
typedef struct gsr_adam_tensor {
    float *param; const float *grad; float *exp_avg; float *exp_avg_sq;
    int64_t n;                  /* elements */
    float lr, lr_tail;          /* lr_tail: for elements [split, row_len) of every row (row_len > 0) */
    int64_t step;               /* this tensor's step count (>= 1), for the bias corrections */
    int32_t row_len, split;     /* row_len == 0: one learning rate for the whole tensor */
} gsr_adam_tensor;
int gsr_adam_step_multi(int32_t count, const gsr_adam_tensor *tensors, float beta1, float beta2, float eps, void *stream);

/* ---- the activations between the optimizer's raw parameters and the rasterizer's inputs, the
 * getters of scene/gaussian_model.py:101-125 (setup_functions :47-60):
 *   scales [P,3] = exp(scaling_raw)   rotations [P,4] = rotation_raw / max(|rotation_raw|_2, 1e-12)   opacities [P] = sigmoid(opacity_raw)
 * in one launch; a NULL input skips that tensor (its output must be NULL too).  Pointers 16-byte aligned. */
int gsr_activations_forward(int64_t P, const float *scaling_raw, const float *rotation_raw, const float *opacity_raw, float *scales,
                            float *rotations, float *opacities, void *stream);
/* Their backward in one launch: scales / opacities are the FORWARD OUTPUTS, rotation_raw the forward input.  A NULL
 * dL_d*_raw output skips that tensor. */
int gsr_activations_backward(int64_t P, const float *scales, const float *rotation_raw, const float *opacities,
                             const float *dL_dscales, const float *dL_drotations, const float *dL_dopacities,
                             float *dL_dscaling_raw, float *dL_drotation_raw, float *dL_dopacity_raw, void *stream);

/* ---- the densification bookkeeping of one training iteration (train.py:127-130,
 * scene/gaussian_model.py:415-417) in one pass without a host synchronisation: for every Gaussian with
 * radii[i] > 0:  max_radii2D[i] = max(max_radii2D[i], radii[i]);  xyz_gradient_accum[i] += |viewspace_grad[i, :2]|;
 * denom[i] += 1.  viewspace_grad is [P,3] (the means2D gradient of the rasterizer). */
int gsr_densify_stats(int32_t P, const int32_t *radii, const float *viewspace_grad, float *max_radii2D,
                      float *xyz_gradient_accum, float *denom, void *stream);

/* Test hook for the hand-written radix sort (csrc/gsr_sort.hip): stable sort of n (key, value) u32 pairs on
 * key bits [0, end_bit).  keys0/vals0 hold the input; *result_buffer says which pair of buffers holds the
 * output.  count_on_device != 0 reads n from a device word (as the progressive binning does). */
int gsr_debug_sort_temp_bytes(size_t *bytes);
int gsr_debug_sort_pairs(uint32_t *keys0, uint32_t *keys1, uint32_t *vals0, uint32_t *vals1, int64_t n, int32_t end_bit,
                         int32_t count_on_device, void *temp, int32_t *result_buffer, void *stream);

/* Per-kernel device timing (hipEvent pairs recorded on the caller's stream around every kernel this
 * library launches, from any thread).  Off by default; the only process-wide state of the library,
 * mutex-protected, meant for benchmarks (one frame in flight).  gsr_profile_enable(1) resets the
 * accumulators.  gsr_profile_read synchronises the recorded events and returns up to `max_entries`
 * (name, total milliseconds, launch count) rows; returns the number of rows written. */
#define GSR_PROFILE_NAME_LEN 32
int gsr_profile_enable(int enable);
int gsr_profile_read(int max_entries, char (*names)[GSR_PROFILE_NAME_LEN], float *total_ms, int32_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* GSRAST_H */
