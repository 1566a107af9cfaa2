/* ags_raster.h — C ABI of libags_raster.so, the MI355X (gfx950) Gaussian-surfel rasterizer.
 *
 * This is the drop-in boundary for ActiveGS's hot path.  The reference binds the path
 * through ONE Python import and ONE call site:
 *   /root/reference/utils/operations.py:22-25   from diff_gaussian_rasterization_2d import
 *                                               GaussianRasterizationSettings, GaussianRasterizer
 *   /root/reference/utils/operations.py:682-713 settings construction + rasterizer(**9 kwargs)
 *   /root/reference/mapping/gaussian_map.py:125 total_loss.backward()  (-> extension backward)
 *   /root/reference/mapping/gaussian_map.py:126 optimizer.step()       (torch.optim.Adam, :259-292)
 * The extension behind that import (pip git+https://github.com/liren-jin/diff-gaussian-rasterization_2d,
 * /root/reference/envs/requirements.txt:15) is CUDA-only and not vendored; this library replaces it.
 * The Python module of the same name (diff_gaussian_rasterization_2d/ in this repo) binds these
 * entry points with ctypes — see INTEGRATION.md.
 *
 * Conventions
 *  - every pointer inside the structs is a DEVICE pointer (HBM) unless stated; the structs
 *    themselves are host memory and are read during the call only;
 *  - all float arrays are contiguous row-major fp32; images are planar (C,H,W);
 *  - matrices are 4x4 row-major in the row-vector convention of operations.py:759-762
 *    (p_view = [x y z 1] * viewmatrix, p_hom = [x y z 1] * projmatrix);
 *  - quaternions are (w,x,y,z), used un-normalised like operations.py:261-278;
 *  - the library never allocates, frees or synchronises: all device memory (including the
 *    workspace) is owned by the caller, every kernel is enqueued on `stream`, and the calls
 *    are hipGraph-capturable.  The instance count is never read back: if the caller's
 *    workspace is too small for the tile instances of a view, the kernels clamp, raise
 *    AgsStatus.overflow on the device and the images are invalid (re-run with more room);
 *  - re-entrant and stateless: one process per GPU, any number of streams;
 *  - errors are return codes (0 = ok, <0 = AGS_E_*), never exceptions.
 */
#ifndef AGS_RASTER_H
#define AGS_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGS_OK 0
#define AGS_E_INVALID (-1)   /* null pointer / non-positive size */
#define AGS_E_WORKSPACE (-2) /* workspace smaller than ags_workspace_bytes() */
#define AGS_E_LAUNCH (-3)    /* hipGetLastError() != hipSuccess after enqueue */

typedef void* ags_stream_t; /* hipStream_t */

/* Replaces GaussianRasterizationSettings (operations.py:682-700). */
typedef struct AgsCamera {
    int32_t image_height;    /* image_height */
    int32_t image_width;     /* image_width  */
    float tanfovx;           /* tanfovx */
    float tanfovy;           /* tanfovy */
    float scale_modifier;    /* scale_modifier */
    float weight_thres;      /* weight_thres: count_i counts pixels with blend weight > this */
    int32_t normalize_depth; /* config[1] */
    int32_t perpix_depth;    /* config[2] */
    int32_t want_stats;      /* config[3]: fill importance / count (1), or AGS_STATS_SEEN */
    int32_t front_only;      /* config[4] */
    const float* viewmatrix; /* viewmatrix, 16 floats */
    const float* projmatrix; /* projmatrix, 16 floats */
    const float* bg;         /* bg, >= 3 floats (the reference passes 4) */
    const float* render_mask;/* render_mask: H*W floats {0,1}, or NULL when unused */
    /* Optional (NULL = off): `config` itself, the DEVICE tensor of 5 floats the reference builds per call
     * (operations.py:697-699: [1, 1, 1, importance?, front_only?]).  When given, the kernels read
     * normalize_depth = config[1] > 0, perpix_depth = config[2] > 0, want_stats = config[3] > 0 and
     * front_only = config[4] > 0 ON THE DEVICE and the four ints above are ignored - the caller does not have to
     * read the tensor back (a stream synchronisation per view).  In this mode per_gaussian->importance / count must
     * always be given and need NOT be zero-filled by the caller: the per-Gaussian kernel clears them. */
    const float* config;
} AgsCamera;

/* Replaces the tensor kwargs of GaussianRasterizer.__call__ (operations.py:703-713).
 * means2D is a gradient placeholder in the reference (always zeros) and is not read. */
typedef struct AgsGaussians {
    int32_t n;
    const float* means3D;     /* (n,3) */
    const float* scales;      /* (n,3) */
    const float* rotations;   /* (n,4) */
    const float* opacities;   /* (n)   */
    const float* colors;      /* (n,3) colors_precomp */
    const float* confidences; /* (n)   */
    /* raw_params != 0: scales/rotations/opacities hold the RAW map parameters and the
     * activations of gaussian_map.py:529-549 (clamp(scale_factor*exp(s),0,max_scale),
     * normalize(q), sigmoid(o)) are applied inside the per-Gaussian kernels; ags_backward then
     * returns gradients wrt the raw parameters. 0 (the drop-in module): activated values. */
    int32_t raw_params;
    float scale_factor;       /* 0.01 */
    float max_scale;          /* 0.05 */
} AgsGaussians;

/* The five image outputs of the 8-tuple (operations.py:703). */
typedef struct AgsImages {
    float* rgb;        /* (3,H,W) */
    float* normal;     /* (3,H,W) un-normalised sum(w n) */
    float* depth;      /* (1,H,W) */
    float* opacity;    /* (1,H,W) */
    float* confidence; /* (1,H,W) */
} AgsImages;

/* Optional sticky row set (all NULL = off) for optimisation loops that see a small part of the
 * map: the reference re-creates its optimiser for every train() call (gaussian_map.py:259-292),
 * so a surfel that no view of the call has shown yet has gradient, exp_avg and exp_avg_sq all
 * exactly zero and torch's dense Adam update moves it by exactly 0 - skipping it is lossless.
 * ags_forward appends every surfel that passes the cull and is not yet a member (so the set
 * only grows); ags_backward and ags_adam_step* then launch work for the listed rows only.
 * The caller zero-fills member, rows and count, the gradient slab and the Adam moments together
 * when it (re)creates the optimiser state.  Several views may insert concurrently from different
 * streams (insertion is atomic). */
typedef struct AgsRowSet {
    int32_t* member; /* (n) 0 / 1 */
    int32_t* rows;   /* (n) member rows in insertion order */
    int32_t* count;  /* (1) number of member rows */
} AgsRowSet;

/* The three per-Gaussian outputs of the 8-tuple. importance/count are written only when
 * want_stats != 0 (they must be zero-filled by the caller before the call).
 * want_stats == AGS_STATS_SEEN: only count is touched (importance may be NULL) and only as a flag - count[i] = 1 for
 * every surfel with at least one pixel whose blend weight exceeds weight_thres (what the full count would make >= 1):
 * all that post_processing reads of its count render (gaussian_map.py:193-194,228-231: `counts[-1] >= 1`,
 * `sum(counts) >= 1`), at a third of the blend kernel's time.  Ignored with AgsCamera.config. */
#define AGS_STATS_SEEN 2
typedef struct AgsPerGaussian {
    float* importance; /* (n) */
    int32_t* count;    /* (n) */
    int32_t* radii;    /* (n) */
    AgsRowSet touched; /* optional: ags_forward inserts the visible surfels */
} AgsPerGaussian;

/* Incoming image gradients of the backward pass (any may be NULL = zeros). */
typedef struct AgsImageGrads {
    const float* d_rgb;
    const float* d_normal;
    const float* d_depth;
    const float* d_opacity;
    const float* d_confidence;
} AgsImageGrads;

struct AgsAdamTensors;

/* Gradients wrt the tensor kwargs; d_means2D is (n,3) with z = 0 and may be NULL. */
typedef struct AgsGaussianGrads {
    float* d_means3D;
    float* d_scales;
    float* d_rotations;
    float* d_opacities;
    float* d_colors;
    float* d_means2D;
    int32_t accumulate; /* 0: overwrite; 1: += (views one after another on a stream, no extra pass);
                         * 2: atomic += into a pre-zeroed slab (views running concurrently on several streams) */
    /* Optional (NULL = off): an Adam device clock (see ags_adam_step_device) that this backward
     * launch advances on the side - pass it with the LAST view of an optimisation step and call
     * ags_adam_step_device(..., pre_ticked = 1): the step then needs no separate clock kernel. */
    void* adam_clock;
    float adam_lr[5];
    float adam_beta1, adam_beta2;
    /* Optional: only the rows of this set are read or written (rows outside it keep whatever the
     * slab holds - zeros, by the AgsRowSet contract).  Must contain every surfel the view shows,
     * i.e. be the set that was passed to this view's ags_forward. */
    AgsRowSet touched;
    /* Optional (NULL = off): fold the optimiser step into this launch.  Give it with the LAST view
     * of a single-GPU optimisation step, together with `touched` and `adam_clock`: after adding
     * this view's gradients the per-Gaussian kernel applies ags_adam_step_device's update to every
     * member row while its gradient is still in registers (fused_adam->grad is ignored, the slab
     * above is still written - unless all five d_* pointers are NULL, which is allowed in this mode
     * with accumulate = 0: the gradient then never leaves the registers).  Do not call ags_adam_step*
     * for that step.  Not with accumulate 2. */
    const struct AgsAdamTensors* fused_adam;
    float adam_eps;
    /* Optional (NULL = off): the data-parallel form of the same idea.  Give it with the LAST view a
     * rank renders in a step, together with `touched` (accumulate 0 or 1, no fused_adam): instead of
     * leaving the rows' totals in the gradient arrays the per-Gaussian kernel writes them as the rank's
     * exchange segment (what ags_rows_pack would produce from them, header included) and leaves the
     * gradient rows zeroed - no separate pack launch.  Like ags_rows_pack, rows beyond pack_capacity do not
     * travel and KEEP their totals in the gradient arrays (header word 1 > pack_capacity tells). */
    float* pack_segment;   /* ags_rows_segment_floats(pack_capacity) floats */
    int32_t pack_capacity;
    /* != 0: ags_backward / ags_backward_batch run the blend backward ONLY - the view's per-surfel gradient records stay in
     * its workspace - and ags_backward_rows later turns the records of ALL the step's views into parameter gradients in
     * one launch.  The five d_* pointers are not used by such a call (they may be NULL); adam_clock still works. */
    int32_t defer_rows;
    /* ags_backward_rows WITHOUT a row set (touched all NULL): the rows [row_begin, row_end) of the map - what a data-parallel
     * rank that exchanges the dense gradient slab uses to cut its per-Gaussian backward into row chunks, so that chunk k's
     * all-reduce (on another stream) runs under chunk k + 1's chain rule.  The five gradient arrays are required and
     * OVERWRITTEN for those rows (zeros where no view shows the row; accumulate 1: added to - a rank with more than
     * AGS_MAX_ROW_VIEWS views calls once per group of views); no fused_adam, no pack_segment.  Ignored elsewhere. */
    int32_t row_begin, row_end;
} AgsGaussianGrads;

#define AGS_BIN_TILE_SORT 0 /* tile counting + bucket scatter + per-tile LDS bitonic sort */
#define AGS_BIN_RADIX 1     /* duplicate-with-keys + global stable LSD radix sort */
#define AGS_BIN_DIRECT 2    /* ONE pass over the surfels: every tile owns max_instances / tiles key slots, the per-Gaussian
                             * kernel takes a slot with one returning atomic and writes the key at once (no counting pass,
                             * no scan, no second emission pass); per-tile sort as in AGS_BIN_TILE_SORT.  Overflows when ONE
                             * tile's list exceeds its slots (AgsStatus.needed_instances says what would do).  Same per-tile
                             * (depth, id) order and bit-identical images as the other two modes. */
/* Kernel-selection knobs, handed over WITH the workspace (AgsWorkspace.tuning; NULL or an all-zero struct = the
 * defaults).  The library itself reads no environment variable and keeps no process-wide setting: two callers in one
 * process can run different selections side by side.  Every selection gives the same images; the gradients differ only
 * where stated.  (The Python / torch bindings of this repository fill the struct from AGS_* environment variables for
 * experiments - that is the bindings' business, see INTEGRATION.md.) */
#define AGS_BWD_F32 0        /* blend backward: per-surfel sums on the matrix cores in EXACT f32 (v_mfma_f32_16x16x4_f32),
                              * the arithmetic of the reference's fp32 atomics up to summation order.  Default. */
#define AGS_BWD_BF16_SPLIT 1 /* the same sums on the bf16 matrix pipe from hi/lo splits of both operands (hi.hi + lo.hi +
                              * hi.lo, f32 accumulation; every dropped term < 2^-16 of its product): ~4 % faster step,
                              * gradients move by ~1e-5 relative.  Opt-in. */
#define AGS_BWD_VALU 2       /* no matrix instructions: per-lane sums + transposed wave reduction (exact f32) */
#define AGS_BWD_BF16X3 3     /* the sums on the bf16 matrix pipe from an EXACT three-way split of both operands (8 + 8 + 8
                              * significand bits), the six products >= 2^-16 of the leading one, f32 accumulation: every
                              * multiply-add is within 2^-24 of the exact product - one f32 rounding, as in AGS_BWD_F32. */
typedef struct AgsTuning {
    int32_t bwd_reduce;        /* AGS_BWD_* */
    int32_t render_slots;      /* 0 = by the number of tiles in flight; 1 / 2 / 4 = 8x8 quadrants per wave in the blend kernels */
    int32_t cull_first_min_n;  /* 0 = default (2^20 rows): from this map size up the per-Gaussian stage of raw-parameter maps
                                * culls on the means first; < 0 = never; 1 = always */
    int32_t tile_sort_no_wave; /* != 0: AGS_BIN_DIRECT always sorts with the 256-thread kernel (never one wave per tile) */
    int32_t bucket_no_scan;    /* != 0: AGS_BIN_TILE_SORT always runs the separate tile-scan launch */
    int32_t view_group;        /* batched forward (ags_forward_batch, AGS_BIN_DIRECT): k > 1 = the per-Gaussian stage loads and
                                * activates a row once for k consecutive views (opt-in: -10 % on a planner's hundred small
                                * views, +4 % on the mapper's eleven); 0 / 1 = one view per workgroup.  Same records either way. */
    int32_t reserved[2];       /* 0 */
} AgsTuning;

typedef struct AgsWorkspace {
    void* ptr;    /* device, 256-byte aligned */
    size_t bytes; /* >= ags_workspace_bytes(n,h,w,max_instances) */
    int64_t max_instances; /* capacity in (Gaussian,tile) instances */
    int32_t binning_mode;  /* AGS_BIN_*; all give the same per-tile (depth, id) order */
    const AgsTuning* tuning; /* optional (NULL = defaults), HOST memory, read during the call only */
    /* Optional (both NULL = off; AGS_BIN_DIRECT, ags_forward only): a caller that must know whether a pass fits its
     * workspace BEFORE it hands the images on - the drop-in module, whose callers have no retry - but does not want to wait
     * for the whole pass: right behind the per-Gaussian kernel (which takes the key slots, so it is the kernel that knows)
     * ags_forward enqueues a copy of the status block to `early_status_host` (page-locked host memory, sizeof(AgsStatus))
     * and records `early_status_event` (a hipEvent_t) behind the copy; the tile sort and the blend are queued behind that.
     * The caller waits for the EVENT - by then the rest of the pass is already in the stream - and reads
     * host->early_tile_need: 0 = every tile's list fitted (the pass is good), else the longest list's length: the pass
     * overflowed and tiles x that many key slots would have held it. */
    void* early_status_host;
    void* early_status_event;
} AgsWorkspace;

/* Device-side status block = the first 64 bytes of the workspace. */
typedef struct AgsStatus {
    uint32_t num_instances; /* tile instances the view needs */
    uint32_t num_sorted;    /* min(num_instances, max_instances) */
    uint32_t overflow;      /* 1 when num_instances > max_instances */
    uint32_t num_visible;   /* Gaussians that passed the cull */
    /* STICKY since the last ags_workspace_init (every forward on this workspace updates, none clears):
     * a loop that renders many views without reading the block back after each one reads these once at
     * the end; an overflow in ANY of its passes shows, with the size that would have been enough. */
    uint32_t peak_instances;  /* max of needed_instances over the passes */
    uint32_t overflow_passes; /* number of passes that overflowed the workspace */
    /* per pass again: */
    uint32_t max_tile_instances; /* longest tile list of the view (0 in AGS_BIN_RADIX mode) */
    uint32_t needed_instances;   /* the max_instances that would have held this view in this binning mode:
                                  * num_instances, or tiles * max_tile_instances for AGS_BIN_DIRECT */
    uint32_t reserved0;          /* (internal: the software-pipelined step's member count) */
    uint32_t early_tile_need;    /* AGS_BIN_DIRECT, valid BEHIND THE PER-GAUSSIAN KERNEL of a pass (AgsWorkspace.early_status_host)
                                  * and cleared again by the pass's blend kernel: 0, or the longest tile list when a list
                                  * outgrew its tile's max_instances / tiles key slots */
    uint32_t reserved[6];
} AgsStatus;

/* Bytes of workspace for n Gaussians, an h x w image and room for max_instances instances.
 * The forward pass leaves its state there; the backward pass of the same view reads it. */
size_t ags_workspace_bytes(int32_t n, int32_t h, int32_t w, int64_t max_instances);

/* Where the forward pass leaves the parts of its state a caller may want to look at (diagnostics, parity tests):
 * byte offset and size of a region inside a workspace of this geometry.  Returns AGS_E_INVALID for an unknown region.
 *   FINAL_T    (H*W) float    transmittance left behind each pixel (= 1 - the opacity image)
 *   N_CONTRIB  (H*W) uint32   1-based position, in the pixel's tile list, of the last surfel the pixel blended (0: none)
 *   GEOM       (n) x 16 float the projected surfel records
 *   RANGES     (tiles) x 2 uint32   AGS_BIN_TILE_SORT / AGS_BIN_RADIX: [begin, end) of tile t's list in the id array;
 *                                   AGS_BIN_DIRECT: {tile, count} of SLOT s, whose list starts at s * (max_instances / tiles)
 *   KEYS       sorted (depth_bits << 32 | id) keys: AGS_BIN_TILE_SORT (list order = RANGES); AGS_BIN_DIRECT (slot order)
 *   IDS        AGS_BIN_RADIX: the sorted ids (uint32) */
#define AGS_REGION_FINAL_T 0
#define AGS_REGION_N_CONTRIB 1
#define AGS_REGION_GEOM 2
#define AGS_REGION_RANGES 3
#define AGS_REGION_KEYS 4
#define AGS_REGION_IDS 5
int ags_workspace_region(int32_t n, int32_t h, int32_t w, int64_t max_instances, int32_t binning_mode, int32_t region,
                         size_t* offset, size_t* bytes);

/* Must be called ONCE on a freshly allocated (or otherwise scribbled-on) workspace before its
 * first ags_forward: clears the counters the forward pass relies on.  Every ags_forward leaves
 * them clean again (each tile's workgroup resets its own counters), so there is no per-pass
 * memset in the default binning mode. */
int ags_workspace_init(const AgsWorkspace* ws, int32_t n, int32_t h, int32_t w, ags_stream_t stream);

/* The same for the `views` consecutive per-view workspaces of a batch (ags_forward_batch's layout: view v at
 * ptr + v * ags_workspace_bytes(n, h, w, max_instances); ws->bytes covers all of them) in one call. */
int ags_workspace_init_batch(const AgsWorkspace* ws, int32_t views, int32_t n, int32_t h, int32_t w, ags_stream_t stream);

/* Forgets a per-Gaussian stage that has run into this workspace without the rest of its pass (the prepared pass of
 * ags_backward_fused_next when the caller leaves the pipeline: it has taken key slots and added to the partial sums):
 * clears the binning counters like ags_workspace_init but KEEPS the status block, whose sticky words
 * (AgsStatus.peak_instances / overflow_passes) a loop has not read yet. */
int ags_workspace_discard_pass(const AgsWorkspace* ws, int32_t n, int32_t h, int32_t w, ags_stream_t stream);

/* Forward: cull+project, tile binning, depth sort, per-tile blend. */
int ags_forward(const AgsCamera* cam, const AgsGaussians* in, const AgsImages* out,
                const AgsPerGaussian* per_gaussian, const AgsWorkspace* ws, ags_stream_t stream);

/* Forward-only render of `views` views of ONE size and field of view in a single set of launches
 * (blockIdx.y = view): the planners' utility pass (~100 candidate views at 128x128,
 * /root/reference/planning/confidence.py:24-46) and the prune pass over all keyframes
 * (/root/reference/mapping/gaussian_map.py:149-192).  One small view cannot fill the GPU; a batch can.
 *   cam:  the common fields; viewmatrix / projmatrix point at (views,16) floats, render_mask (if any)
 *         at (views,H,W);
 *   out:  pointers to (views,C,H,W) image batches; per_gaussian: (views,n) batches;
 *   ws:   `views` consecutive workspaces of ags_workspace_bytes(n,h,w,max_instances) each, every one
 *         initialised with ags_workspace_init; view v's status block is at ptr + v * that size.
 * Tile-sort binning only. */
size_t ags_forward_batch_workspace_bytes(int32_t views, int32_t n, int32_t h, int32_t w, int64_t max_instances);
int ags_forward_batch(const AgsCamera* cam, int32_t views, const AgsGaussians* in, const AgsImages* out,
                      const AgsPerGaussian* per_gaussian, const AgsWorkspace* ws, ags_stream_t stream);

/* Backward of the views of an ags_forward_batch (same argument conventions: batched `fwd` images,
 * batched image gradients, `views` consecutive workspaces).  The views' gradients are SUMMED into
 * the single gradient slab `din` with atomics: din->accumulate must be 2 (pre-zeroed slab) and
 * din->fused_adam NULL; din->touched / adam_clock work as in ags_backward. */
int ags_backward_batch(const AgsCamera* cam, int32_t views, const AgsGaussians* in, const AgsImages* fwd,
                       const AgsPerGaussian* per_gaussian, const AgsImageGrads* dout, const AgsGaussianGrads* din,
                       const AgsWorkspace* ws, ags_stream_t stream);

/* Backward of the same view; `fwd` are the images ags_forward wrote. */
int ags_backward(const AgsCamera* cam, const AgsGaussians* in, const AgsImages* fwd,
                 const AgsPerGaussian* per_gaussian, const AgsImageGrads* dout,
                 const AgsGaussianGrads* din, const AgsWorkspace* ws, ags_stream_t stream);

/* The per-Gaussian backward of ALL the views of an optimisation step in ONE launch (views whose ags_backward /
 * ags_backward_batch ran with din->defer_rows).  One lane per member row of din->touched (required): the row's inputs
 * are loaded and activated once, every view that shows the row contributes its chain rule from that view's gradient
 * record (which is re-zeroed), the views' gradients are summed in registers, and the tail is the one of ags_backward's
 * row kernel: din->fused_adam (single rank: the optimiser step, needs adam_clock already advanced by a view's
 * ags_backward), din->pack_segment (the rank's exchange segment), or the gradient arrays (accumulate 0: overwrite, member
 * rows no view shows get zeros).  Against per-view per-Gaussian launches this saves, per extra view, a walk over the member
 * list, a reload of the row's inputs and a read-modify-write of its 56-byte gradient row.  All views: the same `in`, the
 * same image size is NOT required; 1 <= num_views <= AGS_MAX_ROW_VIEWS; every view needs its own workspace (and radii)
 * until this call. */
#define AGS_MAX_ROW_VIEWS 16
typedef struct AgsViewRef {
    const AgsCamera* cam;      /* the view's camera (matrices in place) */
    const int32_t* radii;      /* the view's radii output (n) */
    const AgsWorkspace* ws;    /* the view's workspace */
} AgsViewRef;
int ags_backward_rows(const AgsViewRef* views, int32_t num_views, const AgsGaussians* in, const AgsGaussianGrads* din,
                      ags_stream_t stream);

/* Software-pipelined optimisation step (single view per step, single rank, AGS_BIN_DIRECT): the per-Gaussian kernels
 * of consecutive steps in ONE launch.  The last kernel of step k (chain rule + Adam over the row set's members) and the
 * first kernel of step k + 1 (cull + project + key emission over all surfels) are neighbours in a training loop
 * (gaussian_map.py:77-127: optimizer.step() of one iteration, render of the next) and the second needs the first only
 * row by row, so
 *   ags_backward_fused_next  = ags_backward with din->fused_adam (required, with state_rows; in->raw_params required), whose per-Gaussian
 *                              launch also runs the per-Gaussian stage of the NEXT forward pass: camera `next_cam` into
 *                              workspace `next_ws` (may be `ws` itself: the blend backward is done with it by then) and
 *                              `next_per_gaussian->radii`; `next_per_gaussian->touched` must be din->touched;
 *   ags_forward_resume       = the rest of that forward pass (tile sort + blend) - call it INSTEAD of ags_forward for
 *                              the view that was handed to ags_backward_fused_next, with the same camera contents.
 * The next view's matrices must be in place when ags_backward_fused_next runs.  To leave the pipeline (the prepared
 * pass is not wanted after all) call ags_workspace_discard_pass on `next_ws` before its next ags_forward - the
 * prepared pass has already taken key slots in it.  The next view's camera takes HOST flags (AgsCamera.config must be NULL
 * in both calls: the prepared per-Gaussian stage does not clear importance / count; want_stats with zero-filled arrays
 * works as in ags_forward).  `rows_hint`: about how many rows the row set holds (it lives on the
 * device; 0 = unknown) - sizes the part of the launch that walks the list, any value is correct.  Results are identical
 * to ags_backward + ags_forward. */
int ags_backward_fused_next(const AgsCamera* cam, const AgsGaussians* in, const AgsImages* fwd,
                            const AgsPerGaussian* per_gaussian, const AgsImageGrads* dout, const AgsGaussianGrads* din,
                            const AgsWorkspace* ws, const AgsCamera* next_cam, const AgsPerGaussian* next_per_gaussian,
                            const AgsWorkspace* next_ws, int32_t rows_hint, ags_stream_t stream);
int ags_forward_resume(const AgsCamera* cam, const AgsGaussians* in, const AgsImages* out,
                       const AgsPerGaussian* per_gaussian, const AgsWorkspace* ws, ags_stream_t stream);

/* Copies the status block to host memory: enqueues a D2H copy and waits for the stream. */
int ags_read_status(const AgsWorkspace* ws, AgsStatus* host_out, ags_stream_t stream);
/* The same without the wait: enqueues the copy and returns.  `host_out` should be page-locked host memory (else the
 * runtime may stage and block); the caller finds out that the copy has landed with an event it records behind this
 * call on the same stream.  Lets a per-view caller (the drop-in module) look at a view's overflow flag / capacity
 * need one call LATE instead of stalling the stream after every forward pass. */
int ags_read_status_async(const AgsWorkspace* ws, AgsStatus* host_out, ags_stream_t stream);

/* Fused Adam over the five parameter tensors of gaussian_map.py:259-292
 * (means 3n, scales 3n, rotations 4n, opacities n, harmonics 3n), torch.optim.Adam
 * semantics (no weight decay, no amsgrad); `step` is 1-based. */
typedef struct AgsAdamTensors {
    float* param[5];
    const float* grad[5];
    float* exp_avg[5];
    float* exp_avg_sq[5];
    int64_t numel[5];
    float lr[5];
    AgsRowSet touched; /* optional: update the member rows only (exact, see AgsRowSet) */
    int32_t zero_grad; /* with `touched`: write 0 over every gradient row once it is consumed, so that a slab the
                        * views accumulate into atomically (accumulate = 2) needs no memset per step */
    /* Optional (NULL = off): the optimiser's moments interleaved per surfel instead of in exp_avg / exp_avg_sq (which
     * are then ignored and may be NULL): (n, 28) floats, row i = { exp_avg of the row's 14 parameters in the order
     * means 0-2, scales 3-5, rotation 6-9, opacity 10, harmonics 11-13; exp_avg_sq in the same order }.  The moments
     * are the optimiser's private state (torch.optim.Adam keeps them per tensor, gaussian_map.py:259-292 never looks at
     * them), so their layout is free: one 112-byte piece per row instead of ten 4..16-byte pieces in ten arrays is
     * what the row-set forms of the step want (a member row costs 2 memory sectors of state instead of 10).
     * Needs numel = { 3n, 3n, 4n, n, 3n }. */
    float* state_rows;
} AgsAdamTensors;
int ags_adam_step(const AgsAdamTensors* t, float beta1, float beta2, float eps, int32_t step,
                  ags_stream_t stream);

/* Graph-replayable form: the 1-based step counter lives on the device.  `state` is a
 * caller-owned, zero-initialised 64-byte device buffer { int32 step; float step_size[5];
 * float inv_sqrt_bc2; int32 skipped_steps (see ags_adam_step_gathered); double beta1^step, beta2^step and their
 * values one step earlier }; each call first bumps the counter and refreshes the bias
 * corrections on the device (in double), then runs the same update.  Capturing this call in
 * a hipGraph and replaying it k times performs Adam steps 1..k. */
int ags_adam_step_device(const AgsAdamTensors* t, float beta1, float beta2, float eps, void* state,
                         int32_t pre_ticked /* 1: ags_backward already advanced `state` for this step */,
                         ags_stream_t stream);

/* Row exchange for the view-parallel optimisation step (one rank per GPU, each rendering its own
 * views of /root/reference/mapping/gaussian_map.py:113-125's batch; the reference itself is single
 * process and sums the views' losses before backward()).  Instead of all-reducing the dense
 * gradient slab a rank ships the rows of its AgsRowSet:
 *   segment = 16 floats of header {int32 count, int32 needed, 0...} + capacity records of 16 floats
 *             {14 gradient floats in the order means, scales, rotation, opacity, colour; int32 row; 0}
 * ags_rows_pack   copies the set's rows of the five gradient arrays into `segment` and writes 0
 *                 behind them (needed > capacity: the excess rows are NOT shipped and stay in the gradient
 *                 arrays - the caller sized the segment too small: it adds its own segment back with
 *                 ags_rows_unpack, packs again with a larger capacity and repeats the exchange);
 * ags_rows_unpack adds a received segment into the gradient arrays and appends rows that are new to
 *                 `union_rows` (the set the optimiser then steps over with zero_grad = 1).  Call it
 *                 once per rank's segment, in rank order, on one stream: every rank then forms
 *                 bit-identical sums, so the replicas cannot drift.
 * The collective in between (an all-gather of equally sized segments) is the host's business. */
size_t ags_rows_segment_floats(int32_t capacity);
int ags_rows_pack(const AgsRowSet* rows, float* const grads[5], int32_t capacity, float* segment,
                  ags_stream_t stream);
int ags_rows_unpack(const float* segment, int32_t capacity, float* const grads[5],
                    const AgsRowSet* union_rows, ags_stream_t stream);

/* The same tail in TWO launches for any number of ranks.  `segments` = the all-gather's output (world
 * segments of ags_rows_segment_floats(capacity) floats, rank order); `slot_table` = n x world int32,
 * zero-filled once by the caller and left zeroed by every ags_adam_step_gathered.
 * ags_rows_index          one launch over all segments: slot_table[row][rank] = record + 1, union set built;
 * ags_adam_step_gathered  Adam over the union rows (t->touched; t->grad is not read) with each row's
 *                         gradient summed from the segments in rank order - bit-identical on every
 *                         rank - on the device clock `state` (see ags_adam_step_device).
 *                         If ANY segment's header says its rank holds more rows than `capacity` (that rank
 *                         shipped only part of its gradient) the step is REFUSED on the device: parameters,
 *                         moments and the step counter are left exactly as they were, the slot table is still
 *                         cleaned, and state->skipped_steps (int32 at byte 28) is incremented.  Row sets only
 *                         grow, so every later step is refused too until the caller - who reads that counter
 *                         every few steps - exchanges with a larger capacity and repeats the refused steps. */
int ags_rows_index(const float* segments, int32_t world, int32_t capacity, int32_t* slot_table,
                   const AgsRowSet* union_rows, ags_stream_t stream);
int ags_adam_step_gathered(const struct AgsAdamTensors* t, const float* segments, int32_t world, int32_t capacity,
                           int32_t* slot_table, float beta1, float beta2, float eps, void* state,
                           int32_t pre_ticked, ags_stream_t stream);

/* Activations of /root/reference/mapping/gaussian_map.py:529-549 (get_scales / get_rotations /
 * get_opacities): scales = clamp(scale_factor*exp(raw), 0, max_scale), rotations =
 * raw/max(|raw|,1e-12), opacities = sigmoid(raw).  One lane per Gaussian. */
typedef struct AgsActivation {
    int32_t n;
    float scale_factor;        /* cfg.scale_factor, 0.01 */
    float max_scale;           /* 0.05 */
    const float* raw_scales;   /* (n,3) */
    const float* raw_rotations;/* (n,4) */
    const float* raw_opacities;/* (n)   */
} AgsActivation;
int ags_activate(const AgsActivation* a, float* scales, float* rotations, float* opacities,
                 ags_stream_t stream);
/* Chain rule through the activations, IN PLACE: on entry d_* hold gradients wrt the
 * activated values, on exit wrt the raw parameters (what Adam consumes). */
int ags_activate_backward(const AgsActivation* a, float* d_scales, float* d_rotations,
                          float* d_opacities, ags_stream_t stream);

/* Fused loss head: the facade's post-processing (operations.py:714-718,
 * 172-219) and the loss of gaussian_map.py:106-124 / mapping/utils.py:14-62,120-121, forward and
 * backward, from the images of ags_forward straight to the image gradients ags_backward takes.
 * Per optimisation step: stage 1 for EVERY view of the batch (it also accumulates the
 * visibility count `msum` the reference's broadcast in the consistency term needs; all-reduce
 * msum across ranks under view parallelism), then per view stage 2 -> ags_backward.
 * accum: AGS_LOSS_ACCUM_ROWS rows of `accum_stride` floats, zeroed by the caller per step; workgroups
 * add into different rows (so their atomics do not serialise on one word) and the reader sums the
 * rows. Per row: [0] rgb-L1 sum, [1] depth-L1 sum, [2] consistency sum, [3] TV sum over all views;
 * [4+2v], [5+2v] rgb / depth L1 sums of view v (per-frame error).
 * total loss = w_rgb*a0/(B*3HW) + w_depth*a1/(B*HW) + w_cons*a2/(B*B*HW) + w_tv*a3/(B*4HW). */
#define AGS_LOSS_ACCUM_ROWS 64
typedef struct AgsLossConfig {
    int32_t image_height, image_width;
    float fov_x, fov_y;    /* radians, as GaussianRenderer.fovs (operations.py:756-757) */
    int32_t batch_total;   /* B: views in the batch over ALL ranks */
    float w_rgb, w_depth, w_cons, w_tv; /* 1, 0.8, 0.1, 0.1 */
    float sigma;           /* 0.3 */
    int32_t accum_stride;  /* floats per accumulator row, >= 4 + 2 * (views in the batch) */
    int32_t num_views;     /* 0 / 1: one view per call.  > 1: the call handles that many views at once
                            * (blockIdx.y): every image argument points at a contiguous (views,C,H,W) batch,
                            * `view` is the index of the first one, and stage 1 needs first_view = -1. */
    /* Optional (NULL = off; with num_views > 1): `num_views` int64 on the DEVICE - view v's ground truth is frame
     * gt_frame_index[v] of gt_rgb / gt_depth, which then point at the WHOLE keyframe store (K,3,H,W) / (K,1,H,W): the sampled
     * frames' images are read in place instead of being gathered into batch buffers first (ags_stage_frames with
     * dst_rgb = dst_depth = NULL then stages the matrices and clears msum only). */
    const int64_t* gt_frame_index;
} AgsLossConfig;
int ags_loss_stage1(const AgsLossConfig* cfg, const AgsImages* fwd, const float* gt_rgb, const float* gt_depth,
                    float* n_img /* (3,H,W) */, float* d_rgb, float* d_depth, int32_t* msum /* (H,W) */,
                    float* accum, int32_t view,
                    int32_t first_view /* 1: msum = v; 0: msum += v; -1: atomic += into a zeroed msum (views on several streams) */,
                    ags_stream_t stream);
int ags_loss_stage2(const AgsLossConfig* cfg, const AgsImages* fwd, const float* n_img, const float* gt_depth,
                    const int32_t* msum, float* d_normal, float* d_depth /* += */, float* accum,
                    ags_stream_t stream);

/* Stage 1 of the loss head as the EPILOGUE of the forward blend kernel (the pixel's nine channels are still in registers
 * there): ags_forward_batch followed by ags_loss_stage1(cfg with num_views = views, ..., view 0, first_view -1) in one
 * set of launches - the same n_img / d_rgb / d_depth / msum bit for bit, the loss sums added per wave instead of per
 * 256-pixel block (same totals up to float summation order).  `msum` must be zero on entry (atomic counting), the images
 * in `out` are written as by ags_forward_batch (stage 2 and the blend backward read them).  cam->want_stats and
 * cam->config must be off.  Replaces one full-image launch and its re-read of four images per training iteration
 * (/root/reference/mapping/gaussian_map.py:94-124 between the render and the loss). */
typedef struct AgsLossEpilogue {
    const AgsLossConfig* cfg;   /* image size = the camera's; batch_total, weights, accum_stride, gt_frame_index as for ags_loss_stage1 */
    const float* gt_rgb;        /* (views,3,H,W), or the whole keyframe store with cfg->gt_frame_index */
    const float* gt_depth;      /* (views,1,H,W), or the store */
    float* n_img;               /* (views,3,H,W) out */
    float* d_rgb;               /* (views,3,H,W) out */
    float* d_depth;             /* (views,1,H,W) out */
    int32_t* msum;              /* (H,W), zero on entry */
    float* accum;               /* AGS_LOSS_ACCUM_ROWS x accum_stride */
} AgsLossEpilogue;
int ags_forward_batch_loss(const AgsCamera* cam, int32_t views, const AgsGaussians* in, const AgsImages* out,
                           const AgsPerGaussian* pg, const AgsWorkspace* ws, const AgsLossEpilogue* loss, ags_stream_t stream);

/* The facade's post-processing alone (render_cuda_core, /root/reference/utils/operations.py:714-718, and
 * depth2normal, :172-219), for callers that keep the reference's loss head in torch:
 *   normal_out = normalize(normal_raw, dim 0, eps 1e-12) * (opacity > 1e-2)
 *   d2n_out    = depth2normal(depth, opacity > 1e-2, fov)   - four-neighbour cross products of the back-projected
 *                points, replicate padding, x focal = H / (2 tan(fov_x / 2)), y focal = W / (2 tan(fov_y / 2)) (the
 *                reference's pairing, kept)
 * in one launch.  Images are (C,H,W) float; tanfov_* = tan(fov / 2).  normal_raw and normal_out may both be NULL
 * (depth -> normal only). */
int ags_facade_post(int32_t h, int32_t w, float tanfov_x, float tanfov_y, const float* normal_raw, const float* depth,
                    const float* opacity, float* normal_out, float* d2n_out, ags_stream_t stream);
/* The same for `views` views of one size and field of view in ONE launch (forward-only consumers: the planners' candidate
 * views, /root/reference/planning/confidence.py:24-46): every image argument points at a contiguous (views,C,H,W) batch. */
int ags_facade_post_batch(int32_t views, int32_t h, int32_t w, float tanfov_x, float tanfov_y, const float* normal_raw,
                          const float* depth, const float* opacity, float* normal_out, float* d2n_out, ags_stream_t stream);
/* Its backward in one launch: g_normal / g_d2n are the gradients wrt normal_out / d2n_out (either may be NULL = zero);
 * d_normal_raw (3,H,W) is written (may be NULL), d_depth (H,W) is ADDED to with atomics - zero it first. */
int ags_facade_post_backward(int32_t h, int32_t w, float tanfov_x, float tanfov_y, const float* normal_raw,
                             const float* depth, const float* opacity, const float* g_normal, const float* g_d2n,
                             float* d_normal_raw, float* d_depth, ags_stream_t stream);

/* ---- Map growth and pruning (replaces the torch/cv2 code of GaussianMap.add_gaussians, cal_mask,
 * prune and voxel_downsample: /root/reference/mapping/gaussian_map.py:234-246,294-489,
 * /root/reference/utils/operations.py:161-169,603-625).  Same rules as everything above: device
 * pointers owned by the caller, no allocation or synchronisation inside, stream-ordered. */

/* get_smooth_depth (operations.py:161-169) = cv2.bilateralFilter(depth, d=15, sigma_color=0.5,
 * sigma_space=20) with invalid (< 0) pixels entering as 0 and leaving as -1.  h, w > d/2. */
int ags_smooth_depth(int32_t h, int32_t w, const float* depth, float* out, int32_t d, float sigma_color,
                     float sigma_space, ags_stream_t stream);

typedef struct AgsKeyframe {          /* dataframe of add_gaussians (gaussian_map.py:294-300) */
    int32_t image_height, image_width;
    const float* rgb;           /* (3,H,W) */
    const float* depth;         /* (1,H,W) sensor depth; <= 0 = no measurement */
    const float* intrinsic_inv; /* 9 floats, device: inverse of the NORMALISED 3x3 intrinsics */
    const float* extrinsic;     /* 16 floats, device: camera-to-world, row-major */
} AgsKeyframe;
typedef struct AgsDensifyPred {       /* render of the current map at the keyframe; all NULL before the */
    const float* rgb;                 /* map is initialised (cal_mask then selects every pixel)          */
    const float* depth;
    const float* opacity;
} AgsDensifyPred;
typedef struct AgsCandidates {        /* per pixel, P = H*W rows */
    float* means;       /* (P,3) unprojected points */
    float* rotations;   /* (P,4) wxyz from the depth-map normal (normal2rotation) */
    float* harmonics;   /* (P,3) pixel colours */
    int32_t* select;    /* (P) 1 = valid measurement that the map lacks (before the voxel filter) */
} AgsCandidates;
/* Everything add_gaussians derives per pixel (gaussian_map.py:301-400); depth_smooth from ags_smooth_depth. */
int ags_densify_candidates(const AgsKeyframe* frame, const float* depth_smooth, const AgsDensifyPred* pred,
                           float error_thres, const AgsCandidates* out, ags_stream_t stream);

/* voxel_downsample (operations.py:603-625): among the rows with select != 0 keep one per occupied
 * voxel - the highest row index (the reference picks a random one; see oracle/densify_oracle.py) -
 * and clear the others.  `ws`: ags_voxel_select_bytes(n) bytes of scratch. */
size_t ags_voxel_select_bytes(int32_t n);
int ags_voxel_select(int32_t n, const float* points, int32_t* select, float voxel_size, void* ws, size_t ws_bytes,
                     ags_stream_t stream);

/* prune's rule (gaussian_map.py:234-236): keep[i] = !(prune_mask[i] != 0 || sigmoid(raw_opacities[i]) < min_opacity);
 * prune_mask may be NULL. */
int ags_prune_keep(int32_t n, const float* prune_mask, const float* raw_opacities, float min_opacity, int32_t* keep,
                   ags_stream_t stream);

/* post_processing's per-surfel bookkeeping (gaussian_map.py:193-232) in one launch: seen = newest_count >= 1 (the
 * count image row of the newest keyframe); view_supports += seen; with use_view_distribution also, for seen surfels,
 * view_means += (direction to the camera - view_means) / max(view_supports, 1) and
 * view_scores += (1 - clamp(dist / far, 0, 1)) * clamp(normal . direction, 0, 1), normal = third column of the
 * normalised raw rotation.  campos: 3 floats on the device (the newest keyframe's camera-to-world translation). */
int ags_view_stats_update(int32_t n, const float* means, const float* raw_rotations, const float* campos, float far,
                          const int32_t* newest_count, int32_t use_view_distribution, float* view_supports,
                          float* view_means /* (n,3) */, float* view_scores, ags_stream_t stream);
/* get_confidences (gaussian_map.py:552-565): clamp(exp(1 - |view_means|) * view_scores, 0, 1) with NaN norms taken
 * as 1 (use_view_distribution), or clamp(1 - exp(-view_supports), 0, 1). */
int ags_confidences(int32_t n, const float* view_supports, const float* view_means, const float* view_scores,
                    int32_t use_view_distribution, float* out, ags_stream_t stream);

/* Stable stream compaction (torch boolean indexing / torch.cat of the selected rows):
 * plan:  dst_index[i] = number of kept rows before i, or -1; *total (device) = kept rows.
 * rows:  dst[dst_index[i]*width + c] = src[i*width + c] for one (n,width) float array. */
size_t ags_compact_plan_bytes(int32_t n);
int ags_compact_plan(int32_t n, const int32_t* keep, int32_t* dst_index, int32_t* total, void* scratch,
                     size_t scratch_bytes, ags_stream_t stream);
int ags_compact_rows(int32_t n, int32_t width, const int32_t* dst_index, const float* src, float* dst,
                     ags_stream_t stream);
/* The map's eight per-surfel arrays (gaussian_map.py:20-33: the five learnable tensors and the three view statistics). */
typedef struct AgsMapArrays {
    float* means;          /* (n,3) */
    float* scales;         /* (n,3) raw */
    float* rotations;      /* (n,4) raw */
    float* opacities;      /* (n)   raw */
    float* harmonics;      /* (n,3) */
    float* view_scores;    /* (n) */
    float* view_supports;  /* (n) */
    float* view_means;     /* (n,3) */
} AgsMapArrays;
/* add_gaussians' torch.cat of the selected candidates (gaussian_map.py:402-462) into arrays that already hold the map
 * with room behind it, ONE launch: new row dst_index[i] (>= 0) of `first_new_row` (pointers at row n of every array)
 * takes candidate i's mean, rotation and colour, raw scales (0, 0, new_z_scale), raw opacity 0 and zeroed statistics.
 * ags_map_compact: prune's boolean indexing of all eight arrays (gaussian_map.py:234-246), ONE launch:
 * dst row dst_index[i] = src row i for the kept rows (src and dst must not overlap). */
int ags_map_append(int32_t candidates, const int32_t* dst_index, const AgsCandidates* c, float new_z_scale,
                   const AgsMapArrays* first_new_row, ags_stream_t stream);
int ags_map_compact(int32_t n, const int32_t* dst_index, const AgsMapArrays* src, const AgsMapArrays* dst,
                    ags_stream_t stream);

/* Two helpers that keep a batched training iteration to a handful of launches:
 * ags_stage_frames gathers the sampled frames (frame_index: `views` int64 indices, device) of the
 * stacked keyframe arrays all_view / all_proj (K,16), all_rgb (K,3,H,W), all_depth (K,1,H,W) into the
 * batch buffers dst_* (what four index_select calls would do; dst_rgb = dst_depth = NULL: the matrices only) and, if
 * msum != NULL, zeroes that (H,W) visibility count; H*W must be a multiple of 4.
 * ags_loss_finish sums the accumulator rows of the loss stages, writes
 * frame_error[frame_index[v]] = mean rgb L1 + mean depth L1 of view v (track_performance,
 * gaussian_map.py:132-139; frame_index NULL: frame_error[v]) and *total_loss, and zeroes the
 * accumulators for the next iteration.  frame_error / total_loss may be NULL. */
/* The weighted draw of the batch sampler (mapping/utils.py:190-228: np.random.choice(older, k, replace=False, p = w / sum w))
 * on the device: out[0..k) = the indices of the k largest log(uniforms[i]) / max(weights[i], 1e-30), largest first
 * (successive sampling without replacement; uniforms ~ U(0,1), n <= 8192 of them, drawn by the caller). */
int ags_weighted_topk(const float* uniforms, const float* weights, int32_t n, int32_t k, int64_t* out, ags_stream_t stream);
int ags_stage_frames(int32_t views, int32_t h, int32_t w, const int64_t* frame_index, const float* all_view,
                     const float* all_proj, const float* all_rgb, const float* all_depth, float* dst_view, float* dst_proj,
                     float* dst_rgb, float* dst_depth, int32_t* msum, ags_stream_t stream);
int ags_loss_finish(const AgsLossConfig* cfg, float* accum, int32_t views, const int64_t* frame_index,
                    float* frame_error, float* total_loss, ags_stream_t stream);
/* The end of one batched training iteration and the set-up of the next in ONE launch: ags_loss_finish (frame_index is
 * required), then - with the errors it has just written as the weights - the next iteration's draw
 * (ags_weighted_topk(uniforms, frame_error, n_weights, k) into frame_index[first_random .. first_random + k); uniforms
 * NULL or k == 0: the indices stay what they are), then ags_stage_frames of the matrices of frame_index[0 .. views)
 * (no images: AgsLossConfig.gt_frame_index reads them in place) with msum cleared.  Replaces, per iteration of the
 * mapper's train() (gaussian_map.py:88-139 with mapping/utils.py:190-228's sampler), three single-workgroup launches
 * and the draw of the uniforms (one torch.rand for all iterations of a train() call instead). */
typedef struct AgsNextIteration {
    const float* uniforms;          /* n_weights numbers ~ U(0,1) for THIS draw, device */
    int32_t n_weights, k;           /* draw k distinct frames of the first n_weights by frame_error */
    int32_t first_random;           /* where in frame_index the drawn indices go */
    int32_t views;                  /* views of the next iteration: frame_index[0 .. views) are staged */
    const float* all_view;          /* (K,16) */
    const float* all_proj;          /* (K,16) */
    float* dst_view;                /* (views,16) */
    float* dst_proj;                /* (views,16) */
    int32_t* msum;                  /* (H,W) visibility count, cleared; H*W a multiple of 4 */
} AgsNextIteration;
/* Zero up to AGS_ZERO_MANY_MAX device regions in ONE launch (16-byte aligned addresses, sizes multiples of 4 bytes): the
 * buffers a train() call starts from - Adam moments, row set, gradient slab, loss accumulators, step clock - are separate
 * allocations, and a fill per buffer is a launch per buffer with the GPU idle in between. */
#define AGS_ZERO_MANY_MAX 16
int ags_zero_many(int32_t count, void* const* regions, const size_t* bytes, ags_stream_t stream);
int ags_loss_finish_next(const AgsLossConfig* cfg, float* accum, int32_t views, int64_t* frame_index, float* frame_error,
                         float* total_loss, const AgsNextIteration* next, ags_stream_t stream);

/* Optional stage timing with library-owned hipEvents (process-global, for bench/profiling
 * only; off by default so the normal path records nothing).  `slots` event pairs are kept
 * per stage; each forward/backward call consumes one slot per stage it runs. */
#define AGS_STAGE_PREPROCESS 0
#define AGS_STAGE_BINNING 1     /* scan + duplicate + radix sort + ranges */
#define AGS_STAGE_RENDER_FWD 2
#define AGS_STAGE_RENDER_BWD 3
#define AGS_STAGE_PREPROCESS_BWD 4
#define AGS_NUM_STAGES 5
int ags_profile_enable(int32_t slots);                 /* 0 frees the events */
/* waits for the events; mean and median stage time over the recorded slots; resets. The median
 * ignores slots in which the HOST stalled between two launches of an eager stage. */
int ags_profile_read(int32_t stage, float* avg_ms, float* median_ms, int32_t* samples);

const char* ags_error_string(int code);
int ags_version(void);

#ifdef __cplusplus
}
#endif
#endif /* AGS_RASTER_H */
