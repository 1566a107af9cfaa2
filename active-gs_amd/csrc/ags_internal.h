// Internal declarations shared by the HIP translation units of libags_raster.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ags_raster.h"
#include "surfel_math.h"

// AGS_BIN_DIRECT: words between two tiles' key-slot counters (AgsLayout.tc_stride).  Device atomics are served per cache
// LINE: with the counters dense, 32 tiles share a 128-byte line and their slot requests queue up behind each other -
// but every line the counters occupy is one more line the atomic units keep, and what pays depends on how many counters
// there are (round 3, compile-time sweeps of 1 .. 32 words, per-Gaussian launch unless noted):
//   64 tiles    (planner views, 128x128, 100 views batched):  whole batch 2.06 ms dense-ish (4) -> 1.80 (8) -> 1.77 (16) -> 1.52 (32)
//   1 024 tiles (a mapper batch of 512x512 views, ~200 requests per tile): 184 us dense -> 124 (2) -> 115 (4) -> 93 (8) -> 114 (16) -> 103 (32)
//   3 225 tiles (1200x680): 15.8 us dense -> 14.5-14.8 (2) -> 14.3-14.9 (4) -> 15.1-15.6 (8)
//   16 384 tiles (2048x2048): +0.5 % of the step at 2, +1.6 % at 4, +24 % of the launch at 8
// i.e. the counters want to cover a few tens of KB, never more than a line each: the largest power of two that keeps them
// within AGS_TC_SPREAD_BYTES, at most 32 words.
#define AGS_TC_SPREAD_BYTES (48 * 1024)
static inline int ags_tc_stride(int num_tiles) {
    int s = 32;
    while (s > 1 && (size_t)num_tiles * 4 * s > AGS_TC_SPREAD_BYTES) s >>= 1;
    return s;
}
#define AGS_SORT_THREADS 256
#define AGS_SORT_ITEMS 16
#define AGS_SORT_TILE (AGS_SORT_THREADS * AGS_SORT_ITEMS) // keys per block per pass
#define AGS_SORT_MAX_PASSES 8
#ifndef AGS_PRE_THREADS
#define AGS_PRE_THREADS 256
#endif

// Byte offsets of the workspace regions (all 256-B aligned). The first three regions are
// contiguous so one hipMemsetAsync clears them at the start of a forward pass.
struct AgsLayout {
    size_t status;      // 256 B: AgsStatus + padding
    size_t totals;      // AGS_SORT_MAX_PASSES * 256 u32 digit totals
    size_t ranges;      // T * uint2
    size_t tile_count;  // T * u32 (tile-sort mode)
    size_t tile_fill;   // T * u32 (tile-sort mode)
    size_t clear_bytes; // status..tile_fill end
    size_t geom;        // n * AgsGeom
    size_t tiles;       // n * u32 tiles touched
    size_t rect;        // n * ushort4
    size_t block_sums;  // (ceil(n/256)+1) * u32
    size_t block_vis;   // ceil(n/256) * u32 visible surfels per preprocess block
    size_t keys0, keys1;// cap * u64
    size_t vals0, vals1;// cap * u32
    size_t hist;        // 256 * nb_cap * u32
    size_t final_T;     // P * f32
    size_t n_contrib;   // P * u32
    size_t dgeom;       // n * AgsGeomGrad (backward)
    size_t total;
    int64_t cap;
    int nb_cap;         // sort blocks at capacity
    int n_blocks;       // preprocess blocks
    int num_tiles;
    int tc_stride;      // words between two tiles' counters in tile_count (ags_tc_stride)
    AgsTuning tune;     // the caller's kernel selection (AgsWorkspace.tuning; zeros = defaults): ags_layout_for()
};

static inline size_t ags_align256(size_t x) { return (x + 255) & ~(size_t)255; }

static inline AgsLayout ags_make_layout(int n, int h, int w, int64_t cap) {
    AgsLayout L;
    const int tiles_x = (w + AGS_TILE - 1) / AGS_TILE, tiles_y = (h + AGS_TILE - 1) / AGS_TILE;
    L.num_tiles = tiles_x * tiles_y;
    L.cap = cap < 1 ? 1 : cap;
    L.nb_cap = (int)((L.cap + AGS_SORT_TILE - 1) / AGS_SORT_TILE);
    L.n_blocks = (n + AGS_PRE_THREADS - 1) / AGS_PRE_THREADS;
    const size_t P = (size_t)h * w;
    size_t o = 0;
    L.status = o; o += 256;
    L.totals = o; o += (size_t)AGS_SORT_MAX_PASSES * 256 * 4;
    L.ranges = o; o += ags_align256((size_t)L.num_tiles * 8);
    L.tc_stride = ags_tc_stride(L.num_tiles);
    L.tile_count = o; o += ags_align256((size_t)L.num_tiles * 4 * L.tc_stride);
    L.tile_fill = o; o += ags_align256((size_t)L.num_tiles * 4);
    L.clear_bytes = o;
    L.geom = o; o += ags_align256((size_t)n * sizeof(AgsGeom));
    L.tiles = o; o += ags_align256((size_t)n * 4);
    L.rect = o; o += ags_align256((size_t)n * 8);
    L.block_sums = o; o += ags_align256((size_t)(L.n_blocks + 1) * 4);
    L.block_vis = o; o += ags_align256((size_t)(L.n_blocks + 1) * 4);
    L.keys0 = o; o += ags_align256((size_t)L.cap * 8);
    L.keys1 = o; o += ags_align256((size_t)L.cap * 8);
    L.vals0 = o; o += ags_align256((size_t)L.cap * 4);
    L.vals1 = o; o += ags_align256((size_t)L.cap * 4);
    L.hist = o; o += ags_align256((size_t)256 * L.nb_cap * 4);
    L.final_T = o; o += ags_align256(P * 4);
    L.n_contrib = o; o += ags_align256(P * 4);
    L.dgeom = o; o += ags_align256((size_t)n * sizeof(AgsGeomGrad));
    L.total = o;
    L.tune = AgsTuning{};
    return L;
}
// the layout of a caller's workspace together with the caller's tuning (what the launchers look at)
static inline AgsLayout ags_layout_for(int n, int h, int w, const AgsWorkspace* ws) {
    AgsLayout L = ags_make_layout(n, h, w, ws->max_instances);
    if (ws->tuning) L.tune = *ws->tuning;
    return L;
}

static inline AgsFrame ags_make_frame(const AgsCamera* c) {
    AgsFrame F;
    F.H = c->image_height; F.W = c->image_width;
    F.tiles_x = (F.W + AGS_TILE - 1) / AGS_TILE; F.tiles_y = (F.H + AGS_TILE - 1) / AGS_TILE;
    F.tanfovx = c->tanfovx; F.tanfovy = c->tanfovy;
    F.fx = F.W / (2.0f * c->tanfovx); F.fy = F.H / (2.0f * c->tanfovy);
    F.scale_mod = c->scale_modifier;
    F.perpix_depth = c->perpix_depth; F.front_only = c->front_only;
    F.cfg = c->config;
    return F;
}

// Device-resident optimiser clock shared by adam.hip, render.hip (tick) and preprocess.hip (fused step)
// `skipped`: optimisation steps ags_adam_step_gathered refused (a rank's row set had outgrown the agreed exchange
// segment): sticky, cleared when the caller zeroes the clock together with the optimiser state
// `pow1/pow2`: beta1^step, beta2^step as running products in double (two multiplies per tick instead of two generic
// pow() calls - several hundred fp64 instructions that used to sit in the blend backward's code); `prev*`: their
// values before the last tick, which a refused step restores.
struct AgsAdamClock { int step; float step_size[5]; float inv_bc2_sqrt; int skipped; double pow1, pow2, prev1, prev2; };
static_assert(sizeof(AgsAdamClock) == 64, "the Adam device clock is a 64-byte block");
struct AgsAdamArgs {
    float* p[5];
    const float* g[5];
    float* m[5];
    float* v[5];
    long long end[5]; // cumulative element counts
    float lr[5];
    float* st;        // AgsAdamTensors.state_rows: (n, 28) interleaved moments, or nullptr (then m / v)
};
// "advance this Adam clock on the side" request carried by a backward launch (clock == nullptr: off)
// (`rows_count` / `count_snap`: the same thread also notes how many rows the optimiser's row set holds right now in the
// workspace's status block - AgsStatus.reserved[0] - for the software-pipelined per-Gaussian kernel that follows, whose
// own workgroups append to the list while others walk it)
struct AgsTick { AgsAdamClock* clock; float lr[5]; float beta1, beta2; const int* rows_count; uint32_t* count_snap; };
#define AGS_STATUS_COUNT_SNAP 8   // word index of AgsStatus.reserved[0]
#define AGS_STATUS_EARLY 9        // word index of AgsStatus.early_tile_need
inline AgsAdamArgs ags_adam_args(const AgsAdamTensors& t) {
    AgsAdamArgs a;
    long long run = 0;
    for (int k = 0; k < 5; ++k) {
        a.p[k] = t.param[k]; a.g[k] = t.grad[k]; a.m[k] = t.exp_avg[k]; a.v[k] = t.exp_avg_sq[k];
        run += t.numel[k];
        a.end[k] = run;
        a.lr[k] = t.lr[k];
    }
    a.st = t.state_rows;
    return a;
}

// Batched forward (ags_forward_batch): blockIdx.y selects the view; every per-view pointer moves by
// these strides (all zero-cost for the single-view launches, which pass gridDim.y == 1).
struct AgsViewStride {
    long long ws;   // bytes between two views' workspaces
    long long px;   // pixels per image plane (H*W): images / masks of consecutive views are contiguous
    long long n;    // Gaussians: per-Gaussian outputs (radii, importance, count) of consecutive views
    int views;      // gridDim.y
};
// pointer arithmetic, NOT an integer round trip: an inttoptr would lose the pointer's global address
// space (flat_load / flat_store instead of global_*, and the uniform `ranges[tile]` reads stop being
// scalar loads - measured +1.3 us on a 5 us kernel)
#define AGS_WS_SHIFT(ptr, off) ptr = (decltype(ptr))((char*)(ptr) + (off))

// ---- launchers (one per translation unit; each enqueues on `s` and never synchronises)
// `ids` + `id_stride`: sorted Gaussian ids per instance; stride 2 when they are the low
// words of the 64-bit (depth|id) keys of the tile-sort mode.
struct AgsIdList { const uint32_t* ids; int stride; };
// emit: 0 = nothing (radix mode counts rect tiles only), 1 = count the reachable tiles (tile-sort mode),
// 2 = AGS_BIN_DIRECT: take a key slot in the tile's own slot range and write the key
// (cam.config != NULL: the kernel also zero-fills pg.importance / pg.count, see AgsCamera.config)
void ags_launch_preprocess(const AgsFrame& F, const AgsCamera& cam, const AgsGaussians& in, char* ws,
                           const AgsLayout& L, const AgsPerGaussian& pg, int emit,
                           const AgsViewStride& vs, hipStream_t s);
void ags_launch_direct_sort(char* ws, const AgsLayout& L, const AgsViewStride& vs, hipStream_t s);
void ags_launch_zero_many(int count, void* const* regions, const size_t* bytes, hipStream_t s);
void ags_launch_clear_regions(char* base, size_t stride, size_t offset, size_t bytes, int views, hipStream_t s);
// AGS_BIN_DIRECT bookkeeping, kept in the (otherwise radix-only) digit-total words of the workspace: 64 partial
// sums of the tiles' list lengths, 64 partial maxima, 64 partial counts of visible surfels - spread so that the
// tiles' / waves' atomics do not serialise on one word; the forward blend kernel's first wave reduces them into
// the status block and leaves them zeroed again
// slot k (0..63) owns one 128-byte line: words 32 k + {0: sum, 1: max, 2: visible}
#ifndef AGS_PART_STRIDE
#define AGS_PART_STRIDE 32
#endif
#define AGS_PART_SUM 0
#define AGS_PART_MAX 1
#define AGS_PART_VIS 2
#define AGS_PART(slot, what) (((slot) & 63) * AGS_PART_STRIDE + (what))
static inline uint32_t ags_direct_tile_cap(const AgsLayout& L) { return (uint32_t)(L.cap / (L.num_tiles > 0 ? L.num_tiles : 1)); }
void ags_launch_binning(const AgsFrame& F, const AgsGaussians& in, char* ws, const AgsLayout& L, hipStream_t s);
void ags_launch_tile_binning(const AgsFrame& F, const AgsGaussians& in, char* ws, const AgsLayout& L,
                             const AgsViewStride& vs, hipStream_t s);
AgsIdList ags_sorted_ids(char* ws, const AgsLayout& L, int binning_mode);
// ags_forward_batch_loss: what the forward blend kernel's epilogue needs of stage 1 of the loss head (loss.hip)
struct AgsLossFuse {
    const float* gt_rgb; const float* gt_depth;
    float* n_img; float* d_rgb; float* d_depth;
    int* msum; float* accum;
    const long long* gt_index;     // view v's ground truth is frame gt_index[v] of the store (NULL: batch order)
    int accum_stride;
    float k_rgb, k_depth;          // w_rgb / (B 3 HW), w_depth / (B HW)
};
void ags_launch_render_fwd(const AgsFrame& F, const AgsCamera& cam, char* ws, const AgsLayout& L,
                           AgsIdList ids, const AgsImages& out, const AgsPerGaussian& pg, const AgsViewStride& vs,
                           bool direct, hipStream_t s, const AgsLossFuse* loss = nullptr);
void ags_launch_rows_adam_preprocess(const AgsFrame& F, const AgsCamera& cam, const AgsGaussians& in, char* ws,
                                     const AgsLayout& L, const int* radii, const AgsGaussianGrads& din,
                                     const AgsFrame& F2, const AgsCamera& cam2, char* ws2, const AgsLayout& L2, int* radii2,
                                     int rows_hint, hipStream_t s);
void ags_launch_render_bwd(const AgsFrame& F, const AgsCamera& cam, char* ws, const AgsLayout& L,
                           AgsIdList ids, const AgsImages& fwd, const AgsImageGrads& dout, const AgsTick& tick,
                           const AgsViewStride& vs, bool direct, hipStream_t s);
void ags_launch_preprocess_bwd(const AgsFrame& F, const AgsCamera& cam, const AgsGaussians& in, char* ws,
                               const AgsLayout& L, const int* radii, const AgsGaussianGrads& din,
                               const AgsViewStride& vs, hipStream_t s);
// ags_backward_rows: per view the frame, the matrices, the gradient records and the radii
struct AgsRowViews {
    int views;
    AgsFrame F[AGS_MAX_ROW_VIEWS];
    const float* V[AGS_MAX_ROW_VIEWS];
    const float* P[AGS_MAX_ROW_VIEWS];
    AgsGeomGrad* dgeom[AGS_MAX_ROW_VIEWS];
    const int* radii[AGS_MAX_ROW_VIEWS];
};
void ags_launch_rows_multi(const AgsRowViews& rv, const AgsGaussians& in, const AgsGaussianGrads& din, hipStream_t s);
void ags_launch_adam(const AgsAdamTensors& t, float beta1, float beta2, float eps, int step, void* dev_state,
                     bool pre_ticked, hipStream_t s);
#if defined(__HIPCC__)
__device__ __forceinline__ void ags_adam_tick(AgsAdamClock* c, const float lr[5], float beta1, float beta2) {
    const int before = c->step;
    const double p1 = before > 0 ? c->pow1 : 1.0, p2 = before > 0 ? c->pow2 : 1.0;
    const double q1 = p1 * (double)beta1, q2 = p2 * (double)beta2;
    c->prev1 = p1; c->prev2 = p2; c->pow1 = q1; c->pow2 = q2;
    c->step = before + 1;
    const double inv_bc1 = 1.0 / (1.0 - q1);
    for (int k = 0; k < 5; ++k) c->step_size[k] = (float)((double)lr[k] * inv_bc1);
    c->inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - q2));
}
// a refused step (ags_adam_step_gathered): as if the last tick had not happened
__device__ __forceinline__ void ags_adam_untick(AgsAdamClock* c) {
    c->step -= 1; c->pow1 = c->prev1; c->pow2 = c->prev2;
}
#endif
void ags_launch_rows_pack(float* const grads[5], const AgsRowSet& rows, float* segment, int capacity, hipStream_t s);
void ags_launch_rows_unpack(const float* segment, int capacity, float* const grads[5], const AgsRowSet& uni, hipStream_t s);
void ags_launch_rows_index(const float* segs, size_t seg_floats, int capacity, int world, int* slot_table, const AgsRowSet& uni, hipStream_t s);
void ags_launch_adam_gathered(const AgsAdamTensors& t, const float* segs, size_t seg_floats, int world, int capacity, int* slot_table, float beta1, float beta2, float eps, void* dev_state, bool pre_ticked, hipStream_t s);
void ags_launch_activate(const AgsActivation& a, float* scales, float* rotations, float* opacities, hipStream_t s);
void ags_launch_activate_bwd(const AgsActivation& a, float* d_scales, float* d_rotations, float* d_opacities,
                             hipStream_t s);
void ags_launch_loss_stage1(const AgsLossConfig& cfg, const AgsImages& img, const float* gt_rgb, const float* gt_depth,
                            float* n_img, float* d_rgb, float* d_depth, int* msum, float* accum, int view,
                            int first_view, hipStream_t s);
void ags_launch_loss_stage2(const AgsLossConfig& cfg, const AgsImages& img, const float* n_img, const float* gt_depth,
                            const int* msum, float* d_normal, float* d_depth, float* accum, hipStream_t s);
void ags_launch_facade_post(int views, int h, int w, float tanx, float tany, const float* normal_raw, const float* depth,
                            const float* opacity, float* normal_out, float* d2n_out, hipStream_t s);
void ags_launch_facade_post_bwd(int h, int w, float tanx, float tany, const float* normal_raw, const float* depth,
                                const float* opacity, const float* g_normal, const float* g_d2n, float* d_normal_raw,
                                float* d_depth, hipStream_t s);
void ags_launch_weighted_topk(const float* u, const float* w, int n, int k, long long* out, hipStream_t s);
void ags_launch_stage_frames(int views, int hw, const long long* frame_index, const float* all_view, const float* all_proj,
                             const float* all_rgb, const float* all_depth, float* dst_view, float* dst_proj,
                             float* dst_rgb, float* dst_depth, int* msum, hipStream_t s);
void ags_launch_loss_finish_next(const AgsLossConfig& cfg, float* accum, int views, long long* frame_index, float* frame_error,
                                 float* total_loss, const AgsNextIteration& nx, hipStream_t s);
void ags_launch_loss_finish(const AgsLossConfig& cfg, float* accum, int views, const long long* frame_index,
                            float* frame_error, float* total_loss, hipStream_t s);
int ags_sort_passes(int num_tiles);
// densify.hip
void ags_launch_bilateral(int h, int w, const float* depth, float* out, int d, float sigma_color, float sigma_space,
                          hipStream_t s);
void ags_launch_candidates(const AgsKeyframe& f, const float* depth_smooth, const AgsDensifyPred& pred,
                           float error_thres, const AgsCandidates& out, hipStream_t s);
size_t ags_voxel_bytes(int n);
void ags_launch_voxel_select(int n, const float* points, int32_t* select, float voxel, void* ws, hipStream_t s);
size_t ags_compact_bytes(int n);
void ags_launch_compact_plan(int n, const int32_t* keep, int32_t* dst_index, int32_t* total, void* scratch, hipStream_t s);
void ags_launch_map_append(int P, const int32_t* dst_index, const AgsCandidates& c, float new_z, const AgsMapArrays& o, hipStream_t s);
void ags_launch_map_compact(int n, const int32_t* dst_index, const AgsMapArrays& a, const AgsMapArrays& o, hipStream_t s);
void ags_launch_compact_rows(int n, int width, const int32_t* dst_index, const float* src, float* dst, hipStream_t s);
void ags_launch_view_stats(int n, const float* means, const float* raw_rotations, const float* campos, float far,
                           const int32_t* newest_count, int use_vd, float* view_supports, float* view_means,
                           float* view_scores, hipStream_t s);
void ags_launch_confidences(int n, const float* view_supports, const float* view_means, const float* view_scores, int use_vd,
                            float* out, hipStream_t s);
void ags_launch_prune_keep(int n, const float* prune_mask, const float* raw_opacities, float min_opacity, int32_t* keep,
                           hipStream_t s);

// instrumentation of experiment builds (per-wave phase timelines, issue-priority and prologue knobs): compiles to nothing
// in the product build
#include "ags_experiments.h"

#if defined(__HIPCC__)
// ---- AgsCamera.config: the reference hands its five configuration floats over as a DEVICE tensor it has just built
// (operations.py:697-699); reading them back would cost the caller a stream synchronisation per view, so the kernels
// take them where they are: wave-uniform scalar loads through the kernel-argument pointer.  cfg == nullptr: the host's
// ints (every other caller).
__device__ __forceinline__ void ags_frame_flags(AgsFrame& F) {
    if (F.cfg) { F.perpix_depth = F.cfg[2] > 0.f ? 1 : 0; F.front_only = F.cfg[4] > 0.f ? 1 : 0; }
}
__device__ __forceinline__ int ags_cfg_flag(const float* cfg, int k, int host_value) {
    return cfg ? (cfg[k] > 0.f ? 1 : 0) : host_value;
}
// ---- wave64 helpers (gfx950): DPP reductions, no LDS, no ds_bpermute
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int ags_dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, true);
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float ags_dpp_f(float v) {
    return __builtin_bit_cast(float, ags_dpp_i<CTRL, ROW_MASK>(__builtin_bit_cast(int, v)));
}
// after this every lane of row 3 (lanes 48..63) holds the wave total; lane 63 is read back
__device__ __forceinline__ float ags_wave_sum_lane63(float v) {
    v += ags_dpp_f<0xB1>(v);       // quad_perm [1,0,3,2]
    v += ags_dpp_f<0x4E>(v);       // quad_perm [2,3,0,1]
    v += ags_dpp_f<0x141>(v);      // row_half_mirror
    v += ags_dpp_f<0x140>(v);      // row_mirror
    v += ags_dpp_f<0x142, 0xA>(v); // row_bcast15 -> rows 1,3
    v += ags_dpp_f<0x143, 0xC>(v); // row_bcast31 -> rows 2,3
    return v;
}
__device__ __forceinline__ float ags_wave_sum(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ags_wave_sum_lane63(v)), 63));
}
__device__ __forceinline__ uint32_t ags_wave_sum_u32(uint32_t x) {
    int v = (int)x;
    v += ags_dpp_i<0xB1>(v);
    v += ags_dpp_i<0x4E>(v);
    v += ags_dpp_i<0x141>(v);
    v += ags_dpp_i<0x140>(v);
    v += ags_dpp_i<0x142, 0xA>(v);
    v += ags_dpp_i<0x143, 0xC>(v);
    return (uint32_t)__builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ uint32_t ags_wave_max_u32(uint32_t x) {
    // values are < 2^31 here, so signed max on the zero-filled DPP sources is exact
    int v = (int)x;
    v = max(v, ags_dpp_i<0xB1>(v));
    v = max(v, ags_dpp_i<0x4E>(v));
    v = max(v, ags_dpp_i<0x141>(v));
    v = max(v, ags_dpp_i<0x140>(v));
    v = max(v, ags_dpp_i<0x142, 0xA>(v));
    v = max(v, ags_dpp_i<0x143, 0xC>(v));
    return (uint32_t)__builtin_amdgcn_readlane(v, 63);
}
// inclusive prefix sum across the wave (Hillis-Steele on DPP row shifts + row broadcasts)
__device__ __forceinline__ uint32_t ags_wave_incl_scan_u32(uint32_t x) {
    int v = (int)x;
    v += ags_dpp_i<0x111>(v); // row_shr:1
    v += ags_dpp_i<0x112>(v); // row_shr:2
    v += ags_dpp_i<0x114>(v); // row_shr:4
    v += ags_dpp_i<0x118>(v); // row_shr:8
    v += ags_dpp_i<0x142, 0xA>(v); // row_bcast15 into rows 1,3
    v += ags_dpp_i<0x143, 0xC>(v); // row_bcast31 into rows 2,3
    return (uint32_t)v;
}
// Visit every tile of a Gaussian's rect. Footprints above COOP tiles are walked by the whole
// wave (lane-strided, coalesced side effects), small ones by their own lane.  `f(tile, a, b)`
// gets the owner's two payload words.  Must be called by all 64 lanes (cnt = 0 when idle).
// (used by the radix path, whose instance slots are ordered by Gaussian)
template <typename Fn>
__device__ __forceinline__ void ags_for_each_tile(uint32_t cnt, uint32_t x0, uint32_t y0, uint32_t wd,
                                                  uint32_t pa, uint32_t pb, int tiles_x, Fn&& f) {
    const uint32_t COOP = 32;
    const int lane = threadIdx.x & 63;
    unsigned long long big = __ballot(cnt > COOP);
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const uint32_t sx = __shfl(x0, src), sy = __shfl(y0, src), sw = __shfl(wd, src), sc = __shfl(cnt, src);
        const uint32_t sa = __shfl(pa, src), sb = __shfl(pb, src);
        for (uint32_t t = lane; t < sc; t += 64) f((sy + t / sw) * tiles_x + sx + t % sw, sa, sb);
    }
    if (cnt && cnt <= COOP)
        for (uint32_t t = 0; t < cnt; ++t) f((y0 + t / wd) * tiles_x + x0 + t % wd, pa, pb);
}

// Balanced form for the tile-sort path: the wave flattens the (surfel, candidate tile) pairs of
// its 64 surfels into one index space (DPP prefix sum) and hands pair j to lane j%64, so a
// surfel touching 400 tiles and one touching 2 cost the same per lane and every atomic in an
// iteration is independent.  The owner of pair j is found by a 6-step binary search over the
// exclusive prefix kept in LDS; the tile must pass the exact reach test (ags_reaches_box) -
// tiles of the D3 rect that no pixel of the surfel can reach are never emitted (they would
// contribute nothing: identical images, fewer instances to sort, gather and blend).
// f(hit, tile, owner's payload word, owner's lane in this wave).
struct AgsEmitRec { uint32_t excl, xy, wd, pa; float mx, my, ca, cb, cc, o; }; // per lane, in LDS

// counter[t] += 1 for every `active` lane, returning the lane's old value (its slot).
// AGG = false: one atomic per lane.  AGG = true (images of few tiles, where a wave's 64 emissions hit
// a handful of counters and same-address atomics serialise in L2): lanes that hit the SAME counter
// are grouped first - up to AGS_AGG_ROUNDS groups per call, found with readlane + ballot, no memory
// traffic - then every group leader issues ONE atomic for the whole group (all leaders in the same
// instruction: a single round trip), and members take base + rank.  RETURN = false drops the result
// (counting pass: fire and forget).  Must be called by all lanes of the wave.
#define AGS_AGG_ROUNDS 6
#ifndef AGS_AGG_MAX_TILES
#define AGS_AGG_MAX_TILES 256
#endif
// AGG = 2 (round 4): the ADAPTIVE form for images of many tiles.  Whether a wave's emissions share tiles depends on the
// map, not on the image: a map grown by the mapper is spatially coherent (a keyframe's surfels are appended pixel by
// pixel, so the 64 rows of a wave sit next to each other and their pairs fall on the same handful of tiles - a 512x512
// training batch asks every tile counter ~200 times per view), a synthetic room in random order shares nothing.  Same
// leader rounds, but the search stops at the first group of fewer than AGS_AGG_MIN_GROUP lanes (a round costs about as
// much issue time as four serialised same-address atomics cost the L2 channel): coherent waves issue a few atomics
// instead of 64, incoherent ones pay one or two short rounds and go lane by lane.
#define AGS_AGG_ADAPT_ROUNDS 8
#define AGS_AGG_MIN_GROUP 4
template <int AGG, bool RETURN>
__device__ __forceinline__ uint32_t ags_wave_agg_inc(uint32_t* __restrict__ counter, uint32_t t, bool active) {
    if (AGG == 0) {
        uint32_t slot = 0;
        if (active) { if (RETURN) slot = atomicAdd(&counter[t], 1u); else atomicAdd(&counter[t], 1u); }
        return slot;
    }
    const int lane = threadIdx.x & 63;
    unsigned long long rem = __ballot(active);
    if (AGG == 2) {
        // probe: does the first active lane's counter have company?  If not, the wave goes lane by lane (nothing below
        // - no ranks, no broadcast of the leaders' results - stands between the lanes and their atomics)
        if (!rem) return 0;
        const uint32_t tp = (uint32_t)__builtin_amdgcn_readlane((int)t, __ffsll((long long)rem) - 1);
        if (__builtin_popcountll(__ballot(active && t == tp)) < AGS_AGG_MIN_GROUP) {
            uint32_t slot = 0;
            if (active) { if (RETURN) slot = atomicAdd(&counter[t], 1u); else atomicAdd(&counter[t], 1u); }
            return slot;
        }
    }
    int leader_of = lane;
    uint32_t rank = 0, size = active ? 1u : 0u;
    constexpr int ROUNDS = AGG == 2 ? AGS_AGG_ADAPT_ROUNDS : AGS_AGG_ROUNDS;
#pragma unroll 1
    for (int round = 0; round < ROUNDS && rem; ++round) {
        const int leader = __ffsll((long long)rem) - 1;
        const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)t, leader);
        const unsigned long long m = __ballot(active && t == t0) & rem;
        if ((m >> lane) & 1ull) {
            leader_of = leader;
            rank = (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
            size = lane == leader ? (uint32_t)__builtin_popcountll(m) : 0u;
        }
        rem &= ~m;
        if (AGG == 2 && __builtin_popcountll(m) < AGS_AGG_MIN_GROUP) break;   // wave-uniform: no sharing worth another round
    }
    uint32_t base = 0;
    if (size) { if (RETURN) base = atomicAdd(&counter[t], size); else atomicAdd(&counter[t], size); }
    if (!RETURN) return 0;
    base = (uint32_t)__shfl((int)base, leader_of);
    return base + rank;
}

template <typename Fn>
__device__ __forceinline__ void ags_emit_tiles_balanced(AgsEmitRec* wave_lds, uint32_t cnt, uint32_t x0, uint32_t y0,
                                                        uint32_t wd, uint32_t pa, const AgsGeom& g, int tiles_x,
                                                        Fn&& f) {
    const int lane = threadIdx.x & 63;
    const uint32_t incl = ags_wave_incl_scan_u32(cnt);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (total == 0) return; // wave-uniform
    AgsEmitRec me;
    me.excl = incl - cnt; me.xy = x0 | (y0 << 16); me.wd = wd; me.pa = pa;
    me.mx = g.mx; me.my = g.my; me.ca = g.ca; me.cb = g.cb; me.cc = g.cc; me.o = g.o;
    wave_lds[lane] = me;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t base = 0; base < total; base += 64) {
        const uint32_t j = base + lane;
        bool hit = false;
        uint32_t tile = 0, owner = 0;
        int lo = 0;
        if (j < total) {
#pragma unroll
            for (int step = 32; step > 0; step >>= 1)
                if (wave_lds[lo + step].excl <= j) lo += step; // largest lane with excl <= j
            const AgsEmitRec r = wave_lds[lo];
            const uint32_t t = j - r.excl;
            const uint32_t tx = (r.xy & 0xFFFF) + t % r.wd, ty = (r.xy >> 16) + t / r.wd;
            AgsGeom og;
            og.mx = r.mx; og.my = r.my; og.ca = r.ca; og.cb = r.cb; og.cc = r.cc; og.o = r.o;
            const float bx = (float)(tx * AGS_TILE), by = (float)(ty * AGS_TILE);
            hit = ags_reaches_box(og, bx, bx + (AGS_TILE - 1), by, by + (AGS_TILE - 1));
            tile = ty * tiles_x + tx; owner = r.pa;
        }
        f(hit, tile, owner, lo); // called by the whole wave (wave-uniform control flow): see ags_wave_agg_inc; lo = owner's lane
    }
    __builtin_amdgcn_wave_barrier();
}

// ---- per-tile sort of the (depth_bits << 32 | id) keys of the tile-sort binning mode.
// ascending-only bitonic network (mirrored first sub-step), so indices >= K behave as +inf
// padding without being stored: works for any K, in LDS or in global memory.
template <int NT, typename Ptr>
__device__ __forceinline__ void ags_bitonic(Ptr a, uint32_t K, int tid) {
    uint32_t Kp = 1;
    while (Kp < K) Kp <<= 1;
    const uint32_t half = Kp >> 1;
    for (uint32_t k = 2; k <= Kp; k <<= 1) {
        const uint32_t hk = k >> 1;
        for (uint32_t t = tid; t < half; t += NT) {
            const uint32_t blk = t / hk, off = t % hk;
            const uint32_t i = blk * k + off, l = blk * k + (k - 1 - off);
            if (l < K) { const uint64_t x = a[i], y = a[l]; if (x > y) { a[i] = y; a[l] = x; } }
        }
        __syncthreads();
        for (uint32_t j = k >> 2; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < half; t += NT) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i + j;
                if (l < K) { const uint64_t x = a[i], y = a[l]; if (x > y) { a[i] = y; a[l] = x; } }
            }
            __syncthreads();
        }
    }
}

// The same network with its SHORT-distance sub-steps in registers.  The LDS network above moves every key through
// the LDS once per sub-step (two 8-byte reads and up to two 8-byte writes per compare-exchange: the LDS store path, ~6
// cycles per wave-instruction, is what bounds it - 45 sub-steps for 512 keys).  Here a thread owns 8 CONSECUTIVE keys:
// the stages 2, 4, 8 run entirely in its registers (one pass over the LDS instead of six) and so do the last three
// sub-steps (distances 4, 2, 1) of every later stage - 28 passes instead of 45 for 512 keys, 45 instead of 66 for 2048.
// Same network, same result (keys are unique).  A thread's 8 keys are four 16-byte granules 64 bytes apart from its
// neighbour's (the array must be 16-byte aligned: ds_read/write_b128): the granule index inside a block is XOR-swizzled with bits 5-6 of the key index so that the 16 lanes the
// hardware serves together (MI355X_MICROARCH.md, LDS) hit 64 different banks.  EVERY access to the array goes through
// ags_sk() - the caller's loads and stores too.
__device__ __forceinline__ uint32_t ags_sk(uint32_t idx) { return idx ^ (((idx >> 5) & 3u) << 1); }
#define AGS_CE(x, y) do { const uint64_t lo_ = v[x] < v[y] ? v[x] : v[y], hi_ = v[x] < v[y] ? v[y] : v[x]; v[x] = lo_; v[y] = hi_; } while (0)
__device__ __forceinline__ void ags_bitonic8_load(const uint64_t* a, uint32_t b, uint32_t K, uint64_t v[8]) {
    typedef unsigned long long u2 __attribute__((ext_vector_type(2)));
    const uint32_t sw = ((b >> 2) & 3u);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const u2 x = *reinterpret_cast<const u2*>(a + 8 * b + 2 * (q ^ sw));
        v[2 * q] = (8 * b + 2 * q < K) ? x.x : ~0ull;
        v[2 * q + 1] = (8 * b + 2 * q + 1 < K) ? x.y : ~0ull;
    }
}
__device__ __forceinline__ void ags_bitonic8_store(uint64_t* a, uint32_t b, const uint64_t v[8]) {
    typedef unsigned long long u2 __attribute__((ext_vector_type(2)));
    const uint32_t sw = ((b >> 2) & 3u);
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<u2*>(a + 8 * b + 2 * (q ^ sw)) = u2{v[2 * q], v[2 * q + 1]};
}
// (slots at and above K hold +inf after a register pass: the array must have room for Kp rounded up to 8 keys)
template <int NT>
__device__ __forceinline__ void ags_bitonic_regs(uint64_t* a, uint32_t K, int tid) {
    uint32_t Kp = 8;
    while (Kp < K) Kp <<= 1;
    const uint32_t nblk = Kp >> 3, half = Kp >> 1;
    for (uint32_t b = tid; b < nblk; b += NT) {          // stages 2, 4, 8
        uint64_t v[8];
        ags_bitonic8_load(a, b, K, v);
        AGS_CE(0, 1); AGS_CE(2, 3); AGS_CE(4, 5); AGS_CE(6, 7);
        AGS_CE(0, 3); AGS_CE(1, 2); AGS_CE(4, 7); AGS_CE(5, 6);
        AGS_CE(0, 1); AGS_CE(2, 3); AGS_CE(4, 5); AGS_CE(6, 7);
        AGS_CE(0, 7); AGS_CE(1, 6); AGS_CE(2, 5); AGS_CE(3, 4);
        AGS_CE(0, 2); AGS_CE(1, 3); AGS_CE(4, 6); AGS_CE(5, 7);
        AGS_CE(0, 1); AGS_CE(2, 3); AGS_CE(4, 5); AGS_CE(6, 7);
        ags_bitonic8_store(a, b, v);
    }
    __syncthreads();
    for (uint32_t k = 16; k <= Kp; k <<= 1) {
        const uint32_t hk = k >> 1;
        for (uint32_t t = tid; t < half; t += NT) {      // mirrored first sub-step of the stage
            const uint32_t blk = t / hk, off = t % hk;
            const uint32_t i = ags_sk(blk * k + off), l = ags_sk(blk * k + (k - 1 - off));
            const uint64_t x = a[i], y = a[l];
            if (x > y) { a[i] = y; a[l] = x; }
        }
        __syncthreads();
        for (uint32_t j = k >> 2; j >= 8; j >>= 1) {
            for (uint32_t t = tid; t < half; t += NT) {
                const uint32_t i0 = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const uint32_t i = ags_sk(i0), l = ags_sk(i0 + j);
                const uint64_t x = a[i], y = a[l];
                if (x > y) { a[i] = y; a[l] = x; }
            }
            __syncthreads();
        }
        for (uint32_t b = tid; b < nblk; b += NT) {      // distances 4, 2, 1
            uint64_t v[8];
            ags_bitonic8_load(a, b, Kp, v);
            AGS_CE(0, 4); AGS_CE(1, 5); AGS_CE(2, 6); AGS_CE(3, 7);
            AGS_CE(0, 2); AGS_CE(1, 3); AGS_CE(4, 6); AGS_CE(5, 7);
            AGS_CE(0, 1); AGS_CE(2, 3); AGS_CE(4, 5); AGS_CE(6, 7);
            ags_bitonic8_store(a, b, v);
        }
        __syncthreads();
    }
}

// Steps j = j0, j0/2, ..., 1 of the network on one LDS-resident chunk (the "finish" of a stage whose
// wide steps ran in global memory); `base` = global index of the chunk's first key.
template <int NT>
__device__ __forceinline__ void ags_bitonic_finish_lds(uint64_t* sk, uint32_t n, uint32_t j0, int tid) {
    for (uint32_t j = j0; j > 0; j >>= 1) {
        for (uint32_t t = tid; t < (n + 1) / 2 + j; t += NT) {
            const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i + j;
            if (l < n) { const uint64_t x = sk[i], y = sk[l]; if (x > y) { sk[i] = y; sk[l] = x; } }
        }
        __syncthreads();
    }
}

// Sorts g[0..K) in place with the NT threads of one workgroup (K is workgroup-uniform) and ends
// with a workgroup barrier, after which every wave of the workgroup sees the sorted keys (a
// workgroup's waves share their CU's L1).
//   K <= 64:        ONE wave, no LDS: keys are unique, so a key's rank is the number of smaller keys;
//                   every other key is broadcast through SGPRs (v_readlane)
//   K <= LDS_KEYS:  bitonic network in LDS
//   larger:         chunks of LDS_KEYS are sorted in LDS, then only the steps whose partners are
//                   >= LDS_KEYS apart run on the (L2-resident) global slice and every stage is finished
//                   chunk by chunk in LDS again: 3 global passes for 8192 keys instead of 91
// `dst`: where the sorted keys go (g itself = in place; elsewhere: g is left in whatever state the sort needed)
template <int NT, int LDS_KEYS, bool BARRIER_AT_END = true>
__device__ __forceinline__ void ags_sort_tile_keys(uint64_t* g, uint32_t K, uint64_t* sk, int tid, uint64_t* dst) {
    if (K < 2) {
        if (K == 1 && dst != g && tid == 0) dst[0] = g[0];
        return;
    }
    if (K <= 64) {
        if (!BARRIER_AT_END && tid >= 64) return; // stand-alone sort kernel: the other waves are done
        if (tid < 64) {
            const uint64_t mine = (tid < (int)K) ? g[tid] : ~0ull;
            const uint32_t lo = (uint32_t)mine, hi = (uint32_t)(mine >> 32);
            uint32_t rank = 0;
            for (uint32_t j = 0; j < K; ++j) {
                const uint64_t other = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hi, j) << 32) |
                                       (uint32_t)__builtin_amdgcn_readlane((int)lo, j);
                rank += (other < mine) ? 1u : 0u;
            }
            if (tid < (int)K) dst[rank] = mine; // all loads happened before the first store (same wave)
        }
    } else if (K <= (uint32_t)LDS_KEYS) {
        for (uint32_t t = tid; t < K; t += NT) sk[ags_sk(t)] = g[t];
        __syncthreads();
        ags_bitonic_regs<NT>(sk, K, tid);
        for (uint32_t t = tid; t < K; t += NT) dst[t] = sk[ags_sk(t)];
    } else {
        constexpr uint32_t C = LDS_KEYS; // power of two
        // phase 1: every chunk sorted ascending on its own
        for (uint32_t c0 = 0; c0 < K; c0 += C) {
            const uint32_t n = (K - c0 < C) ? K - c0 : C;
            for (uint32_t t = tid; t < n; t += NT) sk[t] = g[c0 + t];
            __syncthreads();
            ags_bitonic<NT>(sk, n, tid);
            for (uint32_t t = tid; t < n; t += NT) g[c0 + t] = sk[t];
            __syncthreads();
        }
        // phase 2: merge stages k = 2C, 4C, ...: wide steps in global memory, the rest per chunk in LDS
        volatile uint64_t* a = (volatile uint64_t*)g;
        uint32_t Kp = 1;
        while (Kp < K) Kp <<= 1;
        const uint32_t half = Kp >> 1;
        for (uint32_t k = 2 * C; k <= Kp; k <<= 1) {
            const uint32_t hk = k >> 1;
            for (uint32_t t = tid; t < half; t += NT) { // mirrored first step of the stage
                const uint32_t blk = t / hk, off = t % hk;
                const uint32_t i = blk * k + off, l = blk * k + (k - 1 - off);
                if (l < K) { const uint64_t x = a[i], y = a[l]; if (x > y) { a[i] = y; a[l] = x; } }
            }
            __syncthreads();
            for (uint32_t j = k >> 2; j >= C; j >>= 1) {
                for (uint32_t t = tid; t < half; t += NT) {
                    const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i + j;
                    if (l < K) { const uint64_t x = a[i], y = a[l]; if (x > y) { a[i] = y; a[l] = x; } }
                }
                __syncthreads();
            }
            for (uint32_t c0 = 0; c0 < K; c0 += C) {
                const uint32_t n = (K - c0 < C) ? K - c0 : C;
                for (uint32_t t = tid; t < n; t += NT) sk[t] = g[c0 + t];
                __syncthreads();
                ags_bitonic_finish_lds<NT>(sk, n, C >> 1, tid);
                for (uint32_t t = tid; t < n; t += NT) g[c0 + t] = sk[t];
                __syncthreads();
            }
        }
        if (dst != g)   // (every thread's own stores above are visible to it; other threads' through the barriers)
            for (uint32_t t = tid; t < K; t += NT) dst[t] = a[t];
    }
    if (BARRIER_AT_END) __syncthreads();
}

// Transposed wave reduction of 16 per-lane values (gfx950 v_permlane32_swap / v_permlane16_swap):
// every step halves the lanes a value is spread over while PACKING two values into one register, all
// the way down: 64 -> 32 lanes (8 swap32+add), 32 -> 16 (4 swap16+add), then inside the 16-lane rows
// 16 -> 8 and 8 -> 4 lanes by adding a row-rotated copy and keeping the two registers' results in
// different banks (4-lane groups) of ONE register, and finally two quad-permute adds.  On return every
// lane holds the wave total of exactly one value - value ags_reduce16_field(lane), the same in all
// four lanes of a quad - so there is no read-back, no select and no second register to look at:
// ~36 VALU for 16 values instead of 16 x 6 DPP adds + 16 read-backs.
typedef unsigned int ags_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float ags_swap32_add(float a, float b) {
    const ags_u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y); // lanes 0-31: a[l]+a[l+32]; lanes 32-63: b[l-32]+b[l]
}
__device__ __forceinline__ float ags_swap16_add(float a, float b) {
    const ags_u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y); // rows: a0+a1, b0+b1, a2+a3, b2+b3
}
// x + (x rotated right by ROT lanes within its row of 16: lane i reads lane (i - ROT) mod 16)
template <int ROT>
__device__ __forceinline__ float ags_row_ror_add(float x) { return x + ags_dpp_f<0x120 + ROT>(x); }
// lanes of the banks (4-lane groups of every row) selected by BANKS take `b`, the others keep `a`
template <int BANKS>
__device__ __forceinline__ float ags_bank_merge(float a, float b) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, a), __builtin_bit_cast(int, b),
                                                                 0xE4 /* quad_perm [0,1,2,3] */, 0xF, BANKS, false));
}
__device__ __forceinline__ int ags_reduce16_field(int lane) {
    // row r of the swap16 stage holds value 4k + {0,2,1,3}[r]; quad c of a row ends up with k = {0,2,1,3}[c]
    const int row = lane >> 4, quad = (lane >> 2) & 3;
    const int pr = (row == 1) ? 2 : (row == 2) ? 1 : row, pq = (quad == 1) ? 2 : (quad == 2) ? 1 : quad;
    return 4 * pq + pr;
}
// returns the wave total of value ags_reduce16_field(lane)
__device__ __forceinline__ float ags_wave_reduce16(const float v[16], int lane) {
    float r[8], s[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = ags_swap32_add(v[2 * i], v[2 * i + 1]);
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = ags_swap16_add(r[2 * k], r[2 * k + 1]);
    // 16 -> 8 lanes: lanes 0-7 of a row keep s[0] / s[2], lanes 8-15 take s[1] / s[3]
    const float t0 = ags_bank_merge<0xC>(ags_row_ror_add<8>(s[0]), ags_row_ror_add<8>(s[1]));
    const float t1 = ags_bank_merge<0xC>(ags_row_ror_add<8>(s[2]), ags_row_ror_add<8>(s[3]));
    // 8 -> 4 lanes: banks 0 and 2 keep t0 (lane i + lane i+4), banks 1 and 3 take t1 (lane i + lane i-4)
    float u = ags_bank_merge<0xA>(ags_row_ror_add<12>(t0), ags_row_ror_add<4>(t1));
    u += ags_dpp_f<0xB1>(u); // quad_perm [1,0,3,2]
    u += ags_dpp_f<0x4E>(u); // quad_perm [2,3,0,1]
    return u;
}

// block -> tile map: block b runs on XCD b%8 (observed dispatch order); give every XCD one
// contiguous run of row-major tiles so neighbouring tiles share that XCD's L2. Bijective.
__device__ __forceinline__ int ags_xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7, x = b & 7, k = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}
// first tile and number of tiles of XCD x's band under that map
__device__ __forceinline__ int ags_xcd_band(int x, int n, int& size) {
    const int q = n >> 3, r = n & 7;
    size = q + (x < r ? 1 : 0);
    return x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
}
// Block -> tile map of the blend kernels.
//  * scan-based binning modes (tile_cap == 0): block b blends tile ags_xcd_remap(b), whose sorted ids are
//    ranges[tile] = [begin, end) of the key array.
//  * AGS_BIN_DIRECT (tile_cap > 0): block b blends SLOT ags_xcd_remap(b).  ags_k_tile_sort_direct gave every tile
//    a slot inside its XCD band by DESCENDING list length and left ranges[slot] = {tile, count} and the tile's
//    sorted keys at slot * tile_cap.  Blocks are dispatched in index order, so every XCD starts its heaviest tiles
//    first and fills the tail of the launch with the lightest (longest-processing-time-first: without it a launch
//    ends when the last late-starting heavy tile does) - and a block knows where its ids are before it knows which
//    tile it has: the header and the first 64 ids are requested together (one dependent load level less than
//    slot -> tile -> range -> ids).
__device__ __forceinline__ int ags_slot_tile(const uint2* __restrict__ ranges, int t, uint32_t tile_cap, uint2& rg) {
    const uint2 h = ranges[t];
    if (tile_cap == 0) { rg = h; return t; }
    rg.x = (uint32_t)t * tile_cap; rg.y = rg.x + h.y;
    return (int)h.x;
}
// The blend kernels run ONE WAVE PER WORKGROUP (64 threads): a workgroup's LDS and wave slots are released only when
// its LAST wave ends, and the quadrants of a tile differ a lot in work - with the four quadrant waves of a tile in one
// workgroup a third of the wave slots sat empty behind each tile's slowest quadrant (measured: 3.9 of 6 resident waves
// per SIMD in the blend backward).  Wave-workgroup b of a launch of 8 * ags_wave_blocks_per_xcd(n, W) serves slot
// band(b % 8) + (b / 8) / W, quadrant group (b / 8) % W: the W waves of a tile stay on one XCD (its L2 holds the tile's
// records) and are dispatched back to back; the few workgroups beyond a shorter band exit at once.
static inline int ags_wave_blocks_per_xcd(int num_tiles, int waves_per_tile) { return ((num_tiles + 7) / 8) * waves_per_tile; }
__device__ __forceinline__ bool ags_wave_block(int b, int n, int W, int& slot, int& wave) {
    int size;
    const int band0 = ags_xcd_band(b & 7, n, size), j = b >> 3;
    slot = band0 + j / W; wave = j % W;
    return j < size * W;
}
#endif
