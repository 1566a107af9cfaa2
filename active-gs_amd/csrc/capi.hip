// extern "C" entry points of libags_raster.so (include/ags_raster.h).
#include "ags_internal.h"

#define AGS_VERSION 102   // 102: ags_forward_batch_loss, AGS_BWD_BF16X3

static int ags_check_launch() { return hipGetLastError() == hipSuccess ? AGS_OK : AGS_E_LAUNCH; }

// ---- optional stage timing (process-global; see ags_profile_enable in the header)
#include <algorithm>
#include <vector>
namespace {
struct StageEvents { std::vector<hipEvent_t> a, b; int used = 0; };
StageEvents g_prof[AGS_NUM_STAGES];
int g_prof_slots = 0;
struct StageScope {
    hipStream_t s; int stage; bool on;
    StageScope(int st, hipStream_t stream) : s(stream), stage(st), on(false) {
        if (g_prof_slots > 0 && g_prof[st].used < g_prof_slots) { on = true; hipEventRecord(g_prof[st].a[g_prof[st].used], s); }
    }
    ~StageScope() { if (on) { hipEventRecord(g_prof[stage].b[g_prof[stage].used], s); g_prof[stage].used++; } }
};
} // namespace

extern "C" {

size_t ags_workspace_bytes(int32_t n, int32_t h, int32_t w, int64_t max_instances) {
    if (n < 0 || h <= 0 || w <= 0) return 0;
    return ags_make_layout(n, h, w, max_instances).total;
}

int ags_workspace_init(const AgsWorkspace* ws, int32_t n, int32_t h, int32_t w, ags_stream_t stream) {
    if (!ws || !ws->ptr || n < 0 || h <= 0 || w <= 0 || ws->max_instances < 1) return AGS_E_INVALID;
    const AgsLayout L = ags_make_layout(n, h, w, ws->max_instances);
    if (ws->bytes < L.total) return AGS_E_WORKSPACE;
    if (hipMemsetAsync((char*)ws->ptr + L.status, 0, L.clear_bytes, (hipStream_t)stream) != hipSuccess) return AGS_E_LAUNCH;
    return AGS_OK;
}

int ags_workspace_init_batch(const AgsWorkspace* ws, int32_t views, int32_t n, int32_t h, int32_t w, ags_stream_t stream) {
    if (!ws || !ws->ptr || views < 1 || n < 0 || h <= 0 || w <= 0 || ws->max_instances < 1) return AGS_E_INVALID;
    const AgsLayout L = ags_make_layout(n, h, w, ws->max_instances);
    if (ws->bytes / (size_t)views < L.total) return AGS_E_WORKSPACE;
    // one launch for all views (a memset per view: ~7 us of host time each; one hipMemset2DAsync: ~170 us on ROCm 7.2 - round 4)
    if (views == 1) {
        if (hipMemsetAsync((char*)ws->ptr + L.status, 0, L.clear_bytes, (hipStream_t)stream) != hipSuccess) return AGS_E_LAUNCH;
        return AGS_OK;
    }
    ags_launch_clear_regions((char*)ws->ptr, L.total, L.status, L.clear_bytes, views, (hipStream_t)stream);
    return ags_check_launch();
}

int ags_workspace_region(int32_t n, int32_t h, int32_t w, int64_t max_instances, int32_t binning_mode, int32_t region,
                         size_t* offset, size_t* bytes) {
    if (n < 0 || h <= 0 || w <= 0 || max_instances < 1 || !offset || !bytes) return AGS_E_INVALID;
    const AgsLayout L = ags_make_layout(n, h, w, max_instances);
    const size_t P = (size_t)h * w;
    switch (region) {
        case AGS_REGION_FINAL_T: *offset = L.final_T; *bytes = P * 4; return AGS_OK;
        case AGS_REGION_N_CONTRIB: *offset = L.n_contrib; *bytes = P * 4; return AGS_OK;
        case AGS_REGION_GEOM: *offset = L.geom; *bytes = (size_t)n * sizeof(AgsGeom); return AGS_OK;
        case AGS_REGION_RANGES: *offset = L.ranges; *bytes = (size_t)L.num_tiles * 8; return AGS_OK;
        case AGS_REGION_KEYS:
            if (binning_mode == AGS_BIN_RADIX) return AGS_E_INVALID;
            *offset = binning_mode == AGS_BIN_DIRECT ? L.keys1 : L.keys0; *bytes = (size_t)L.cap * 8; return AGS_OK;
        case AGS_REGION_IDS:
            if (binning_mode != AGS_BIN_RADIX) return AGS_E_INVALID;
            *offset = (ags_sort_passes(L.num_tiles) & 1) ? L.vals1 : L.vals0; *bytes = (size_t)L.cap * 4; return AGS_OK;
        default: return AGS_E_INVALID;
    }
}

int ags_workspace_discard_pass(const AgsWorkspace* ws, int32_t n, int32_t h, int32_t w, ags_stream_t stream) {
    if (!ws || !ws->ptr || n < 0 || h <= 0 || w <= 0 || ws->max_instances < 1) return AGS_E_INVALID;
    const AgsLayout L = ags_make_layout(n, h, w, ws->max_instances);
    if (ws->bytes < L.total) return AGS_E_WORKSPACE;
    // everything behind the status block: partial sums / digit totals, tile ranges, tile counters, tile fills
    if (hipMemsetAsync((char*)ws->ptr + L.totals, 0, L.clear_bytes - L.totals, (hipStream_t)stream) != hipSuccess) return AGS_E_LAUNCH;
    // ... and the discarded pass's early overflow note (the blend kernel that would have cleared it never ran)
    if (hipMemsetAsync((char*)ws->ptr + L.status + 4 * AGS_STATUS_EARLY, 0, 4, (hipStream_t)stream) != hipSuccess) return AGS_E_LAUNCH;
    return AGS_OK;
}

static const AgsViewStride kOneView = {0, 0, 0, 1};

static bool ags_tuning_ok(const AgsWorkspace* ws) {
    const AgsTuning* t = ws->tuning;
    if (!t) return true;
    if (t->bwd_reduce < AGS_BWD_F32 || t->bwd_reduce > AGS_BWD_BF16X3) return false;
    return t->render_slots == 0 || t->render_slots == 1 || t->render_slots == 2 || t->render_slots == 4;
}

int ags_forward(const AgsCamera* cam, const AgsGaussians* in, const AgsImages* out,
                const AgsPerGaussian* pg, const AgsWorkspace* ws, ags_stream_t stream) {
    if (!cam || !in || !out || !pg || !ws || !ws->ptr) return AGS_E_INVALID;
    if (in->n < 0 || cam->image_height <= 0 || cam->image_width <= 0) return AGS_E_INVALID;
    if (!cam->viewmatrix || !cam->projmatrix || !cam->bg) return AGS_E_INVALID;
    if (in->n > 0 && !pg->radii) return AGS_E_INVALID;
    if (!out->rgb || !out->normal || !out->depth || !out->opacity || !out->confidence) return AGS_E_INVALID;
    if (in->n > 0 && (cam->want_stats || cam->config) && (!pg->count || (!pg->importance && (cam->want_stats != AGS_STATS_SEEN || cam->config))))
        return AGS_E_INVALID;
    if (in->n > 0 && (!in->means3D || !in->scales || !in->rotations || !in->opacities || !in->colors || !in->confidences))
        return AGS_E_INVALID;
    if (ws->max_instances < 1 || ws->max_instances > 0xFFFFFFFFll || !ags_tuning_ok(ws)) return AGS_E_INVALID;
    if (ws->binning_mode != AGS_BIN_TILE_SORT && ws->binning_mode != AGS_BIN_RADIX && ws->binning_mode != AGS_BIN_DIRECT)
        return AGS_E_INVALID;
    const AgsLayout L = ags_layout_for(in->n, cam->image_height, cam->image_width, ws);
    if (ws->bytes < L.total) return AGS_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char* base = (char*)ws->ptr;
    const AgsFrame F = ags_make_frame(cam);
    const bool radix = ws->binning_mode == AGS_BIN_RADIX, direct = ws->binning_mode == AGS_BIN_DIRECT;
    if (direct && ags_direct_tile_cap(L) < 1) return AGS_E_WORKSPACE;   // not even one key slot per tile
    // tile-sort mode is self-cleaning (see ags_workspace_init); the radix path re-zeroes its digit
    // totals and tile ranges every pass
    // (status words 0-3 and everything behind the status block; the sticky words 4-5 of AgsStatus survive)
    if (radix || in->n == 0) {
        if (hipMemsetAsync(base + L.status, 0, 16, s) != hipSuccess) return AGS_E_LAUNCH;
        if (hipMemsetAsync(base + L.totals, 0, L.clear_bytes - L.totals, s) != hipSuccess) return AGS_E_LAUNCH;
    }
    if (in->n > 0) {
        { StageScope t(AGS_STAGE_PREPROCESS, s);
          ags_launch_preprocess(F, *cam, *in, base, L, *pg, radix ? 0 : direct ? 2 : 1, kOneView, s); }
        if (direct && ws->early_status_host && ws->early_status_event) {
            // the key slots are taken: whether a tile's list outgrew its range is known NOW (AgsStatus.early_tile_need)
            if (hipMemcpyAsync(ws->early_status_host, base + L.status, sizeof(AgsStatus), hipMemcpyDeviceToHost, s) != hipSuccess) return AGS_E_LAUNCH;
            if (hipEventRecord((hipEvent_t)ws->early_status_event, s) != hipSuccess) return AGS_E_LAUNCH;
        }
        { StageScope t(AGS_STAGE_BINNING, s);
          if (radix) ags_launch_binning(F, *in, base, L, s);
          else if (direct) ags_launch_direct_sort(base, L, kOneView, s);
          else ags_launch_tile_binning(F, *in, base, L, kOneView, s); }
    }
    { StageScope t(AGS_STAGE_RENDER_FWD, s);
      ags_launch_render_fwd(F, *cam, base, L, ags_sorted_ids(base, L, ws->binning_mode), *out, *pg, kOneView,
                            direct && in->n > 0, s); }
    return ags_check_launch();
}

size_t ags_forward_batch_workspace_bytes(int32_t views, int32_t n, int32_t h, int32_t w, int64_t max_instances) {
    if (views < 1 || n < 0 || h <= 0 || w <= 0) return 0;
    return (size_t)views * ags_make_layout(n, h, w, max_instances).total;
}

static int ags_forward_batch_impl(const AgsCamera* cam, int32_t views, const AgsGaussians* in, const AgsImages* out,
                                  const AgsPerGaussian* pg, const AgsWorkspace* ws, const AgsLossFuse* loss, ags_stream_t stream) {
    if (!cam || !in || !out || !pg || !ws || !ws->ptr || views < 1 || views > 65535) return AGS_E_INVALID;
    if (in->n <= 0 || cam->image_height <= 0 || cam->image_width <= 0) return AGS_E_INVALID;
    if (!cam->viewmatrix || !cam->projmatrix || !cam->bg || !pg->radii) return AGS_E_INVALID;
    if (!out->rgb || !out->normal || !out->depth || !out->opacity || !out->confidence) return AGS_E_INVALID;
    if ((cam->want_stats || cam->config) && (!pg->count || (!pg->importance && (cam->want_stats != AGS_STATS_SEEN || cam->config))))
        return AGS_E_INVALID;
    if (!in->means3D || !in->scales || !in->rotations || !in->opacities || !in->colors || !in->confidences) return AGS_E_INVALID;
    if (ws->max_instances < 1 || ws->max_instances > 0xFFFFFFFFll || !ags_tuning_ok(ws)) return AGS_E_INVALID;
    if (ws->binning_mode != AGS_BIN_TILE_SORT && ws->binning_mode != AGS_BIN_DIRECT) return AGS_E_INVALID;
    const AgsLayout L = ags_layout_for(in->n, cam->image_height, cam->image_width, ws);
    if (ws->bytes < (size_t)views * L.total) return AGS_E_WORKSPACE;
    const bool direct = ws->binning_mode == AGS_BIN_DIRECT;
    if (direct && ags_direct_tile_cap(L) < 1) return AGS_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char* base = (char*)ws->ptr;
    const AgsFrame F = ags_make_frame(cam);
    AgsViewStride vs;
    vs.ws = (long long)L.total; vs.px = (long long)cam->image_height * cam->image_width; vs.n = in->n; vs.views = views;
    ags_launch_preprocess(F, *cam, *in, base, L, *pg, direct ? 2 : 1, vs, s);
    if (direct) ags_launch_direct_sort(base, L, vs, s); else ags_launch_tile_binning(F, *in, base, L, vs, s);
    ags_launch_render_fwd(F, *cam, base, L, ags_sorted_ids(base, L, ws->binning_mode), *out, *pg, vs, direct, s, loss);
    return ags_check_launch();
}

int ags_forward_batch(const AgsCamera* cam, int32_t views, const AgsGaussians* in, const AgsImages* out,
                      const AgsPerGaussian* pg, const AgsWorkspace* ws, ags_stream_t stream) {
    return ags_forward_batch_impl(cam, views, in, out, pg, ws, nullptr, stream);
}

int ags_forward_batch_loss(const AgsCamera* cam, int32_t views, const AgsGaussians* in, const AgsImages* out,
                           const AgsPerGaussian* pg, const AgsWorkspace* ws, const AgsLossEpilogue* loss, ags_stream_t stream) {
    if (!cam || !loss || !loss->cfg || cam->want_stats || cam->config) return AGS_E_INVALID;
    const AgsLossConfig* c = loss->cfg;
    if (c->image_height != cam->image_height || c->image_width != cam->image_width || c->batch_total < 1 ||
        views < 1 || 5 + 2 * (views - 1) >= c->accum_stride)
        return AGS_E_INVALID;
    if (!loss->gt_rgb || !loss->gt_depth || !loss->n_img || !loss->d_rgb || !loss->d_depth || !loss->msum || !loss->accum)
        return AGS_E_INVALID;
    const float hw = (float)c->image_height * (float)c->image_width;
    AgsLossFuse lf;
    lf.gt_rgb = loss->gt_rgb; lf.gt_depth = loss->gt_depth; lf.n_img = loss->n_img; lf.d_rgb = loss->d_rgb;
    lf.d_depth = loss->d_depth; lf.msum = loss->msum; lf.accum = loss->accum;
    lf.gt_index = (const long long*)c->gt_frame_index; lf.accum_stride = c->accum_stride;
    // the constants exactly as ags_k_loss_stage1 forms them: w / ((float)B * 3.f * (float)HW), w / ((float)B * (float)HW)
    lf.k_rgb = c->w_rgb / ((float)c->batch_total * 3.f * hw);
    lf.k_depth = c->w_depth / ((float)c->batch_total * hw);
    return ags_forward_batch_impl(cam, views, in, out, pg, ws, &lf, stream);
}

static bool ags_adam_map_shaped(const AgsAdamTensors* t) {   // the five map tensors: 3n, 3n, 4n, n, 3n
    const int64_t n = t->numel[3];
    return t->numel[0] == 3 * n && t->numel[1] == 3 * n && t->numel[2] == 4 * n && t->numel[4] == 3 * n && n <= 0x7FFFFFFF;
}
static int ags_adam_check(const AgsAdamTensors* t, bool need_grad = true) {
    if (!t) return AGS_E_INVALID;
    for (int k = 0; k < 5; ++k) {
        if (t->numel[k] < 0) return AGS_E_INVALID;
        if (t->numel[k] > 0 && (!t->param[k] || (need_grad && !t->grad[k]))) return AGS_E_INVALID;
        if (t->numel[k] > 0 && !t->state_rows && (!t->exp_avg[k] || !t->exp_avg_sq[k])) return AGS_E_INVALID;
    }
    if (t->state_rows && !ags_adam_map_shaped(t)) return AGS_E_INVALID;
    return AGS_OK;
}

} // extern "C"

// ags_backward and its software-pipelined form (next_* != NULL: the per-Gaussian kernel also runs the per-Gaussian
// stage of the NEXT forward pass)
static int ags_backward_impl(const AgsCamera* cam, const AgsGaussians* in, const AgsImages* fwd,
                             const AgsPerGaussian* pg, const AgsImageGrads* dout, const AgsGaussianGrads* din,
                             const AgsWorkspace* ws, const AgsCamera* next_cam, const AgsPerGaussian* next_pg,
                             const AgsWorkspace* next_ws, int32_t rows_hint, ags_stream_t stream) {
    if (!cam || !in || !fwd || !pg || !dout || !din || !ws || !ws->ptr || !ags_tuning_ok(ws)) return AGS_E_INVALID;
    if (in->n == 0) return next_cam ? AGS_E_INVALID : AGS_OK;
    if (!pg->radii || !fwd->depth || !fwd->opacity) return AGS_E_INVALID;
    if (!din->defer_rows) {   // all five gradient arrays, or - fused optimiser step, overwrite mode - none at all
        const int have = (din->d_means3D != nullptr) + (din->d_scales != nullptr) + (din->d_rotations != nullptr) +
                         (din->d_opacities != nullptr) + (din->d_colors != nullptr);
        if (have != 5 && !(have == 0 && din->fused_adam && din->accumulate == 0 && !din->d_means2D)) return AGS_E_INVALID;
    } else if (next_cam || din->fused_adam || din->pack_segment) {
        return AGS_E_INVALID;   // a deferred view runs its blend backward only; the optimiser / exchange tail belongs to ags_backward_rows
    }
    const AgsLayout L = ags_layout_for(in->n, cam->image_height, cam->image_width, ws);
    if (ws->bytes < L.total) return AGS_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char* base = (char*)ws->ptr;
    const AgsFrame F = ags_make_frame(cam);
    const AgsIdList vals_sorted = ags_sorted_ids(base, L, ws->binning_mode); // where the forward left them
    // dgeom needs no memset: ags_forward zeroed the records of the visible surfels and every
    // ags_backward leaves them zeroed again
    AgsTick tick = {};
    if (din->adam_clock) {
        tick.clock = (AgsAdamClock*)din->adam_clock;
        for (int k = 0; k < 5; ++k) tick.lr[k] = din->adam_lr[k];
        tick.beta1 = din->adam_beta1; tick.beta2 = din->adam_beta2;
        if (next_cam && din->touched.count) {   // the pipelined per-Gaussian kernel reads the member count of NOW
            tick.rows_count = din->touched.count;
            tick.count_snap = (uint32_t*)(base + L.status) + AGS_STATUS_COUNT_SNAP;
        }
    }
    if (din->fused_adam) {
        if (!din->touched.rows || !din->touched.count || !din->adam_clock || din->accumulate == 2) return AGS_E_INVALID;
        if (int e = ags_adam_check(din->fused_adam, false)) return e;
        const int64_t* ne = din->fused_adam->numel; // the five map tensors: 3n, 3n, 4n, n, 3n
        const int64_t n64 = in->n;
        if (ne[0] != 3 * n64 || ne[1] != 3 * n64 || ne[2] != 4 * n64 || ne[3] != n64 || ne[4] != 3 * n64) return AGS_E_INVALID;
    }
    if (din->pack_segment && (!din->touched.rows || !din->touched.count || din->fused_adam || din->accumulate == 2 ||
                              din->pack_capacity < 0)) return AGS_E_INVALID;
    if (next_cam) {
        // pipelined: single view, single rank, fused Adam on raw parameters, one-pass binning on both workspaces
        // (all of this is checked BEFORE the blend backward is enqueued: a refused call must leave dgeom and the Adam
        // clock exactly as they were)
        if (!next_pg || !next_ws || !next_ws->ptr || !next_pg->radii || !din->fused_adam || !in->raw_params) return AGS_E_INVALID;
        if (next_cam->config) return AGS_E_INVALID;   // (see ags_forward_resume: the prepared pass does not clear the statistics)
        if (!din->fused_adam->state_rows) return AGS_E_INVALID;   // the pipelined kernel reads the interleaved moments
        if (!next_cam->viewmatrix || !next_cam->projmatrix || next_cam->image_height <= 0 || next_cam->image_width <= 0) return AGS_E_INVALID;
        if (ws->binning_mode != AGS_BIN_DIRECT || next_ws->binning_mode != AGS_BIN_DIRECT) return AGS_E_INVALID;
        if (next_ws->max_instances < 1 || next_ws->max_instances > 0xFFFFFFFFll) return AGS_E_INVALID;
        if (!in->means3D || !in->scales || !in->rotations || !in->opacities || !in->colors || !in->confidences) return AGS_E_INVALID;
        // the next pass's row set is the optimiser's (new members are appended while this step's rows are walked)
        if (next_pg->touched.member != din->touched.member || next_pg->touched.rows != din->touched.rows ||
            next_pg->touched.count != din->touched.count) return AGS_E_INVALID;
        const AgsLayout L2 = ags_layout_for(in->n, next_cam->image_height, next_cam->image_width, next_ws);
        if (next_ws->bytes < L2.total) return AGS_E_WORKSPACE;
        if (ags_direct_tile_cap(L2) < 1) return AGS_E_WORKSPACE;
    }
    { StageScope t(AGS_STAGE_RENDER_BWD, s);
      ags_launch_render_bwd(F, *cam, base, L, vals_sorted, *fwd, *dout, tick, kOneView, ws->binning_mode == AGS_BIN_DIRECT, s); }
    if (din->defer_rows) return ags_check_launch();
    if (next_cam) {
        const AgsLayout L2 = ags_layout_for(in->n, next_cam->image_height, next_cam->image_width, next_ws);
        const AgsFrame F2 = ags_make_frame(next_cam);
        StageScope t(AGS_STAGE_PREPROCESS_BWD, s);
        ags_launch_rows_adam_preprocess(F, *cam, *in, base, L, pg->radii, *din, F2, *next_cam, (char*)next_ws->ptr, L2,
                                        next_pg->radii, rows_hint, s);
        return ags_check_launch();
    }
    { StageScope t(AGS_STAGE_PREPROCESS_BWD, s); ags_launch_preprocess_bwd(F, *cam, *in, base, L, pg->radii, *din, kOneView, s); }
    return ags_check_launch();
}

extern "C" {

int ags_backward(const AgsCamera* cam, const AgsGaussians* in, const AgsImages* fwd,
                 const AgsPerGaussian* pg, const AgsImageGrads* dout, const AgsGaussianGrads* din,
                 const AgsWorkspace* ws, ags_stream_t stream) {
    return ags_backward_impl(cam, in, fwd, pg, dout, din, ws, nullptr, nullptr, nullptr, 0, stream);
}

int ags_backward_fused_next(const AgsCamera* cam, const AgsGaussians* in, const AgsImages* fwd,
                            const AgsPerGaussian* pg, const AgsImageGrads* dout, const AgsGaussianGrads* din,
                            const AgsWorkspace* ws, const AgsCamera* next_cam, const AgsPerGaussian* next_pg,
                            const AgsWorkspace* next_ws, int32_t rows_hint, ags_stream_t stream) {
    if (!next_cam || !next_pg || !next_ws || rows_hint < 0) return AGS_E_INVALID;
    return ags_backward_impl(cam, in, fwd, pg, dout, din, ws, next_cam, next_pg, next_ws, rows_hint, stream);
}

int ags_forward_resume(const AgsCamera* cam, const AgsGaussians* in, const AgsImages* out,
                       const AgsPerGaussian* pg, const AgsWorkspace* ws, ags_stream_t stream) {
    if (!cam || !in || !out || !pg || !ws || !ws->ptr) return AGS_E_INVALID;
    if (in->n <= 0 || cam->image_height <= 0 || cam->image_width <= 0) return AGS_E_INVALID;
    if (!cam->viewmatrix || !cam->projmatrix || !cam->bg || !pg->radii) return AGS_E_INVALID;
    if (!out->rgb || !out->normal || !out->depth || !out->opacity || !out->confidence) return AGS_E_INVALID;
    // device-side configuration is not available in the pipelined form: the prepared per-Gaussian stage ran inside
    // ags_k_rows_adam_preprocess, which does not clear importance / count (with config they would accumulate onto stale
    // contents); the optimisation loops that pipeline pass host flags
    if (cam->config) return AGS_E_INVALID;
    if (cam->want_stats && (!pg->importance || !pg->count)) return AGS_E_INVALID;
    if (ws->max_instances < 1 || ws->max_instances > 0xFFFFFFFFll || ws->binning_mode != AGS_BIN_DIRECT || !ags_tuning_ok(ws)) return AGS_E_INVALID;
    const AgsLayout L = ags_layout_for(in->n, cam->image_height, cam->image_width, ws);
    if (ws->bytes < L.total) return AGS_E_WORKSPACE;
    if (ags_direct_tile_cap(L) < 1) return AGS_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char* base = (char*)ws->ptr;
    const AgsFrame F = ags_make_frame(cam);
    { StageScope t(AGS_STAGE_BINNING, s); ags_launch_direct_sort(base, L, kOneView, s); }
    { StageScope t(AGS_STAGE_RENDER_FWD, s);
      ags_launch_render_fwd(F, *cam, base, L, ags_sorted_ids(base, L, ws->binning_mode), *out, *pg, kOneView, true, s); }
    return ags_check_launch();
}

int ags_backward_batch(const AgsCamera* cam, int32_t views, const AgsGaussians* in, const AgsImages* fwd,
                       const AgsPerGaussian* pg, const AgsImageGrads* dout, const AgsGaussianGrads* din,
                       const AgsWorkspace* ws, ags_stream_t stream) {
    if (!cam || !in || !fwd || !pg || !dout || !din || !ws || !ws->ptr || views < 1 || views > 65535) return AGS_E_INVALID;
    if (in->n <= 0 || !pg->radii || !fwd->depth || !fwd->opacity) return AGS_E_INVALID;
    if (!din->defer_rows) {
        if (!din->d_means3D || !din->d_scales || !din->d_rotations || !din->d_opacities || !din->d_colors) return AGS_E_INVALID;
        if (din->accumulate != 2) return AGS_E_INVALID;     // views sum with atomics into a pre-zeroed slab
    }
    if (din->fused_adam || din->pack_segment) return AGS_E_INVALID;
    if ((ws->binning_mode != AGS_BIN_TILE_SORT && ws->binning_mode != AGS_BIN_DIRECT) || !ags_tuning_ok(ws)) return AGS_E_INVALID;
    const AgsLayout L = ags_layout_for(in->n, cam->image_height, cam->image_width, ws);
    if (ws->bytes < (size_t)views * L.total) return AGS_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char* base = (char*)ws->ptr;
    const AgsFrame F = ags_make_frame(cam);
    AgsViewStride vs;
    vs.ws = (long long)L.total; vs.px = (long long)cam->image_height * cam->image_width; vs.n = in->n; vs.views = views;
    AgsTick tick = {};
    if (din->adam_clock) {
        tick.clock = (AgsAdamClock*)din->adam_clock;
        for (int k = 0; k < 5; ++k) tick.lr[k] = din->adam_lr[k];
        tick.beta1 = din->adam_beta1; tick.beta2 = din->adam_beta2;
    }
    ags_launch_render_bwd(F, *cam, base, L, ags_sorted_ids(base, L, ws->binning_mode), *fwd, *dout, tick, vs,
                          ws->binning_mode == AGS_BIN_DIRECT, s);
    if (!din->defer_rows) ags_launch_preprocess_bwd(F, *cam, *in, base, L, pg->radii, *din, vs, s);
    return ags_check_launch();
}

int ags_backward_rows(const AgsViewRef* views, int32_t num_views, const AgsGaussians* in, const AgsGaussianGrads* din,
                      ags_stream_t stream) {
    if (!views || !in || !din || num_views < 1 || num_views > AGS_MAX_ROW_VIEWS) return AGS_E_INVALID;
    if (in->n == 0) return AGS_OK;
    const bool ranged = !din->touched.member && !din->touched.rows && !din->touched.count;   // rows [row_begin, row_end) of the map
    if (ranged) {
        if (din->row_begin < 0 || din->row_end <= din->row_begin || din->row_end > in->n) return AGS_E_INVALID;
        if (din->fused_adam || din->pack_segment || (din->accumulate != 0 && din->accumulate != 1)) return AGS_E_INVALID;   // 1: += (a further group of views)
        if (!din->d_means3D || !din->d_scales || !din->d_rotations || !din->d_opacities || !din->d_colors) return AGS_E_INVALID;
    } else if (!din->touched.member || !din->touched.rows || !din->touched.count || din->accumulate == 2) {
        return AGS_E_INVALID;
    }
    if (!in->means3D || !in->scales || !in->rotations || !in->opacities) return AGS_E_INVALID;
    {
        const int have = (din->d_means3D != nullptr) + (din->d_scales != nullptr) + (din->d_rotations != nullptr) +
                         (din->d_opacities != nullptr) + (din->d_colors != nullptr);
        if (have != 5 && !(have == 0 && (din->fused_adam || din->pack_segment) && !din->d_means2D)) return AGS_E_INVALID;
        if (din->accumulate != 0 && have != 5) return AGS_E_INVALID;
    }
    if (din->fused_adam) {
        if (!din->adam_clock || din->pack_segment) return AGS_E_INVALID;
        if (int e = ags_adam_check(din->fused_adam, false)) return e;
        const int64_t* ne = din->fused_adam->numel;
        const int64_t n64 = in->n;
        if (ne[0] != 3 * n64 || ne[1] != 3 * n64 || ne[2] != 4 * n64 || ne[3] != n64 || ne[4] != 3 * n64) return AGS_E_INVALID;
    }
    if (din->pack_segment && din->pack_capacity < 0) return AGS_E_INVALID;
    AgsRowViews rv;
    rv.views = num_views;
    for (int v = 0; v < num_views; ++v) {
        const AgsViewRef& r = views[v];
        if (!r.cam || !r.radii || !r.ws || !r.ws->ptr || !r.cam->viewmatrix || !r.cam->projmatrix) return AGS_E_INVALID;
        if (r.cam->image_height <= 0 || r.cam->image_width <= 0) return AGS_E_INVALID;
        const AgsLayout L = ags_make_layout(in->n, r.cam->image_height, r.cam->image_width, r.ws->max_instances);
        if (r.ws->bytes < L.total) return AGS_E_WORKSPACE;
        rv.F[v] = ags_make_frame(r.cam);
        rv.V[v] = r.cam->viewmatrix; rv.P[v] = r.cam->projmatrix;
        rv.dgeom[v] = (AgsGeomGrad*)((char*)r.ws->ptr + L.dgeom);
        rv.radii[v] = r.radii;
    }
    { StageScope t(AGS_STAGE_PREPROCESS_BWD, (hipStream_t)stream); ags_launch_rows_multi(rv, *in, *din, (hipStream_t)stream); }
    return ags_check_launch();
}

int ags_read_status(const AgsWorkspace* ws, AgsStatus* host_out, ags_stream_t stream) {
    if (!ws || !ws->ptr || !host_out) return AGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemcpyAsync(host_out, ws->ptr, sizeof(AgsStatus), hipMemcpyDeviceToHost, s) != hipSuccess) return AGS_E_LAUNCH;
    if (hipStreamSynchronize(s) != hipSuccess) return AGS_E_LAUNCH;
    return AGS_OK;
}

int ags_read_status_async(const AgsWorkspace* ws, AgsStatus* host_out, ags_stream_t stream) {
    if (!ws || !ws->ptr || !host_out) return AGS_E_INVALID;
    if (hipMemcpyAsync(host_out, ws->ptr, sizeof(AgsStatus), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return AGS_E_LAUNCH;
    return AGS_OK;
}

int ags_adam_step(const AgsAdamTensors* t, float beta1, float beta2, float eps, int32_t step, ags_stream_t stream) {
    if (ags_adam_check(t) != AGS_OK || step < 1) return AGS_E_INVALID;
    ags_launch_adam(*t, beta1, beta2, eps, step, nullptr, false, (hipStream_t)stream);
    return ags_check_launch();
}

int ags_adam_step_device(const AgsAdamTensors* t, float beta1, float beta2, float eps, void* state,
                         int32_t pre_ticked, ags_stream_t stream) {
    if (ags_adam_check(t) != AGS_OK || !state) return AGS_E_INVALID;
    ags_launch_adam(*t, beta1, beta2, eps, 0, state, pre_ticked != 0, (hipStream_t)stream);
    return ags_check_launch();
}

size_t ags_rows_segment_floats(int32_t capacity) { return capacity < 0 ? 0 : 16 * ((size_t)capacity + 1); }

static bool ags_rows_ok(const AgsRowSet* r) { return r && r->member && r->rows && r->count; }

int ags_rows_pack(const AgsRowSet* rows, float* const grads[5], int32_t capacity, float* segment, ags_stream_t stream) {
    if (!ags_rows_ok(rows) || !grads || !segment || capacity < 0) return AGS_E_INVALID;
    for (int k = 0; k < 5; ++k)
        if (!grads[k]) return AGS_E_INVALID;
    ags_launch_rows_pack(grads, *rows, segment, capacity, (hipStream_t)stream);
    return ags_check_launch();
}

int ags_rows_unpack(const float* segment, int32_t capacity, float* const grads[5], const AgsRowSet* union_rows,
                    ags_stream_t stream) {
    if (!ags_rows_ok(union_rows) || !grads || !segment || capacity < 0) return AGS_E_INVALID;
    for (int k = 0; k < 5; ++k)
        if (!grads[k]) return AGS_E_INVALID;
    ags_launch_rows_unpack(segment, capacity, grads, *union_rows, (hipStream_t)stream);
    return ags_check_launch();
}

int ags_rows_index(const float* segments, int32_t world, int32_t capacity, int32_t* slot_table, const AgsRowSet* union_rows,
                   ags_stream_t stream) {
    if (!ags_rows_ok(union_rows) || !segments || !slot_table || capacity < 0 || world < 1) return AGS_E_INVALID;
    ags_launch_rows_index(segments, ags_rows_segment_floats(capacity), capacity, world, slot_table, *union_rows, (hipStream_t)stream);
    return ags_check_launch();
}

int ags_adam_step_gathered(const AgsAdamTensors* t, const float* segments, int32_t world, int32_t capacity, int32_t* slot_table,
                           float beta1, float beta2, float eps, void* state, int32_t pre_ticked, ags_stream_t stream) {
    if (!t || !state || !segments || !slot_table || capacity < 0 || world < 1 || !ags_rows_ok(&t->touched)) return AGS_E_INVALID;
    for (int k = 0; k < 5; ++k)
        if (t->numel[k] < 0 || (t->numel[k] > 0 && (!t->param[k] || (!t->state_rows && (!t->exp_avg[k] || !t->exp_avg_sq[k]))))) return AGS_E_INVALID;
    if (t->state_rows && !ags_adam_map_shaped(t)) return AGS_E_INVALID;
    ags_launch_adam_gathered(*t, segments, ags_rows_segment_floats(capacity), world, capacity, slot_table, beta1, beta2, eps, state,
                             pre_ticked != 0, (hipStream_t)stream);
    return ags_check_launch();
}

int ags_activate(const AgsActivation* a, float* scales, float* rotations, float* opacities, ags_stream_t stream) {
    if (!a || a->n < 0) return AGS_E_INVALID;
    if (a->n > 0 && (!a->raw_scales || !a->raw_rotations || !a->raw_opacities || !scales || !rotations || !opacities))
        return AGS_E_INVALID;
    ags_launch_activate(*a, scales, rotations, opacities, (hipStream_t)stream);
    return ags_check_launch();
}

int ags_activate_backward(const AgsActivation* a, float* d_scales, float* d_rotations, float* d_opacities,
                          ags_stream_t stream) {
    if (!a || a->n < 0) return AGS_E_INVALID;
    if (a->n > 0 && (!a->raw_scales || !a->raw_rotations || !a->raw_opacities || !d_scales || !d_rotations || !d_opacities))
        return AGS_E_INVALID;
    ags_launch_activate_bwd(*a, d_scales, d_rotations, d_opacities, (hipStream_t)stream);
    return ags_check_launch();
}

static int ags_loss_check(const AgsLossConfig* c, const AgsImages* f) {
    if (!c || !f || c->image_height <= 0 || c->image_width <= 0 || c->batch_total < 1 || !(c->sigma > 0.f) ||
        c->accum_stride < 6)
        return AGS_E_INVALID;
    if (!f->rgb || !f->normal || !f->depth || !f->opacity) return AGS_E_INVALID;
    return AGS_OK;
}

int ags_loss_stage1(const AgsLossConfig* cfg, const AgsImages* fwd, const float* gt_rgb, const float* gt_depth,
                    float* n_img, float* d_rgb, float* d_depth, int32_t* msum, float* accum, int32_t view,
                    int32_t first_view, ags_stream_t stream) {
    if (ags_loss_check(cfg, fwd) != AGS_OK || !gt_rgb || !gt_depth || !n_img || !d_rgb || !d_depth || !msum || !accum ||
        view < 0 || 5 + 2 * (view + (cfg->num_views > 1 ? cfg->num_views - 1 : 0)) >= cfg->accum_stride ||
        cfg->num_views < 0 || cfg->num_views > 65535 || (cfg->num_views > 1 && first_view >= 0))
        return AGS_E_INVALID;
    ags_launch_loss_stage1(*cfg, *fwd, gt_rgb, gt_depth, n_img, d_rgb, d_depth, msum, accum, view, first_view,
                           (hipStream_t)stream);
    return ags_check_launch();
}

int ags_loss_stage2(const AgsLossConfig* cfg, const AgsImages* fwd, const float* n_img, const float* gt_depth,
                    const int32_t* msum, float* d_normal, float* d_depth, float* accum, ags_stream_t stream) {
    if (ags_loss_check(cfg, fwd) != AGS_OK || !n_img || !gt_depth || !msum || !d_normal || !d_depth || !accum)
        return AGS_E_INVALID;
    ags_launch_loss_stage2(*cfg, *fwd, n_img, gt_depth, msum, d_normal, d_depth, accum, (hipStream_t)stream);
    return ags_check_launch();
}

int ags_facade_post(int32_t h, int32_t w, float tanfov_x, float tanfov_y, const float* normal_raw, const float* depth,
                    const float* opacity, float* normal_out, float* d2n_out, ags_stream_t stream) {
    if (h <= 0 || w <= 0 || !(tanfov_x > 0.f) || !(tanfov_y > 0.f) || !depth || !opacity || !d2n_out) return AGS_E_INVALID;
    if ((normal_raw == nullptr) != (normal_out == nullptr)) return AGS_E_INVALID;
    ags_launch_facade_post(1, h, w, tanfov_x, tanfov_y, normal_raw, depth, opacity, normal_out, d2n_out, (hipStream_t)stream);
    return ags_check_launch();
}

int ags_facade_post_batch(int32_t views, int32_t h, int32_t w, float tanfov_x, float tanfov_y, const float* normal_raw,
                          const float* depth, const float* opacity, float* normal_out, float* d2n_out, ags_stream_t stream) {
    if (views < 1 || views > 65535 || h <= 0 || w <= 0 || !(tanfov_x > 0.f) || !(tanfov_y > 0.f) || !depth || !opacity || !d2n_out)
        return AGS_E_INVALID;
    if ((normal_raw == nullptr) != (normal_out == nullptr)) return AGS_E_INVALID;
    ags_launch_facade_post(views, h, w, tanfov_x, tanfov_y, normal_raw, depth, opacity, normal_out, d2n_out, (hipStream_t)stream);
    return ags_check_launch();
}

int ags_facade_post_backward(int32_t h, int32_t w, float tanfov_x, float tanfov_y, const float* normal_raw,
                             const float* depth, const float* opacity, const float* g_normal, const float* g_d2n,
                             float* d_normal_raw, float* d_depth, ags_stream_t stream) {
    if (h <= 0 || w <= 0 || !(tanfov_x > 0.f) || !(tanfov_y > 0.f) || !depth || !opacity) return AGS_E_INVALID;
    if (g_d2n && !d_depth) return AGS_E_INVALID;
    if (d_normal_raw && g_normal && !normal_raw) return AGS_E_INVALID;
    ags_launch_facade_post_bwd(h, w, tanfov_x, tanfov_y, normal_raw, depth, opacity, g_normal, g_d2n, d_normal_raw, d_depth,
                               (hipStream_t)stream);
    return ags_check_launch();
}

int ags_weighted_topk(const float* uniforms, const float* weights, int32_t n, int32_t k, int64_t* out, ags_stream_t stream) {
    if (!uniforms || !weights || !out || n < 1 || n > 8192 || k < 1 || k > n) return AGS_E_INVALID;
    ags_launch_weighted_topk(uniforms, weights, n, k, (long long*)out, (hipStream_t)stream);
    return ags_check_launch();
}

int ags_stage_frames(int32_t views, int32_t h, int32_t w, const int64_t* frame_index, const float* all_view,
                     const float* all_proj, const float* all_rgb, const float* all_depth, float* dst_view, float* dst_proj,
                     float* dst_rgb, float* dst_depth, int32_t* msum, ags_stream_t stream) {
    if (views < 1 || views > 65535 || h <= 0 || w <= 0 || ((long long)h * w) % 4 != 0) return AGS_E_INVALID;
    if (!frame_index || !all_view || !all_proj || !dst_view || !dst_proj) return AGS_E_INVALID;
    if ((dst_rgb == nullptr) != (dst_depth == nullptr)) return AGS_E_INVALID;
    if (dst_rgb && (!all_rgb || !all_depth)) return AGS_E_INVALID;
    ags_launch_stage_frames(views, h * w, (const long long*)frame_index, all_view, all_proj, all_rgb, all_depth, dst_view,
                            dst_proj, dst_rgb, dst_depth, msum, (hipStream_t)stream);
    return ags_check_launch();
}

int ags_loss_finish(const AgsLossConfig* cfg, float* accum, int32_t views, const int64_t* frame_index,
                    float* frame_error, float* total_loss, ags_stream_t stream) {
    if (!cfg || !accum || views < 0 || cfg->accum_stride < 4 + 2 * views || cfg->accum_stride > 256 ||
        cfg->batch_total < 1 || cfg->image_height <= 0 || cfg->image_width <= 0)
        return AGS_E_INVALID;
    ags_launch_loss_finish(*cfg, accum, views, (const long long*)frame_index, frame_error, total_loss, (hipStream_t)stream);
    return ags_check_launch();
}

int ags_zero_many(int32_t count, void* const* regions, const size_t* bytes, ags_stream_t stream) {
    if (count < 0 || count > AGS_ZERO_MANY_MAX) return AGS_E_INVALID;
    if (count == 0) return AGS_OK;
    if (!regions || !bytes) return AGS_E_INVALID;
    for (int k = 0; k < count; ++k)
        if ((bytes[k] && !regions[k]) || ((uintptr_t)regions[k] & 15) || (bytes[k] & 3)) return AGS_E_INVALID;
    ags_launch_zero_many(count, regions, bytes, (hipStream_t)stream);
    return ags_check_launch();
}

int ags_loss_finish_next(const AgsLossConfig* cfg, float* accum, int32_t views, int64_t* frame_index, float* frame_error,
                         float* total_loss, const AgsNextIteration* next, ags_stream_t stream) {
    if (!cfg || !accum || !next || !frame_index || views < 0 || cfg->accum_stride < 4 + 2 * views || cfg->accum_stride > 256 ||
        cfg->batch_total < 1 || cfg->image_height <= 0 || cfg->image_width <= 0 ||
        ((long long)cfg->image_height * cfg->image_width) % 4 != 0)
        return AGS_E_INVALID;
    if (next->views < 1 || next->views > 4096 || !next->all_view || !next->all_proj || !next->dst_view || !next->dst_proj || !next->msum)
        return AGS_E_INVALID;
    if (next->uniforms && next->k > 0 &&
        (!frame_error || next->n_weights < 1 || next->n_weights > 8192 || next->k > next->n_weights || next->first_random < 0 ||
         next->first_random + next->k > next->views))
        return AGS_E_INVALID;
    ags_launch_loss_finish_next(*cfg, accum, views, (long long*)frame_index, frame_error, total_loss, *next, (hipStream_t)stream);
    return ags_check_launch();
}

int ags_profile_enable(int32_t slots) {
    if (slots < 0) return AGS_E_INVALID;
    for (int st = 0; st < AGS_NUM_STAGES; ++st) {
        for (hipEvent_t e : g_prof[st].a) hipEventDestroy(e);
        for (hipEvent_t e : g_prof[st].b) hipEventDestroy(e);
        g_prof[st].a.clear(); g_prof[st].b.clear(); g_prof[st].used = 0;
        for (int k = 0; k < slots; ++k) {
            hipEvent_t x, y;
            if (hipEventCreate(&x) != hipSuccess || hipEventCreate(&y) != hipSuccess) return AGS_E_LAUNCH;
            g_prof[st].a.push_back(x); g_prof[st].b.push_back(y);
        }
    }
    g_prof_slots = slots;
    return AGS_OK;
}

int ags_profile_read(int32_t stage, float* avg_ms, float* median_ms, int32_t* samples) {
    if (stage < 0 || stage >= AGS_NUM_STAGES || !avg_ms || !median_ms || !samples) return AGS_E_INVALID;
    StageEvents& e = g_prof[stage];
    double sum = 0.0;
    std::vector<float> all;
    for (int k = 0; k < e.used; ++k) {
        float ms = 0.f;
        if (hipEventSynchronize(e.b[k]) != hipSuccess) return AGS_E_LAUNCH;
        if (hipEventElapsedTime(&ms, e.a[k], e.b[k]) != hipSuccess) return AGS_E_LAUNCH;
        sum += ms;
        all.push_back(ms);
    }
    std::sort(all.begin(), all.end());
    *samples = e.used;
    *avg_ms = e.used ? (float)(sum / e.used) : 0.f;
    *median_ms = e.used ? all[all.size() / 2] : 0.f;
    e.used = 0;
    return AGS_OK;
}

int ags_smooth_depth(int32_t h, int32_t w, const float* depth, float* out, int32_t d, float sigma_color,
                     float sigma_space, ags_stream_t stream) {
    if (h <= 0 || w <= 0 || !depth || !out || d < 1 || !(sigma_color > 0.f) || !(sigma_space > 0.f)) return AGS_E_INVALID;
    if (d / 2 >= h || d / 2 >= w || d > 31) return AGS_E_INVALID; // reflect-101 needs radius < size; LDS tile bound
    ags_launch_bilateral(h, w, depth, out, d, sigma_color, sigma_space, (hipStream_t)stream);
    return ags_check_launch();
}

int ags_densify_candidates(const AgsKeyframe* f, const float* depth_smooth, const AgsDensifyPred* pred,
                           float error_thres, const AgsCandidates* out, ags_stream_t stream) {
    if (!f || !depth_smooth || !pred || !out) return AGS_E_INVALID;
    if (f->image_height <= 0 || f->image_width <= 0 || !f->rgb || !f->depth || !f->intrinsic_inv || !f->extrinsic)
        return AGS_E_INVALID;
    if (!out->means || !out->rotations || !out->harmonics || !out->select) return AGS_E_INVALID;
    const int have = (pred->rgb != nullptr) + (pred->depth != nullptr) + (pred->opacity != nullptr);
    if (have != 0 && have != 3) return AGS_E_INVALID;
    ags_launch_candidates(*f, depth_smooth, *pred, error_thres, *out, (hipStream_t)stream);
    return ags_check_launch();
}

size_t ags_voxel_select_bytes(int32_t n) { return ags_voxel_bytes(n < 0 ? 0 : n); }
int ags_voxel_select(int32_t n, const float* points, int32_t* select, float voxel_size, void* ws, size_t ws_bytes,
                     ags_stream_t stream) {
    if (n < 0 || !(voxel_size > 0.f)) return AGS_E_INVALID;
    if (n == 0) return AGS_OK;
    if (!points || !select || !ws) return AGS_E_INVALID;
    if (ws_bytes < ags_voxel_bytes(n)) return AGS_E_WORKSPACE;
    ags_launch_voxel_select(n, points, select, voxel_size, ws, (hipStream_t)stream);
    return ags_check_launch();
}

int ags_prune_keep(int32_t n, const float* prune_mask, const float* raw_opacities, float min_opacity, int32_t* keep,
                   ags_stream_t stream) {
    if (n < 0) return AGS_E_INVALID;
    if (n == 0) return AGS_OK;
    if (!raw_opacities || !keep) return AGS_E_INVALID;
    ags_launch_prune_keep(n, prune_mask, raw_opacities, min_opacity, keep, (hipStream_t)stream);
    return ags_check_launch();
}

int ags_view_stats_update(int32_t n, const float* means, const float* raw_rotations, const float* campos, float far,
                          const int32_t* newest_count, int32_t use_view_distribution, float* view_supports,
                          float* view_means, float* view_scores, ags_stream_t stream) {
    if (n < 0 || !(far > 0.f)) return AGS_E_INVALID;
    if (n == 0) return AGS_OK;
    if (!newest_count || !view_supports) return AGS_E_INVALID;
    if (use_view_distribution && (!means || !raw_rotations || !campos || !view_means || !view_scores)) return AGS_E_INVALID;
    ags_launch_view_stats(n, means, raw_rotations, campos, far, newest_count, use_view_distribution ? 1 : 0, view_supports,
                          view_means, view_scores, (hipStream_t)stream);
    return ags_check_launch();
}

int ags_confidences(int32_t n, const float* view_supports, const float* view_means, const float* view_scores,
                    int32_t use_view_distribution, float* out, ags_stream_t stream) {
    if (n < 0) return AGS_E_INVALID;
    if (n == 0) return AGS_OK;
    if (!out || (use_view_distribution ? (!view_means || !view_scores) : !view_supports)) return AGS_E_INVALID;
    ags_launch_confidences(n, view_supports, view_means, view_scores, use_view_distribution ? 1 : 0, out, (hipStream_t)stream);
    return ags_check_launch();
}

size_t ags_compact_plan_bytes(int32_t n) { return ags_compact_bytes(n < 0 ? 0 : n); }
int ags_compact_plan(int32_t n, const int32_t* keep, int32_t* dst_index, int32_t* total, void* scratch,
                     size_t scratch_bytes, ags_stream_t stream) {
    if (n < 0 || !total) return AGS_E_INVALID;
    if (n == 0) return hipMemsetAsync(total, 0, 4, (hipStream_t)stream) == hipSuccess ? AGS_OK : AGS_E_LAUNCH;
    if (!keep || !dst_index || !scratch) return AGS_E_INVALID;
    if (scratch_bytes < ags_compact_bytes(n)) return AGS_E_WORKSPACE;
    ags_launch_compact_plan(n, keep, dst_index, total, scratch, (hipStream_t)stream);
    return ags_check_launch();
}
int ags_compact_rows(int32_t n, int32_t width, const int32_t* dst_index, const float* src, float* dst,
                     ags_stream_t stream) {
    if (n < 0 || width < 1) return AGS_E_INVALID;
    if (n == 0) return AGS_OK;
    if (!dst_index || !src || !dst) return AGS_E_INVALID;
    ags_launch_compact_rows(n, width, dst_index, src, dst, (hipStream_t)stream);
    return ags_check_launch();
}

static bool ags_map_arrays_ok(const AgsMapArrays* a) {
    return a && a->means && a->scales && a->rotations && a->opacities && a->harmonics && a->view_scores && a->view_supports &&
           a->view_means;
}

int ags_map_append(int32_t candidates, const int32_t* dst_index, const AgsCandidates* c, float new_z_scale,
                   const AgsMapArrays* first_new_row, ags_stream_t stream) {
    if (candidates < 0) return AGS_E_INVALID;
    if (candidates == 0) return AGS_OK;
    if (!dst_index || !c || !c->means || !c->rotations || !c->harmonics || !ags_map_arrays_ok(first_new_row)) return AGS_E_INVALID;
    ags_launch_map_append(candidates, dst_index, *c, new_z_scale, *first_new_row, (hipStream_t)stream);
    return ags_check_launch();
}

int ags_map_compact(int32_t n, const int32_t* dst_index, const AgsMapArrays* src, const AgsMapArrays* dst, ags_stream_t stream) {
    if (n < 0) return AGS_E_INVALID;
    if (n == 0) return AGS_OK;
    if (!dst_index || !ags_map_arrays_ok(src) || !ags_map_arrays_ok(dst)) return AGS_E_INVALID;
    if (src->means == dst->means || src->scales == dst->scales || src->rotations == dst->rotations) return AGS_E_INVALID;   // not in place
    ags_launch_map_compact(n, dst_index, *src, *dst, (hipStream_t)stream);
    return ags_check_launch();
}

#ifdef AGS_TIMELINE
// experiment builds only: buffer of 8 kernels x AGS_TL_WAVES waves x 8 uint64 (see AGS_TL in ags_internal.h)
int ags_debug_timeline(void* device_buffer) {
    ags_tl_set_preprocess(device_buffer); ags_tl_set_binning(device_buffer); ags_tl_set_render(device_buffer);
    return AGS_OK;
}
#endif

const char* ags_error_string(int code) {
    switch (code) {
        case AGS_OK: return "ok";
        case AGS_E_INVALID: return "invalid argument";
        case AGS_E_WORKSPACE: return "workspace too small (see ags_workspace_bytes)";
        case AGS_E_LAUNCH: return "HIP enqueue failed";
        default: return "unknown error";
    }
}

int ags_version(void) { return AGS_VERSION; }
}
