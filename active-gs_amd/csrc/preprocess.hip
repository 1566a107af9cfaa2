// Per-Gaussian kernels: F1 (cull + project + conic + tile rect) and B2+B3 fused
// (conic/mean2D/depth/normal gradients -> means3D, scales, rotations, opacity, colour).
// One lane per Gaussian; the camera matrices are wave-uniform (scalar loads).
// Counterpart of the preprocess / preprocess-backward stages listed in SURVEY.md §2.3.

#include "ags_internal.h"

AGS_TL_DEFINE(preprocess)

// Coalesced access to the reference's (N,3) row-major arrays: the block's 256 rows are 3072
// contiguous bytes, moved as 192 x 16-byte lane accesses and transposed through LDS (a stride-3
// LDS access is conflict-free: 3 is coprime with the 32 banks), instead of three 12-byte-stride
// dword accesses per lane.  `rows` = valid rows of this block; the ragged last block falls back
// to dword accesses.
__device__ __forceinline__ void ags_load_rows3(const float* __restrict__ base, int first_row, int rows, float* lds,
                                               float out[3]) {
    const int t = threadIdx.x;
    const float* src = base + (size_t)first_row * 3;
    if (rows == AGS_PRE_THREADS && ((uintptr_t)src & 15) == 0) { // block-uniform
        if (t < AGS_PRE_THREADS * 3 / 4) reinterpret_cast<float4*>(lds)[t] = reinterpret_cast<const float4*>(src)[t];
    } else {
        for (int k = t; k < rows * 3; k += AGS_PRE_THREADS) lds[k] = src[k];
    }
    __syncthreads();
    out[0] = lds[3 * t]; out[1] = lds[3 * t + 1]; out[2] = lds[3 * t + 2];
    __syncthreads();
}

template <bool ACCUMULATE>
__device__ __forceinline__ void ags_store_rows3(float* __restrict__ base, int first_row, int rows, float* lds,
                                                const float v[3]) {
    const int t = threadIdx.x;
    float* dst = base + (size_t)first_row * 3;
    lds[3 * t] = v[0]; lds[3 * t + 1] = v[1]; lds[3 * t + 2] = v[2];
    __syncthreads();
    if (rows == AGS_PRE_THREADS && ((uintptr_t)dst & 15) == 0) {
        if (t < AGS_PRE_THREADS * 3 / 4) {
            float4 x = reinterpret_cast<const float4*>(lds)[t];
            if (ACCUMULATE) {
                const float4 o = reinterpret_cast<const float4*>(dst)[t];
                x.x += o.x; x.y += o.y; x.z += o.z; x.w += o.w;
            }
            reinterpret_cast<float4*>(dst)[t] = x;
        }
    } else {
        for (int k = t; k < rows * 3; k += AGS_PRE_THREADS) dst[k] = ACCUMULATE ? dst[k] + lds[k] : lds[k];
    }
    __syncthreads();
}

// Activations of /root/reference/mapping/gaussian_map.py:529-549, applied in registers when the
// caller hands over raw map parameters (AgsGaussians.raw_params).
__device__ __forceinline__ void ags_activate_inplace(const AgsGaussians& in, float sc[3], float q[4], float& opacity,
                                                     float raw_v[3], float& qinv) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        raw_v[k] = in.scale_factor * expf(sc[k]);
        sc[k] = fminf(fmaxf(raw_v[k], 0.f), in.max_scale);
    }
    qinv = 1.0f / fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] *= qinv;
    opacity = 1.0f / (1.0f + expf(-opacity));
}

// The block's rows of three (N,3) arrays at once: all three global reads are issued before the first wait and ONE
// barrier separates the LDS writes from the transposed reads (three ags_load_rows3 calls cost six barriers and
// three load latencies in a row - this kernel is a latency chain).
__device__ __forceinline__ void ags_load_rows3x3(const float* __restrict__ b0, const float* __restrict__ b1,
                                                 const float* __restrict__ b2, int first_row, int rows, float* lds,
                                                 float o0[3], float o1[3], float o2[3]) {
    const int t = threadIdx.x;
    const float *s0 = b0 + (size_t)first_row * 3, *s1 = b1 + (size_t)first_row * 3, *s2 = b2 + (size_t)first_row * 3;
    float *l0 = lds, *l1 = lds + 3 * AGS_PRE_THREADS, *l2 = lds + 6 * AGS_PRE_THREADS;
    if (rows == AGS_PRE_THREADS && (((uintptr_t)s0 | (uintptr_t)s1 | (uintptr_t)s2) & 15) == 0) { // block-uniform
        if (t < AGS_PRE_THREADS * 3 / 4) {
            const float4 a = reinterpret_cast<const float4*>(s0)[t], b = reinterpret_cast<const float4*>(s1)[t],
                         c = reinterpret_cast<const float4*>(s2)[t];
            reinterpret_cast<float4*>(l0)[t] = a; reinterpret_cast<float4*>(l1)[t] = b; reinterpret_cast<float4*>(l2)[t] = c;
        }
    } else {
        for (int k = t; k < rows * 3; k += AGS_PRE_THREADS) { l0[k] = s0[k]; l1[k] = s1[k]; l2[k] = s2[k]; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) { o0[k] = l0[3 * t + k]; o1[k] = l1[3 * t + k]; o2[k] = l2[3 * t + k]; }
}

// `early`: AgsStatus.early_tile_need - a lane whose slot lies beyond its tile's range notes slot + 1 there (atomicMax;
// only overflowing passes ever touch it), so that the pass's overflow is known behind THIS kernel, not behind the blend
struct AgsDirectEmit { uint64_t* keys; uint32_t tile_cap; uint32_t* partial; uint32_t tc_stride; uint32_t* early; };
#ifndef AGS_DIRECT_AGG_LARGE
#define AGS_DIRECT_AGG_LARGE 2   // slot atomics of images with more than AGS_AGG_MAX_TILES tiles: 2 = adaptive grouping, 0 = per lane (probe builds)
#endif
#ifndef AGS_DIRECT_AGG_MIN_PAIRS
#define AGS_DIRECT_AGG_MIN_PAIRS 64    // (surfel, candidate tile) pairs of a wave from which the grouping is tried at all
#endif

// EMIT: 0 = nothing, 1 = count the tiles a surfel reaches (tile-sort binning), 2 = AGS_BIN_DIRECT: take a slot in
// the tile's own key range with ONE returning atomic and write the (depth | id) key at once
// Body of the forward per-Gaussian kernel for workgroup `bx` of the pass (pointers already moved to the view).
// SKIP_MEMBERS (the software-pipelined optimisation step, ags_k_rows_adam_preprocess): rows that are already members of
// the row set are left alone - the lane that applied their Adam update runs this stage for them from its registers.
template <int EMIT, bool AGG, bool SKIP_MEMBERS>
__device__ __forceinline__ void ags_preprocess_block(
    const AgsFrame& F, const float* __restrict__ Vp, const float* __restrict__ Pp, const AgsGaussians& in,
    AgsGeom* __restrict__ geom, uint32_t* __restrict__ tiles, ushort4* __restrict__ rect,
    int* __restrict__ radii, uint32_t* __restrict__ block_sums, uint32_t* __restrict__ block_vis,
    uint32_t* __restrict__ tile_count, float4* __restrict__ dgeom, const AgsRowSet& touched, const AgsDirectEmit& direct,
    const int bx) {
    __shared__ uint32_t wsum[AGS_PRE_THREADS / 64], wvis[AGS_PRE_THREADS / 64];
    __shared__ AgsEmitRec emit[EMIT ? AGS_PRE_THREADS : 1];
    __shared__ __attribute__((aligned(16))) float rows3[9 * AGS_PRE_THREADS];
    float V[16], P[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { V[k] = Vp[k]; P[k] = Pp[k]; }
    [[maybe_unused]] const int tl_w = bx * (AGS_PRE_THREADS / 64) + (threadIdx.x >> 6);
    AGS_TL(0, tl_w, 0);
    const int first = bx * AGS_PRE_THREADS;
    const int i = first + threadIdx.x;
    const int rows = min(AGS_PRE_THREADS, in.n - first);
    uint32_t cnt = 0, vis = 0;
    uint32_t rx0 = 0, ry0 = 0, rwd = 1;
    AgsGeom g;
    g.mx = g.my = g.ca = g.cb = g.cc = g.o = 0.f; g.dc = 0.f;
    float p[3], sc[3], col[3];
    // the row's other inputs are requested before the barrier inside the transposed load
    const int ic = i < in.n ? i : in.n - 1;
    const float4 q4 = reinterpret_cast<const float4*>(in.rotations)[ic];
    float opacity = in.opacities[ic];
    const float conf = in.confidences[ic];
    const int was_member = touched.member ? touched.member[ic] : 1;   // requested with the inputs, not behind the projection
    ags_load_rows3x3(in.means3D, in.scales, in.colors, first, rows, rows3, p, sc, col);
    AGS_TL(0, tl_w, 1);
    if (i < in.n && !(SKIP_MEMBERS && was_member != 0)) {
        float q[4] = {q4.x, q4.y, q4.z, q4.w};
        if (in.raw_params) { float rv[3], qi; ags_activate_inplace(in, sc, q, opacity, rv, qi); }
        int radius = 0, rc[4];
        if (ags_preprocess_fwd(F, V, P, p, sc, q, opacity, col, conf, 0.f, 0.f, g, radius, rc)) {
            float4* dst = reinterpret_cast<float4*>(geom + i);
            dst[0] = make_float4(g.mx, g.my, g.ca, g.cb);
            dst[1] = make_float4(g.cc, g.o, g.dc, g.gx);
            dst[2] = make_float4(g.gy, g.r, g.g, g.b);
            dst[3] = make_float4(g.nx, g.ny, g.nz, g.conf);
            // the gradient record the blend backward accumulates into starts at zero (no memset pass)
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            dgeom[4 * (size_t)i + 0] = z4; dgeom[4 * (size_t)i + 1] = z4;
            dgeom[4 * (size_t)i + 2] = z4; dgeom[4 * (size_t)i + 3] = z4;
            if (EMIT != 2) rect[i] = make_ushort4((unsigned short)rc[0], (unsigned short)rc[1], (unsigned short)rc[2], (unsigned short)rc[3]);
            cnt = (uint32_t)((rc[2] - rc[0]) * (rc[3] - rc[1]));
            rx0 = (uint32_t)rc[0]; ry0 = (uint32_t)rc[1]; rwd = (uint32_t)(rc[2] - rc[0]);
            vis = 1;
        }
        radii[i] = radius;
        if (EMIT != 2) tiles[i] = cnt;
    }
    AGS_TL(0, tl_w, 2);
    if (touched.member) { // sticky row set of the optimisation loop: insert first-time-visible surfels
        // (after the first few steps of a keyframe nothing is new and this is one sparse read)
        const bool fresh = vis && was_member == 0 && atomicExch(&touched.member[i], 1) == 0;
        const unsigned long long mask = __ballot(fresh);
        if (mask) { // wave-uniform
            const int lane = threadIdx.x & 63;
            int base = 0;
            if (lane == (int)__builtin_ctzll(mask)) base = atomicAdd(touched.count, (int)__builtin_popcountll(mask));
            base = __shfl(base, (int)__builtin_ctzll(mask));
            if (fresh) touched.rows[base + (int)__builtin_popcountll(mask & ((1ull << lane) - 1ull))] = i;
        }
    }
    AGS_TL(0, tl_w, 3);
    if (EMIT == 1)  // tile-sort binning: how many surfels can reach each tile
        ags_emit_tiles_balanced(emit + (threadIdx.x & ~63), cnt, rx0, ry0, rwd, 0u, g, F.tiles_x,
                                [&](bool hit, uint32_t t, uint32_t, int) {
                                    ags_wave_agg_inc<AGG, false>(tile_count, t, hit);
                                });
    const uint32_t ws = ags_wave_sum_u32(cnt), wv = ags_wave_sum_u32(vis);
    if (EMIT == 2) { // one-pass binning: the tile's counter hands out the slot, the key is written at once
        const uint32_t wave_first = (uint32_t)(first + (threadIdx.x & ~63));
        // images of many tiles: lanes that ask the same counter share an atomic only in waves that are DENSE with
        // candidate tiles (a mapper-grown map: the wave's 64 rows are neighbours in space and most are visible - hundreds of
        // pairs on a handful of tiles); a sparse wave (a random-order map: a few visible rows, ~30 pairs, all on different
        // tiles) takes the plain per-lane path without looking
        const bool dense = AGS_DIRECT_AGG_LARGE != 0 && ws >= AGS_DIRECT_AGG_MIN_PAIRS;     // wave-uniform
        ags_emit_tiles_balanced(emit + (threadIdx.x & ~63), cnt, rx0, ry0, rwd, __float_as_uint(g.dc), g, F.tiles_x,
                                [&](bool hit, uint32_t t, uint32_t depth_bits, int owner_lane) {
                                    uint32_t got;
                                    if (AGG) got = ags_wave_agg_inc<1, true>(tile_count, t * direct.tc_stride, hit);
                                    else if (dense) got = ags_wave_agg_inc<2, true>(tile_count, t * direct.tc_stride, hit);
                                    else got = ags_wave_agg_inc<0, true>(tile_count, t * direct.tc_stride, hit);
                                    if (hit && got < direct.tile_cap)
                                        direct.keys[(size_t)t * direct.tile_cap + got] =
                                            ((uint64_t)depth_bits << 32) | (wave_first + (uint32_t)owner_lane);
                                    else if (hit) atomicMax(direct.early, got + 1u);
                                });
    }
    AGS_TL(0, tl_w, 4);
    if (EMIT == 2) { // no block-level reduction (and no barrier): one spread atomic per wave that shows anything
        if ((threadIdx.x & 63) == 0 && wv) atomicAdd(&direct.partial[AGS_PART(bx, AGS_PART_VIS)], wv);
    } else {
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { wsum[wave] = ws; wvis[wave] = wv; }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t a = 0, b = 0;
#pragma unroll
            for (int k = 0; k < AGS_PRE_THREADS / 64; ++k) { a += wsum[k]; b += wvis[k]; }
            block_sums[bx] = a;
            block_vis[bx] = b;  // summed by the (single-workgroup) scan kernel: no fan-in atomics
        }
    }
    AGS_TL(0, tl_w, 5);
    AGS_TL_VAL(0, tl_w, 6, ws | ((unsigned long long)wv << 32));
    AGS_TL_VAL(0, tl_w, 7, (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)(__builtin_amdgcn_s_getreg(63508) & 15) << 32));
}

template <int EMIT, bool AGG>
__global__ __launch_bounds__(AGS_PRE_THREADS) void ags_k_preprocess(
    AgsFrame F, const float* __restrict__ Vp, const float* __restrict__ Pp, AgsGaussians in,
    AgsGeom* __restrict__ geom, uint32_t* __restrict__ tiles, ushort4* __restrict__ rect,
    int* __restrict__ radii, uint32_t* __restrict__ block_sums, uint32_t* __restrict__ block_vis,
    uint32_t* __restrict__ tile_count, float4* __restrict__ dgeom, AgsRowSet touched, AgsDirectEmit direct,
    float* __restrict__ zero_importance, int* __restrict__ zero_count, AgsViewStride vs) {
    ags_frame_flags(F);
    { // batched forward: this workgroup's view // (offsets are 0 for a single view)
        const size_t wo = (size_t)blockIdx.y * (size_t)vs.ws;
        Vp += 16 * blockIdx.y; Pp += 16 * blockIdx.y;
        radii += (size_t)blockIdx.y * (size_t)vs.n;
        if (zero_importance) { // device-side configuration (AgsCamera.config): the statistics start at zero without a caller-side memset
            const int i = blockIdx.x * AGS_PRE_THREADS + threadIdx.x;
            if (i < in.n) {
                zero_importance[(size_t)blockIdx.y * (size_t)vs.n + i] = 0.f;
                zero_count[(size_t)blockIdx.y * (size_t)vs.n + i] = 0;
            }
        }
        AGS_WS_SHIFT(geom, wo); AGS_WS_SHIFT(tiles, wo); AGS_WS_SHIFT(rect, wo); AGS_WS_SHIFT(block_sums, wo);
        AGS_WS_SHIFT(block_vis, wo); AGS_WS_SHIFT(tile_count, wo); AGS_WS_SHIFT(dgeom, wo);
        if (EMIT == 2) { AGS_WS_SHIFT(direct.keys, wo); AGS_WS_SHIFT(direct.partial, wo); AGS_WS_SHIFT(direct.early, wo); }
    }
    ags_preprocess_block<EMIT, AGG, false>(F, Vp, Pp, in, geom, tiles, rect, radii, block_sums, block_vis, tile_count, dgeom,
                                           touched, direct, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------
// The forward per-Gaussian stage of a BATCH of views (ags_forward_batch: the views of a training iteration, a planner's
// candidate views, the keyframes of a prune pass) with the rows' inputs loaded and activated ONCE for a group of views.
// In ags_k_preprocess blockIdx.y is the view: every view's workgroup loads the same 256 rows (60 bytes each) and runs the
// same activations (three exp, a normalisation, a sigmoid) before its own cull - eleven times per training iteration of
// the mapper, a hundred times per planner step - and most of those row-views end at the cull.  Here blockIdx.y is a GROUP
// of consecutive views: the workgroup keeps its rows in registers and walks the group's views - view transform and cull,
// and only for the rows a view shows the exact stage (the same ags_preprocess_fwd on the same inputs: bit-identical
// records), the record / radius writes and the wave-balanced key emission into THAT view's workspace.  One-pass binning
// only (no block-level reductions: nothing between two views of the loop but a wave barrier).
template <bool AGG>
__global__ __launch_bounds__(AGS_PRE_THREADS) void ags_k_preprocess_views(
    AgsFrame F, const float* __restrict__ Vp0, const float* __restrict__ Pp0, AgsGaussians in,
    AgsGeom* __restrict__ geom0, int* __restrict__ radii0, uint32_t* __restrict__ tile_count0, float4* __restrict__ dgeom0,
    AgsRowSet touched, AgsDirectEmit direct0, float* __restrict__ zero_importance, int* __restrict__ zero_count,
    AgsViewStride vs, int views_per_group) {
    ags_frame_flags(F);
    __shared__ AgsEmitRec emit[AGS_PRE_THREADS];
    __shared__ __attribute__((aligned(16))) float rows3[9 * AGS_PRE_THREADS];
    const int bx = (int)blockIdx.x;
    const int first = bx * AGS_PRE_THREADS;
    const int i = first + threadIdx.x;
    const int rows = min(AGS_PRE_THREADS, in.n - first);
    float p[3], sc[3], col[3];
    const int ic = i < in.n ? i : in.n - 1;
    const float4 q4 = reinterpret_cast<const float4*>(in.rotations)[ic];
    float opacity = in.opacities[ic];
    const float conf = in.confidences[ic];
    int was_member = touched.member ? touched.member[ic] : 1;
    ags_load_rows3x3(in.means3D, in.scales, in.colors, first, rows, rows3, p, sc, col);
    float q[4] = {q4.x, q4.y, q4.z, q4.w};
    if (in.raw_params) { float rv[3], qi; ags_activate_inplace(in, sc, q, opacity, rv, qi); }
    const int v0 = (int)blockIdx.y * views_per_group;
    const int v1 = min(v0 + views_per_group, vs.views);
    const uint32_t wave_first = (uint32_t)(first + (threadIdx.x & ~63));
    for (int v = v0; v < v1; ++v) {          // (workgroup-uniform)
        const size_t wo = (size_t)v * (size_t)vs.ws;
        const float* Vp = Vp0 + 16 * v;
        const float* Pp = Pp0 + 16 * v;
        int* radii = radii0 + (size_t)v * (size_t)vs.n;
        AgsGeom* geom = geom0; uint32_t* tile_count = tile_count0; float4* dgeom = dgeom0;
        AgsDirectEmit direct = direct0;
        AGS_WS_SHIFT(geom, wo); AGS_WS_SHIFT(tile_count, wo); AGS_WS_SHIFT(dgeom, wo);
        AGS_WS_SHIFT(direct.keys, wo); AGS_WS_SHIFT(direct.partial, wo); AGS_WS_SHIFT(direct.early, wo);
        float V[16], P[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { V[k] = Vp[k]; P[k] = Pp[k]; }
        if (zero_importance && i < in.n) {
            zero_importance[(size_t)v * (size_t)vs.n + i] = 0.f;
            zero_count[(size_t)v * (size_t)vs.n + i] = 0;
        }
        uint32_t cnt = 0, vis = 0;
        uint32_t rx0 = 0, ry0 = 0, rwd = 1;
        AgsGeom g;
        g.mx = g.my = g.ca = g.cb = g.cc = g.o = 0.f; g.dc = 0.f;
        if (i < in.n) {
            int radius = 0, rc[4];
            if (ags_preprocess_fwd(F, V, P, p, sc, q, opacity, col, conf, 0.f, 0.f, g, radius, rc)) {
                float4* dst = reinterpret_cast<float4*>(geom + i);
                dst[0] = make_float4(g.mx, g.my, g.ca, g.cb);
                dst[1] = make_float4(g.cc, g.o, g.dc, g.gx);
                dst[2] = make_float4(g.gy, g.r, g.g, g.b);
                dst[3] = make_float4(g.nx, g.ny, g.nz, g.conf);
                const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                dgeom[4 * (size_t)i + 0] = z4; dgeom[4 * (size_t)i + 1] = z4;
                dgeom[4 * (size_t)i + 2] = z4; dgeom[4 * (size_t)i + 3] = z4;
                cnt = (uint32_t)((rc[2] - rc[0]) * (rc[3] - rc[1]));
                rx0 = (uint32_t)rc[0]; ry0 = (uint32_t)rc[1]; rwd = (uint32_t)(rc[2] - rc[0]);
                vis = 1;
            }
            radii[i] = radius;
        }
        if (touched.member) { // sticky row set of the optimisation loop: a row is inserted by the first view that shows it
            const bool fresh = vis && was_member == 0 && atomicExch(&touched.member[i], 1) == 0;
            if (vis) was_member = 1;
            const unsigned long long mask = __ballot(fresh);
            if (mask) { // wave-uniform
                const int lane = threadIdx.x & 63;
                int base = 0;
                if (lane == (int)__builtin_ctzll(mask)) base = atomicAdd(touched.count, (int)__builtin_popcountll(mask));
                base = __shfl(base, (int)__builtin_ctzll(mask));
                if (fresh) touched.rows[base + (int)__builtin_popcountll(mask & ((1ull << lane) - 1ull))] = i;
            }
        }
        const uint32_t ws = ags_wave_sum_u32(cnt), wv = ags_wave_sum_u32(vis);
        const bool dense = AGS_DIRECT_AGG_LARGE != 0 && ws >= AGS_DIRECT_AGG_MIN_PAIRS;     // wave-uniform
        ags_emit_tiles_balanced(emit + (threadIdx.x & ~63), cnt, rx0, ry0, rwd, __float_as_uint(g.dc), g, F.tiles_x,
                                [&](bool hit, uint32_t t, uint32_t depth_bits, int owner_lane) {
                                    uint32_t got;
                                    if (AGG) got = ags_wave_agg_inc<1, true>(tile_count, t * direct.tc_stride, hit);
                                    else if (dense) got = ags_wave_agg_inc<2, true>(tile_count, t * direct.tc_stride, hit);
                                    else got = ags_wave_agg_inc<0, true>(tile_count, t * direct.tc_stride, hit);
                                    if (hit && got < direct.tile_cap)
                                        direct.keys[(size_t)t * direct.tile_cap + got] =
                                            ((uint64_t)depth_bits << 32) | (wave_first + (uint32_t)owner_lane);
                                    else if (hit) atomicMax(direct.early, got + 1u);
                                });
        if ((threadIdx.x & 63) == 0 && wv) atomicAdd(&direct.partial[AGS_PART(bx, AGS_PART_VIS)], wv);
    }
}

// ---------------------------------------------------------------------------------------
// The forward per-Gaussian stage for LARGE maps of which a view shows little (one-pass binning, raw parameters: the
// trainers' configuration 4 / 5 shape - 1.5 M / 5 M surfels, 11-20 % visible, in random order).  In ags_k_preprocess a
// lane owns a row from load to key emission: with one row in nine visible nearly every wave still has a visible lane and
// runs the ~900-instruction projection at a tenth of its lanes (config 5: ~110 of the kernel's 215 us are that).  Here
// the workgroup's 512 rows are CULLED first with the mean alone - view depth, projected centre, and a radius bound that
// needs no other input (scales are clamped to max_scale, the rotation is normalised in-kernel: lambda_max <=
// max_scale^2 |J|_F^2 |A|_F^2 + 0.92) - the survivors (the visible rows plus a margin of ~130 px around the image) are
// compacted, and the exact stage (the same ags_preprocess_fwd: bit-identical records) runs on full waves, gathering
// the survivors' other inputs by row.  Key emission is balanced over the WORKGROUP's survivors.
#define AGS_CULL_ROWS 512
#define AGS_CULL_MIN_N (1 << 20)     // below this the plain kernel's single pass is the shorter chain
__global__ __launch_bounds__(AGS_PRE_THREADS) void ags_k_preprocess_cull(
    AgsFrame F, const float* __restrict__ Vp, const float* __restrict__ Pp, AgsGaussians in,
    AgsGeom* __restrict__ geom, int* __restrict__ radii, uint32_t* __restrict__ tile_count, float4* __restrict__ dgeom,
    AgsRowSet touched, AgsDirectEmit direct, float* __restrict__ zero_importance, int* __restrict__ zero_count,
    AgsViewStride vs) {
    ags_frame_flags(F);
    {
        const size_t wo = (size_t)blockIdx.y * (size_t)vs.ws;
        Vp += 16 * blockIdx.y; Pp += 16 * blockIdx.y;
        radii += (size_t)blockIdx.y * (size_t)vs.n;
        if (zero_importance) { zero_importance += (size_t)blockIdx.y * (size_t)vs.n; zero_count += (size_t)blockIdx.y * (size_t)vs.n; }
        AGS_WS_SHIFT(geom, wo); AGS_WS_SHIFT(tile_count, wo); AGS_WS_SHIFT(dgeom, wo);
        AGS_WS_SHIFT(direct.keys, wo); AGS_WS_SHIFT(direct.partial, wo); AGS_WS_SHIFT(direct.early, wo);
    }
    constexpr int R = AGS_CULL_ROWS, NT = AGS_PRE_THREADS;
    __shared__ __attribute__((aligned(16))) float lmeans[3 * R];
    __shared__ unsigned short cand[R];
    __shared__ AgsEmitRec emit[NT];
    __shared__ uint32_t wtot[NT / 64], wcnt[2][NT / 64], s_total;
    float V[16], P[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { V[k] = Vp[k]; P[k] = Pp[k]; }
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int first = blockIdx.x * R;
    const int rows = min(R, in.n - first);
    {   // the block's means: 6144 contiguous bytes, coalesced, kept in LDS for the exact stage
        const float* src = in.means3D + (size_t)first * 3;
        if (rows == R && ((uintptr_t)src & 15) == 0) {
            for (int k = t; k < R * 3 / 4; k += NT) reinterpret_cast<float4*>(lmeans)[k] = reinterpret_cast<const float4*>(src)[k];
        } else {
            for (int k = t; k < rows * 3; k += NT) lmeans[k] = src[k];
        }
    }
    __syncthreads();
    // ---- the cull: conservative (never drops a row the exact stage would keep)
    const float limx = AGS_FRUSTUM_CLAMP * F.tanfovx, limy = AGS_FRUSTUM_CLAMP * F.tanfovy;
    float a2 = 0.f;   // |A|_F^2, A = the view matrix's 3x3 part
#pragma unroll
    for (int k = 0; k < 3; ++k) a2 += V[k * 4 + 0] * V[k * 4 + 0] + V[k * 4 + 1] * V[k * 4 + 1] + V[k * 4 + 2] * V[k * 4 + 2];
    const float smax = in.max_scale * F.scale_mod;
    const float cb = 1.0001f * smax * smax * a2 * (F.fx * F.fx * (1.f + limx * limx) + F.fy * F.fy * (1.f + limy * limy));
    bool is_cand[R / NT];
#pragma unroll
    for (int u = 0; u < R / NT; ++u) {
        const int r = t + u * NT, i = first + r;
        bool c = false;
        if (r < rows) {
            const float x = lmeans[3 * r], y = lmeans[3 * r + 1], z = lmeans[3 * r + 2];
            const float tz = ags_affine(V, 2, x, y, z);
            if (tz > AGS_NEAR_CULL) {
                const float phx = ags_affine(P, 0, x, y, z), phy = ags_affine(P, 1, x, y, z), phw = ags_affine(P, 3, x, y, z);
                const float pw = 1.0f / (phw + 1e-7f);
                const float mx = ((phx * pw + 1.0f) * F.W - 1.0f) * 0.5f, my = ((phy * pw + 1.0f) * F.H - 1.0f) * 0.5f;
                const float rb = 3.0f * sqrtf(cb / (tz * tz) + 0.92f) + 2.0f;       // >= ceil(3 sqrt(lambda_max)) + 1
                // the exact rect [ (m - r) / 16, (m + r + 15) / 16 ) clamped to the grid is empty unless all four hold
                c = (mx + rb >= 0.f) && (mx - rb < (float)(AGS_TILE * F.tiles_x)) && (my + rb >= 0.f) &&
                    (my - rb < (float)(AGS_TILE * F.tiles_y));
                c = c || !(mx == mx) || !(my == my);                                 // (never drop on a NaN: the exact stage decides)
            }
            if (!c) radii[i] = 0;
            if (zero_importance) { zero_importance[i] = 0.f; zero_count[i] = 0; }
        }
        is_cand[u] = c;
    }
    // ---- compaction: survivors' local row numbers, in row order
    uint32_t pre[R / NT];
#pragma unroll
    for (int u = 0; u < R / NT; ++u) {
        const unsigned long long m = __ballot(is_cand[u]);
        pre[u] = (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[u][wave] = (uint32_t)__builtin_popcountll(m);
    }
    __syncthreads();
    uint32_t ncand = 0;
    {
        uint32_t base = 0;
#pragma unroll
        for (int u = 0; u < R / NT; ++u) {
#pragma unroll
            for (int k = 0; k < NT / 64; ++k) {
                if (k == wave && is_cand[u]) cand[base + pre[u]] = (unsigned short)(t + u * NT);
                base += wcnt[u][k];
            }
        }
        ncand = base;    // workgroup-uniform
    }
    __syncthreads();
    // ---- the exact stage on the survivors + workgroup-balanced key emission, 256 survivors at a time
    uint32_t vis_total = 0;
    for (uint32_t c0 = 0; c0 < ncand; c0 += NT) {
        const bool have = c0 + t < ncand;
        const int r = have ? (int)cand[c0 + t] : 0, i = first + r;
        uint32_t cnt = 0, vis = 0, rx0 = 0, ry0 = 0, rwd = 1;
        AgsGeom g;
        g.mx = g.my = g.ca = g.cb = g.cc = g.o = 0.f; g.dc = 0.f;
        int was_member = 1;
        if (have) {
            const float p[3] = {lmeans[3 * r], lmeans[3 * r + 1], lmeans[3 * r + 2]};
            float sc[3] = {in.scales[3 * (size_t)i], in.scales[3 * (size_t)i + 1], in.scales[3 * (size_t)i + 2]};
            const float4 q4 = reinterpret_cast<const float4*>(in.rotations)[i];
            float q[4] = {q4.x, q4.y, q4.z, q4.w};
            float opacity = in.opacities[i];
            const float col[3] = {in.colors[3 * (size_t)i], in.colors[3 * (size_t)i + 1], in.colors[3 * (size_t)i + 2]};
            const float conf = in.confidences[i];
            was_member = touched.member ? touched.member[i] : 1;
            { float rv[3], qi; ags_activate_inplace(in, sc, q, opacity, rv, qi); }
            int radius = 0, rc[4];
            if (ags_preprocess_fwd(F, V, P, p, sc, q, opacity, col, conf, 0.f, 0.f, g, radius, rc)) {
                float4* dst = reinterpret_cast<float4*>(geom + i);
                dst[0] = make_float4(g.mx, g.my, g.ca, g.cb);
                dst[1] = make_float4(g.cc, g.o, g.dc, g.gx);
                dst[2] = make_float4(g.gy, g.r, g.g, g.b);
                dst[3] = make_float4(g.nx, g.ny, g.nz, g.conf);
                const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                dgeom[4 * (size_t)i + 0] = z4; dgeom[4 * (size_t)i + 1] = z4;
                dgeom[4 * (size_t)i + 2] = z4; dgeom[4 * (size_t)i + 3] = z4;
                cnt = (uint32_t)((rc[2] - rc[0]) * (rc[3] - rc[1]));
                rx0 = (uint32_t)rc[0]; ry0 = (uint32_t)rc[1]; rwd = (uint32_t)(rc[2] - rc[0]);
                vis = 1;
            }
            radii[i] = radius;
        }
        if (touched.member) { // sticky row set of the optimisation loop: insert first-time-visible surfels
            const bool fresh = vis && was_member == 0 && atomicExch(&touched.member[i], 1) == 0;
            const unsigned long long mask = __ballot(fresh);
            if (mask) { // wave-uniform
                int base = 0;
                if (lane == (int)__builtin_ctzll(mask)) base = atomicAdd(touched.count, (int)__builtin_popcountll(mask));
                base = __shfl(base, (int)__builtin_ctzll(mask));
                if (fresh) touched.rows[base + (int)__builtin_popcountll(mask & ((1ull << lane) - 1ull))] = i;
            }
        }
        vis_total += ags_wave_sum_u32(vis);
        // the batch's (surfel, candidate tile) pairs flattened over the whole workgroup
        const uint32_t incl = ags_wave_incl_scan_u32(cnt);
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();          // (also: the previous batch's emit[] reads are done)
        uint32_t excl = incl - cnt, total = 0;
#pragma unroll
        for (int k = 0; k < NT / 64; ++k) { if (k < wave) excl += wtot[k]; total += wtot[k]; }
        AgsEmitRec me;
        me.excl = excl; me.xy = rx0 | (ry0 << 16); me.wd = rwd; me.pa = __float_as_uint(g.dc);
        me.mx = g.mx; me.my = g.my; me.ca = g.ca; me.cb = g.cb; me.cc = g.cc; me.o = g.o;
        emit[t] = me;
        __syncthreads();
        constexpr int NR = 2;     // rounds of 256 pairs with their slot atomics in flight together
        for (uint32_t base = 0; base < total; base += (uint32_t)NT * NR) {
            bool hit[NR];
            uint32_t tile[NR], got[NR], owner_depth[NR], owner_row[NR];
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                hit[rr] = false; tile[rr] = got[rr] = owner_depth[rr] = owner_row[rr] = 0u;
                const uint32_t j = base + (uint32_t)NT * rr + t;
                if (j < total) {
                    int l = 0;
#pragma unroll
                    for (int step = NT / 2; step > 0; step >>= 1)
                        if (emit[l + step].excl <= j) l += step;       // largest entry with excl <= j (entries without tiles share their successor's excl and lose to it)
                    const AgsEmitRec rec = emit[l];
                    const uint32_t tt = j - rec.excl;
                    const uint32_t tx = (rec.xy & 0xFFFF) + tt % rec.wd, ty = (rec.xy >> 16) + tt / rec.wd;
                    AgsGeom og;
                    og.mx = rec.mx; og.my = rec.my; og.ca = rec.ca; og.cb = rec.cb; og.cc = rec.cc; og.o = rec.o;
                    const float bx = (float)(tx * AGS_TILE), by = (float)(ty * AGS_TILE);
                    hit[rr] = ags_reaches_box(og, bx, bx + (AGS_TILE - 1), by, by + (AGS_TILE - 1));
                    tile[rr] = ty * F.tiles_x + tx; owner_depth[rr] = rec.pa;
                    owner_row[rr] = (uint32_t)first + (uint32_t)cand[c0 + l];
                }
            }
#pragma unroll
            for (int rr = 0; rr < NR; ++rr)
                if (hit[rr]) got[rr] = atomicAdd(&tile_count[(size_t)tile[rr] * direct.tc_stride], 1u);
#pragma unroll
            for (int rr = 0; rr < NR; ++rr)
                if (hit[rr] && got[rr] < direct.tile_cap)
                    direct.keys[(size_t)tile[rr] * direct.tile_cap + got[rr]] = ((uint64_t)owner_depth[rr] << 32) | owner_row[rr];
                else if (hit[rr]) atomicMax(direct.early, got[rr] + 1u);
        }
    }
    if (lane == 0 && vis_total) atomicAdd(&direct.partial[AGS_PART(blockIdx.x * (NT / 64) + wave, AGS_PART_VIS)], vis_total);
}

__global__ __launch_bounds__(AGS_PRE_THREADS) void ags_k_preprocess_bwd(
    AgsFrame F, const float* __restrict__ Vp, const float* __restrict__ Pp, AgsGaussians in,
    const int* __restrict__ radii, AgsGeomGrad* __restrict__ dgeom, AgsGaussianGrads out, AgsViewStride vs) {
    ags_frame_flags(F);
    { // batched backward (accumulate == 2): this workgroup's view // (offsets are 0 for a single view)
        Vp += 16 * blockIdx.y; Pp += 16 * blockIdx.y;
        radii += (size_t)blockIdx.y * (size_t)vs.n;
        AGS_WS_SHIFT(dgeom, (size_t)blockIdx.y * (size_t)vs.ws);
    }
    __shared__ __attribute__((aligned(16))) float rows3[3 * AGS_PRE_THREADS];
    float V[16], P[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { V[k] = Vp[k]; P[k] = Pp[k]; }
    const int first = blockIdx.x * AGS_PRE_THREADS;
    const int i = first + threadIdx.x;
    const int rows = min(AGS_PRE_THREADS, in.n - first);
    float dm[3] = {0, 0, 0}, ds[3] = {0, 0, 0}, dq[4] = {0, 0, 0, 0}, dop = 0, dcol[3] = {0, 0, 0}, dm2[2] = {0, 0};
    const bool vis = (i < in.n) && radii[i] > 0;
    // a block with no visible surfel has nothing to add (accumulate) / only zeros to write
    const bool block_has_vis = __syncthreads_or(vis ? 1 : 0) != 0;
    float p[3] = {0, 0, 0}, sc[3] = {0, 0, 0};
    if (block_has_vis) {
        ags_load_rows3(in.means3D, first, rows, rows3, p);
        ags_load_rows3(in.scales, first, rows, rows3, sc);
    }
    if (vis) {
        const float4 q4 = reinterpret_cast<const float4*>(in.rotations)[i];
        float q[4] = {q4.x, q4.y, q4.z, q4.w};
        float opacity = in.opacities[i];
        float raw_v[3] = {0, 0, 0}, qinv = 1.f;
        if (in.raw_params) ags_activate_inplace(in, sc, q, opacity, raw_v, qinv);
        float4* src = reinterpret_cast<float4*>(dgeom + i);
        const float4 a = src[0], b = src[1], c = src[2], d = src[3];
        // leave the record zeroed again: a second ags_backward on the same forward state is valid
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        src[0] = z4; src[1] = z4; src[2] = z4; src[3] = z4;
        AgsGeomGrad dg;
        dg.m1x = a.x; dg.m1y = a.y; dg.m2xx = a.z; dg.m2xy = a.w;
        dg.m2yy = b.x; dg.m0 = b.y; dg.ddc = b.z; dg.dgx = b.w;
        dg.dgy = c.x; dg.dr = c.y; dg.dg = c.z; dg.db = c.w;
        dg.dnx = d.x; dg.dny = d.y; dg.dnz = d.z; dg.pad = 0.f;
        ags_preprocess_bwd(F, V, P, p, sc, q, opacity, dg, dm, ds, dq, &dop, dcol, dm2);
        if (in.raw_params) { // chain rule through clamp(exp), normalize, sigmoid
#pragma unroll
            for (int k = 0; k < 3; ++k) ds[k] = (raw_v[k] >= 0.f && raw_v[k] <= in.max_scale) ? ds[k] * raw_v[k] : 0.f;
            const float dot = q[0] * dq[0] + q[1] * dq[1] + q[2] * dq[2] + q[3] * dq[3];
#pragma unroll
            for (int k = 0; k < 4; ++k) dq[k] = (dq[k] - q[k] * dot) * qinv;
            dop *= opacity * (1.f - opacity);
        }
    }
    if (out.accumulate == 2) { // concurrent views: only visible rows touch memory, atomically
        if (vis) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                unsafeAtomicAdd(&out.d_means3D[3 * i + k], dm[k]);
                unsafeAtomicAdd(&out.d_scales[3 * i + k], ds[k]);
                unsafeAtomicAdd(&out.d_colors[3 * i + k], dcol[k]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) unsafeAtomicAdd(&out.d_rotations[4 * i + k], dq[k]);
            unsafeAtomicAdd(&out.d_opacities[i], dop);
            if (out.d_means2D) { unsafeAtomicAdd(&out.d_means2D[3 * i], dm2[0]); unsafeAtomicAdd(&out.d_means2D[3 * i + 1], dm2[1]); }
        }
    } else if (out.accumulate) {
        if (!block_has_vis) return; // block-uniform
        ags_store_rows3<true>(out.d_means3D, first, rows, rows3, dm);
        ags_store_rows3<true>(out.d_scales, first, rows, rows3, ds);
        ags_store_rows3<true>(out.d_colors, first, rows, rows3, dcol);
        if (vis) {
            float4* dr = reinterpret_cast<float4*>(out.d_rotations) + i;
            const float4 o = *dr;
            *dr = make_float4(o.x + dq[0], o.y + dq[1], o.z + dq[2], o.w + dq[3]);
            out.d_opacities[i] += dop;
            if (out.d_means2D) { out.d_means2D[3 * i] += dm2[0]; out.d_means2D[3 * i + 1] += dm2[1]; }
        }
    } else {
        ags_store_rows3<false>(out.d_means3D, first, rows, rows3, dm);
        ags_store_rows3<false>(out.d_scales, first, rows, rows3, ds);
        ags_store_rows3<false>(out.d_colors, first, rows, rows3, dcol);
        if (i < in.n) {
            reinterpret_cast<float4*>(out.d_rotations)[i] = make_float4(dq[0], dq[1], dq[2], dq[3]);
            out.d_opacities[i] = dop;
        }
        if (out.d_means2D) {
            const float m2[3] = {dm2[0], dm2[1], 0.f};
            ags_store_rows3<false>(out.d_means2D, first, rows, rows3, m2);
        }
    }
}

// Row-set form of the kernel above (AgsGaussianGrads.touched): one lane per MEMBER row instead of
// one per map row.  At the headline view 6.7 % of the map is visible, so this reads and writes
// ~13 k scattered rows instead of streaming all 200 k.
//
// FUSED_ADAM (AgsGaussianGrads.fused_adam): the row's 14 gradient values are final once this view
// is added, so the same lane applies the Adam update to the row's parameters right here - no
// second kernel and no gradient round trip.  The clock was advanced by the blend backward that
// ran just before on this stream (AgsTick), so every lane reads finished scalars.
#define AGS_ROWS_THREADS 64 // one wave per workgroup: the few member rows spread over all CUs
// MODE 2 (AgsGaussianGrads.pack_segment): the data-parallel step's last view per rank - the row's
// totals go straight into the rank's exchange segment (record = the row's position in the list: one
// 64-byte store per lane, see ags_rows_pack in adam.hip for the layout) and the slab is left zeroed.
// a row's exp_avg / exp_avg_sq (14 + 14 floats)
__device__ __forceinline__ void ags_load_moments(const AgsAdamArgs& adam, int i, float am[14], float av[14]) {
    if (adam.st) { // interleaved moments: the row's 28 floats as seven 16-byte accesses of one 112-byte piece
        const float4* sr = reinterpret_cast<const float4*>(adam.st + (size_t)i * 28);
        float mv[28];
#pragma unroll
        for (int q = 0; q < 7; ++q) { const float4 x = sr[q]; mv[4 * q] = x.x; mv[4 * q + 1] = x.y; mv[4 * q + 2] = x.z; mv[4 * q + 3] = x.w; }
#pragma unroll
        for (int e = 0; e < 14; ++e) { am[e] = mv[e]; av[e] = mv[14 + e]; }
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            am[k] = adam.m[0][3 * i + k]; av[k] = adam.v[0][3 * i + k];
            am[3 + k] = adam.m[1][3 * i + k]; av[3 + k] = adam.v[1][3 * i + k];
            am[11 + k] = adam.m[4][3 * i + k]; av[11 + k] = adam.v[4][3 * i + k];
        }
        const float4 m4 = reinterpret_cast<const float4*>(adam.m[2])[i], v4 = reinterpret_cast<const float4*>(adam.v[2])[i];
        am[6] = m4.x; am[7] = m4.y; am[8] = m4.z; am[9] = m4.w;
        av[6] = v4.x; av[7] = v4.y; av[8] = v4.z; av[9] = v4.w;
        am[10] = adam.m[3][i]; av[10] = adam.v[3][i];
    }
}

// What the software-pipelined optimisation step (ags_backward_fused_next) hands the per-Gaussian kernel besides the
// backward's own arguments: the NEXT forward pass's camera and where its per-Gaussian stage writes.
struct AgsNextPre {
    AgsFrame F;
    const float* V; const float* P;           // next view: view / projection matrices
    AgsGeom* geom; float4* dgeom; int* radii; // next pass's workspace records and radii
    uint32_t* tile_count;
    AgsDirectEmit direct;
    const uint32_t* count_snap;               // members of the row set when this launch started (noted by render_bwd)
};

// Body of the row-set per-Gaussian backward for wave `wave_index` of `num_waves` (pointers already moved to the view).
// NEXT (with MODE 1): once a row's Adam update is applied the same lane runs the forward per-Gaussian stage of the
// next view for it, from the updated parameters it holds in registers (emit_wave: 64 AgsEmitRec of LDS for the wave).
// (NEXT: AGS_NEXT_ROWS member rows per wave instead of 64.  The member list is COMPACT - every lane's row is visible and
// its tile rect holds ~12 candidate tiles - so a full wave would emit in a dozen rounds of 64 candidates, each with a
// search through the wave's records in LDS and a returning atomic: measured 29 us per wave.  The kernel is a latency
// chain and idle lanes cost nothing, but every wave holds a wave slot that the forward workgroups of the same launch
// need too.  Measured on the 1200x680 view (step time, five-launch form 78.5 us): 4 rows per wave 76.1, 6: 73.8,
// 8: 74.4, 12: 75.7, 16: 76.0, 32: 82.6, 64: 109.)
#ifndef AGS_NEXT_ROWS
#define AGS_NEXT_ROWS 8
#endif

// Key emission of a wave whose lanes hold (surfel, tile rect) records, AGS_EMIT_ROUNDS rounds of 64 (surfel, tile)
// candidates per iteration with ALL those rounds' slot atomics in flight before the first key is stored (the plain loop of
// ags_emit_tiles_balanced waits for every round's returning atomic: ~2 us each).  One atomic per lane (no same-tile
// grouping: for images of more than AGS_AGG_MAX_TILES tiles).  row_of_lane: the surfel id a lane's record belongs to.
#ifndef AGS_EMIT_ROUNDS
#define AGS_EMIT_ROUNDS 2     // eight surfels of a compact list have ~100 candidate tiles: one iteration
#endif
__device__ __forceinline__ void ags_emit_keys_rounds(AgsEmitRec* wave_lds, uint32_t cnt, uint32_t x0, uint32_t y0, uint32_t wd,
                                                         uint32_t depth_bits, const AgsGeom& g, int tiles_x, int row_of_lane,
                                                         uint32_t* __restrict__ tile_count, const AgsDirectEmit& direct) {
    const int lane = threadIdx.x & 63;
    const uint32_t incl = ags_wave_incl_scan_u32(cnt);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (total == 0) return; // wave-uniform
    AgsEmitRec me;
    me.excl = incl - cnt; me.xy = x0 | (y0 << 16); me.wd = wd; me.pa = depth_bits;
    me.mx = g.mx; me.my = g.my; me.ca = g.ca; me.cb = g.cb; me.cc = g.cc; me.o = g.o;
    wave_lds[lane] = me;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    constexpr int NR = AGS_EMIT_ROUNDS;
    for (uint32_t base = 0; base < total; base += 64u * NR) {
        bool hit[NR];
        uint32_t tile[NR], owner_depth[NR], got[NR];
        int lo[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) { hit[r] = false; tile[r] = owner_depth[r] = got[r] = 0u; lo[r] = 0; }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (base + 64u * r >= total) break;   // wave-uniform
            const uint32_t j = base + 64u * r + lane;
            if (j < total) {
                int l = 0;
#pragma unroll
                for (int step = 32; step > 0; step >>= 1)
                    if (wave_lds[l + step].excl <= j) l += step; // largest lane with excl <= j
                const AgsEmitRec rec = wave_lds[l];
                const uint32_t t = j - rec.excl;
                const uint32_t tx = (rec.xy & 0xFFFF) + t % rec.wd, ty = (rec.xy >> 16) + t / rec.wd;
                AgsGeom og;
                og.mx = rec.mx; og.my = rec.my; og.ca = rec.ca; og.cb = rec.cb; og.cc = rec.cc; og.o = rec.o;
                const float bx = (float)(tx * AGS_TILE), by = (float)(ty * AGS_TILE);
                hit[r] = ags_reaches_box(og, bx, bx + (AGS_TILE - 1), by, by + (AGS_TILE - 1));
                tile[r] = ty * tiles_x + tx; owner_depth[r] = rec.pa; lo[r] = l;
            }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (hit[r]) got[r] = atomicAdd(&tile_count[(size_t)tile[r] * direct.tc_stride], 1u);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (base + 64u * r >= total) break;   // wave-uniform
            const uint32_t owner_row = (uint32_t)__shfl(row_of_lane, lo[r]);
            if (hit[r] && got[r] < direct.tile_cap)
                direct.keys[(size_t)tile[r] * direct.tile_cap + got[r]] = ((uint64_t)owner_depth[r] << 32) | owner_row;
            else if (hit[r]) atomicMax(direct.early, got[r] + 1u);
        }
    }
    __builtin_amdgcn_wave_barrier();
}
template <int MODE, bool NEXT, bool AGG>
__device__ __forceinline__ void ags_rows_body(
    const AgsFrame& F, const float* __restrict__ Vp, const float* __restrict__ Pp, const AgsGaussians& in,
    const int* __restrict__ radii, AgsGeomGrad* __restrict__ dgeom, const AgsGaussianGrads& out, const AgsAdamArgs& adam,
    const int wave_index, const int num_waves, const AgsNextPre* nx, AgsEmitRec* emit_wave) {
    constexpr bool FUSED_ADAM = MODE == 1, PACK = MODE == 2;
    static_assert(!NEXT || FUSED_ADAM, "the pipelined form continues from the fused Adam update");
    float V[16], P[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { V[k] = Vp[k]; P[k] = Pp[k]; }
    const int lane = threadIdx.x & 63;
    AGS_TL(4, wave_index, 0);
    // This kernel is a short chain of dependent loads on few rows (latency, not bandwidth), so every
    // load that can be issued early is: the first batch of row ids is fetched while the member count
    // is still in flight (the list is zero-filled past `count`, so any slot holds a valid row), and
    // a row's inputs are fetched together with its radius instead of behind the visibility test.
    constexpr int RPW = NEXT ? AGS_NEXT_ROWS : 64;    // member rows per wave and pass
    const int slot0 = min(wave_index * RPW + min(lane, RPW - 1), in.n - 1);
    int i_next = out.touched.rows[slot0];
    // (pipelined form: this launch's other workgroups may be appending to the list - the rows of THIS step are the
    // ones that were members when the launch started)
    const int count = NEXT ? (int)*nx->count_snap : *out.touched.count;
    if (PACK && wave_index == 0 && lane < 16) // segment header: rows shipped, rows the set holds
        out.pack_segment[lane] = __int_as_float(lane == 0 ? min(count, out.pack_capacity) : lane == 1 ? count : 0);
    for (int base = wave_index * RPW; base < count; base += num_waves * RPW) { // wave-uniform
        const bool valid = lane < RPW && base + lane < count;
        const int i = valid ? i_next : 0;
        {
            const int nb = base + num_waves * RPW;
            if (nb < count) i_next = out.touched.rows[min(nb + min(lane, RPW - 1), in.n - 1)];
        }
        const int rad = radii[i];
        float p[3], sc[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { p[k] = in.means3D[3 * i + k]; sc[k] = in.scales[3 * i + k]; }
        const float4 q4 = reinterpret_cast<const float4*>(in.rotations)[i];
        float opacity = in.opacities[i];
        [[maybe_unused]] const float conf_next = NEXT ? in.confidences[i] : 0.f;   // requested with the row's other inputs
        float4* src = reinterpret_cast<float4*>(dgeom + i);
        const float4 a = src[0], b = src[1], c = src[2], d = src[3];
        float am[14], av[14], ap[14]; // FUSED_ADAM: the row's exp_avg / exp_avg_sq / parameters, requested now
        if (FUSED_ADAM) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                ap[k] = adam.p[0][3 * i + k]; ap[3 + k] = adam.p[1][3 * i + k]; ap[11 + k] = adam.p[4][3 * i + k];
            }
            const float4 p4 = reinterpret_cast<const float4*>(adam.p[2])[i];
            ap[6] = p4.x; ap[7] = p4.y; ap[8] = p4.z; ap[9] = p4.w;
            ap[10] = adam.p[3][i];
            if (!NEXT) ags_load_moments(adam, i, am, av);
        }
        const bool vis = valid && rad > 0;
        float dm[3] = {0, 0, 0}, ds[3] = {0, 0, 0}, dq[4] = {0, 0, 0, 0}, dop = 0, dcol[3] = {0, 0, 0}, dm2[2] = {0, 0};
        if (vis) {
            float q[4] = {q4.x, q4.y, q4.z, q4.w};
            float raw_v[3] = {0, 0, 0}, qinv = 1.f;
            if (in.raw_params) ags_activate_inplace(in, sc, q, opacity, raw_v, qinv);
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            src[0] = z4; src[1] = z4; src[2] = z4; src[3] = z4;
            AgsGeomGrad dg;
            dg.m1x = a.x; dg.m1y = a.y; dg.m2xx = a.z; dg.m2xy = a.w;
            dg.m2yy = b.x; dg.m0 = b.y; dg.ddc = b.z; dg.dgx = b.w;
            dg.dgy = c.x; dg.dr = c.y; dg.dg = c.z; dg.db = c.w;
            dg.dnx = d.x; dg.dny = d.y; dg.dnz = d.z; dg.pad = 0.f;
            ags_preprocess_bwd(F, V, P, p, sc, q, opacity, dg, dm, ds, dq, &dop, dcol, dm2);
            if (in.raw_params) {
#pragma unroll
                for (int k = 0; k < 3; ++k) ds[k] = (raw_v[k] >= 0.f && raw_v[k] <= in.max_scale) ? ds[k] * raw_v[k] : 0.f;
                const float dot = q[0] * dq[0] + q[1] * dq[1] + q[2] * dq[2] + q[3] * dq[3];
#pragma unroll
                for (int k = 0; k < 4; ++k) dq[k] = (dq[k] - q[k] * dot) * qinv;
                dop *= opacity * (1.f - opacity);
            }
        }
        if (out.accumulate == 2) {
            if (vis) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    unsafeAtomicAdd(&out.d_means3D[3 * i + k], dm[k]);
                    unsafeAtomicAdd(&out.d_scales[3 * i + k], ds[k]);
                    unsafeAtomicAdd(&out.d_colors[3 * i + k], dcol[k]);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) unsafeAtomicAdd(&out.d_rotations[4 * i + k], dq[k]);
                unsafeAtomicAdd(&out.d_opacities[i], dop);
                if (out.d_means2D) { unsafeAtomicAdd(&out.d_means2D[3 * i], dm2[0]); unsafeAtomicAdd(&out.d_means2D[3 * i + 1], dm2[1]); }
            }
        } else if (out.accumulate) {
            if (vis || ((FUSED_ADAM || PACK) && valid)) { // the fused step / the exchange need the totals of rows this view does not show too
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    dm[k] += out.d_means3D[3 * i + k]; ds[k] += out.d_scales[3 * i + k]; dcol[k] += out.d_colors[3 * i + k];
                }
                float4* dr = reinterpret_cast<float4*>(out.d_rotations) + i;
                const float4 o = *dr;
                dq[0] += o.x; dq[1] += o.y; dq[2] += o.z; dq[3] += o.w;
                dop += out.d_opacities[i];
            }
            if (PACK) { // the totals leave in the segment: the earlier views' partial sums are cleared
                if (valid && base + lane < out.pack_capacity) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) { out.d_means3D[3 * i + k] = 0.f; out.d_scales[3 * i + k] = 0.f; out.d_colors[3 * i + k] = 0.f; }
                    reinterpret_cast<float4*>(out.d_rotations)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    out.d_opacities[i] = 0.f;
                }
            } else if (vis) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    out.d_means3D[3 * i + k] = dm[k]; out.d_scales[3 * i + k] = ds[k]; out.d_colors[3 * i + k] = dcol[k];
                }
                reinterpret_cast<float4*>(out.d_rotations)[i] = make_float4(dq[0], dq[1], dq[2], dq[3]);
                out.d_opacities[i] = dop;
                if (out.d_means2D) { out.d_means2D[3 * i] += dm2[0]; out.d_means2D[3 * i + 1] += dm2[1]; }
            }
        } else if (valid && !PACK && out.d_means3D) { // overwrite: member rows this view does not show get their zeros
            // (d_* all NULL: fused optimiser step whose gradient never leaves the registers)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                out.d_means3D[3 * i + k] = dm[k]; out.d_scales[3 * i + k] = ds[k]; out.d_colors[3 * i + k] = dcol[k];
            }
            reinterpret_cast<float4*>(out.d_rotations)[i] = make_float4(dq[0], dq[1], dq[2], dq[3]);
            out.d_opacities[i] = dop;
            if (out.d_means2D) { out.d_means2D[3 * i] = dm2[0]; out.d_means2D[3 * i + 1] = dm2[1]; out.d_means2D[3 * i + 2] = 0.f; }
        }
        if (PACK && valid) {
            if (base + lane < out.pack_capacity) {
                float4* rec = reinterpret_cast<float4*>(out.pack_segment + 16 + (size_t)(base + lane) * 16);
                rec[0] = make_float4(dm[0], dm[1], dm[2], ds[0]);
                rec[1] = make_float4(ds[1], ds[2], dq[0], dq[1]);
                rec[2] = make_float4(dq[2], dq[3], dop, dcol[0]);
                rec[3] = make_float4(dcol[1], dcol[2], __int_as_float(i), 0.f);
            } else {
                // the segment is full (the caller sized it too small: header word 1 > capacity says so): like
                // ags_rows_pack, rows that do not travel keep their totals in the gradient arrays, so nothing
                // is lost - the caller restores the shipped rows (ags_rows_unpack of its own segment), agrees
                // on a larger segment and exchanges again, or repeats the step
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    out.d_means3D[3 * i + k] = dm[k]; out.d_scales[3 * i + k] = ds[k]; out.d_colors[3 * i + k] = dcol[k];
                }
                reinterpret_cast<float4*>(out.d_rotations)[i] = make_float4(dq[0], dq[1], dq[2], dq[3]);
                out.d_opacities[i] = dop;
            }
        }
        if (NEXT) AGS_TL(4, wave_index, 2);
        // (pipelined form: the moments are requested only now - 28 registers less across the chain rule keeps the launch
        // at three waves per SIMD without spills, which the forward workgroups of the same launch need; one load level
        // more on the member rows' chain)
        if (NEXT && FUSED_ADAM && valid) {   // (interleaved moments only: ags_backward_fused_next requires state_rows)
            const float4* sr = reinterpret_cast<const float4*>(adam.st + (size_t)i * 28);
            float mv[28];
#pragma unroll
            for (int q = 0; q < 7; ++q) { const float4 x = sr[q]; mv[4 * q] = x.x; mv[4 * q + 1] = x.y; mv[4 * q + 2] = x.z; mv[4 * q + 3] = x.w; }
#pragma unroll
            for (int e = 0; e < 14; ++e) { am[e] = mv[e]; av[e] = mv[14 + e]; }
        }
        if (FUSED_ADAM && valid) {
            // the row's Adam state was requested together with its inputs (am / av / ap above), so it is
            // here by now: update in the owning lane and store each tensor's 3-4 floats as one access
            const AgsAdamClock* clk = (const AgsAdamClock*)out.adam_clock;
            const float ib = clk->inv_bc2_sqrt, b1 = out.adam_beta1, b2 = out.adam_beta2, eps = out.adam_eps;
            const float gr[14] = {dm[0], dm[1], dm[2], ds[0], ds[1], ds[2], dq[0], dq[1], dq[2], dq[3], dop,
                                  dcol[0], dcol[1], dcol[2]};
#pragma unroll
            for (int e = 0; e < 14; ++e) {
                const int seg = (e >= 3) + (e >= 6) + (e >= 10) + (e >= 11);
                const float g = gr[e];
                const float m = am[e] + (1.f - b1) * (g - am[e]);
                const float v = av[e] * b2 + (1.f - b2) * g * g;
                const float denom = sqrtf(v) * ib + eps;
                am[e] = m; av[e] = v;
                ap[e] -= clk->step_size[seg] * (m / denom);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                adam.p[0][3 * i + k] = ap[k]; adam.p[1][3 * i + k] = ap[3 + k]; adam.p[4][3 * i + k] = ap[11 + k];
            }
            reinterpret_cast<float4*>(adam.p[2])[i] = make_float4(ap[6], ap[7], ap[8], ap[9]);
            adam.p[3][i] = ap[10];
            if (adam.st) {
                float4* sr = reinterpret_cast<float4*>(adam.st + (size_t)i * 28);
                float mv[28];
#pragma unroll
                for (int e = 0; e < 14; ++e) { mv[e] = am[e]; mv[14 + e] = av[e]; }
#pragma unroll
                for (int q = 0; q < 7; ++q) sr[q] = make_float4(mv[4 * q], mv[4 * q + 1], mv[4 * q + 2], mv[4 * q + 3]);
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    adam.m[0][3 * i + k] = am[k]; adam.v[0][3 * i + k] = av[k];
                    adam.m[1][3 * i + k] = am[3 + k]; adam.v[1][3 * i + k] = av[3 + k];
                    adam.m[4][3 * i + k] = am[11 + k]; adam.v[4][3 * i + k] = av[11 + k];
                }
                reinterpret_cast<float4*>(adam.m[2])[i] = make_float4(am[6], am[7], am[8], am[9]);
                reinterpret_cast<float4*>(adam.v[2])[i] = make_float4(av[6], av[7], av[8], av[9]);
                adam.m[3][i] = am[10]; adam.v[3][i] = av[10];
            }
        }
        if constexpr (NEXT) {
            AGS_TL(4, wave_index, 3);
            // ---- the next view's per-Gaussian stage for this row (what ags_k_preprocess<2> does for non-member rows),
            // from the parameters the Adam update above left in registers
            float V2[16], P2[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) { V2[k] = nx->V[k]; P2[k] = nx->P[k]; }
            uint32_t cnt = 0, vis2 = 0, rx0 = 0, ry0 = 0, rwd = 1;
            AgsGeom g2;
            g2.mx = g2.my = g2.ca = g2.cb = g2.cc = g2.o = 0.f; g2.dc = 0.f;
            if (valid) {
                float p2[3] = {ap[0], ap[1], ap[2]}, sc2[3] = {ap[3], ap[4], ap[5]}, q2[4] = {ap[6], ap[7], ap[8], ap[9]};
                float op2 = ap[10];
                const float col2[3] = {ap[11], ap[12], ap[13]};
                if (in.raw_params) { float rv[3], qi; ags_activate_inplace(in, sc2, q2, op2, rv, qi); }
                int radius = 0, rc[4];
                if (ags_preprocess_fwd(nx->F, V2, P2, p2, sc2, q2, op2, col2, conf_next, 0.f, 0.f, g2, radius, rc)) {
                    float4* dst = reinterpret_cast<float4*>(nx->geom + i);
                    dst[0] = make_float4(g2.mx, g2.my, g2.ca, g2.cb);
                    dst[1] = make_float4(g2.cc, g2.o, g2.dc, g2.gx);
                    dst[2] = make_float4(g2.gy, g2.r, g2.g, g2.b);
                    dst[3] = make_float4(g2.nx, g2.ny, g2.nz, g2.conf);
                    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    nx->dgeom[4 * (size_t)i + 0] = z4; nx->dgeom[4 * (size_t)i + 1] = z4;
                    nx->dgeom[4 * (size_t)i + 2] = z4; nx->dgeom[4 * (size_t)i + 3] = z4;
                    cnt = (uint32_t)((rc[2] - rc[0]) * (rc[3] - rc[1]));
                    rx0 = (uint32_t)rc[0]; ry0 = (uint32_t)rc[1]; rwd = (uint32_t)(rc[2] - rc[0]);
                    vis2 = 1;
                }
                nx->radii[i] = radius;
            }
            AGS_TL(4, wave_index, 4);
            const int my_row = i;
            if constexpr (AGG)   // images of few tiles: same-tile lanes share an atomic (ags_wave_agg_inc), round by round
                ags_emit_tiles_balanced(emit_wave, cnt, rx0, ry0, rwd, __float_as_uint(g2.dc), g2, nx->F.tiles_x,
                                        [&](bool hit, uint32_t t, uint32_t depth_bits, int owner_lane) {
                                            const uint32_t got = ags_wave_agg_inc<AGG, true>(nx->tile_count, t * nx->direct.tc_stride, hit);
                                            const uint32_t owner_row = (uint32_t)__shfl(my_row, owner_lane);
                                            if (hit && got < nx->direct.tile_cap)
                                                nx->direct.keys[(size_t)t * nx->direct.tile_cap + got] = ((uint64_t)depth_bits << 32) | owner_row;
                                            else if (hit) atomicMax(nx->direct.early, got + 1u);
                                        });
            else
                ags_emit_keys_rounds(emit_wave, cnt, rx0, ry0, rwd, __float_as_uint(g2.dc), g2, nx->F.tiles_x, my_row,
                                     nx->tile_count, nx->direct);
            const uint32_t wv = ags_wave_sum_u32(vis2);
            if (lane == 0 && wv) atomicAdd(&nx->direct.partial[AGS_PART(wave_index, AGS_PART_VIS)], wv);
            AGS_TL(4, wave_index, 5);
        }
    }
    AGS_TL(4, wave_index, 1);
    AGS_TL_VAL(4, wave_index, 6, count);
    AGS_TL_VAL(4, wave_index, 7, (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)(__builtin_amdgcn_s_getreg(63508) & 15) << 32));
}

// ---------------------------------------------------------------------------------------
// ags_backward_rows: the per-Gaussian backward of ALL the views of an optimisation step in one launch.  One lane per
// member row; the row's inputs are loaded and activated once; every view that shows the row adds its chain rule (in the
// space of the ACTIVATED parameters: the activations' own chain rule is view-independent and applied once to the sum);
// then the tail of the single-view row kernel: MODE 1 fused Adam, MODE 2 exchange segment, MODE 0 gradient arrays.
#ifndef AGS_ROWS_MULTI_WAVES
#define AGS_ROWS_MULTI_WAVES 3   // resident waves per SIMD the register budget is cut for (512 / 3 -> 168 VGPRs; 4 spills: measured slower)
#endif
template <int MODE>
__global__ __launch_bounds__(AGS_ROWS_THREADS) __attribute__((amdgpu_waves_per_eu(AGS_ROWS_MULTI_WAVES, AGS_ROWS_MULTI_WAVES))) void ags_k_rows_multi(AgsRowViews rv, AgsGaussians in, AgsGaussianGrads out,
                                                                    AgsAdamArgs adam) {
    // MODE 3 (data-parallel ranks that exchange the dense gradient slab in row chunks): no row set - the rows
    // [out.row_begin, out.row_end) of the MAP, gradient arrays overwritten (zeros where no view shows the row)
    constexpr bool FUSED_ADAM = MODE == 1, PACK = MODE == 2, RANGE = MODE == 3;
    const int lane = threadIdx.x & 63;
    const int wave_index = (int)blockIdx.x, num_waves = (int)gridDim.x;
    const int slot0 = min(wave_index * 64 + lane, in.n - 1);
    int i_next = RANGE ? 0 : out.touched.rows[slot0];
    const int count = RANGE ? 0 : *out.touched.count;
    if (PACK && wave_index == 0 && lane < 16) // segment header: rows shipped, rows the set holds
        out.pack_segment[lane] = __int_as_float(lane == 0 ? min(count, out.pack_capacity) : lane == 1 ? count : 0);
    // When most of the map is listed (a room seen from inside: config 4's four views list 73 % of the rows) walking the
    // MAP in row order with a membership test beats walking the list: every access of a wave is then a stream instead
    // of 64 scattered 12-16-byte pieces per array (six sectors per row for 52 useful bytes).  Not for the exchange
    // segment, whose record positions are list positions.
    const bool dense = RANGE || (!PACK && 2 * (long long)count > (long long)in.n);
    const int total = RANGE ? out.row_end : (dense ? in.n : count);
    for (int base = (RANGE ? out.row_begin : 0) + wave_index * 64; base < total; base += num_waves * 64) { // wave-uniform
        bool valid = base + lane < total;
        int i = dense ? min(base + lane, in.n - 1) : (valid ? i_next : 0);
        if (dense) {
            if (!RANGE) valid = valid && out.touched.member[i] != 0;
            if (!__any(valid)) continue;            // wave-uniform
            if (!valid) i = 0;
        } else {
            const int nb = base + num_waves * 64;
            if (nb < count) i_next = out.touched.rows[min(nb + lane, in.n - 1)];
        }
        float p[3] = {0, 0, 0}, sc[3] = {0, 0, 0};
        float4 q4 = make_float4(1.f, 0.f, 0.f, 0.f);
        float opacity = 0.f;
        if (valid || !dense) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { p[k] = in.means3D[3 * (size_t)i + k]; sc[k] = in.scales[3 * (size_t)i + k]; }
            q4 = reinterpret_cast<const float4*>(in.rotations)[i];
            opacity = in.opacities[i];
        }
        float am[14], av[14], ap[14];
        if (FUSED_ADAM && (valid || !dense)) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                ap[k] = adam.p[0][3 * (size_t)i + k]; ap[3 + k] = adam.p[1][3 * (size_t)i + k]; ap[11 + k] = adam.p[4][3 * (size_t)i + k];
            }
            const float4 p4 = reinterpret_cast<const float4*>(adam.p[2])[i];
            ap[6] = p4.x; ap[7] = p4.y; ap[8] = p4.z; ap[9] = p4.w;
            ap[10] = adam.p[3][i];
        }
        float q[4] = {q4.x, q4.y, q4.z, q4.w};
        float raw_v[3] = {0, 0, 0}, qinv = 1.f;
        if (in.raw_params) ags_activate_inplace(in, sc, q, opacity, raw_v, qinv);
        float dm[3] = {0, 0, 0}, ds[3] = {0, 0, 0}, dq[4] = {0, 0, 0, 0}, dop = 0, dcol[3] = {0, 0, 0}, dm2[2] = {0, 0};
        bool vis_any = false;
#pragma unroll 1
        for (int v = 0; v < rv.views; ++v) {
            const bool vis = valid && rv.radii[v][i] > 0;
            if (!__any(vis)) continue;      // wave-uniform
            if (vis) {
                AgsFrame F = rv.F[v];
                ags_frame_flags(F);
                float V[16], P[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) { V[k] = rv.V[v][k]; P[k] = rv.P[v][k]; }
                float4* src = reinterpret_cast<float4*>(rv.dgeom[v] + i);
                const float4 a = src[0], b = src[1], c = src[2], d = src[3];
                const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                src[0] = z4; src[1] = z4; src[2] = z4; src[3] = z4;
                AgsGeomGrad dg;
                dg.m1x = a.x; dg.m1y = a.y; dg.m2xx = a.z; dg.m2xy = a.w;
                dg.m2yy = b.x; dg.m0 = b.y; dg.ddc = b.z; dg.dgx = b.w;
                dg.dgy = c.x; dg.dr = c.y; dg.dg = c.z; dg.db = c.w;
                dg.dnx = d.x; dg.dny = d.y; dg.dnz = d.z; dg.pad = 0.f;
                float vm[3], vs_[3], vq[4], vop, vcol[3], vm2[2];
                ags_preprocess_bwd(F, V, P, p, sc, q, opacity, dg, vm, vs_, vq, &vop, vcol, vm2);
#pragma unroll
                for (int k = 0; k < 3; ++k) { dm[k] += vm[k]; ds[k] += vs_[k]; dcol[k] += vcol[k]; }
#pragma unroll
                for (int k = 0; k < 4; ++k) dq[k] += vq[k];
                dop += vop; dm2[0] += vm2[0]; dm2[1] += vm2[1];
                vis_any = true;
            }
        }
        // (the moments are requested only now: 28 registers less across the views' chain rules)
        if (FUSED_ADAM && (valid || !dense)) ags_load_moments(adam, i, am, av);
        if (in.raw_params && vis_any) { // chain rule through clamp(exp), normalize, sigmoid: linear, applied to the views' sum
#pragma unroll
            for (int k = 0; k < 3; ++k) ds[k] = (raw_v[k] >= 0.f && raw_v[k] <= in.max_scale) ? ds[k] * raw_v[k] : 0.f;
            const float dot = q[0] * dq[0] + q[1] * dq[1] + q[2] * dq[2] + q[3] * dq[3];
#pragma unroll
            for (int k = 0; k < 4; ++k) dq[k] = (dq[k] - q[k] * dot) * qinv;
            dop *= opacity * (1.f - opacity);
        }
        if (valid && out.d_means3D && !(PACK && base + lane < out.pack_capacity)) {
            // the gradient arrays: overwrite, or += what earlier calls of the step left (accumulate 1)
            if (out.accumulate) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    dm[k] += out.d_means3D[3 * (size_t)i + k]; ds[k] += out.d_scales[3 * (size_t)i + k]; dcol[k] += out.d_colors[3 * (size_t)i + k];
                }
                const float4 o = reinterpret_cast<float4*>(out.d_rotations)[i];
                dq[0] += o.x; dq[1] += o.y; dq[2] += o.z; dq[3] += o.w;
                dop += out.d_opacities[i];
            }
            if (!PACK) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    out.d_means3D[3 * (size_t)i + k] = dm[k]; out.d_scales[3 * (size_t)i + k] = ds[k]; out.d_colors[3 * (size_t)i + k] = dcol[k];
                }
                reinterpret_cast<float4*>(out.d_rotations)[i] = make_float4(dq[0], dq[1], dq[2], dq[3]);
                out.d_opacities[i] = dop;
                if (out.d_means2D) { out.d_means2D[3 * (size_t)i] = dm2[0]; out.d_means2D[3 * (size_t)i + 1] = dm2[1]; out.d_means2D[3 * (size_t)i + 2] = 0.f; }
            }
        }
        if (PACK && valid) {
            if (base + lane < out.pack_capacity) {
                if (out.d_means3D && out.accumulate) {   // earlier partial sums of the step travel too, and their rows are cleared
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        dm[k] += out.d_means3D[3 * (size_t)i + k]; ds[k] += out.d_scales[3 * (size_t)i + k]; dcol[k] += out.d_colors[3 * (size_t)i + k];
                        out.d_means3D[3 * (size_t)i + k] = 0.f; out.d_scales[3 * (size_t)i + k] = 0.f; out.d_colors[3 * (size_t)i + k] = 0.f;
                    }
                    float4* dr = reinterpret_cast<float4*>(out.d_rotations) + i;
                    const float4 o = *dr;
                    dq[0] += o.x; dq[1] += o.y; dq[2] += o.z; dq[3] += o.w;
                    *dr = make_float4(0.f, 0.f, 0.f, 0.f);
                    dop += out.d_opacities[i]; out.d_opacities[i] = 0.f;
                }
                float4* rec = reinterpret_cast<float4*>(out.pack_segment + 16 + (size_t)(base + lane) * 16);
                rec[0] = make_float4(dm[0], dm[1], dm[2], ds[0]);
                rec[1] = make_float4(ds[1], ds[2], dq[0], dq[1]);
                rec[2] = make_float4(dq[2], dq[3], dop, dcol[0]);
                rec[3] = make_float4(dcol[1], dcol[2], __int_as_float(i), 0.f);
            } else if (out.d_means3D) {
                // the segment is full: like ags_rows_pack, rows that do not travel keep their totals in the gradient arrays
                // (already += above when accumulate is set)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    out.d_means3D[3 * (size_t)i + k] = dm[k]; out.d_scales[3 * (size_t)i + k] = ds[k]; out.d_colors[3 * (size_t)i + k] = dcol[k];
                }
                reinterpret_cast<float4*>(out.d_rotations)[i] = make_float4(dq[0], dq[1], dq[2], dq[3]);
                out.d_opacities[i] = dop;
            }
        }
        if (FUSED_ADAM && valid) {
            const AgsAdamClock* clk = (const AgsAdamClock*)out.adam_clock;
            const float ib = clk->inv_bc2_sqrt, b1 = out.adam_beta1, b2 = out.adam_beta2, eps = out.adam_eps;
            const float gr[14] = {dm[0], dm[1], dm[2], ds[0], ds[1], ds[2], dq[0], dq[1], dq[2], dq[3], dop,
                                  dcol[0], dcol[1], dcol[2]};
#pragma unroll
            for (int e = 0; e < 14; ++e) {
                const int seg = (e >= 3) + (e >= 6) + (e >= 10) + (e >= 11);
                const float g = gr[e];
                const float m = am[e] + (1.f - b1) * (g - am[e]);
                const float vv = av[e] * b2 + (1.f - b2) * g * g;
                const float denom = sqrtf(vv) * ib + eps;
                am[e] = m; av[e] = vv;
                ap[e] -= clk->step_size[seg] * (m / denom);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                adam.p[0][3 * (size_t)i + k] = ap[k]; adam.p[1][3 * (size_t)i + k] = ap[3 + k]; adam.p[4][3 * (size_t)i + k] = ap[11 + k];
            }
            reinterpret_cast<float4*>(adam.p[2])[i] = make_float4(ap[6], ap[7], ap[8], ap[9]);
            adam.p[3][i] = ap[10];
            if (adam.st) {
                float4* sr = reinterpret_cast<float4*>(adam.st + (size_t)i * 28);
                float mv[28];
#pragma unroll
                for (int e = 0; e < 14; ++e) { mv[e] = am[e]; mv[14 + e] = av[e]; }
#pragma unroll
                for (int qq = 0; qq < 7; ++qq) sr[qq] = make_float4(mv[4 * qq], mv[4 * qq + 1], mv[4 * qq + 2], mv[4 * qq + 3]);
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    adam.m[0][3 * (size_t)i + k] = am[k]; adam.v[0][3 * (size_t)i + k] = av[k];
                    adam.m[1][3 * (size_t)i + k] = am[3 + k]; adam.v[1][3 * (size_t)i + k] = av[3 + k];
                    adam.m[4][3 * (size_t)i + k] = am[11 + k]; adam.v[4][3 * (size_t)i + k] = av[11 + k];
                }
                reinterpret_cast<float4*>(adam.m[2])[i] = make_float4(am[6], am[7], am[8], am[9]);
                reinterpret_cast<float4*>(adam.v[2])[i] = make_float4(av[6], av[7], av[8], av[9]);
                adam.m[3][i] = am[10]; adam.v[3][i] = av[10];
            }
        }
    }
}

void ags_launch_rows_multi(const AgsRowViews& rv, const AgsGaussians& in, const AgsGaussianGrads& din, hipStream_t s) {
    if (!din.touched.rows) {   // a row range of the map, no row set (capi.hip checked the range)
        int blocks = (din.row_end - din.row_begin + AGS_ROWS_THREADS - 1) / AGS_ROWS_THREADS;
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(ags_k_rows_multi<3>, dim3(blocks), dim3(AGS_ROWS_THREADS), 0, s, rv, in, din, AgsAdamArgs());
        return;
    }
    int blocks = (in.n + AGS_ROWS_THREADS - 1) / AGS_ROWS_THREADS;
    if (blocks > 16384) blocks = 16384; // fixed grid (the member count lives on the device): idle blocks exit at once
    if (din.fused_adam)
        hipLaunchKernelGGL(ags_k_rows_multi<1>, dim3(blocks), dim3(AGS_ROWS_THREADS), 0, s, rv, in, din, ags_adam_args(*din.fused_adam));
    else if (din.pack_segment)
        hipLaunchKernelGGL(ags_k_rows_multi<2>, dim3(blocks), dim3(AGS_ROWS_THREADS), 0, s, rv, in, din, AgsAdamArgs());
    else
        hipLaunchKernelGGL(ags_k_rows_multi<0>, dim3(blocks), dim3(AGS_ROWS_THREADS), 0, s, rv, in, din, AgsAdamArgs());
}

template <int MODE>
__global__ __launch_bounds__(AGS_ROWS_THREADS) void ags_k_preprocess_bwd_rows(
    AgsFrame F, const float* __restrict__ Vp, const float* __restrict__ Pp, AgsGaussians in,
    const int* __restrict__ radii, AgsGeomGrad* __restrict__ dgeom, AgsGaussianGrads out, AgsAdamArgs adam,
    AgsViewStride vs) {
    ags_frame_flags(F);
    { // (offsets are 0 for a single view)
        Vp += 16 * blockIdx.y; Pp += 16 * blockIdx.y;
        radii += (size_t)blockIdx.y * (size_t)vs.n;
        AGS_WS_SHIFT(dgeom, (size_t)blockIdx.y * (size_t)vs.ws);
    }
    ags_rows_body<MODE, false, false>(F, Vp, Pp, in, radii, dgeom, out, adam, (int)blockIdx.x, (int)gridDim.x, nullptr, nullptr);
}

// The software-pipelined optimisation step's per-Gaussian kernel (single view, single rank, one-pass binning): ONE
// launch does what ags_k_preprocess_bwd_rows<1> of step k and ags_k_preprocess<2> of step k + 1 do in two.  The two
// are consecutive kernels of the training loop, both are one generation of waves that live for a chain of dependent
// loads, and the second depends on the first only ROW BY ROW (a surfel's projection needs that surfel's updated
// parameters and nothing else), so:
//   * workgroups [0, rows_blocks): one lane per MEMBER row - the gradient chain rule and the Adam update as before, then
//     the next view's per-Gaussian stage for that row from the registers the update left (no reload);
//   * workgroups [rows_blocks, ...): the forward kernel's body over all rows, skipping member rows.
// Every row is projected exactly once; no workgroup waits for another.  Saved per step: one kernel boundary (~2.5 us
// inside a replayed graph), one launch ramp and drain (~4 us), and the shorter of the two latency chains.
template <bool AGG>
#ifndef AGS_FUSED_SUB
#define AGS_FUSED_SUB 1       // (2 and 4 measured the same step time: the launch's ramp and drain, not its waves, are the rest)
#endif
#ifndef AGS_FUSED_WAVES
#define AGS_FUSED_WAVES 3     // register budget: 3 waves per SIMD (168 VGPRs)
#endif
__global__ __launch_bounds__(AGS_PRE_THREADS) __attribute__((amdgpu_waves_per_eu(AGS_FUSED_WAVES, AGS_FUSED_WAVES))) void ags_k_rows_adam_preprocess(
    AgsFrame F, const float* __restrict__ Vp, const float* __restrict__ Pp, AgsGaussians in,
    const int* __restrict__ radii, AgsGeomGrad* __restrict__ dgeom, AgsGaussianGrads out, AgsAdamArgs adam,
    AgsNextPre nx, int rows_blocks, int n_blocks) {
    ags_frame_flags(F);
    ags_frame_flags(nx.F);
    __shared__ AgsEmitRec emit_rows[AGS_PRE_THREADS];
    if ((int)blockIdx.x < rows_blocks) {
        const int wave = threadIdx.x >> 6;
        ags_rows_body<1, true, AGG>(F, Vp, Pp, in, radii, dgeom, out, adam, (int)blockIdx.x * (AGS_PRE_THREADS / 64) + wave,
                                    rows_blocks * (AGS_PRE_THREADS / 64), &nx, emit_rows + (threadIdx.x & ~63));
        return;
    }
    // AGS_FUSED_SUB blocks of 256 rows per forward workgroup, one after the other: the member-row waves hold most wave
    // slots for the whole launch, so the forward rows want FEW waves (in steady state their lanes are all culled - every
    // visible surfel is a member - and a pass over 256 rows is ~4.5 us)
    const int fb = ((int)blockIdx.x - rows_blocks) * AGS_FUSED_SUB;
#pragma unroll 1
    for (int sub = 0; sub < AGS_FUSED_SUB; ++sub) {
        if (fb + sub >= n_blocks) break;       // workgroup-uniform
        if (sub) __syncthreads();              // the body's LDS is reused
        ags_preprocess_block<2, AGG, true>(nx.F, nx.V, nx.P, in, nx.geom, nullptr, nullptr, nx.radii, nullptr, nullptr,
                                           nx.tile_count, nx.dgeom, out.touched, nx.direct, fb + sub);
    }
}

void ags_launch_preprocess(const AgsFrame& F, const AgsCamera& cam, const AgsGaussians& in, char* ws,
                           const AgsLayout& L, const AgsPerGaussian& pg, int emit,
                           const AgsViewStride& vs, hipStream_t s) {
    int* radii = pg.radii;
    const AgsRowSet& touched = pg.touched;
    float* zero_importance = cam.config ? pg.importance : nullptr;
    int* zero_count = cam.config ? pg.count : nullptr;
    const AgsDirectEmit direct = {(uint64_t*)(ws + L.keys0), ags_direct_tile_cap(L), (uint32_t*)(ws + L.totals), (uint32_t)L.tc_stride,
                                  (uint32_t*)(ws + L.status) + AGS_STATUS_EARLY};
#define AGS_LAUNCH_PRE(EMIT, AGG)                                                                                        \
    hipLaunchKernelGGL((ags_k_preprocess<EMIT, AGG>), dim3(L.n_blocks, vs.views), dim3(AGS_PRE_THREADS), 0, s, F,          \
                       cam.viewmatrix, cam.projmatrix, in, (AgsGeom*)(ws + L.geom), (uint32_t*)(ws + L.tiles),             \
                       (ushort4*)(ws + L.rect), radii, (uint32_t*)(ws + L.block_sums), (uint32_t*)(ws + L.block_vis),      \
                       (uint32_t*)(ws + L.tile_count), (float4*)(ws + L.dgeom), touched, direct, zero_importance,         \
                       zero_count, vs)
    const bool agg = L.num_tiles <= AGS_AGG_MAX_TILES;
    // large maps, raw parameters, one-pass binning: cull with the means first, project the survivors on full waves
    // (AgsTuning.cull_first_min_n: 0 = AGS_CULL_MIN_N rows, < 0 = never)
    const int cull_min_n = L.tune.cull_first_min_n ? L.tune.cull_first_min_n : AGS_CULL_MIN_N;
    if (emit == 2 && !agg && in.raw_params && cull_min_n > 0 && in.n >= cull_min_n && in.max_scale > 0.f) {
        hipLaunchKernelGGL(ags_k_preprocess_cull, dim3((in.n + AGS_CULL_ROWS - 1) / AGS_CULL_ROWS, vs.views), dim3(AGS_PRE_THREADS), 0, s, F,
                           cam.viewmatrix, cam.projmatrix, in, (AgsGeom*)(ws + L.geom), radii, (uint32_t*)(ws + L.tile_count),
                           (float4*)(ws + L.dgeom), touched, direct, zero_importance, zero_count, vs);
        return;
    }
    // a BATCH of views under one-pass binning, OPT-IN (AgsTuning.view_group = k > 1): the rows' inputs are loaded and
    // activated once per group of k views (ags_k_preprocess_views).  Measured (profiles/r06_view_group_ab.md): the mapper's
    // batch of eleven 512x512 views 80.9 -> 84.4 us (the time is the visible rows' projection and key emission, which one
    // view per workgroup spreads over eleven times as many workgroups; the shared loads hit the L2 anyway), a planner's
    // hundred 128x128 views 1.54 -> 1.39 ms.  0 / 1 = one view per workgroup (ags_k_preprocess, blockIdx.y = view).
    if (emit == 2 && vs.views > 1 && L.tune.view_group > 1) {
        const int per = L.tune.view_group;
        {
            const dim3 grid(L.n_blocks, (vs.views + per - 1) / per);
#define AGS_LAUNCH_VIEWS(AGG)                                                                                             \
    hipLaunchKernelGGL((ags_k_preprocess_views<AGG>), grid, dim3(AGS_PRE_THREADS), 0, s, F, cam.viewmatrix, cam.projmatrix, in, \
                       (AgsGeom*)(ws + L.geom), radii, (uint32_t*)(ws + L.tile_count), (float4*)(ws + L.dgeom), touched,    \
                       direct, zero_importance, zero_count, vs, per)
            if (agg) AGS_LAUNCH_VIEWS(true); else AGS_LAUNCH_VIEWS(false);
#undef AGS_LAUNCH_VIEWS
            return;
        }
    }
    if (emit == 0) AGS_LAUNCH_PRE(0, false);
    else if (emit == 1) { if (agg) AGS_LAUNCH_PRE(1, true); else AGS_LAUNCH_PRE(1, false); }
    else { if (agg) AGS_LAUNCH_PRE(2, true); else AGS_LAUNCH_PRE(2, false); }
#undef AGS_LAUNCH_PRE
}

void ags_launch_rows_adam_preprocess(const AgsFrame& F, const AgsCamera& cam, const AgsGaussians& in, char* ws,
                                     const AgsLayout& L, const int* radii, const AgsGaussianGrads& din,
                                     const AgsFrame& F2, const AgsCamera& cam2, char* ws2, const AgsLayout& L2, int* radii2,
                                     int rows_hint, hipStream_t s) {
    AgsNextPre nx;
    nx.F = F2; nx.V = cam2.viewmatrix; nx.P = cam2.projmatrix;
    nx.geom = (AgsGeom*)(ws2 + L2.geom); nx.dgeom = (float4*)(ws2 + L2.dgeom); nx.radii = radii2;
    nx.tile_count = (uint32_t*)(ws2 + L2.tile_count);
    nx.direct = AgsDirectEmit{(uint64_t*)(ws2 + L2.keys0), ags_direct_tile_cap(L2), (uint32_t*)(ws2 + L2.totals), (uint32_t)L2.tc_stride,
                              (uint32_t*)(ws2 + L2.status) + AGS_STATUS_EARLY};
    nx.count_snap = (const uint32_t*)(ws + L.status) + AGS_STATUS_COUNT_SNAP;
    // the member count lives on the device: the member workgroups stride over the list, so any number of them is
    // correct; `rows_hint` (what the caller last saw, 0 = no idea) sizes them for one pass
    constexpr int rows_per_block = (AGS_PRE_THREADS / 64) * AGS_NEXT_ROWS;
    const long long want = rows_hint > 0 ? (long long)rows_hint : (long long)in.n;
    int rows_blocks = (int)((want + rows_per_block - 1) / rows_per_block);
    if (rows_blocks < 1) rows_blocks = 1;
    if (rows_blocks > 4096) rows_blocks = 4096;
    const dim3 grid(rows_blocks + (L2.n_blocks + AGS_FUSED_SUB - 1) / AGS_FUSED_SUB), block(AGS_PRE_THREADS);
    if (L2.num_tiles <= AGS_AGG_MAX_TILES)
        hipLaunchKernelGGL(ags_k_rows_adam_preprocess<true>, grid, block, 0, s, F, cam.viewmatrix, cam.projmatrix, in, radii,
                           (AgsGeomGrad*)(ws + L.dgeom), din, ags_adam_args(*din.fused_adam), nx, rows_blocks, L2.n_blocks);
    else
        hipLaunchKernelGGL(ags_k_rows_adam_preprocess<false>, grid, block, 0, s, F, cam.viewmatrix, cam.projmatrix, in, radii,
                           (AgsGeomGrad*)(ws + L.dgeom), din, ags_adam_args(*din.fused_adam), nx, rows_blocks, L2.n_blocks);
}

void ags_launch_preprocess_bwd(const AgsFrame& F, const AgsCamera& cam, const AgsGaussians& in, char* ws,
                               const AgsLayout& L, const int* radii, const AgsGaussianGrads& din,
                               const AgsViewStride& vs, hipStream_t s) {
    if (din.touched.rows) {
        // the member count lives on the device: a fixed grid strides over the list
        int blocks = (in.n + AGS_ROWS_THREADS - 1) / AGS_ROWS_THREADS;
        if (blocks > 16384) blocks = 16384; // fixed grid (the member count lives on the device): idle blocks exit at once
        if (din.fused_adam)
            hipLaunchKernelGGL(ags_k_preprocess_bwd_rows<1>, dim3(blocks), dim3(AGS_ROWS_THREADS), 0, s, F,
                               cam.viewmatrix, cam.projmatrix, in, radii, (AgsGeomGrad*)(ws + L.dgeom), din,
                               ags_adam_args(*din.fused_adam), vs);
        else if (din.pack_segment)
            hipLaunchKernelGGL(ags_k_preprocess_bwd_rows<2>, dim3(blocks), dim3(AGS_ROWS_THREADS), 0, s, F,
                               cam.viewmatrix, cam.projmatrix, in, radii, (AgsGeomGrad*)(ws + L.dgeom), din,
                               AgsAdamArgs(), vs);
        else
            hipLaunchKernelGGL(ags_k_preprocess_bwd_rows<0>, dim3(blocks, vs.views), dim3(AGS_ROWS_THREADS), 0, s, F,
                               cam.viewmatrix, cam.projmatrix, in, radii, (AgsGeomGrad*)(ws + L.dgeom), din,
                               AgsAdamArgs(), vs);
        return;
    }
    hipLaunchKernelGGL(ags_k_preprocess_bwd, dim3(L.n_blocks, vs.views), dim3(AGS_PRE_THREADS), 0, s, F, cam.viewmatrix,
                       cam.projmatrix, in, radii, (AgsGeomGrad*)(ws + L.dgeom), din, vs);
}
