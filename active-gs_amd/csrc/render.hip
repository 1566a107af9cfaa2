// Per-tile alpha compositing, forward (F6) and backward (B1).
//
// Wave64-native shape, not a 16x16-thread CUDA block: ONE wavefront owns a 16x16 tile and
// every lane owns four pixels (rows y0+{0,4,8,12}), so
//   * a round stages 64 projected surfels (64 B each, one per lane) into 4 KiB of LDS and
//     needs no cross-wave barrier;
//   * each staged record is read from LDS once per wave (broadcast ds_read_b128) and
//     serves four pixel evaluations — a quarter of the LDS traffic of a 256-thread tile;
//   * the four 16x4 strips keep their own reject: `__any` over the 64-bit ballot skips the
//     accumulate (forward) or the whole gradient block (backward) when no lane takes the
//     surfel, and `__all(done)` ends the tile early;
//   * the backward reduces each surfel's 15 partial gradients over the wave with a transposed
//     permlane-swap/DPP reduction (ags_wave_reduce16) and issues ONE 15-lane atomic per
//     (surfel,tile) instead of 256 x 15 atomics.
// Blocks are mapped to tiles XCD-aware (ags_xcd_remap) so one XCD's L2 serves a contiguous
// band of tiles.  Counterpart of renderCUDA fwd/bwd in SURVEY.md §2.3; arithmetic in
// surfel_math.h.
#include "ags_internal.h"

template <bool STATS>
__global__ __launch_bounds__(64) void ags_k_render_fwd(
    AgsFrame F, int normalize_depth, float weight_thres, const float* __restrict__ bgp,
    const float* __restrict__ mask, const uint2* __restrict__ ranges, const uint32_t* __restrict__ vals,
    int id_stride, const AgsGeom* __restrict__ geom, AgsImages out, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, float* __restrict__ importance, int* __restrict__ count, int num_tiles) {
    __shared__ AgsGeom sg[64];
    __shared__ uint32_t sid[64];
    __shared__ uint32_t smask[64];
    const int tile = ags_xcd_remap(blockIdx.x, num_tiles);
    const int tx = tile % F.tiles_x, ty = tile / F.tiles_x;
    const int lane = threadIdx.x;
    const int px = tx * AGS_TILE + (lane & 15);
    const int py0 = ty * AGS_TILE + (lane >> 4);
    const float fpx = (float)px;
    const uint2 rg = ranges[tile];
    const float bx0 = (float)(tx * AGS_TILE), by0 = (float)(ty * AGS_TILE);
    // four named accumulators (not an array): keeps every field in VGPRs
    AgsPix pix0, pix1, pix2, pix3;
    float mk0 = 1.f, mk1 = 1.f, mk2 = 1.f, mk3 = 1.f;
#define AGS_SLOT_INIT(S, PIX, MK)                                                                   \
    {                                                                                               \
        const bool inside = (px < F.W) && (py0 + 4 * S < F.H);                                      \
        ags_pix_init(PIX, inside);                                                                  \
        if (STATS) {                                                                                \
            MK = inside ? 1.f : 0.f;                                                                \
            if (mask != nullptr && inside) MK = mask[(size_t)(py0 + 4 * S) * F.W + px] > 0.f ? 1.f : 0.f; \
        }                                                                                           \
    }
    AGS_SLOT_INIT(0, pix0, mk0) AGS_SLOT_INIT(1, pix1, mk1) AGS_SLOT_INIT(2, pix2, mk2) AGS_SLOT_INIT(3, pix3, mk3)
#undef AGS_SLOT_INIT
    for (uint32_t base = rg.x; base < rg.y; base += 64) {
        if (__all(pix0.done & pix1.done & pix2.done & pix3.done)) break;
        __syncthreads();
        const uint32_t idx = base + lane;
        if (idx < rg.y) {
            const uint32_t gid = vals[(size_t)idx * id_stride];
            const float4* src = reinterpret_cast<const float4*>(geom + gid);
            const float4 r0 = src[0], r1 = src[1], r2 = src[2], r3 = src[3];
            float4* dst = reinterpret_cast<float4*>(&sg[lane]);
            dst[0] = r0; dst[1] = r1; dst[2] = r2; dst[3] = r3;
            if (STATS) sid[lane] = gid;
            // which of the tile's four 16x4 strips can this surfel reach at all? (one lane per surfel)
            AgsGeom me;
            me.mx = r0.x; me.my = r0.y; me.ca = r0.z; me.cb = r0.w; me.cc = r1.x; me.o = r1.y;
            uint32_t m = 0;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                m |= ags_reaches_box(me, bx0, bx0 + 15.f, by0 + 4.f * s, by0 + 4.f * s + 3.f) ? (1u << s) : 0u;
            smask[lane] = m;
        }
        __syncthreads();
        const int cnt = (int)min(64u, rg.y - base);
        for (int k = 0; k < cnt; ++k) {
            const uint32_t m = __builtin_amdgcn_readfirstlane(smask[k]); // wave-uniform
            if (m == 0) continue;
            const AgsGeom g = sg[k];
            float dx0 = 0, dy0 = 0, al0 = 0, dx1 = 0, dy1 = 0, al1 = 0, dx2 = 0, dy2 = 0, al2 = 0, dx3 = 0, dy3 = 0, al3 = 0;
            bool ok0 = false, ok1 = false, ok2 = false, ok3 = false;
            if (m & 1u) ok0 = ags_alpha(g, fpx, (float)(py0), dx0, dy0, al0) && !pix0.done;
            if (m & 2u) ok1 = ags_alpha(g, fpx, (float)(py0 + 4), dx1, dy1, al1) && !pix1.done;
            if (m & 4u) ok2 = ags_alpha(g, fpx, (float)(py0 + 8), dx2, dy2, al2) && !pix2.done;
            if (m & 8u) ok3 = ags_alpha(g, fpx, (float)(py0 + 12), dx3, dy3, al3) && !pix3.done;
            if (!__any(ok0 | ok1 | ok2 | ok3)) continue;
            const uint32_t pos1 = base - rg.x + k + 1;
            float wsum = 0.f;
            uint32_t wcnt = 0;
#define AGS_SLOT_BLEND(OK, PIX, DX, DY, AL, MK)                                                     \
    if (OK) {                                                                                       \
        const float w = ags_blend_apply(PIX, g, DX, DY, AL, pos1);                                  \
        if (STATS) { const float wm = w * MK; wsum += wm; wcnt += (wm > weight_thres) ? 1u : 0u; }  \
    }
            AGS_SLOT_BLEND(ok0, pix0, dx0, dy0, al0, mk0) AGS_SLOT_BLEND(ok1, pix1, dx1, dy1, al1, mk1)
            AGS_SLOT_BLEND(ok2, pix2, dx2, dy2, al2, mk2) AGS_SLOT_BLEND(ok3, pix3, dx3, dy3, al3, mk3)
#undef AGS_SLOT_BLEND
            if (STATS) {
                const float ts = ags_wave_sum(wsum);
                const uint32_t tc = ags_wave_sum_u32(wcnt);
                if (lane == 0 && ts > 0.f) {
                    const uint32_t gid = sid[k];
                    atomicAdd(&importance[gid], ts);
                    if (tc) atomicAdd(&count[gid], (int)tc);
                }
            }
        }
    }
    const float bg0 = bgp[0], bg1 = bgp[1], bg2 = bgp[2];
    const size_t HW = (size_t)F.H * F.W;
#define AGS_SLOT_STORE(S, PIX)                                                                      \
    {                                                                                               \
        const int py = py0 + 4 * S;                                                                 \
        if (px < F.W && py < F.H) {                                                                 \
            const size_t o = (size_t)py * F.W + px;                                                 \
            const float A = 1.f - PIX.T;                                                            \
            out.rgb[o] = PIX.c0 + PIX.T * bg0; out.rgb[HW + o] = PIX.c1 + PIX.T * bg1;              \
            out.rgb[2 * HW + o] = PIX.c2 + PIX.T * bg2;                                             \
            out.normal[o] = PIX.n0; out.normal[HW + o] = PIX.n1; out.normal[2 * HW + o] = PIX.n2;   \
            out.depth[o] = normalize_depth ? PIX.d / fmaxf(A, AGS_DEPTH_A_EPS) : PIX.d;             \
            out.opacity[o] = A;                                                                     \
            out.confidence[o] = PIX.cf;                                                             \
            final_T[o] = PIX.T;                                                                     \
            n_contrib[o] = PIX.last;                                                                \
        }                                                                                           \
    }
    AGS_SLOT_STORE(0, pix0) AGS_SLOT_STORE(1, pix1) AGS_SLOT_STORE(2, pix2) AGS_SLOT_STORE(3, pix3)
#undef AGS_SLOT_STORE
}

__global__ __launch_bounds__(64) void ags_k_render_bwd(
    AgsFrame F, int normalize_depth, const float* __restrict__ bgp, const uint2* __restrict__ ranges,
    const uint32_t* __restrict__ vals, int id_stride, const AgsGeom* __restrict__ geom,
    const float* __restrict__ depth_out,
    const float* __restrict__ opac_out, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    AgsImageGrads dout, float* __restrict__ dgeom, int num_tiles) {
    __shared__ AgsGeom sg[64];
    __shared__ uint32_t sid[64];
    __shared__ uint32_t smask[64];
    const int tile = ags_xcd_remap(blockIdx.x, num_tiles);
    const uint2 rg = ranges[tile];
    if (rg.y <= rg.x) return;
    const int tx = tile % F.tiles_x, ty = tile / F.tiles_x;
    const int lane = threadIdx.x;
    const int px = tx * AGS_TILE + (lane & 15);
    const int py0 = ty * AGS_TILE + (lane >> 4);
    const float fpx = (float)px;
    const float bx0 = (float)(tx * AGS_TILE), by0 = (float)(ty * AGS_TILE);
    const float bg[3] = {bgp[0], bgp[1], bgp[2]};
    const size_t HW = (size_t)F.H * F.W;
    AgsPixGrad pg[4];
    uint32_t mymax = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int py = py0 + 4 * s;
        float dC[3] = {0, 0, 0}, dN[3] = {0, 0, 0}, dD = 0, dO = 0, dCf = 0, dep = 0, opa = 0, Tf = 1.f;
        uint32_t last = 0;
        if (px < F.W && py < F.H) {
            const size_t o = (size_t)py * F.W + px;
            last = n_contrib[o];
            if (last) {
                if (dout.d_rgb) { dC[0] = dout.d_rgb[o]; dC[1] = dout.d_rgb[HW + o]; dC[2] = dout.d_rgb[2 * HW + o]; }
                if (dout.d_normal) { dN[0] = dout.d_normal[o]; dN[1] = dout.d_normal[HW + o]; dN[2] = dout.d_normal[2 * HW + o]; }
                if (dout.d_depth) dD = dout.d_depth[o];
                if (dout.d_opacity) dO = dout.d_opacity[o];
                if (dout.d_confidence) dCf = dout.d_confidence[o];
                dep = depth_out[o]; opa = opac_out[o]; Tf = final_T[o];
            }
        }
        ags_pixgrad_init(pg[s], dC, dN, dD, dO, dCf, dep, opa, Tf, last, bg, normalize_depth);
        mymax = max(mymax, last);
    }
    const uint32_t maxlast = ags_wave_max_u32(mymax);
    if (maxlast == 0) return;
    // lanes (lane&15) < 4 each own one field of the reduced gradient record (15 is padding)
    const int my_field = ((lane & 15) < 4 && ags_reduce16_field(lane) < 15) ? ags_reduce16_field(lane) : -1;
    for (int r = (int)((maxlast - 1) >> 6); r >= 0; --r) {
        const uint32_t k0 = (uint32_t)r << 6;
        __syncthreads();
        if (k0 + lane < maxlast) {
            const uint32_t gid = vals[(size_t)(rg.x + k0 + lane) * id_stride];
            const float4* src = reinterpret_cast<const float4*>(geom + gid);
            const float4 r0 = src[0], r1 = src[1], r2 = src[2], r3 = src[3];
            float4* dst = reinterpret_cast<float4*>(&sg[lane]);
            dst[0] = r0; dst[1] = r1; dst[2] = r2; dst[3] = r3;
            sid[lane] = gid;
            AgsGeom me;
            me.mx = r0.x; me.my = r0.y; me.ca = r0.z; me.cb = r0.w; me.cc = r1.x; me.o = r1.y;
            uint32_t m = 0;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                m |= ags_reaches_box(me, bx0, bx0 + 15.f, by0 + 4.f * s, by0 + 4.f * s + 3.f) ? (1u << s) : 0u;
            smask[lane] = m;
        }
        __syncthreads();
        const int kend = (int)min(63u, maxlast - 1 - k0);
        for (int k = kend; k >= 0; --k) {
            const uint32_t m = __builtin_amdgcn_readfirstlane(smask[k]); // wave-uniform strip mask
            if (m == 0) continue;
            const AgsGeom g = sg[k];
            const uint32_t pos1 = k0 + k + 1;
            float dx[4] = {0, 0, 0, 0}, dy[4] = {0, 0, 0, 0}, al[4] = {0, 0, 0, 0};
            bool ok[4] = {false, false, false, false};
            bool any = false;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (m & (1u << s))
                    ok[s] = ags_alpha(g, fpx, (float)(py0 + 4 * s), dx[s], dy[s], al[s]) && (pos1 <= pg[s].last);
                any |= ok[s];
            }
            if (!__any(any)) continue;
            AgsGeomGrad acc;
            float* a = reinterpret_cast<float*>(&acc);
#pragma unroll
            for (int j = 0; j < 16; ++j) a[j] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (ok[s]) ags_blend_bwd_apply(pg[s], g, dx[s], dy[s], al[s], acc);
            const float mine = ags_wave_reduce16(a, lane); // 16 lanes end up owning one total each
            if (my_field >= 0) unsafeAtomicAdd(dgeom + (size_t)sid[k] * 16 + my_field, mine);
        }
    }
}

void ags_launch_render_fwd(const AgsFrame& F, const AgsCamera& cam, char* ws, const AgsLayout& L,
                           AgsIdList ids, const AgsImages& out, const AgsPerGaussian& pg, hipStream_t s) {
    const uint2* ranges = (const uint2*)(ws + L.ranges);
    const AgsGeom* geom = (const AgsGeom*)(ws + L.geom);
    float* fT = (float*)(ws + L.final_T);
    uint32_t* nc = (uint32_t*)(ws + L.n_contrib);
    if (cam.want_stats)
        hipLaunchKernelGGL(ags_k_render_fwd<true>, dim3(L.num_tiles), dim3(64), 0, s, F, cam.normalize_depth,
                           cam.weight_thres, cam.bg, cam.render_mask, ranges, ids.ids, ids.stride, geom, out, fT, nc,
                           pg.importance, pg.count, L.num_tiles);
    else
        hipLaunchKernelGGL(ags_k_render_fwd<false>, dim3(L.num_tiles), dim3(64), 0, s, F, cam.normalize_depth,
                           cam.weight_thres, cam.bg, cam.render_mask, ranges, ids.ids, ids.stride, geom, out, fT, nc,
                           pg.importance, pg.count, L.num_tiles);
}

void ags_launch_render_bwd(const AgsFrame& F, const AgsCamera& cam, char* ws, const AgsLayout& L,
                           AgsIdList ids, const AgsImages& fwd, const AgsImageGrads& dout, hipStream_t s) {
    hipLaunchKernelGGL(ags_k_render_bwd, dim3(L.num_tiles), dim3(64), 0, s, F, cam.normalize_depth, cam.bg,
                       (const uint2*)(ws + L.ranges), ids.ids, ids.stride, (const AgsGeom*)(ws + L.geom), fwd.depth,
                       fwd.opacity, (const float*)(ws + L.final_T), (const uint32_t*)(ws + L.n_contrib), dout,
                       (float*)(ws + L.dgeom), L.num_tiles);
}
