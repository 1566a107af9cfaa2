// Per-Gaussian kernels: F1 (cull + project + conic + tile rect) and B2+B3 fused
// (conic/mean2D/depth/normal gradients -> means3D, scales, rotations, opacity, colour).
// One lane per Gaussian; the camera matrices are wave-uniform (scalar loads).
// Counterpart of the preprocess / preprocess-backward stages listed in SURVEY.md §2.3.
#include "ags_internal.h"

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
        if (t < 192) reinterpret_cast<float4*>(lds)[t] = reinterpret_cast<const float4*>(src)[t];
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
        if (t < 192) {
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

template <bool COUNT_TILES>
__global__ __launch_bounds__(AGS_PRE_THREADS) void ags_k_preprocess(
    AgsFrame F, const float* __restrict__ Vp, const float* __restrict__ Pp, AgsGaussians in,
    AgsGeom* __restrict__ geom, uint32_t* __restrict__ tiles, ushort4* __restrict__ rect,
    int* __restrict__ radii, uint32_t* __restrict__ block_sums, uint32_t* __restrict__ block_vis,
    uint32_t* __restrict__ tile_count, float4* __restrict__ dgeom) {
    __shared__ uint32_t wsum[AGS_PRE_THREADS / 64], wvis[AGS_PRE_THREADS / 64];
    __shared__ AgsEmitRec emit[COUNT_TILES ? AGS_PRE_THREADS : 1];
    __shared__ __attribute__((aligned(16))) float rows3[3 * AGS_PRE_THREADS];
    float V[16], P[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { V[k] = Vp[k]; P[k] = Pp[k]; }
    const int first = blockIdx.x * AGS_PRE_THREADS;
    const int i = first + threadIdx.x;
    const int rows = min(AGS_PRE_THREADS, in.n - first);
    uint32_t cnt = 0, vis = 0;
    uint32_t rx0 = 0, ry0 = 0, rwd = 1;
    AgsGeom g;
    g.mx = g.my = g.ca = g.cb = g.cc = g.o = 0.f;
    float p[3], sc[3], col[3];
    ags_load_rows3(in.means3D, first, rows, rows3, p);
    ags_load_rows3(in.scales, first, rows, rows3, sc);
    ags_load_rows3(in.colors, first, rows, rows3, col);
    if (i < in.n) {
        const float4 q4 = reinterpret_cast<const float4*>(in.rotations)[i];
        float q[4] = {q4.x, q4.y, q4.z, q4.w};
        float opacity = in.opacities[i];
        if (in.raw_params) { float rv[3], qi; ags_activate_inplace(in, sc, q, opacity, rv, qi); }
        int radius = 0, rc[4];
        if (ags_preprocess_fwd(F, V, P, p, sc, q, opacity, col, in.confidences[i], 0.f, 0.f, g, radius, rc)) {
            float4* dst = reinterpret_cast<float4*>(geom + i);
            dst[0] = make_float4(g.mx, g.my, g.ca, g.cb);
            dst[1] = make_float4(g.cc, g.o, g.dc, g.gx);
            dst[2] = make_float4(g.gy, g.r, g.g, g.b);
            dst[3] = make_float4(g.nx, g.ny, g.nz, g.conf);
            // the gradient record the blend backward accumulates into starts at zero (no memset pass)
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            dgeom[4 * (size_t)i + 0] = z4; dgeom[4 * (size_t)i + 1] = z4;
            dgeom[4 * (size_t)i + 2] = z4; dgeom[4 * (size_t)i + 3] = z4;
            rect[i] = make_ushort4((unsigned short)rc[0], (unsigned short)rc[1], (unsigned short)rc[2], (unsigned short)rc[3]);
            cnt = (uint32_t)((rc[2] - rc[0]) * (rc[3] - rc[1]));
            rx0 = (uint32_t)rc[0]; ry0 = (uint32_t)rc[1]; rwd = (uint32_t)(rc[2] - rc[0]);
            vis = 1;
        }
        radii[i] = radius;
        tiles[i] = cnt;
    }
    if (COUNT_TILES)  // tile-sort binning: how many surfels can reach each tile
        ags_emit_tiles_balanced(emit + (threadIdx.x & ~63), cnt, rx0, ry0, rwd, 0u, g, F.tiles_x,
                                [&](uint32_t t, uint32_t) { atomicAdd(&tile_count[t], 1u); });
    const uint32_t ws = ags_wave_sum_u32(cnt), wv = ags_wave_sum_u32(vis);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { wsum[wave] = ws; wvis[wave] = wv; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = 0, b = 0;
#pragma unroll
        for (int k = 0; k < AGS_PRE_THREADS / 64; ++k) { a += wsum[k]; b += wvis[k]; }
        block_sums[blockIdx.x] = a;
        block_vis[blockIdx.x] = b;  // summed by the (single-workgroup) scan kernel: no fan-in atomics
    }
}

__global__ __launch_bounds__(AGS_PRE_THREADS) void ags_k_preprocess_bwd(
    AgsFrame F, const float* __restrict__ Vp, const float* __restrict__ Pp, AgsGaussians in,
    const int* __restrict__ radii, AgsGeomGrad* __restrict__ dgeom, AgsGaussianGrads out) {
    __shared__ __attribute__((aligned(16))) float rows3[3 * AGS_PRE_THREADS];
    float V[16], P[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { V[k] = Vp[k]; P[k] = Pp[k]; }
    const int first = blockIdx.x * AGS_PRE_THREADS;
    const int i = first + threadIdx.x;
    const int rows = min(AGS_PRE_THREADS, in.n - first);
    // side job of the step's last backward: advance the Adam device clock (this launch completes
    // before the Adam kernel starts, so every Adam block sees the new scalars)
    if (out.adam_clock && i == 0) ags_adam_tick((AgsAdamClock*)out.adam_clock, out.adam_lr, out.adam_beta1, out.adam_beta2, 0);
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
        dg.dmx = a.x; dg.dmy = a.y; dg.dca = a.z; dg.dcb = a.w;
        dg.dcc = b.x; dg.dop = b.y; dg.ddc = b.z; dg.dgx = b.w;
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

void ags_launch_preprocess(const AgsFrame& F, const AgsCamera& cam, const AgsGaussians& in, char* ws,
                           const AgsLayout& L, int* radii, bool count_tiles, hipStream_t s) {
    if (count_tiles)
        hipLaunchKernelGGL(ags_k_preprocess<true>, dim3(L.n_blocks), dim3(AGS_PRE_THREADS), 0, s, F, cam.viewmatrix,
                           cam.projmatrix, in, (AgsGeom*)(ws + L.geom), (uint32_t*)(ws + L.tiles),
                           (ushort4*)(ws + L.rect), radii, (uint32_t*)(ws + L.block_sums),
                           (uint32_t*)(ws + L.block_vis), (uint32_t*)(ws + L.tile_count), (float4*)(ws + L.dgeom));
    else
        hipLaunchKernelGGL(ags_k_preprocess<false>, dim3(L.n_blocks), dim3(AGS_PRE_THREADS), 0, s, F, cam.viewmatrix,
                           cam.projmatrix, in, (AgsGeom*)(ws + L.geom), (uint32_t*)(ws + L.tiles),
                           (ushort4*)(ws + L.rect), radii, (uint32_t*)(ws + L.block_sums),
                           (uint32_t*)(ws + L.block_vis), (uint32_t*)(ws + L.tile_count), (float4*)(ws + L.dgeom));
}

void ags_launch_preprocess_bwd(const AgsFrame& F, const AgsCamera& cam, const AgsGaussians& in, char* ws,
                               const AgsLayout& L, const int* radii, const AgsGaussianGrads& din, hipStream_t s) {
    hipLaunchKernelGGL(ags_k_preprocess_bwd, dim3(L.n_blocks), dim3(AGS_PRE_THREADS), 0, s, F, cam.viewmatrix,
                       cam.projmatrix, in, radii, (AgsGeomGrad*)(ws + L.dgeom), din);
}
