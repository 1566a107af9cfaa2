// Stage 1 of the loss head for ONE pixel (/root/reference/utils/operations.py:714-715 normalise * mask;
// /root/reference/mapping/gaussian_map.py:106-112 with mapping/utils.py:14-16 masked L1 of rgb and depth): shared by
// ags_k_loss_stage1 (loss.hip) and the epilogue of the forward blend kernel (render.hip, ags_forward_batch_loss), which
// must give the same bits.  The two translation units are compiled with different floating-point flags, so the contraction
// of multiply-adds is pinned HERE (off): every operation below rounds on its own in both.
#pragma once
#include <hip/hip_runtime.h>

struct AgsStage1Pixel {
    float n[3];          // post-processed normal
    float d_rgb[3];      // dL/d rgb
    float d_depth;       // dL/d depth (the direct L1 part; stage 2 adds the stencil terms)
    float s_rgb, s_dep;  // this pixel's L1 sums
    int vis;             // opacity > 1e-3: counted in msum
};

__device__ __forceinline__ AgsStage1Pixel ags_loss_stage1_pixel(float o, float nx, float ny, float nz, float r, float g,
                                                                float b, float gr, float gg, float gb, float depth,
                                                                float dg, float k_rgb, float k_depth) {
#pragma clang fp contract(off)
    AgsStage1Pixel out;
    const float mvis = o > 1e-3f ? 1.f : 0.f, mn = o > 1e-2f ? 1.f : 0.f;
    const float inv = mn / fmaxf(sqrtf(nx * nx + ny * ny + nz * nz), 1e-12f);
    out.n[0] = nx * inv; out.n[1] = ny * inv; out.n[2] = nz * inv;
    const float c[3] = {r, g, b}, t[3] = {gr, gg, gb};
    out.s_rgb = 0.f;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float e = (c[ch] - t[ch]) * mvis;
        out.s_rgb += fabsf(e);
        out.d_rgb[ch] = (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) * mvis * k_rgb;
    }
    const float md = dg > 0.f ? 1.f : 0.f;
    const float e = (depth - dg) * md;
    out.s_dep = fabsf(e);
    out.d_depth = (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) * md * k_depth;
    out.vis = o > 1e-3f ? 1 : 0;
    return out;
}
