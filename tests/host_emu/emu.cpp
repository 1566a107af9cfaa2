// TEST-ONLY sequential driver for active-gs_amd/csrc/surfel_math.h.
//
// Compiled with g++ by tests/conftest.py into tests/host_emu/libags_emu.so.  It lets
// the CPU test-suite check the hand-derived per-Gaussian and per-pixel backward (the
// exact inline functions the HIP kernels call) against autograd of the oracle without
// a GPU.  It is NOT a product path: nothing under active-gs_amd/ loads it.
#include "surfel_math.h"

#include <algorithm>
#include <cstring>
#include <vector>

namespace {
struct Inst { uint64_t key; uint32_t gid; };

struct Scene {
    AgsFrame F;
    std::vector<AgsGeom> geom;
    std::vector<int> radius;
    std::vector<int> rect;
    std::vector<uint32_t> sorted;        // gid per instance, sorted
    std::vector<uint32_t> range;         // 2*T
};

AgsFrame make_frame(int H, int W, float tanfovx, float tanfovy, float scale_mod, int perpix, int front_only) {
    AgsFrame F;
    F.H = H; F.W = W;
    F.tiles_x = (W + AGS_TILE - 1) / AGS_TILE; F.tiles_y = (H + AGS_TILE - 1) / AGS_TILE;
    F.tanfovx = tanfovx; F.tanfovy = tanfovy;
    F.fx = W / (2.0f * tanfovx); F.fy = H / (2.0f * tanfovy);
    F.scale_mod = scale_mod; F.perpix_depth = perpix; F.front_only = front_only; F.cfg = nullptr;
    return F;
}

void build(Scene& S, const float* V, const float* P, int N, const float* means, const float* scales,
           const float* rots, const float* opac, const float* colors, const float* conf) {
    S.geom.assign(N, AgsGeom());
    S.radius.assign(N, 0);
    S.rect.assign(4 * N, 0);
    std::vector<Inst> inst;
    for (int i = 0; i < N; ++i) {
        int rad = 0, rc[4] = {0, 0, 0, 0};
        AgsGeom g;
        if (!ags_preprocess_fwd(S.F, V, P, means + 3 * i, scales + 3 * i, rots + 4 * i, opac[i], colors + 3 * i,
                                conf[i], 0.f, 0.f, g, rad, rc))
            continue;
        S.geom[i] = g; S.radius[i] = rad;
        for (int k = 0; k < 4; ++k) S.rect[4 * i + k] = rc[k];
        uint32_t dbits; std::memcpy(&dbits, &g.dc, 4);
        for (int y = rc[1]; y < rc[3]; ++y)
            for (int x = rc[0]; x < rc[2]; ++x)
                inst.push_back({(uint64_t)(y * S.F.tiles_x + x) << 32 | dbits, (uint32_t)i});
    }
    std::stable_sort(inst.begin(), inst.end(), [](const Inst& a, const Inst& b) { return a.key < b.key; });
    const int T = S.F.tiles_x * S.F.tiles_y;
    S.range.assign(2 * T, 0);
    S.sorted.resize(inst.size());
    for (size_t k = 0; k < inst.size(); ++k) {
        S.sorted[k] = inst[k].gid;
        const uint32_t t = inst[k].key >> 32;
        if (k == 0 || (inst[k - 1].key >> 32) != t) S.range[2 * t] = k;
        S.range[2 * t + 1] = k + 1;
    }
}
} // namespace

extern "C" {

// images are planar (C,H,W); final_T (H,W); n_contrib (H,W)
long emu_forward(int H, int W, float tanfovx, float tanfovy, float scale_mod, int normalize_depth, int perpix,
                 int front_only, int want_stats, float weight_thres, const float* mask, const float* V,
                 const float* P, const float* bg, int N, const float* means, const float* scales,
                 const float* rots, const float* opac, const float* colors, const float* conf, float* rgb,
                 float* normal, float* depth, float* opacity, float* confidence, float* final_T,
                 int32_t* n_contrib, float* importance, int32_t* count, int32_t* radii, float* geom_out) {
    Scene S;
    S.F = make_frame(H, W, tanfovx, tanfovy, scale_mod, perpix, front_only);
    build(S, V, P, N, means, scales, rots, opac, colors, conf);
    for (int i = 0; i < N; ++i) {
        radii[i] = S.radius[i];
        importance[i] = 0.f; count[i] = 0;
        if (geom_out) std::memcpy(geom_out + 16 * i, &S.geom[i], 64);
    }
    const size_t HW = (size_t)H * W;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int t = (y / AGS_TILE) * S.F.tiles_x + x / AGS_TILE;
            AgsPix s; ags_pix_init(s, true);
            const uint32_t b = S.range[2 * t], e = S.range[2 * t + 1];
            const bool m = !mask || mask[y * W + x] > 0.f;
            for (uint32_t k = b; k < e && !s.done; ++k) {
                const uint32_t gid = S.sorted[k];
                const float w = ags_blend_fwd(s, S.geom[gid], (float)x, (float)y, k - b + 1);
                if (want_stats && m && w > 0.f) {
                    importance[gid] += w;
                    if (w > weight_thres) count[gid] += 1;
                }
            }
            const size_t o = (size_t)y * W + x;
            const float A = 1.f - s.T;
            rgb[o] = s.c0 + s.T * bg[0]; rgb[HW + o] = s.c1 + s.T * bg[1]; rgb[2 * HW + o] = s.c2 + s.T * bg[2];
            normal[o] = s.n0; normal[HW + o] = s.n1; normal[2 * HW + o] = s.n2;
            depth[o] = normalize_depth ? s.d / fmaxf(A, AGS_DEPTH_A_EPS) : s.d;
            opacity[o] = A; confidence[o] = s.cf; final_T[o] = s.T; n_contrib[o] = (int32_t)s.last;
        }
    return (long)S.sorted.size();
}

void emu_backward(int H, int W, float tanfovx, float tanfovy, float scale_mod, int normalize_depth, int perpix,
                  int front_only, const float* V, const float* P, const float* bg, int N, const float* means,
                  const float* scales, const float* rots, const float* opac, const float* colors,
                  const float* conf, const float* depth_out, const float* opac_out, const float* final_T,
                  const int32_t* n_contrib, const float* d_rgb, const float* d_normal, const float* d_depth,
                  const float* d_opacity, const float* d_conf, float* dmeans, float* dscales, float* drots,
                  float* dopac, float* dcolors, float* dmeans2d, float* dgeom_out) {
    Scene S;
    S.F = make_frame(H, W, tanfovx, tanfovy, scale_mod, perpix, front_only);
    build(S, V, P, N, means, scales, rots, opac, colors, conf);
    std::vector<AgsGeomGrad> acc(N);
    std::memset(acc.data(), 0, sizeof(AgsGeomGrad) * N);
    const size_t HW = (size_t)H * W;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int t = (y / AGS_TILE) * S.F.tiles_x + x / AGS_TILE;
            const size_t o = (size_t)y * W + x;
            const float dC[3] = {d_rgb[o], d_rgb[HW + o], d_rgb[2 * HW + o]};
            const float dN[3] = {d_normal[o], d_normal[HW + o], d_normal[2 * HW + o]};
            AgsPixGrad s;
            ags_pixgrad_init(s, dC, dN, d_depth[o], d_opacity[o], d_conf[o], depth_out[o], opac_out[o], final_T[o],
                             (uint32_t)n_contrib[o], bg, normalize_depth);
            const uint32_t b = S.range[2 * t];
            for (uint32_t pos1 = s.last; pos1 >= 1; --pos1) {
                const uint32_t gid = S.sorted[b + pos1 - 1];
                ags_blend_bwd(s, S.geom[gid], (float)x, (float)y, pos1, acc[gid]);
            }
        }
    for (int i = 0; i < N; ++i) {
        if (dgeom_out) std::memcpy(dgeom_out + 16 * i, &acc[i], 64);
        if (S.radius[i] > 0) {
            ags_preprocess_bwd(S.F, V, P, means + 3 * i, scales + 3 * i, rots + 4 * i, opac[i], acc[i], dmeans + 3 * i,
                               dscales + 3 * i, drots + 4 * i, dopac + i, dcolors + 3 * i, dmeans2d + 3 * i);
            dmeans2d[3 * i + 2] = 0.f;
        } else {
            for (int k = 0; k < 3; ++k) dmeans[3 * i + k] = dscales[3 * i + k] = dcolors[3 * i + k] = dmeans2d[3 * i + k] = 0.f;
            for (int k = 0; k < 4; ++k) drots[4 * i + k] = 0.f;
            dopac[i] = 0.f;
        }
    }
}
}
