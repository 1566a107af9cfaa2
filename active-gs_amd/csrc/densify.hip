// Map growth and pruning as device kernels:
//   get_smooth_depth      /root/reference/utils/operations.py:161-169   -> ags_k_bilateral
//   add_gaussians         /root/reference/mapping/gaussian_map.py:294-400 -> ags_k_candidates
//   cal_mask              /root/reference/mapping/gaussian_map.py:470-489 -> ags_k_candidates
//   voxel_downsample      /root/reference/utils/operations.py:603-625   -> ags_k_voxel_insert / _resolve
//   prune + boolean index /root/reference/mapping/gaussian_map.py:234-246 -> ags_k_prune_keep + compaction
// All of it is HBM/latency-bound per-pixel or per-row work (no contraction): one lane per pixel or
// per row, coalesced planar image reads, a hash table in HBM for the voxel filter, and a stable
// two-level scan for the compaction so that the surviving rows keep the reference's order.
#include "ags_internal.h"

#define AGS_BIL_TILE 16

// ---------------------------------------------------------------------------------------------
// cv2.bilateralFilter(depth, d, sigma_color, sigma_space) for one float channel, restated from the
// published filter (oracle/densify_oracle.py explains what is pinned): circular support of radius
// d/2, BORDER_REFLECT_101, weights exp(-r^2/(2 ss^2)) * exp(-dv^2/(2 sc^2)).  Fused with
// get_smooth_depth's handling of invalid (< 0) pixels: they enter as 0 and leave as -1.
// A 16x16 workgroup stages its (16+2r)^2 neighbourhood in LDS once.
__global__ __launch_bounds__(AGS_BIL_TILE * AGS_BIL_TILE) void ags_k_bilateral(
    int H, int W, const float* __restrict__ depth, float* __restrict__ out, int radius, float coef_space,
    float coef_color) {
    extern __shared__ float tile[];
    const int span = AGS_BIL_TILE + 2 * radius;
    const int x0 = blockIdx.x * AGS_BIL_TILE - radius, y0 = blockIdx.y * AGS_BIL_TILE - radius;
    for (int k = threadIdx.y * AGS_BIL_TILE + threadIdx.x; k < span * span; k += AGS_BIL_TILE * AGS_BIL_TILE) {
        int sx = x0 + k % span, sy = y0 + k / span;
        // reflect-101 (valid while radius < image size; the host checks)
        sx = sx < 0 ? -sx : (sx >= W ? 2 * W - 2 - sx : sx);
        sy = sy < 0 ? -sy : (sy >= H ? 2 * H - 2 - sy : sy);
        sx = min(max(sx, 0), W - 1); sy = min(max(sy, 0), H - 1); // tiles hanging over the far border
        const float v = depth[(size_t)sy * W + sx];
        tile[k] = v < 0.f ? 0.f : v;
    }
    __syncthreads();
    const int x = blockIdx.x * AGS_BIL_TILE + threadIdx.x, y = blockIdx.y * AGS_BIL_TILE + threadIdx.y;
    if (x >= W || y >= H) return;
    const float* c = tile + (threadIdx.y + radius) * span + threadIdx.x + radius;
    const float v0 = *c;
    float acc = 0.f, wsum = 0.f;
    const int r2max = radius * radius;
    for (int dy = -radius; dy <= radius; ++dy)
        for (int dx = -radius; dx <= radius; ++dx) {
            const int r2 = dy * dy + dx * dx;
            if (r2 > r2max) continue;
            const float v = c[dy * span + dx];
            const float dv = v - v0;
            const float w = expf((float)r2 * coef_space) * expf(dv * dv * coef_color);
            acc += v * w;
            wsum += w;
        }
    out[(size_t)y * W + x] = depth[(size_t)y * W + x] < 0.f ? -1.0f : acc / wsum;
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void ags_cam_point(int x, int y, int H, int W, float d, float ik00, float ik11, float p[3]) {
    // depth2normal's back-projection (operations.py:184-193): centred pixel grid, K00 from H, K11 from W
    p[0] = ((float)x - 0.5f * (float)W) * d * ik00;
    p[1] = ((float)y - 0.5f * (float)H) * d * ik11;
    p[2] = d;
}
__device__ __forceinline__ void ags_cross_acc(const float a[3], const float b[3], float n[3]) {
    n[0] += a[1] * b[2] - a[2] * b[1];
    n[1] += a[2] * b[0] - a[0] * b[2];
    n[2] += a[0] * b[1] - a[1] * b[0];
}

// One lane per pixel of a new keyframe: everything add_gaussians derives per pixel before the
// voxel filter.  Kinv = inverse of the normalised intrinsics, E = camera-to-world (row-major 4x4).
__global__ __launch_bounds__(256) void ags_k_candidates(AgsKeyframe f, const float* __restrict__ depth_smooth,
                                                        AgsDensifyPred pred, float error_thres, float ik00, float ik11,
                                                        AgsCandidates out) {
    const int P = f.image_height * f.image_width;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int H = f.image_height, W = f.image_width;
    const int y = i / W, x = i - y * W;
    float Ki[9], R[9], t[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) Ki[k] = f.intrinsic_inv[k];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) R[r * 3 + c] = f.extrinsic[r * 4 + c];
        t[r] = f.extrinsic[r * 4 + 3];
    }
    const float d = f.depth[i];
    bool valid = d > 0.0f;
    // ---- normal from the smoothed depth (depth2normal with mask = depth > 0, replicate border)
    const int xl = max(x - 1, 0), xr = min(x + 1, W - 1), yu = max(y - 1, 0), yb = min(y + 1, H - 1);
    const float mc = valid ? 1.f : 0.f;
    const float mu = f.depth[yu * W + x] > 0.f ? 1.f : 0.f, mb = f.depth[yb * W + x] > 0.f ? 1.f : 0.f;
    const float ml = f.depth[y * W + xl] > 0.f ? 1.f : 0.f, mr = f.depth[y * W + xr] > 0.f ? 1.f : 0.f;
    float pc[3], pu[3], pl[3], pb[3], pr[3];
    ags_cam_point(x, y, H, W, depth_smooth[i], ik00, ik11, pc);
    ags_cam_point(x, yu, H, W, depth_smooth[yu * W + x], ik00, ik11, pu);
    ags_cam_point(x, yb, H, W, depth_smooth[yb * W + x], ik00, ik11, pb);
    ags_cam_point(xl, y, H, W, depth_smooth[y * W + xl], ik00, ik11, pl);
    ags_cam_point(xr, y, H, W, depth_smooth[y * W + xr], ik00, ik11, pr);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        pc[k] *= mc;
        pu[k] = (pu[k] - pc[k]) * mu; pl[k] = (pl[k] - pc[k]) * ml;
        pb[k] = (pb[k] - pc[k]) * mb; pr[k] = (pr[k] - pc[k]) * mr;
    }
    float n[3] = {0.f, 0.f, 0.f};
    ags_cross_acc(pu, pl, n); ags_cross_acc(pr, pu, n); ags_cross_acc(pb, pr, n); ags_cross_acc(pl, pb, n);
    const float inv = mc / fmaxf(sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), 1e-12f);
    n[0] *= inv; n[1] *= inv; n[2] *= inv;
    valid = valid && (n[0] * n[0] + n[1] * n[1] + n[2] * n[2]) > 0.0f;
    // ---- ray, point, world normal
    const float u = ((float)x + 0.5f) / (float)W, v = ((float)y + 0.5f) / (float)H;
    const float dc[3] = {Ki[0] * u + Ki[1] * v + Ki[2], Ki[3] * u + Ki[4] * v + Ki[5], Ki[6] * u + Ki[7] * v + Ki[8]};
    float dw[3], nw[3] = {0.f, 0.f, 1.f};
#pragma unroll
    for (int r = 0; r < 3; ++r) dw[r] = R[r * 3] * dc[0] + R[r * 3 + 1] * dc[1] + R[r * 3 + 2] * dc[2];
    if (valid) {
#pragma unroll
        for (int r = 0; r < 3; ++r) nw[r] = R[r * 3] * n[0] + R[r * 3 + 1] * n[1] + R[r * 3 + 2] * n[2];
    }
    const float dl = 1.0f / fmaxf(sqrtf(dw[0] * dw[0] + dw[1] * dw[1] + dw[2] * dw[2]), 1e-12f);
    const float cs = (dw[0] * nw[0] + dw[1] * nw[1] + dw[2] * nw[2]) * dl;
    valid = valid && cs < -0.01f;
    // ---- normal2rotation + rotmat2quaternion (operations.py:481-500,526-541)
    float z[3];
    {
        const float zl = sqrtf(nw[0] * nw[0] + nw[1] * nw[1] + nw[2] * nw[2]);
        z[0] = nw[0] / zl; z[1] = nw[1] / zl; z[2] = nw[2] / zl;
    }
    const bool par = fabsf(z[0]) > 0.99f;
    const float ref[3] = {par ? 0.f : 1.f, par ? 1.f : 0.f, 0.f};
    const float rz = ref[0] * z[0] + ref[1] * z[1] + ref[2] * z[2];
    float ax[3] = {ref[0] - rz * z[0], ref[1] - rz * z[1], ref[2] - rz * z[2]};
    const float al = sqrtf(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    ax[0] /= al; ax[1] /= al; ax[2] /= al;
    float ay[3] = {z[1] * ax[2] - z[2] * ax[1], z[2] * ax[0] - z[0] * ax[2], z[0] * ax[1] - z[1] * ax[0]};
    const float yl = sqrtf(ay[0] * ay[0] + ay[1] * ay[1] + ay[2] * ay[2]);
    ay[0] /= yl; ay[1] /= yl; ay[2] /= yl;
    // R = [x y z] as columns: R[r][0] = ax[r], R[r][1] = ay[r], R[r][2] = z[r]
    const float tr = ax[0] + ay[1] + z[2] + 1e-6f;
    const float qr = sqrtf(1.f + tr) * 0.5f;
    float q[4] = {qr, (ay[2] - z[1]) / (4.f * qr), (z[0] - ax[2]) / (4.f * qr), (ax[1] - ay[0]) / (4.f * qr)};
    const float ql = 1.0f / fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
    q[0] *= ql; q[1] *= ql; q[2] *= ql; q[3] *= ql;
    valid = valid && !(isnan(q[0]) || isnan(q[1]) || isnan(q[2]) || isnan(q[3]));
    // ---- where does the map need new surfels (cal_mask)
    const size_t HW = (size_t)P;
    const float c0 = f.rgb[i], c1 = f.rgb[HW + i], c2 = f.rgb[2 * HW + i];
    bool want = true;
    if (pred.rgb) {
        const float e0 = c0 - pred.rgb[i], e1 = c1 - pred.rgb[HW + i], e2 = c2 - pred.rgb[2 * HW + i];
        want = (e0 * e0 + e1 * e1 + e2 * e2) / 3.0f > error_thres;
        want = want || pred.opacity[i] < 0.5f;
        want = want || (d - pred.depth[i]) < -0.05f * d;
    }
    out.means[3 * (size_t)i + 0] = t[0] + dw[0] * d;
    out.means[3 * (size_t)i + 1] = t[1] + dw[1] * d;
    out.means[3 * (size_t)i + 2] = t[2] + dw[2] * d;
    reinterpret_cast<float4*>(out.rotations)[i] = make_float4(q[0], q[1], q[2], q[3]);
    out.harmonics[3 * (size_t)i + 0] = c0; out.harmonics[3 * (size_t)i + 1] = c1; out.harmonics[3 * (size_t)i + 2] = c2;
    out.select[i] = (want && valid) ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// voxel_downsample: one point per occupied voxel.  Open-addressing table of 64-bit voxel keys in
// HBM (capacity = power of two >= 2 * points), value = highest point index seen (the rule the
// pinned reference run follows, see oracle/densify_oracle.py).
#define AGS_VOX_EMPTY 0xFFFFFFFFFFFFFFFFull
__device__ __forceinline__ unsigned long long ags_voxel_key(const float* p, float voxel) {
    // torch.floor(point / voxel).long() per axis; 21 bits each (|coordinate| < 2^20 voxels = 20 km at 2 cm)
    const long long ix = (long long)floorf(p[0] / voxel) + (1 << 20);
    const long long iy = (long long)floorf(p[1] / voxel) + (1 << 20);
    const long long iz = (long long)floorf(p[2] / voxel) + (1 << 20);
    return ((unsigned long long)(ix & 0x1FFFFF) << 42) | ((unsigned long long)(iy & 0x1FFFFF) << 21) |
           (unsigned long long)(iz & 0x1FFFFF);
}
__device__ __forceinline__ uint32_t ags_voxel_hash(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (uint32_t)k;
}

__global__ __launch_bounds__(256) void ags_k_voxel_insert(int n, const float* __restrict__ points,
                                                          const int32_t* __restrict__ select, float voxel,
                                                          unsigned long long* keys, int32_t* vals, uint32_t mask) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !select[i]) return;
    const unsigned long long key = ags_voxel_key(points + 3 * (size_t)i, voxel);
    uint32_t slot = ags_voxel_hash(key) & mask;
    for (;;) {
        const unsigned long long prev = atomicCAS(&keys[slot], AGS_VOX_EMPTY, key);
        if (prev == AGS_VOX_EMPTY || prev == key) break;
        slot = (slot + 1) & mask;
    }
    atomicMax(&vals[slot], i);
}

__global__ __launch_bounds__(256) void ags_k_voxel_resolve(int n, const float* __restrict__ points,
                                                           int32_t* __restrict__ select, float voxel,
                                                           const unsigned long long* __restrict__ keys,
                                                           const int32_t* __restrict__ vals, uint32_t mask) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !select[i]) return;
    const unsigned long long key = ags_voxel_key(points + 3 * (size_t)i, voxel);
    uint32_t slot = ags_voxel_hash(key) & mask;
    while (keys[slot] != key) slot = (slot + 1) & mask; // inserted by the previous launch
    select[i] = vals[slot] == i ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// prune's drop rule: prune_mask | sigmoid(raw opacity) < min_opacity  ->  keep flags
__global__ __launch_bounds__(256) void ags_k_prune_keep(int n, const float* __restrict__ prune_mask,
                                                        const float* __restrict__ raw_opacities, float min_opacity,
                                                        int32_t* __restrict__ keep) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const bool drop = (prune_mask && prune_mask[i] != 0.f) || (1.0f / (1.0f + expf(-raw_opacities[i])) < min_opacity);
    keep[i] = drop ? 0 : 1;
}

// ---------------------------------------------------------------------------------------------
// post_processing's per-surfel bookkeeping (/root/reference/mapping/gaussian_map.py:193-232): which surfels the newest
// keyframe sees (count >= 1), their support count, the running mean of the unit directions they were seen from and
// the view score (1 - dist / far) * max(cos(normal, direction), 0) - and get_confidences (:552-565) from them.  One
// launch instead of ~25 torch ops per keyframe.  Plain f32 expressions in the reference's order.
__global__ __launch_bounds__(256) void ags_k_view_stats(int n, const float* __restrict__ means,
                                                        const float* __restrict__ raw_rotations,
                                                        const float* __restrict__ campos, float far,
                                                        const int32_t* __restrict__ newest_count, int use_vd,
                                                        float* __restrict__ view_supports, float* __restrict__ view_means,
                                                        float* __restrict__ view_scores) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const bool seen = newest_count[i] >= 1;
    const float sup = view_supports[i] + (seen ? 1.f : 0.f);
    view_supports[i] = sup;
    if (!use_vd || !seen) return;
    const float4 q4 = reinterpret_cast<const float4*>(raw_rotations)[i];
    const float ql = fmaxf(sqrtf(q4.x * q4.x + q4.y * q4.y + q4.z * q4.z + q4.w * q4.w), 1e-12f);
    const float r = q4.x / ql, x = q4.y / ql, y = q4.z / ql, z = q4.w / ql;
    float nx = 2.f * (x * z + r * y), ny = 2.f * (y * z - r * x), nz = 1.f - 2.f * (x * x + y * y);
    const float nl = fmaxf(sqrtf(nx * nx + ny * ny + nz * nz), 1e-12f);
    nx /= nl; ny /= nl; nz /= nl;
    float tx = campos[0] - means[3 * (size_t)i], ty = campos[1] - means[3 * (size_t)i + 1], tz = campos[2] - means[3 * (size_t)i + 2];
    const float dist = sqrtf(tx * tx + ty * ty + tz * tz);
    tx /= dist; ty /= dist; tz /= dist;
    const float den = fmaxf(sup, 1.f);
    float* vm = view_means + 3 * (size_t)i;
    vm[0] = vm[0] + (tx - vm[0]) / den; vm[1] = vm[1] + (ty - vm[1]) / den; vm[2] = vm[2] + (tz - vm[2]) / den;
    const float c = fminf(fmaxf(nx * tx + ny * ty + nz * tz, 0.f), 1.f);
    view_scores[i] = view_scores[i] + (1.f - fminf(fmaxf(dist / far, 0.f), 1.f)) * c;
}

__global__ __launch_bounds__(256) void ags_k_confidences(int n, const float* __restrict__ view_supports,
                                                         const float* __restrict__ view_means,
                                                         const float* __restrict__ view_scores, int use_vd,
                                                         float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float c;
    if (use_vd) {
        const float* vm = view_means + 3 * (size_t)i;
        float var = sqrtf(vm[0] * vm[0] + vm[1] * vm[1] + vm[2] * vm[2]);
        if (var != var) var = 1.f;                              // NaN -> 1 (gaussian_map.py:556)
        c = expf(1.f - var) * view_scores[i];
    } else {
        c = 1.f - 1.f / expf(view_supports[i]);
    }
    out[i] = fminf(fmaxf(c, 0.f), 1.f);
}

// ---------------------------------------------------------------------------------------------
// Stable compaction plan: dst_index[i] = number of kept rows before i (or -1).  Level 1: per
// 1024-row chunk counts; level 2: one workgroup scans the chunk counts; level 3: each chunk
// rescans its flags from its base.  Order-preserving like boolean indexing in torch.
#define AGS_CHUNK 1024
__global__ __launch_bounds__(256) void ags_k_chunk_count(int n, const int32_t* __restrict__ keep,
                                                         int32_t* __restrict__ chunk_sum) {
    __shared__ uint32_t ws[4];
    const int base = blockIdx.x * AGS_CHUNK;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < AGS_CHUNK / 256; ++k) {
        const int i = base + k * 256 + threadIdx.x;
        c += (i < n && keep[i]) ? 1u : 0u;
    }
    c = ags_wave_sum_u32(c);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) chunk_sum[blockIdx.x] = (int32_t)(ws[0] + ws[1] + ws[2] + ws[3]);
}

__global__ __launch_bounds__(1024) void ags_k_chunk_scan(int chunks, int32_t* __restrict__ chunk_sum,
                                                         int32_t* __restrict__ total) {
    __shared__ uint32_t ws[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < chunks; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < chunks ? (uint32_t)chunk_sum[i] : 0u;
        const uint32_t incl = ags_wave_incl_scan_u32(v);
        if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t pre = carry;
        for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) pre += ws[k];
        if (i < chunks) chunk_sum[i] = (int32_t)(pre + incl - v); // exclusive
        __syncthreads();
        if (threadIdx.x == 1023) carry = pre + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = (int32_t)carry;
}

__global__ __launch_bounds__(256) void ags_k_chunk_index(int n, const int32_t* __restrict__ keep,
                                                         const int32_t* __restrict__ chunk_base,
                                                         int32_t* __restrict__ dst_index) {
    __shared__ uint32_t ws[4];
    const int base = blockIdx.x * AGS_CHUNK;
    uint32_t run = (uint32_t)chunk_base[blockIdx.x];
    for (int k = 0; k < AGS_CHUNK / 256; ++k) {
        const int i = base + k * 256 + threadIdx.x;
        const uint32_t f = (i < n && keep[i]) ? 1u : 0u;
        const uint32_t incl = ags_wave_incl_scan_u32(f);
        if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t pre = run;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) pre += ws[w];
        if (i < n) dst_index[i] = f ? (int32_t)(pre + incl - 1u) : -1;
        run += ws[0] + ws[1] + ws[2] + ws[3];
        __syncthreads();
    }
}

// dst[dst_index[i]] = src[i] for the kept rows of one (n, width) array; lanes run over elements so
// that reads are fully coalesced and writes nearly so (kept rows stay in order).
__global__ __launch_bounds__(256) void ags_k_compact_rows(long long elems, int width,
                                                          const int32_t* __restrict__ dst_index,
                                                          const float* __restrict__ src, float* __restrict__ dst) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= elems) return;
    const long long row = e / width;
    const int col = (int)(e - row * width);
    const int32_t d = dst_index[row];
    if (d >= 0) dst[(size_t)d * width + col] = src[e];
}

// All eight arrays of the map in one launch each way (one lane per candidate / per row): the growth used to be eight
// zero-fills, eight copies of the old map into fresh arrays, three row compactions and a strided fill per keyframe.
__global__ __launch_bounds__(256) void ags_k_map_append(int P, const int32_t* __restrict__ dst_index, AgsCandidates c,
                                                        float new_z, AgsMapArrays o) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int32_t d = dst_index[i];
    if (d < 0) return;
    const size_t r = (size_t)d;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o.means[3 * r + k] = c.means[3 * (size_t)i + k];
        o.harmonics[3 * r + k] = c.harmonics[3 * (size_t)i + k];
        o.view_means[3 * r + k] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) o.rotations[4 * r + k] = c.rotations[4 * (size_t)i + k];
    o.scales[3 * r] = 0.f; o.scales[3 * r + 1] = 0.f; o.scales[3 * r + 2] = new_z;
    o.opacities[r] = 0.f; o.view_scores[r] = 0.f; o.view_supports[r] = 0.f;
}
__global__ __launch_bounds__(256) void ags_k_map_compact(int n, const int32_t* __restrict__ dst_index, AgsMapArrays a,
                                                         AgsMapArrays o) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t d = dst_index[i];
    if (d < 0) return;
    const size_t r = (size_t)d, q = (size_t)i;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o.means[3 * r + k] = a.means[3 * q + k];
        o.scales[3 * r + k] = a.scales[3 * q + k];
        o.harmonics[3 * r + k] = a.harmonics[3 * q + k];
        o.view_means[3 * r + k] = a.view_means[3 * q + k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) o.rotations[4 * r + k] = a.rotations[4 * q + k];
    o.opacities[r] = a.opacities[q]; o.view_scores[r] = a.view_scores[q]; o.view_supports[r] = a.view_supports[q];
}
void ags_launch_map_append(int P, const int32_t* dst_index, const AgsCandidates& c, float new_z, const AgsMapArrays& o, hipStream_t s) {
    hipLaunchKernelGGL(ags_k_map_append, dim3((P + 255) / 256), dim3(256), 0, s, P, dst_index, c, new_z, o);
}
void ags_launch_map_compact(int n, const int32_t* dst_index, const AgsMapArrays& a, const AgsMapArrays& o, hipStream_t s) {
    hipLaunchKernelGGL(ags_k_map_compact, dim3((n + 255) / 256), dim3(256), 0, s, n, dst_index, a, o);
}

// ---------------------------------------------------------------------------------------------
void ags_launch_bilateral(int h, int w, const float* depth, float* out, int d, float sigma_color, float sigma_space,
                          hipStream_t s) {
    const int radius = d / 2;
    const int span = AGS_BIL_TILE + 2 * radius;
    hipLaunchKernelGGL(ags_k_bilateral, dim3((w + AGS_BIL_TILE - 1) / AGS_BIL_TILE, (h + AGS_BIL_TILE - 1) / AGS_BIL_TILE),
                       dim3(AGS_BIL_TILE, AGS_BIL_TILE), (size_t)span * span * sizeof(float), s, h, w, depth, out, radius,
                       -0.5f / (sigma_space * sigma_space), -0.5f / (sigma_color * sigma_color));
}

void ags_launch_candidates(const AgsKeyframe& f, const float* depth_smooth, const AgsDensifyPred& pred,
                           float error_thres, const AgsCandidates& out, hipStream_t s) {
    const int P = f.image_height * f.image_width;
    // add_gaussians always calls depth2normal with fov = (pi/3, pi/3) (gaussian_map.py:318-320), and
    // depth2normal pairs fov[0] with H and fov[1] with W (operations.py:188-189)
    const double tan30 = 0.57735026918962576451;
    const float k00 = (float)(f.image_height / (2.0 * tan30)), k11 = (float)(f.image_width / (2.0 * tan30));
    hipLaunchKernelGGL(ags_k_candidates, dim3((P + 255) / 256), dim3(256), 0, s, f, depth_smooth, pred, error_thres,
                       1.0f / k00, 1.0f / k11, out);
}

static uint32_t ags_voxel_capacity(int n) {
    uint32_t cap = 1024;
    while (cap < 2u * (uint32_t)n) cap <<= 1;
    return cap;
}
size_t ags_voxel_bytes(int n) { return (size_t)ags_voxel_capacity(n) * 12; }

void ags_launch_voxel_select(int n, const float* points, int32_t* select, float voxel, void* ws, hipStream_t s) {
    const uint32_t cap = ags_voxel_capacity(n);
    unsigned long long* keys = (unsigned long long*)ws;
    int32_t* vals = (int32_t*)((char*)ws + (size_t)cap * 8);
    (void)hipMemsetAsync(ws, 0xFF, (size_t)cap * 12, s); // keys = EMPTY, vals = -1
    hipLaunchKernelGGL(ags_k_voxel_insert, dim3((n + 255) / 256), dim3(256), 0, s, n, points, select, voxel, keys, vals, cap - 1);
    hipLaunchKernelGGL(ags_k_voxel_resolve, dim3((n + 255) / 256), dim3(256), 0, s, n, points, select, voxel, keys, vals, cap - 1);
}

size_t ags_compact_bytes(int n) { return ((size_t)(n + AGS_CHUNK - 1) / AGS_CHUNK + 1) * 4; }

void ags_launch_compact_plan(int n, const int32_t* keep, int32_t* dst_index, int32_t* total, void* scratch, hipStream_t s) {
    const int chunks = (n + AGS_CHUNK - 1) / AGS_CHUNK;
    int32_t* sums = (int32_t*)scratch;
    hipLaunchKernelGGL(ags_k_chunk_count, dim3(chunks), dim3(256), 0, s, n, keep, sums);
    hipLaunchKernelGGL(ags_k_chunk_scan, dim3(1), dim3(1024), 0, s, chunks, sums, total);
    hipLaunchKernelGGL(ags_k_chunk_index, dim3(chunks), dim3(256), 0, s, n, keep, sums, dst_index);
}

void ags_launch_compact_rows(int n, int width, const int32_t* dst_index, const float* src, float* dst, hipStream_t s) {
    const long long elems = (long long)n * width;
    hipLaunchKernelGGL(ags_k_compact_rows, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, s, elems, width, dst_index,
                       src, dst);
}

void ags_launch_view_stats(int n, const float* means, const float* raw_rotations, const float* campos, float far,
                           const int32_t* newest_count, int use_vd, float* view_supports, float* view_means,
                           float* view_scores, hipStream_t s) {
    hipLaunchKernelGGL(ags_k_view_stats, dim3((n + 255) / 256), dim3(256), 0, s, n, means, raw_rotations, campos, far,
                       newest_count, use_vd, view_supports, view_means, view_scores);
}
void ags_launch_confidences(int n, const float* view_supports, const float* view_means, const float* view_scores, int use_vd,
                            float* out, hipStream_t s) {
    hipLaunchKernelGGL(ags_k_confidences, dim3((n + 255) / 256), dim3(256), 0, s, n, view_supports, view_means, view_scores,
                       use_vd, out);
}
void ags_launch_prune_keep(int n, const float* prune_mask, const float* raw_opacities, float min_opacity, int32_t* keep,
                           hipStream_t s) {
    hipLaunchKernelGGL(ags_k_prune_keep, dim3((n + 255) / 256), dim3(256), 0, s, n, prune_mask, raw_opacities, min_opacity, keep);
}
