// Fused loss head: facade post-processing + depth->normal + the four training losses, forward
// and hand-derived backward, straight from the rasterizer's images to dL/d{rgb, normal, depth}.
//
// Replaces, per view, the torch ops of
//   /root/reference/utils/operations.py:714-718   normal = normalize(normal)*(opacity>1e-2); d2n = depth2normal(...)
//   /root/reference/utils/operations.py:172-219   depth2normal (4-neighbour cross products, replicate padding,
//                                                 fov_x paired with H and fov_y with W — kept)
//   /root/reference/mapping/gaussian_map.py:106-124 masked L1 rgb/depth, consistency, normal TV, weights 1/.8/.1/.1
//   /root/reference/mapping/utils.py:14-62,120-121  cons_loss_fc, normal_tv_loss_fc/central_diff, l1_loss_fc_mask
// including the reference's (B,H,W)x(B,1,H,W) broadcast in the consistency term: that term is
//   sum_b' sum_p (1 - n_b'(p).d2n_b'(p)) * Msum(p) / (B*B*H*W),  Msum = sum over ALL views of (opacity>1e-3),
// so stage 1 of every view of the batch runs (and Msum is all-reduced under view parallelism)
// before stage 2 of any view.
//   stage 1 (per pixel): n = normalised masked normal -> n image; direct L1 gradients d_rgb, d_depth;
//                        Msum += visibility; loss / per-frame error sums
//   stage 2 (per pixel): TV gradient wrt n(p) gathered from the 4 neighbours (each unordered pair counted
//                        from both sides), consistency gradient, chain rule through the normalisation ->
//                        d_normal; consistency gradient wrt d2n(p) through the cross products onto the
//                        depths of p and its 4 neighbours (gathered per 32x8 tile through LDS, added to d_depth)
#include "ags_internal.h"
#include "loss_pixel.h"

struct AgsLossDev {
    int H, W, B;
    float fxq, fyq;          // H/(2 tan(fov_x/2)), W/(2 tan(fov_y/2))  (the reference's pairing)
    float w_rgb, w_depth, w_cons, w_tv, inv_2sig2;
    const long long* gt_index;   // AgsLossConfig.gt_frame_index: batched calls read view v's ground truth at frame gt_index[v]
};

__device__ __forceinline__ float3 f3(float x, float y, float z) { return make_float3(x, y, z); }
__device__ __forceinline__ float3 operator-(float3 a, float3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float3 operator+(float3 a, float3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ float3 operator*(float3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
// back-projected point ray*d with ROUNDED products (no FMA contraction with a later subtraction): where
// a replicated neighbour is the pixel itself, P(neighbour) - P(centre) must cancel to exactly 0 as it
// does in the reference's torch ops, or the normalisation turns rounding residue into a unit vector
__device__ __forceinline__ float3 ags_point(float3 ray, float d) {
    return f3(__fmul_rn(ray.x, d), __fmul_rn(ray.y, d), __fmul_rn(ray.z, d));
}
// (this file is compiled with -ffp-contract=off for ags_point's sake; dot and cross products fuse explicitly: the stencil
// kernels are bound by vector-instruction issue, and nothing downstream relies on an unfused rounding of these)
__device__ __forceinline__ float dot3(float3 a, float3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
__device__ __forceinline__ float3 cross3(float3 a, float3 b) {
    return f3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
// b - a * s
__device__ __forceinline__ float3 fnma3(float3 a, float s, float3 b) { return f3(fmaf(-a.x, s, b.x), fmaf(-a.y, s, b.y), fmaf(-a.z, s, b.z)); }
// b + a * s
__device__ __forceinline__ float3 fma3(float3 a, float s, float3 b) { return f3(fmaf(a.x, s, b.x), fmaf(a.y, s, b.y), fmaf(a.z, s, b.z)); }

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
    v = ags_wave_sum(v);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[wave] = v;
    __syncthreads();
    const float t = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return t;
}

// accum layout (floats), AGS_LOSS_ACCUM_ROWS rows of `accum_stride` floats; a block adds into row
// (blockIdx & 63) so the thousands of per-block atomics do not serialise on one word; the
// reader sums the rows:  [0] rgb L1 sum, [1] depth L1 sum, [2] cons sum, [3] tv sum (all views);
//                        [4 + 2*view] rgb L1 sum of the view, [5 + 2*view] depth L1 sum of the view
__global__ __launch_bounds__(256) void ags_k_loss_stage1(
    AgsLossDev c, const float* __restrict__ rgb, const float* __restrict__ normal_raw,
    const float* __restrict__ depth, const float* __restrict__ opacity, const float* __restrict__ gt_rgb,
    const float* __restrict__ gt_depth, float* __restrict__ n_img, float* __restrict__ d_rgb,
    float* __restrict__ d_depth, int* __restrict__ msum, float* __restrict__ accum, int accum_stride, int view,
    int first_view) {
    __shared__ float sh[4];
    const int HW = c.H * c.W;
    if (gridDim.y > 1) { // batched: blockIdx.y = view of the batch, image batches are contiguous
        const size_t po = (size_t)blockIdx.y * (size_t)HW;
        const size_t go = c.gt_index ? (size_t)c.gt_index[blockIdx.y] * (size_t)HW : po;   // ground truth straight from the keyframe store
        rgb += 3 * po; normal_raw += 3 * po; depth += po; opacity += po; gt_rgb += 3 * go; gt_depth += go;
        n_img += 3 * po; d_rgb += 3 * po; d_depth += po;
        view += blockIdx.y;
    }
    const int p = blockIdx.x * 256 + threadIdx.x;
    float s_rgb = 0.f, s_dep = 0.f;
    if (p < HW) {
        const float k_rgb = c.w_rgb / ((float)c.B * 3.f * (float)HW), k_depth = c.w_depth / ((float)c.B * (float)HW);
        const AgsStage1Pixel px = ags_loss_stage1_pixel(opacity[p], normal_raw[p], normal_raw[HW + p], normal_raw[2 * HW + p],
                                                        rgb[p], rgb[HW + p], rgb[2 * HW + p], gt_rgb[p], gt_rgb[HW + p],
                                                        gt_rgb[2 * HW + p], depth[p], gt_depth[p], k_rgb, k_depth);
        n_img[p] = px.n[0]; n_img[HW + p] = px.n[1]; n_img[2 * HW + p] = px.n[2];
        d_rgb[p] = px.d_rgb[0]; d_rgb[HW + p] = px.d_rgb[1]; d_rgb[2 * HW + p] = px.d_rgb[2];
        d_depth[p] = px.d_depth;
        s_rgb = px.s_rgb; s_dep = px.s_dep;
        const int v = px.vis;
        if (first_view < 0) { if (v) atomicAdd(&msum[p], 1); }   // concurrent views into a pre-zeroed count
        else msum[p] = first_view ? v : msum[p] + v;
    }
    const float t_rgb = block_sum_256(s_rgb, sh), t_dep = block_sum_256(s_dep, sh);
    if (threadIdx.x == 0) {
        float* row = accum + (size_t)(blockIdx.x & (AGS_LOSS_ACCUM_ROWS - 1)) * accum_stride;
        atomicAdd(&row[0], t_rgb); atomicAdd(&row[1], t_dep);
        atomicAdd(&row[4 + 2 * view], t_rgb); atomicAdd(&row[5 + 2 * view], t_dep);
    }
}

__device__ __forceinline__ float3 load_n(const float* __restrict__ n_img, int HW, int q) {
    return f3(n_img[q], n_img[HW + q], n_img[2 * HW + q]);
}

// Stage 2 works on 32x8-pixel tiles.  The consistency term's gradient reaches the depths of a pixel AND of its four
// neighbours; until round 4 every pixel added its five contributions with atomics (14 M float atomics per batch of eleven
// 512x512 views, half the kernel's time, and a last-bit dependence on their order).  Now a workgroup first evaluates the
// stencil of every pixel of its tile and of the one-pixel ring around it (340 pixels: the ring's stencils reach into the
// tile) into LDS - the five depth contributions and d2n - and every tile pixel then GATHERS what is addressed to it in a
// fixed order (replicate padding: at an image border a pixel's "neighbour" is the pixel itself) and adds it to d_depth
// with one plain read-modify-write.  1.33 x the stencil arithmetic, no atomics, deterministic.
#define AGS_S2_TW 32
#define AGS_S2_TH 8
#define AGS_S2_RW (AGS_S2_TW + 2)
#define AGS_S2_RN (AGS_S2_RW * (AGS_S2_TH + 2))
__global__ __launch_bounds__(256) void ags_k_loss_stage2(
    AgsLossDev c, const float* __restrict__ depth, const float* __restrict__ opacity,
    const float* __restrict__ normal_raw, const float* __restrict__ n_img, const float* __restrict__ gt_depth,
    const int* __restrict__ msum, float* __restrict__ d_normal, float* __restrict__ d_depth,
    float* __restrict__ accum, int accum_stride) {
    __shared__ float sh[4];
    __shared__ float st[8][AGS_S2_RN];      // [0..4]: contributions to the depths of (centre, up, left, right, below); [5..7]: d2n
    const int H = c.H, W = c.W, HW = H * W;
    if (gridDim.z > 1) {
        const size_t po = (size_t)blockIdx.z * (size_t)HW;
        const size_t go = c.gt_index ? (size_t)c.gt_index[blockIdx.z] * (size_t)HW : po;
        depth += po; opacity += po; normal_raw += 3 * po; n_img += 3 * po; gt_depth += go;
        d_normal += 3 * po; d_depth += po;
    }
    const int x0 = blockIdx.x * AGS_S2_TW, y0 = blockIdx.y * AGS_S2_TH;
    const float k_c = c.w_cons / ((float)c.B * (float)c.B * (float)HW);
    // the ray slopes of the tile's columns and rows (x0 - 2 .. x0 + 33, y0 - 2 .. y0 + 9, clamped to the image): 48 correctly
    // rounded divisions per workgroup instead of six per pixel
    __shared__ float rxs[AGS_S2_TW + 4], rys[AGS_S2_TH + 4];
    if (threadIdx.x < AGS_S2_TW + 4) rxs[threadIdx.x] = ((float)min(max(x0 - 2 + (int)threadIdx.x, 0), W - 1) - 0.5f * (float)W) / c.fxq;
    else if (threadIdx.x >= 64 && threadIdx.x < 64 + AGS_S2_TH + 4)
        rys[threadIdx.x - 64] = ((float)min(max(y0 - 2 + (int)threadIdx.x - 64, 0), H - 1) - 0.5f * (float)H) / c.fyq;
    __syncthreads();
    // ---------------- depth -> normal (replicate padding) and the consistency term's gradient onto the five depths
    for (int i = threadIdx.x; i < AGS_S2_RN; i += 256) {
        const int x = x0 + (i % AGS_S2_RW) - 1, y = y0 + (i / AGS_S2_RW) - 1;
        float gc = 0.f, gu = 0.f, gl = 0.f, gr = 0.f, gb = 0.f;
        float3 d2n = f3(0.f, 0.f, 0.f);
        if (x >= 0 && x < W && y >= 0 && y < H) {
            const int p = y * W + x;
            const float mn = opacity[p] > 1e-2f ? 1.f : 0.f;
            const float ms = (float)msum[p];
            if (mn > 0.f) {
                const int yu = max(y - 1, 0), yb = min(y + 1, H - 1), xl = max(x - 1, 0), xr = min(x + 1, W - 1);
                const int iu = yu * W + x, ib = yb * W + x, il = y * W + xl, ir = y * W + xr;
                const float rx = rxs[x - x0 + 2], ry = rys[y - y0 + 2];
                const float rxl = rxs[xl - x0 + 2], rxr = rxs[xr - x0 + 2], ryu = rys[yu - y0 + 2], ryb = rys[yb - y0 + 2];
                const float dp = depth[p], du = depth[iu], db = depth[ib], dl = depth[il], dr = depth[ir];
                const float Mu = opacity[iu] > 1e-2f ? 1.f : 0.f, Mb = opacity[ib] > 1e-2f ? 1.f : 0.f;
                const float Ml = opacity[il] > 1e-2f ? 1.f : 0.f, Mr = opacity[ir] > 1e-2f ? 1.f : 0.f;
                const float3 ray_c = f3(rx, ry, 1.f), ray_u = f3(rx, ryu, 1.f), ray_b = f3(rx, ryb, 1.f);
                const float3 ray_l = f3(rxl, ry, 1.f), ray_r = f3(rxr, ry, 1.f);
                const float3 pc = ags_point(ray_c, dp) * mn;
                const float3 pu = (ags_point(ray_u, du) - pc) * Mu, pl = (ags_point(ray_l, dl) - pc) * Ml;
                const float3 pb = (ags_point(ray_b, db) - pc) * Mb, pr = (ags_point(ray_r, dr) - pc) * Mr;
                const float3 m = cross3(pu, pl) + cross3(pr, pu) + cross3(pb, pr) + cross3(pl, pb);
                const float ml = fmaxf(sqrtf(dot3(m, m)), 1e-12f);
                const float3 mh = m * (1.f / ml);
                d2n = mh * mn;
                // consistency gradient wrt d2n(p) -> m -> the five depths
                if (sqrtf(dot3(m, m)) > 1e-12f && ms > 0.f) {
                    const float3 G = load_n(n_img, HW, p) * (-k_c * ms);
                    const float3 Gm = fnma3(mh, dot3(mh, G), G) * (1.f / ml);
                    const float3 gpu = cross3(pl, Gm) + cross3(Gm, pr);
                    const float3 gpl = cross3(Gm, pu) + cross3(pb, Gm);
                    const float3 gpr = cross3(pu, Gm) + cross3(Gm, pb);
                    const float3 gpb = cross3(pr, Gm) + cross3(Gm, pl);
                    const float3 gpc = (gpu * Mu + gpl * Ml + gpr * Mr + gpb * Mb) * -1.f;
                    gc = dot3(gpc, ray_c) * mn; gu = dot3(gpu, ray_u) * Mu; gl = dot3(gpl, ray_l) * Ml;
                    gr = dot3(gpr, ray_r) * Mr; gb = dot3(gpb, ray_b) * Mb;
                }
            }
        }
        st[0][i] = gc; st[1][i] = gu; st[2][i] = gl; st[3][i] = gr; st[4][i] = gb;
        st[5][i] = d2n.x; st[6][i] = d2n.y; st[7][i] = d2n.z;
    }
    __syncthreads();
    const int lx = threadIdx.x & (AGS_S2_TW - 1), ly = threadIdx.x / AGS_S2_TW;
    const int x = x0 + lx, y = y0 + ly;
    float s_cons = 0.f, s_tv = 0.f;
    if (x < W && y < H) {
        const int p = y * W + x, ri = (ly + 1) * AGS_S2_RW + lx + 1;
        const float3 n = load_n(n_img, HW, p);
        const float dp = depth[p];
        const float mdp = gt_depth[p] > 0.f ? 1.f : 0.f;
        // ---------------- normal TV: gather over the (up to) 4 neighbours
        float3 gn = f3(0.f, 0.f, 0.f);
        const float k_tv = c.w_tv / ((float)c.B * 4.f * (float)HW);
        const int qx[4] = {x + 1, x - 1, x, x};
        const int qy[4] = {y, y, y + 1, y - 1};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            if (qx[d] < 0 || qx[d] >= W || qy[d] < 0 || qy[d] >= H) continue;
            const int q = qy[d] * W + qx[d];
            const float3 df = n - load_n(n_img, HW, q);
            const float nd = dot3(df, df);
            const float dd = dp - depth[q];
            if (dd * dd <= 1e-4f) {
                const float ex = __expf(-nd * c.inv_2sig2);
                s_tv += ex * nd * mdp;                                   // this pixel's own term
                const float mdq = gt_depth[q] > 0.f ? 1.f : 0.f;
                const float fp = ex * (1.f - nd * c.inv_2sig2) * (mdp + mdq) * 2.f * k_tv; // both orientations of the pair
                gn = fma3(df, fp, gn);
            }
        }
        // ---------------- the consistency term
        const float mn = opacity[p] > 1e-2f ? 1.f : 0.f;
        const float3 d2n = f3(st[5][ri], st[6][ri], st[7][ri]);
        const float ms = (float)msum[p];
        s_cons = (1.f - dot3(n, d2n)) * ms;
        gn = fma3(d2n, -k_c * ms, gn);
        // chain rule through n = normalize(N) * mask
        const float3 N = f3(normal_raw[p], normal_raw[HW + p], normal_raw[2 * HW + p]);
        const float Nl = sqrtf(dot3(N, N));
        float3 dN = f3(0.f, 0.f, 0.f);
        if (mn > 0.f && Nl > 1e-12f) {
            const float iNl = 1.f / Nl;
            const float3 nh = N * iNl;
            dN = fnma3(nh, dot3(nh, gn), gn) * iNl;
        }
        d_normal[p] = dN.x; d_normal[HW + p] = dN.y; d_normal[2 * HW + p] = dN.z;
        // what the stencils of this pixel and of its neighbours address to this pixel's depth (replicate padding: at a
        // border the pixel is its own neighbour)
        float add = st[0][ri];
        add += (y + 1 < H ? st[1][ri + AGS_S2_RW] : 0.f) + (y == 0 ? st[1][ri] : 0.f);       // "up" of the pixel below
        add += (x + 1 < W ? st[2][ri + 1] : 0.f) + (x == 0 ? st[2][ri] : 0.f);               // "left" of the pixel to the right
        add += (x > 0 ? st[3][ri - 1] : 0.f) + (x == W - 1 ? st[3][ri] : 0.f);               // "right" of the pixel to the left
        add += (y > 0 ? st[4][ri - AGS_S2_RW] : 0.f) + (y == H - 1 ? st[4][ri] : 0.f);       // "below" of the pixel above
        d_depth[p] += add;
    }
    const float t_c = block_sum_256(s_cons, sh), t_t = block_sum_256(s_tv, sh);
    if (threadIdx.x == 0) {
        float* row = accum + (size_t)((blockIdx.y * gridDim.x + blockIdx.x) & (AGS_LOSS_ACCUM_ROWS - 1)) * accum_stride;
        atomicAdd(&row[2], t_c); atomicAdd(&row[3], t_t);
    }
}

// ---------------------------------------------------------------------------------------------
// Two helpers that keep a batched training iteration down to a handful of launches (the host of the
// GPU box needs 20-40 us per torch op; an iteration used to issue ~35 of them):
//   ags_k_stage_frames: gathers the sampled frames' poses and ground truth into the batch buffers
//                       (replaces four index_select launches) and zeroes the visibility count
//   ags_k_loss_finish:  sums the 64 accumulator rows, writes the per-frame errors
//                       (track_performance, gaussian_map.py:132-139) and the total loss, and leaves
//                       the accumulators zeroed for the next iteration
__global__ __launch_bounds__(256) void ags_k_stage_frames(
    int HW, const long long* __restrict__ frame_index, const float* __restrict__ all_view,
    const float* __restrict__ all_proj, const float* __restrict__ all_rgb, const float* __restrict__ all_depth,
    float* __restrict__ dst_view, float* __restrict__ dst_proj, float* __restrict__ dst_rgb,
    float* __restrict__ dst_depth, int* __restrict__ msum) {
    const int v = blockIdx.y;
    const size_t f = (size_t)frame_index[v];
    const int q = blockIdx.x * 256 + threadIdx.x;          // float4 index within a plane
    const int HW4 = HW >> 2;                                 // HW is a multiple of 4 (checked by the caller)
    if (q < HW4) {
        if (dst_rgb) {   // (NULL: the loss stages read the ground truth in place, AgsLossConfig.gt_frame_index)
            const float4* r = reinterpret_cast<const float4*>(all_rgb + f * 3 * (size_t)HW);
            float4* o = reinterpret_cast<float4*>(dst_rgb + (size_t)v * 3 * (size_t)HW);
            o[q] = r[q]; o[HW4 + q] = r[HW4 + q]; o[2 * HW4 + q] = r[2 * HW4 + q];
            reinterpret_cast<float4*>(dst_depth + (size_t)v * HW)[q] = reinterpret_cast<const float4*>(all_depth + f * (size_t)HW)[q];
        }
        if (v == 0 && msum) reinterpret_cast<int4*>(msum)[q] = make_int4(0, 0, 0, 0);
    }
    if (blockIdx.x == 0 && threadIdx.x < 32) {
        const int k = threadIdx.x & 15;
        if (threadIdx.x < 16) dst_view[v * 16 + k] = all_view[f * 16 + k];
        else dst_proj[v * 16 + k] = all_proj[f * 16 + k];
    }
}

__global__ __launch_bounds__(256) void ags_k_loss_finish(AgsLossDev c, float* __restrict__ accum, int accum_stride,
                                                         int views, const long long* __restrict__ frame_index,
                                                         float* __restrict__ frame_error, float* __restrict__ total_loss) {
    __shared__ float sums[256];
    const int t = threadIdx.x;
    float s = 0.f;
    if (t < accum_stride) {
        // all 64 rows are requested before the first one is added (one memory latency, not a chain of them: this workgroup
        // is alone on the GPU while it runs), summed in row order, then cleared
        float v[AGS_LOSS_ACCUM_ROWS];
#pragma unroll
        for (int r = 0; r < AGS_LOSS_ACCUM_ROWS; ++r) v[r] = accum[(size_t)r * accum_stride + t];
#pragma unroll
        for (int r = 0; r < AGS_LOSS_ACCUM_ROWS; ++r) s += v[r];
#pragma unroll
        for (int r = 0; r < AGS_LOSS_ACCUM_ROWS; ++r) accum[(size_t)r * accum_stride + t] = 0.f;
    }
    sums[t] = s;
    __syncthreads();
    const float hw = (float)c.H * (float)c.W, b = (float)c.B;
    if (t < views && frame_error) frame_error[frame_index ? frame_index[t] : t] = sums[4 + 2 * t] / (3.f * hw) + sums[5 + 2 * t] / hw;
    if (t == 0 && total_loss)
        *total_loss = c.w_rgb * sums[0] / (b * 3.f * hw) + c.w_depth * sums[1] / (b * hw) + c.w_cons * sums[2] / (b * b * hw) +
                      c.w_tv * sums[3] / (b * 4.f * hw);
}

// ---------------------------------------------------------------------------------------------
// The facade's post-processing on its own (render_cuda_core, /root/reference/utils/operations.py:714-718 with
// depth2normal :172-219), for callers that keep the reference's loss head in torch: n = normalize(N) * (opacity > 1e-2)
// and d2n = depth2normal(depth, that mask, fov) in one launch, and their backward in one launch (what ~35 torch ops
// and their autograd nodes do per view).  Same arithmetic as the fused loss stages above.
struct AgsPostDev { int H, W; float fxq, fyq; };

struct AgsD2n {          // everything the backward needs again
    float3 ray_c, ray_u, ray_b, ray_l, ray_r, pu, pl, pb, pr, m;
    float Mu, Mb, Ml, Mr, mn, ml;
    int iu, ib, il, ir;
};
__device__ __forceinline__ AgsD2n ags_d2n_at(const AgsPostDev& c, const float* __restrict__ depth,
                                             const float* __restrict__ opacity, int x, int y, int p) {
    AgsD2n r;
    const int H = c.H, W = c.W;
    r.mn = opacity[p] > 1e-2f ? 1.f : 0.f;
    const int yu = max(y - 1, 0), yb = min(y + 1, H - 1), xl = max(x - 1, 0), xr = min(x + 1, W - 1);
    r.iu = yu * W + x; r.ib = yb * W + x; r.il = y * W + xl; r.ir = y * W + xr;
    const float rx = ((float)x - 0.5f * (float)W) / c.fxq, ry = ((float)y - 0.5f * (float)H) / c.fyq;
    const float rxl = ((float)xl - 0.5f * (float)W) / c.fxq, rxr = ((float)xr - 0.5f * (float)W) / c.fxq;
    const float ryu = ((float)yu - 0.5f * (float)H) / c.fyq, ryb = ((float)yb - 0.5f * (float)H) / c.fyq;
    r.Mu = opacity[r.iu] > 1e-2f ? 1.f : 0.f; r.Mb = opacity[r.ib] > 1e-2f ? 1.f : 0.f;
    r.Ml = opacity[r.il] > 1e-2f ? 1.f : 0.f; r.Mr = opacity[r.ir] > 1e-2f ? 1.f : 0.f;
    r.ray_c = f3(rx, ry, 1.f); r.ray_u = f3(rx, ryu, 1.f); r.ray_b = f3(rx, ryb, 1.f);
    r.ray_l = f3(rxl, ry, 1.f); r.ray_r = f3(rxr, ry, 1.f);
    const float3 pc = ags_point(r.ray_c, depth[p]) * r.mn;
    r.pu = (ags_point(r.ray_u, depth[r.iu]) - pc) * r.Mu; r.pl = (ags_point(r.ray_l, depth[r.il]) - pc) * r.Ml;
    r.pb = (ags_point(r.ray_b, depth[r.ib]) - pc) * r.Mb; r.pr = (ags_point(r.ray_r, depth[r.ir]) - pc) * r.Mr;
    r.m = cross3(r.pu, r.pl) + cross3(r.pr, r.pu) + cross3(r.pb, r.pr) + cross3(r.pl, r.pb);
    r.ml = fmaxf(sqrtf(dot3(r.m, r.m)), 1e-12f);
    return r;
}

__global__ __launch_bounds__(256) void ags_k_facade_post(AgsPostDev c, const float* __restrict__ normal_raw,
                                                         const float* __restrict__ depth, const float* __restrict__ opacity,
                                                         float* __restrict__ normal_out, float* __restrict__ d2n_out) {
    const int HW = c.H * c.W, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    { // a batch of views (ags_facade_post_batch): blockIdx.y = view, images contiguous (views,C,H,W)
        const size_t v = blockIdx.y;
        depth += v * HW; opacity += v * HW; d2n_out += 3 * v * HW;
        if (normal_raw && normal_out) { normal_raw += 3 * v * HW; normal_out += 3 * v * HW; }
    }
    const int y = p / c.W, x = p - y * c.W;
    const AgsD2n r = ags_d2n_at(c, depth, opacity, x, y, p);
    if (normal_raw && normal_out) {
        const float nx = normal_raw[p], ny = normal_raw[HW + p], nz = normal_raw[2 * HW + p];
        const float inv = r.mn / fmaxf(sqrtf(nx * nx + ny * ny + nz * nz), 1e-12f);
        normal_out[p] = nx * inv; normal_out[HW + p] = ny * inv; normal_out[2 * HW + p] = nz * inv;
    }
    const float3 d2n = r.m * (r.mn / r.ml);
    d2n_out[p] = d2n.x; d2n_out[HW + p] = d2n.y; d2n_out[2 * HW + p] = d2n.z;
}

__global__ __launch_bounds__(256) void ags_k_facade_post_bwd(AgsPostDev c, const float* __restrict__ normal_raw,
                                                             const float* __restrict__ depth, const float* __restrict__ opacity,
                                                             const float* __restrict__ g_normal, const float* __restrict__ g_d2n,
                                                             float* __restrict__ d_normal_raw, float* __restrict__ d_depth) {
    const int HW = c.H * c.W, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int y = p / c.W, x = p - y * c.W;
    const float mn = opacity[p] > 1e-2f ? 1.f : 0.f;
    if (d_normal_raw) {   // n = normalize(N) * mask
        float3 dN = f3(0.f, 0.f, 0.f);
        if (g_normal && normal_raw) {
            const float3 N = f3(normal_raw[p], normal_raw[HW + p], normal_raw[2 * HW + p]);
            const float3 gn = f3(g_normal[p], g_normal[HW + p], g_normal[2 * HW + p]);
            const float Nl = sqrtf(dot3(N, N));
            if (mn > 0.f && Nl > 1e-12f) {
                const float3 nh = N * (1.f / Nl);
                dN = (gn - nh * dot3(nh, gn)) * (1.f / Nl);
            }
        }
        d_normal_raw[p] = dN.x; d_normal_raw[HW + p] = dN.y; d_normal_raw[2 * HW + p] = dN.z;
    }
    if (!g_d2n || !d_depth || mn == 0.f) return;
    const float3 G = f3(g_d2n[p], g_d2n[HW + p], g_d2n[2 * HW + p]);
    if (G.x == 0.f && G.y == 0.f && G.z == 0.f) return;
    const AgsD2n r = ags_d2n_at(c, depth, opacity, x, y, p);
    if (sqrtf(dot3(r.m, r.m)) <= 1e-12f) return;
    const float3 mh = r.m * (1.f / r.ml);
    const float3 Gm = (G - mh * dot3(mh, G)) * (1.f / r.ml);
    const float3 gpu = cross3(r.pl, Gm) + cross3(Gm, r.pr);
    const float3 gpl = cross3(Gm, r.pu) + cross3(r.pb, Gm);
    const float3 gpr = cross3(r.pu, Gm) + cross3(Gm, r.pb);
    const float3 gpb = cross3(r.pr, Gm) + cross3(Gm, r.pl);
    const float3 gpc = (gpu * r.Mu + gpl * r.Ml + gpr * r.Mr + gpb * r.Mb) * -1.f;
    atomicAdd(&d_depth[p], dot3(gpc, r.ray_c) * r.mn);
    atomicAdd(&d_depth[r.iu], dot3(gpu, r.ray_u) * r.Mu);
    atomicAdd(&d_depth[r.il], dot3(gpl, r.ray_l) * r.Ml);
    atomicAdd(&d_depth[r.ir], dot3(gpr, r.ray_r) * r.Mr);
    atomicAdd(&d_depth[r.ib], dot3(gpb, r.ray_b) * r.Mb);
}

void ags_launch_facade_post(int views, int h, int w, float tanx, float tany, const float* normal_raw, const float* depth,
                            const float* opacity, float* normal_out, float* d2n_out, hipStream_t s) {
    const AgsPostDev c = {h, w, (float)h / (2.0f * tanx), (float)w / (2.0f * tany)};   // the reference pairs fov_x with H (sic)
    hipLaunchKernelGGL(ags_k_facade_post, dim3((h * w + 255) / 256, views), dim3(256), 0, s, c, normal_raw, depth, opacity,
                       normal_out, d2n_out);
}
void ags_launch_facade_post_bwd(int h, int w, float tanx, float tany, const float* normal_raw, const float* depth,
                                const float* opacity, const float* g_normal, const float* g_d2n, float* d_normal_raw,
                                float* d_depth, hipStream_t s) {
    const AgsPostDev c = {h, w, (float)h / (2.0f * tanx), (float)w / (2.0f * tany)};
    hipLaunchKernelGGL(ags_k_facade_post_bwd, dim3((h * w + 255) / 256), dim3(256), 0, s, c, normal_raw, depth, opacity,
                       g_normal, g_d2n, d_normal_raw, d_depth);
}

// ---------------------------------------------------------------------------------------------
// The weighted frame draw of the batch sampler (mapping/utils.py:190-228: np.random.choice(older, k, replace=False,
// p = error / sum)) as ONE launch behind torch.rand: successive sampling without replacement == the k largest
// log(u_i) / w_i (Efraimidis & Spirakis); keys in LDS, k rounds of a workgroup arg-max, indices written in descending key
// order (what torch.topk returned when this was log + clamp + div + topk + a slice assignment: five launches and
// 0.26 ms of host time per iteration).
#define AGS_WTOPK_MAX 8192
// (the body of ags_k_weighted_topk for one workgroup of 256 threads; key / best_v / best_i: its LDS)
// (w and out carry no __restrict__: ags_k_loss_finish_next hands in the per-frame errors it has just written through another
// pointer, and reads the indices written here through its own)
__device__ __forceinline__ void ags_wtopk_block(const float* __restrict__ u, const float* w, int n, int k,
                                                long long* out, float* key, float* best_v, int* best_i) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // every key stays FINITE (u = 0 from torch.rand would give log(0) = -inf, a NaN weight a NaN key: both become the lowest
    // finite key), so that a drawn slot can be marked with NaN and never be selected again: the k indices are distinct
    // whatever the inputs, like np.random.choice(replace=False) and torch.topk
    for (int i = t; i < n; i += 256) {
        const float kv = logf(fmaxf(u[i], 1.17549435e-38f)) / fmaxf(w[i], 1e-30f);
        key[i] = (kv == kv && kv > -3.4e38f) ? kv : -3.4e38f;
    }
    __syncthreads();
    for (int j = 0; j < k; ++j) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = t; i < n; i += 256) {
            const float v = key[i];                      // NaN = already drawn: no comparison with it is true
            if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(bi, off);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { best_v[wave] = bv; best_i[wave] = bi; }
        __syncthreads();
        if (t == 0) {
            float v = best_v[0];
            int i = best_i[0];
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (best_v[q] > v || (best_v[q] == v && best_i[q] < i)) { v = best_v[q]; i = best_i[q]; }
            if (i == 0x7fffffff) i = 0;          // (cannot happen for k <= n: every undrawn key is finite)
            out[j] = (long long)i;
            if (i < n) key[i] = __builtin_nanf("");
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void ags_k_weighted_topk(const float* __restrict__ u, const float* __restrict__ w, int n,
                                                           int k, long long* __restrict__ out) {
    __shared__ float key[AGS_WTOPK_MAX];
    __shared__ float best_v[4];
    __shared__ int best_i[4];
    ags_wtopk_block(u, w, n, k, out, key, best_v, best_i);
}

// The end of one batched training iteration and the set-up of the next in ONE launch (ags_loss_finish_next): workgroup 0
// runs ags_k_loss_finish's sums, then the next iteration's weighted draw over the errors it has just written
// (ags_k_weighted_topk), then gathers the drawn frames' matrices (ags_k_stage_frames without images); the other
// workgroups clear the visibility count.  Three launches and a torch.rand fewer per iteration: each of them was a
// single-workgroup kernel that the GPU finished before the next one arrived (~25 us of an 0.58-ms iteration).
struct AgsNextDev {
    const float* u; int n_w, k, first_random, views;
    const float* all_view; const float* all_proj; float* dst_view; float* dst_proj; int* msum;
};
__global__ __launch_bounds__(256) void ags_k_loss_finish_next(AgsLossDev c, float* __restrict__ accum, int accum_stride,
                                                              int views, long long* frame_index,
                                                              float* frame_error, float* __restrict__ total_loss,
                                                              AgsNextDev nx) {
    __shared__ float key[AGS_WTOPK_MAX];
    __shared__ float best_v[4];
    __shared__ int best_i[4];
    const int t = threadIdx.x;
    if (blockIdx.x > 0) {            // the visibility count of the next iteration: (H*W / 4) int4 over gridDim.x - 1 workgroups
        const int HW4 = (c.H * c.W) >> 2;
        for (int q = (blockIdx.x - 1) * 256 + t; q < HW4; q += (gridDim.x - 1) * 256) reinterpret_cast<int4*>(nx.msum)[q] = make_int4(0, 0, 0, 0);
        return;
    }
    float s = 0.f;
    if (t < accum_stride) {
        // all 64 rows are requested before the first one is added (one memory latency, not a chain of them: this workgroup
        // is alone on the GPU while it runs), summed in row order, then cleared
        float v[AGS_LOSS_ACCUM_ROWS];
#pragma unroll
        for (int r = 0; r < AGS_LOSS_ACCUM_ROWS; ++r) v[r] = accum[(size_t)r * accum_stride + t];
#pragma unroll
        for (int r = 0; r < AGS_LOSS_ACCUM_ROWS; ++r) s += v[r];
#pragma unroll
        for (int r = 0; r < AGS_LOSS_ACCUM_ROWS; ++r) accum[(size_t)r * accum_stride + t] = 0.f;
    }
    key[t] = s;                      // (the draw's key array doubles as the sums' scratch)
    __syncthreads();
    const float hw = (float)c.H * (float)c.W, b = (float)c.B;
    if (t < views && frame_error) frame_error[frame_index[t]] = key[4 + 2 * t] / (3.f * hw) + key[5 + 2 * t] / hw;
    if (t == 0 && total_loss)
        *total_loss = c.w_rgb * key[0] / (b * 3.f * hw) + c.w_depth * key[1] / (b * hw) + c.w_cons * key[2] / (b * b * hw) +
                      c.w_tv * key[3] / (b * 4.f * hw);
    __threadfence_block();
    __syncthreads();                 // the errors are written: the draw reads them
    if (nx.u && nx.k > 0) ags_wtopk_block(nx.u, frame_error, nx.n_w, nx.k, frame_index + nx.first_random, key, best_v, best_i);
    __threadfence_block();
    __syncthreads();
    for (int i = t; i < nx.views * 32; i += 256) {
        const int v = i >> 5, e = i & 31;
        const size_t f = (size_t)frame_index[v];
        if (e < 16) nx.dst_view[v * 16 + e] = nx.all_view[f * 16 + e];
        else nx.dst_proj[v * 16 + e - 16] = nx.all_proj[f * 16 + e - 16];
    }
}

void ags_launch_weighted_topk(const float* u, const float* w, int n, int k, long long* out, hipStream_t s) {
    hipLaunchKernelGGL(ags_k_weighted_topk, dim3(1), dim3(256), 0, s, u, w, n, k, out);
}

static AgsLossDev make_dev(const AgsLossConfig& cfg) {
    AgsLossDev c;
    c.H = cfg.image_height; c.W = cfg.image_width; c.B = cfg.batch_total;
    c.fxq = (float)cfg.image_height / (2.0f * tanf(0.5f * cfg.fov_x));
    c.fyq = (float)cfg.image_width / (2.0f * tanf(0.5f * cfg.fov_y));
    c.w_rgb = cfg.w_rgb; c.w_depth = cfg.w_depth; c.w_cons = cfg.w_cons; c.w_tv = cfg.w_tv;
    c.inv_2sig2 = 1.0f / (2.0f * cfg.sigma * cfg.sigma);
    c.gt_index = cfg.num_views > 1 ? (const long long*)cfg.gt_frame_index : nullptr;
    return c;
}

void ags_launch_loss_stage1(const AgsLossConfig& cfg, const AgsImages& img, const float* gt_rgb, const float* gt_depth,
                            float* n_img, float* d_rgb, float* d_depth, int* msum, float* accum, int view,
                            int first_view, hipStream_t s) {
    const int HW = cfg.image_height * cfg.image_width;
    const int views = cfg.num_views > 1 ? cfg.num_views : 1;
    hipLaunchKernelGGL(ags_k_loss_stage1, dim3((HW + 255) / 256, views), dim3(256), 0, s, make_dev(cfg), img.rgb, img.normal,
                       img.depth, img.opacity, gt_rgb, gt_depth, n_img, d_rgb, d_depth, msum, accum, cfg.accum_stride,
                       view, first_view);
}

void ags_launch_loss_stage2(const AgsLossConfig& cfg, const AgsImages& img, const float* n_img, const float* gt_depth,
                            const int* msum, float* d_normal, float* d_depth, float* accum, hipStream_t s) {
    const int HW = cfg.image_height * cfg.image_width;
    const int views = cfg.num_views > 1 ? cfg.num_views : 1;
    (void)HW;
    hipLaunchKernelGGL(ags_k_loss_stage2, dim3((cfg.image_width + AGS_S2_TW - 1) / AGS_S2_TW, (cfg.image_height + AGS_S2_TH - 1) / AGS_S2_TH, views),
                       dim3(256), 0, s, make_dev(cfg), img.depth, img.opacity, img.normal, n_img, gt_depth, msum, d_normal,
                       d_depth, accum, cfg.accum_stride);
}

void ags_launch_stage_frames(int views, int hw, const long long* frame_index, const float* all_view, const float* all_proj,
                             const float* all_rgb, const float* all_depth, float* dst_view, float* dst_proj,
                             float* dst_rgb, float* dst_depth, int* msum, hipStream_t s) {
    // without image copies only view 0's workgroups have pixels to touch (the visibility count) - one workgroup per other view
    hipLaunchKernelGGL(ags_k_stage_frames, dim3(((hw >> 2) + 255) / 256, views), dim3(256), 0, s, hw, frame_index, all_view,
                       all_proj, all_rgb, all_depth, dst_view, dst_proj, dst_rgb, dst_depth, msum);
}

void ags_launch_loss_finish_next(const AgsLossConfig& cfg, float* accum, int views, long long* frame_index, float* frame_error,
                                 float* total_loss, const AgsNextIteration& nx, hipStream_t s) {
    const AgsNextDev d = {nx.uniforms, nx.n_weights, nx.k, nx.first_random, nx.views, nx.all_view, nx.all_proj, nx.dst_view,
                          nx.dst_proj, nx.msum};
    const int hw4 = (cfg.image_height * cfg.image_width) >> 2;
    const int clear_blocks = (hw4 + 1023) / 1024 < 1 ? 1 : (hw4 + 1023) / 1024;     // four int4 per thread
    hipLaunchKernelGGL(ags_k_loss_finish_next, dim3(1 + clear_blocks), dim3(256), 0, s, make_dev(cfg), accum, cfg.accum_stride,
                       views, frame_index, frame_error, total_loss, d);
}

void ags_launch_loss_finish(const AgsLossConfig& cfg, float* accum, int views, const long long* frame_index,
                            float* frame_error, float* total_loss, hipStream_t s) {
    hipLaunchKernelGGL(ags_k_loss_finish, dim3(1), dim3(256), 0, s, make_dev(cfg), accum, cfg.accum_stride, views,
                       frame_index, frame_error, total_loss);
}
