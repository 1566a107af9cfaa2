// Fused Adam over the five Gaussian parameter tensors (means, scales, rotations,
// opacities, harmonics) in ONE launch: the 14*N floats are one flat index space, each
// element reads g,m,v,p once and writes m,v,p once (28 B/float = 392 B per Gaussian).
// Semantics: torch.optim.Adam as configured at /root/reference/mapping/gaussian_map.py:259-292
// (betas 0.9/0.999 by default, eps 1e-15, no weight decay, no amsgrad; per-tensor lr from
// /root/reference/config/mapper/incremental.yaml:27-32).  Zero-gradient rows still decay
// their moments and move, exactly like the dense torch update.
#include "ags_internal.h"

// Device-resident optimiser clock AgsAdamClock { int step; float step_size[5]; float inv_sqrt_bc2; }
// (ags_internal.h); advanced either by this 1-thread kernel or on the side of the step's last
// ags_backward launch (AgsGaussianGrads.adam_clock).
__global__ void ags_k_adam_tick(AgsAdamClock* c, float lr0, float lr1, float lr2, float lr3, float lr4,
                                float beta1, float beta2) {
    const float lr[5] = {lr0, lr1, lr2, lr3, lr4};
    ags_adam_tick(c, lr, beta1, beta2);
}

__global__ __launch_bounds__(256) void ags_k_adam(AgsAdamArgs a, AgsAdamClock host_clk,
                                                  const AgsAdamClock* __restrict__ dev_clk, float beta1,
                                                  float beta2, float eps, long long total) {
    // the step's scalars, from the device clock (global loads) or the by-value host copy - not through
    // one pointer to either, which would be a flat access per use
    float ss[5], inv_bc2_sqrt;
    if (dev_clk) {
#pragma unroll
        for (int k = 0; k < 5; ++k) ss[k] = dev_clk->step_size[k];
        inv_bc2_sqrt = dev_clk->inv_bc2_sqrt;
    } else {
#pragma unroll
        for (int k = 0; k < 5; ++k) ss[k] = host_clk.step_size[k];
        inv_bc2_sqrt = host_clk.inv_bc2_sqrt;
    }
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int seg = (i >= a.end[0]) + (i >= a.end[1]) + (i >= a.end[2]) + (i >= a.end[3]);
        const long long j = i - (seg ? a.end[seg - 1] : 0);
        // the four arrays of the element's tensor, selected with conditional moves: indexing the
        // kernel-argument pointer arrays with a runtime `seg` would make these flat_* accesses
        const float* pg = seg == 0 ? a.g[0] : seg == 1 ? a.g[1] : seg == 2 ? a.g[2] : seg == 3 ? a.g[3] : a.g[4];
        float* pm = seg == 0 ? a.m[0] : seg == 1 ? a.m[1] : seg == 2 ? a.m[2] : seg == 3 ? a.m[3] : a.m[4];
        float* pv = seg == 0 ? a.v[0] : seg == 1 ? a.v[1] : seg == 2 ? a.v[2] : seg == 3 ? a.v[3] : a.v[4];
        float* pp = seg == 0 ? a.p[0] : seg == 1 ? a.p[1] : seg == 2 ? a.p[2] : seg == 3 ? a.p[3] : a.p[4];
        const float g = pg[j];
        float m = pm[j], v = pv[j];
        m = m + (1.f - beta1) * (g - m);                 // exp_avg.lerp_(grad, 1-beta1)
        v = v * beta2 + (1.f - beta2) * g * g;           // mul_(beta2).addcmul_(g, g, 1-beta2)
        const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
        pm[j] = m;
        pv[j] = v;
        const float step_size = seg == 0 ? ss[0] : seg == 1 ? ss[1] : seg == 2 ? ss[2] : seg == 3 ? ss[3] : ss[4];
        pp[j] -= step_size * (m / denom); // addcdiv_(m, denom, -step_size)
    }
}

// Row-set form (AgsAdamTensors.touched): 16 lanes per member row, lane k < 14 owns one of the
// row's 14 floats (means 0-2, scales 3-5, rotation 6-9, opacity 10, harmonics 11-13).  Rows
// outside the set have g = m = v = 0, for which the dense update above is exactly 0.
__global__ __launch_bounds__(256) void ags_k_adam_rows(AgsAdamArgs a, const AgsAdamClock* __restrict__ clk,
                                                       AgsAdamClock host_clk, AgsRowSet touched, float beta1,
                                                       float beta2, float eps, int zero_grad, int n_rows) {
    const int k = threadIdx.x & 15;
    const int seg = (k >= 3) + (k >= 6) + (k >= 10) + (k >= 11);
    const int width = seg == 2 ? 4 : (seg == 3 ? 1 : 3);
    const int off = k - (seg == 0 ? 0 : seg == 1 ? 3 : seg == 2 ? 6 : seg == 3 ? 10 : 11);
    float* pg = const_cast<float*>(seg == 0 ? a.g[0] : seg == 1 ? a.g[1] : seg == 2 ? a.g[2] : seg == 3 ? a.g[3] : a.g[4]);
    float* pm = seg == 0 ? a.m[0] : seg == 1 ? a.m[1] : seg == 2 ? a.m[2] : seg == 3 ? a.m[3] : a.m[4];
    float* pv = seg == 0 ? a.v[0] : seg == 1 ? a.v[1] : seg == 2 ? a.v[2] : seg == 3 ? a.v[3] : a.v[4];
    float* pp = seg == 0 ? a.p[0] : seg == 1 ? a.p[1] : seg == 2 ? a.p[2] : seg == 3 ? a.p[3] : a.p[4];
    float ss[5], inv_bc2_sqrt;
    if (clk) {
#pragma unroll
        for (int q = 0; q < 5; ++q) ss[q] = clk->step_size[q];
        inv_bc2_sqrt = clk->inv_bc2_sqrt;
    } else {
#pragma unroll
        for (int q = 0; q < 5; ++q) ss[q] = host_clk.step_size[q];
        inv_bc2_sqrt = host_clk.inv_bc2_sqrt;
    }
    const float step_size = seg == 0 ? ss[0] : seg == 1 ? ss[1] : seg == 2 ? ss[2] : seg == 3 ? ss[3] : ss[4];
    // touched.rows == nullptr: every row (the dense step over interleaved moments: same lane layout, row = r)
    const int count = touched.rows ? *touched.count : n_rows;
    const int stride = gridDim.x * 16;
    for (int r = blockIdx.x * 16 + (threadIdx.x >> 4); r < count; r += stride) {
        if (k >= 14) continue;
        const int row = touched.rows ? touched.rows[r] : r;
        const long long j = (long long)row * width + off;
        // interleaved moments (AgsAdamTensors.state_rows): the row's 28 floats are one 112-byte piece
        float* qm = a.st ? a.st + (long long)row * 28 + k : pm + j;
        float* qv = a.st ? a.st + (long long)row * 28 + 14 + k : pv + j;
        const float g = pg[j];
        float m = *qm, v = *qv;
        m = m + (1.f - beta1) * (g - m);
        v = v * beta2 + (1.f - beta2) * g * g;
        const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
        *qm = m;
        *qv = v;
        pp[j] -= step_size * (m / denom);
        if (zero_grad) pg[j] = 0.f; // consumed: the slab is clean for the next step
    }
}

void ags_launch_adam(const AgsAdamTensors& t, float beta1, float beta2, float eps, int step, void* dev_state,
                     bool pre_ticked, hipStream_t s) {
    const AgsAdamArgs a = ags_adam_args(t);
    const long long run = a.end[4];
    if (run <= 0) return;
    AgsAdamClock* clk = (AgsAdamClock*)dev_state;
    AgsAdamClock hc = {};
    if (clk && pre_ticked) {
        // the clock was advanced by the step's last ags_backward launch
    } else if (clk) {
        hipLaunchKernelGGL(ags_k_adam_tick, dim3(1), dim3(1), 0, s, clk, t.lr[0], t.lr[1], t.lr[2], t.lr[3], t.lr[4],
                           beta1, beta2);
    } else { // host-side clock: scalars travel as kernel arguments
        const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
        hc.step = step;
        for (int k = 0; k < 5; ++k) hc.step_size[k] = (float)((double)t.lr[k] / bc1);
        hc.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    }
    if (t.touched.rows || t.state_rows) { // row-set step, or the dense step over interleaved moments (all rows)
        long long rb = (t.numel[3] + 15) / 16; // 16 rows per block
        if (rb > 16384) rb = 16384;
        hipLaunchKernelGGL(ags_k_adam_rows, dim3((unsigned)rb), dim3(256), 0, s, a, (const AgsAdamClock*)clk, hc,
                           t.touched, beta1, beta2, eps, t.touched.rows ? t.zero_grad : 0, (int)t.numel[3]);
        return;
    }
    long long blocks = (run + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(ags_k_adam, dim3((unsigned)blocks), dim3(256), 0, s, a, hc, (const AgsAdamClock*)clk, beta1,
                       beta2, eps, run);
}

// ---------------------------------------------------------------------------------------
// Row exchange of the view-parallel step (the reference sums the views' losses before
// backward(), gaussian_map.py:113-125): instead of all-reducing the dense 14*N-float slab, a rank
// ships only the rows its views have shown.  A SEGMENT is 16 floats of header {count, needed, ...}
// followed by `capacity` 64-byte records {g[0..13], row id, 0}; 16 lanes move one record.
__device__ __forceinline__ float* ags_row_field(float* const g[5], int k, int row) {
    const int seg = (k >= 3) + (k >= 6) + (k >= 10) + (k >= 11);
    const int width = seg == 2 ? 4 : (seg == 3 ? 1 : 3);
    const int off = k - (seg == 0 ? 0 : seg == 1 ? 3 : seg == 2 ? 6 : seg == 3 ? 10 : 11);
    float* base = seg == 0 ? g[0] : seg == 1 ? g[1] : seg == 2 ? g[2] : seg == 3 ? g[3] : g[4];
    return base + (long long)row * width + off;
}

struct AgsGradPtrs { float* g[5]; };

// slab rows of `rows` -> segment; the rows are zeroed behind the copy, so the slab is all-zero when
// the unpack launches start adding (every rank then forms the same sums in the same order)
__global__ __launch_bounds__(256) void ags_k_rows_pack(AgsGradPtrs gp, AgsRowSet rows, float* __restrict__ seg,
                                                       int capacity) {
    const int k = threadIdx.x & 15;
    const int needed = *rows.count;
    const int count = min(needed, capacity);
    if (blockIdx.x == 0 && threadIdx.x < 16)
        seg[threadIdx.x] = __int_as_float(threadIdx.x == 0 ? count : threadIdx.x == 1 ? needed : 0);
    const int stride = gridDim.x * 16;
    for (int r = blockIdx.x * 16 + (threadIdx.x >> 4); r < count; r += stride) {
        const int row = rows.rows[r];
        float v = 0.f;
        if (k < 14) {
            float* p = ags_row_field(gp.g, k, row);
            v = *p;
            *p = 0.f;
        } else if (k == 14) {
            v = __int_as_float(row);
        }
        seg[16 + (long long)r * 16 + k] = v;
    }
}

// one rank's segment += into the slab; rows new to `uni` (the union the optimiser steps over) are
// appended.  Launched once per rank IN RANK ORDER on one stream: a row occurs at most once per
// segment, so there is no race inside a launch and the summation order is the same on every rank.
__global__ __launch_bounds__(256) void ags_k_rows_unpack(const float* __restrict__ seg, int capacity, AgsGradPtrs gp,
                                                         AgsRowSet uni) {
    const int k = threadIdx.x & 15;
    const int count = min(__float_as_int(seg[0]), capacity);
    const int stride = gridDim.x * 16;
    for (int r = blockIdx.x * 16 + (threadIdx.x >> 4); r < count; r += stride) {
        const float* rec = seg + 16 + (long long)r * 16;
        const int row = __float_as_int(rec[14]);
        if (k < 14) {
            float* p = ags_row_field(gp.g, k, row);
            *p += rec[k];
        } else if (k == 14 && uni.member[row] == 0) {
            uni.member[row] = 1;
            uni.rows[atomicAdd(uni.count, 1)] = row;
        }
    }
}

void ags_launch_rows_pack(float* const grads[5], const AgsRowSet& rows, float* segment, int capacity, hipStream_t s) {
    AgsGradPtrs gp;
    for (int k = 0; k < 5; ++k) gp.g[k] = grads[k];
    int blocks = (capacity + 15) / 16;
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    hipLaunchKernelGGL(ags_k_rows_pack, dim3(blocks), dim3(256), 0, s, gp, rows, segment, capacity);
}
void ags_launch_rows_unpack(const float* segment, int capacity, float* const grads[5], const AgsRowSet& uni,
                            hipStream_t s) {
    AgsGradPtrs gp;
    for (int k = 0; k < 5; ++k) gp.g[k] = grads[k];
    int blocks = (capacity + 15) / 16;
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    hipLaunchKernelGGL(ags_k_rows_unpack, dim3(blocks), dim3(256), 0, s, segment, capacity, gp, uni);
}

// Two-launch tail of the row exchange for any number of ranks (instead of one unpack launch per rank
// plus the row-set Adam): ags_k_rows_index notes, for every record of every segment, where its row
// sits (slot_table[row * world + rank] = record + 1) and builds the union; ags_k_adam_rows_gathered
// then runs one 16-lane group per union row, sums the row's gradient from the segments IN RANK ORDER
// (so every rank forms bit-identical sums), applies the Adam update and leaves the table zeroed.
__global__ __launch_bounds__(256) void ags_k_rows_index(const float* __restrict__ segs, size_t seg_floats, int capacity,
                                                        int world, int* __restrict__ slot_table, AgsRowSet uni) {
    const int rank = blockIdx.y;
    const float* seg = segs + (size_t)rank * seg_floats;
    const int count = min(__float_as_int(seg[0]), capacity);
    for (int r = blockIdx.x * 256 + threadIdx.x; r < count; r += gridDim.x * 256) {
        const int row = __float_as_int(seg[16 + (size_t)r * 16 + 14]);
        slot_table[(size_t)row * world + rank] = r + 1;
        if (atomicCAS(&uni.member[row], 0, 1) == 0) uni.rows[atomicAdd(uni.count, 1)] = row; // several ranks may bring the row
    }
}

__global__ __launch_bounds__(256) void ags_k_adam_rows_gathered(AgsAdamArgs a, AgsAdamClock* __restrict__ clk,
                                                                AgsRowSet uni, const float* __restrict__ segs,
                                                                size_t seg_floats, int world, int capacity,
                                                                int* __restrict__ slot_table,
                                                                float beta1, float beta2, float eps) {
    // A rank whose row set has outgrown the agreed segment shipped only part of its gradient (header word 1 =
    // rows it holds > capacity).  Every rank sees every header, so all of them take the same decision with no
    // further exchange: the step is NOT applied (parameters, moments and the step counter stay as they were,
    // the slot table is still cleaned) and counted in clk->skipped; the host reads that counter every few steps,
    // agrees on a larger segment and repeats the refused steps (trainer.SurfelTrainer).
    bool over = false;
    for (int s = 0; s < world; ++s) over |= __float_as_int(segs[(size_t)s * seg_floats + 1]) > capacity;   // wave-uniform
    if (over && blockIdx.x == 0 && threadIdx.x == 0) { ags_adam_untick(clk); clk->skipped += 1; }   // nobody reads the clock in a refused step
    const int k = threadIdx.x & 15;
    const int seg = (k >= 3) + (k >= 6) + (k >= 10) + (k >= 11);
    const int width = seg == 2 ? 4 : (seg == 3 ? 1 : 3);
    const int off = k - (seg == 0 ? 0 : seg == 1 ? 3 : seg == 2 ? 6 : seg == 3 ? 10 : 11);
    float* pm = seg == 0 ? a.m[0] : seg == 1 ? a.m[1] : seg == 2 ? a.m[2] : seg == 3 ? a.m[3] : a.m[4];
    float* pv = seg == 0 ? a.v[0] : seg == 1 ? a.v[1] : seg == 2 ? a.v[2] : seg == 3 ? a.v[3] : a.v[4];
    float* pp = seg == 0 ? a.p[0] : seg == 1 ? a.p[1] : seg == 2 ? a.p[2] : seg == 3 ? a.p[3] : a.p[4];
    float ss[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) ss[q] = clk->step_size[q];
    const float inv_bc2_sqrt = clk->inv_bc2_sqrt;
    const float step_size = seg == 0 ? ss[0] : seg == 1 ? ss[1] : seg == 2 ? ss[2] : seg == 3 ? ss[3] : ss[4];
    const int count = *uni.count;
    const int stride = gridDim.x * 16;
    for (int r = blockIdx.x * 16 + (threadIdx.x >> 4); r < count; r += stride) {
        const int row = uni.rows[r];
        int* tab = slot_table + (size_t)row * world;
        float g = 0.f;
        for (int s = 0; s < world; ++s) { // rank order: the same sum on every rank
            const int t = tab[s];
            if (t) g += segs[(size_t)s * seg_floats + 16 + (size_t)(t - 1) * 16 + (k < 14 ? k : 0)];
        }
        if (k < 14 && !over) {
            const long long j = (long long)row * width + off;
            float* qm = a.st ? a.st + (long long)row * 28 + k : pm + j;
            float* qv = a.st ? a.st + (long long)row * 28 + 14 + k : pv + j;
            float m = *qm, v = *qv;
            m = m + (1.f - beta1) * (g - m);
            v = v * beta2 + (1.f - beta2) * g * g;
            const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
            *qm = m;
            *qv = v;
            pp[j] -= step_size * (m / denom);
        }
        // the update above needed g, i.e. every lane's table reads have returned (vmcnt is per wave); keep the
        // compiler from moving the clearing stores ahead of them
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int s = k; s < world; s += 16) tab[s] = 0;
    }
}

void ags_launch_rows_index(const float* segs, size_t seg_floats, int capacity, int world, int* slot_table,
                           const AgsRowSet& uni, hipStream_t s) {
    int blocks = (capacity + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    hipLaunchKernelGGL(ags_k_rows_index, dim3(blocks, world), dim3(256), 0, s, segs, seg_floats, capacity, world, slot_table, uni);
}
void ags_launch_adam_gathered(const AgsAdamTensors& t, const float* segs, size_t seg_floats, int world, int capacity,
                              int* slot_table, float beta1, float beta2, float eps, void* dev_state, bool pre_ticked,
                              hipStream_t s) {
    const AgsAdamArgs a = ags_adam_args(t);
    AgsAdamClock* clk = (AgsAdamClock*)dev_state;
    if (!pre_ticked)
        hipLaunchKernelGGL(ags_k_adam_tick, dim3(1), dim3(1), 0, s, clk, t.lr[0], t.lr[1], t.lr[2], t.lr[3], t.lr[4], beta1, beta2);
    long long rb = (t.numel[3] + 15) / 16; // 16 rows per block
    if (rb > 16384) rb = 16384;
    if (rb < 1) return;
    hipLaunchKernelGGL(ags_k_adam_rows_gathered, dim3((unsigned)rb), dim3(256), 0, s, a, clk, t.touched,
                       segs, seg_floats, world, capacity, slot_table, beta1, beta2, eps);
}

// ---------------------------------------------------------------------------------------
// Activations (gaussian_map.py:529-549) and their chain rule, one lane per Gaussian.
__global__ __launch_bounds__(256) void ags_k_activate(AgsActivation a, float* __restrict__ scales,
                                                      float* __restrict__ rotations, float* __restrict__ opacities) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v = a.scale_factor * expf(a.raw_scales[3 * i + k]);
        scales[3 * i + k] = fminf(fmaxf(v, 0.f), a.max_scale);
    }
    const float4 q = reinterpret_cast<const float4*>(a.raw_rotations)[i];
    const float inv = 1.0f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    reinterpret_cast<float4*>(rotations)[i] = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
    opacities[i] = 1.0f / (1.0f + expf(-a.raw_opacities[i]));
}

__global__ __launch_bounds__(256) void ags_k_activate_bwd(AgsActivation a, float* __restrict__ d_scales,
                                                          float* __restrict__ d_rotations,
                                                          float* __restrict__ d_opacities) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v = a.scale_factor * expf(a.raw_scales[3 * i + k]);
        // torch.clamp passes the gradient on the closed interval [0, max]
        d_scales[3 * i + k] = (v >= 0.f && v <= a.max_scale) ? d_scales[3 * i + k] * v : 0.f;
    }
    const float4 q = reinterpret_cast<const float4*>(a.raw_rotations)[i];
    const float4 d = reinterpret_cast<const float4*>(d_rotations)[i];
    const float nrm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    const float inv = 1.0f / fmaxf(nrm, 1e-12f);
    const float hx = q.x * inv, hy = q.y * inv, hz = q.z * inv, hw = q.w * inv;
    const float dot = hx * d.x + hy * d.y + hz * d.z + hw * d.w;
    reinterpret_cast<float4*>(d_rotations)[i] =
        make_float4((d.x - hx * dot) * inv, (d.y - hy * dot) * inv, (d.z - hz * dot) * inv, (d.w - hw * dot) * inv);
    const float o = 1.0f / (1.0f + expf(-a.raw_opacities[i]));
    d_opacities[i] = d_opacities[i] * o * (1.0f - o);
}

void ags_launch_activate(const AgsActivation& a, float* scales, float* rotations, float* opacities, hipStream_t s) {
    if (a.n <= 0) return;
    hipLaunchKernelGGL(ags_k_activate, dim3((a.n + 255) / 256), dim3(256), 0, s, a, scales, rotations, opacities);
}
void ags_launch_activate_bwd(const AgsActivation& a, float* d_scales, float* d_rotations, float* d_opacities,
                             hipStream_t s) {
    if (a.n <= 0) return;
    hipLaunchKernelGGL(ags_k_activate_bwd, dim3((a.n + 255) / 256), dim3(256), 0, s, a, d_scales, d_rotations,
                       d_opacities);
}
