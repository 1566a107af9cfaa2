// Tile binning: F2 scan of tiles-touched, F3 duplicate-with-keys, F4 radix sort of
// (tile | depth-bits) keys with gaussian-id payloads, F5 per-tile ranges.
//
// Counterpart of the cub::DeviceScan / duplicateWithKeys / cub::DeviceRadixSort /
// identifyTileRanges stages of the CUDA lineage — written from scratch
// for wave64: the instance count lives ONLY on the device (status[1]); every kernel is
// launched for the workspace capacity and trims itself, so the host never reads it back.
// The sort is a stable LSD radix sort, 8-bit digits: per pass a block histogram, a
// digit-major scan (one workgroup per digit) and a scatter whose in-block ranks come from
// wave-wide digit matching with __ballot (64-bit) — no LDS sort, no atomics in the ranks.
#include "ags_internal.h"

AGS_TL_DEFINE(binning)

// AgsStatus.peak_instances / overflow_passes: written by the ONE thread that publishes a pass's status block;
// passes on one workspace are stream-ordered, so plain read-modify-write is enough (no atomics)
__device__ __forceinline__ void ags_status_sticky(uint32_t* status, uint32_t total, uint32_t cap) {
    if (total > status[4]) status[4] = total;
    if (total > cap) status[5] += 1u;
    status[6] = 0u;      // max_tile_instances: not tracked in the scan-based modes
    status[7] = total;   // needed_instances
}

// ------------------------------------------------------------------ F2: scan of block sums
// number of visible surfels = sum of the per-block counts the preprocess kernel left
__device__ __forceinline__ uint32_t ags_sum_block_vis(const uint32_t* __restrict__ block_vis, int nblk, uint32_t* sh16) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t v = 0;
    for (int i = tid; i < nblk; i += 1024) v += block_vis[i];
    v = ags_wave_sum_u32(v);
    if (lane == 0) sh16[wave] = v;
    __syncthreads();
    uint32_t t = 0;
    for (int k = 0; k < 16; ++k) t += sh16[k];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(1024) void ags_k_scan_blocks(uint32_t* __restrict__ block_sums, int nblk,
                                                          uint32_t* __restrict__ status, uint32_t cap,
                                                          const uint32_t* __restrict__ block_vis) {
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nvis = ags_sum_block_vis(block_vis, nblk, wtot);
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nblk; base += 1024) {
        const int i = base + tid;
        const uint32_t x = (i < nblk) ? block_sums[i] : 0u;
        const uint32_t inc = ags_wave_incl_scan_u32(x);
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (int k = 0; k < wave; ++k) woff += wtot[k];
        const uint32_t carry = carry_s;
        if (i < nblk) block_sums[i] = carry + woff + inc - x; // exclusive
        __syncthreads();
        if (tid == 1023) carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (tid == 0) {
        const uint32_t total = carry_s;
        status[0] = total;
        status[1] = total < cap ? total : cap;
        status[2] = total > cap ? 1u : 0u;
        status[3] = nvis;
        ags_status_sticky(status, total, cap);
    }
}

// ------------------------------------------------------------------ F3: duplicate with keys
__global__ __launch_bounds__(AGS_PRE_THREADS) void ags_k_duplicate(
    int n, int tiles_x, const uint32_t* __restrict__ tiles, const ushort4* __restrict__ rect,
    const AgsGeom* __restrict__ geom, const uint32_t* __restrict__ block_prefix, uint64_t* __restrict__ keys,
    uint32_t* __restrict__ vals, uint32_t cap) {
    __shared__ uint32_t wtot[AGS_PRE_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x * AGS_PRE_THREADS + tid;
    const uint32_t cnt = (i < n) ? tiles[i] : 0u;
    const uint32_t inc = ags_wave_incl_scan_u32(cnt);
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    uint32_t off = block_prefix[blockIdx.x] + inc - cnt;
    for (int k = 0; k < wave; ++k) off += wtot[k];
    // Large footprints are emitted by the whole wave (coalesced), small ones per lane.
    uint4 mine = make_uint4(0, 0, 0, 0); // x0 | y0<<16, width, off, depth bits
    if (cnt) {
        const ushort4 rc = rect[i];
        mine = make_uint4((uint32_t)rc.x | ((uint32_t)rc.y << 16), (uint32_t)(rc.z - rc.x), off,
                          __float_as_uint(geom[i].dc));
    }
    const uint32_t COOP = 32;
    unsigned long long big = __ballot(cnt > COOP);
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const uint32_t xy = __shfl(mine.x, src), wd = __shfl(mine.y, src), o = __shfl(mine.z, src);
        const uint32_t db = __shfl(mine.w, src), c = __shfl(cnt, src);
        const uint32_t gid = (uint32_t)(blockIdx.x * AGS_PRE_THREADS + (wave << 6) + src);
        for (uint32_t t = lane; t < c; t += 64) {
            const uint32_t ty = (xy >> 16) + t / wd, tx = (xy & 0xFFFF) + t % wd;
            const uint32_t idx = o + t;
            if (idx < cap) {
                keys[idx] = ((uint64_t)(ty * tiles_x + tx) << 32) | db;
                vals[idx] = gid;
            }
        }
    }
    if (cnt && cnt <= COOP) {
        const uint32_t x0 = mine.x & 0xFFFF, y0 = mine.x >> 16, wd = mine.y;
        for (uint32_t t = 0; t < cnt; ++t) {
            const uint32_t ty = y0 + t / wd, tx = x0 + t % wd;
            const uint32_t idx = off + t;
            if (idx < cap) {
                keys[idx] = ((uint64_t)(ty * tiles_x + tx) << 32) | mine.w;
                vals[idx] = (uint32_t)i;
            }
        }
    }
}

// ------------------------------------------------------------------ F4: radix sort passes
// Layout of one block's keys: wave w owns [w*1024, (w+1)*1024) of the block's 4096 keys,
// item j of lane l sits at w*1024 + j*64 + l, so (wave, item, lane) order == memory order
// and the per-wave running counters give a stable rank.
__global__ __launch_bounds__(AGS_SORT_THREADS) void ags_k_sort_hist(
    const uint64_t* __restrict__ keys, const uint32_t* __restrict__ status, uint32_t* __restrict__ hist,
    uint32_t* __restrict__ totals, int nb_cap, int shift, uint32_t mask) {
    const uint32_t n = status[1];
    const uint32_t nb = (n + AGS_SORT_TILE - 1) / AGS_SORT_TILE;
    if (blockIdx.x >= nb) return;
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * AGS_SORT_TILE + (threadIdx.x >> 6) * (64 * AGS_SORT_ITEMS) + (threadIdx.x & 63);
#pragma unroll
    for (int j = 0; j < AGS_SORT_ITEMS; ++j) {
        const uint32_t idx = base + j * 64;
        if (idx < n) atomicAdd(&h[(uint32_t)(keys[idx] >> shift) & mask], 1u);
    }
    __syncthreads();
    const uint32_t c = h[threadIdx.x];
    hist[(size_t)threadIdx.x * nb_cap + blockIdx.x] = c;
    if (c) atomicAdd(&totals[threadIdx.x], c);
}

// one workgroup per digit: digit base = sum of totals of smaller digits, then an exclusive
// scan along the digit's row of per-block counts.
__global__ __launch_bounds__(256) void ags_k_sort_scan(const uint32_t* __restrict__ status,
                                                       uint32_t* __restrict__ hist,
                                                       const uint32_t* __restrict__ totals, int nb_cap) {
    const uint32_t n = status[1];
    const int nb = (int)((n + AGS_SORT_TILE - 1) / AGS_SORT_TILE);
    const int d = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ uint32_t wtot[4];
    __shared__ uint32_t carry_s;
    const uint32_t part = ags_wave_sum_u32(tid < d ? totals[tid] : 0u);
    if (lane == 0) wtot[wave] = part;
    __syncthreads();
    if (tid == 0) carry_s = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    __syncthreads();
    uint32_t* row = hist + (size_t)d * nb_cap;
    for (int base = 0; base < nb; base += 256) {
        const int i = base + tid;
        const uint32_t x = (i < nb) ? row[i] : 0u;
        const uint32_t inc = ags_wave_incl_scan_u32(x);
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (int k = 0; k < wave; ++k) woff += wtot[k];
        const uint32_t carry = carry_s;
        if (i < nb) row[i] = carry + woff + inc - x;
        __syncthreads();
        if (tid == 255) carry_s = carry + woff + inc;
        __syncthreads();
    }
}

__global__ __launch_bounds__(AGS_SORT_THREADS) void ags_k_sort_scatter(
    const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint64_t* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, const uint32_t* __restrict__ status, const uint32_t* __restrict__ hist,
    int nb_cap, int shift, uint32_t mask, int bits) {
    const uint32_t n = status[1];
    const uint32_t nb = (n + AGS_SORT_TILE - 1) / AGS_SORT_TILE;
    if (blockIdx.x >= nb) return;
    __shared__ uint32_t cnt[4][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int w = 0; w < 4; ++w) cnt[w][tid] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * AGS_SORT_TILE + wave * (64 * AGS_SORT_ITEMS) + lane;
    uint64_t k[AGS_SORT_ITEMS];
    uint32_t v[AGS_SORT_ITEMS];
    uint32_t rk[AGS_SORT_ITEMS];
#pragma unroll
    for (int j = 0; j < AGS_SORT_ITEMS; ++j) {
        const uint32_t idx = base + j * 64;
        const bool valid = idx < n;
        k[j] = valid ? keys_in[idx] : ~0ull;
        v[j] = valid ? vals_in[idx] : 0u;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < AGS_SORT_ITEMS; ++j) {
        const bool valid = (base + j * 64) < n;
        const uint32_t d = (uint32_t)(k[j] >> shift) & mask;
        unsigned long long peers = __ballot(valid);
        for (int b = 0; b < bits; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t r = __popcll(peers & lt);
        uint32_t prev = 0;
        if (valid && r == 0) { // lowest lane of each digit group bumps the wave's counter
            prev = cnt[wave][d];
            cnt[wave][d] = prev + __popcll(peers);
        }
        const int leader = valid ? (__ffsll((long long)peers) - 1) : lane;
        prev = __shfl(prev, leader);
        rk[j] = prev + r;
    }
    __syncthreads();
    { // per digit: global base of this block + exclusive offsets of the four waves
        uint32_t run = hist[(size_t)tid * nb_cap + blockIdx.x];
#pragma unroll
        for (int w = 0; w < 4; ++w) { const uint32_t c = cnt[w][tid]; cnt[w][tid] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < AGS_SORT_ITEMS; ++j) {
        if ((base + j * 64) < n) {
            const uint32_t d = (uint32_t)(k[j] >> shift) & mask;
            const uint32_t pos = cnt[wave][d] + rk[j];
            keys_out[pos] = k[j];
            vals_out[pos] = v[j];
        }
    }
}

// ------------------------------------------------------------------ F5: tile ranges
__global__ __launch_bounds__(256) void ags_k_ranges(const uint64_t* __restrict__ keys,
                                                    const uint32_t* __restrict__ status,
                                                    uint2* __restrict__ ranges) {
    const uint32_t n = status[1];
    const uint32_t idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const uint32_t t = (uint32_t)(keys[idx] >> 32);
    if (idx == 0) ranges[t].x = 0;
    else {
        const uint32_t p = (uint32_t)(keys[idx - 1] >> 32);
        if (p != t) { ranges[p].y = idx; ranges[t].x = idx; }
    }
    if (idx == n - 1) ranges[t].y = n;
}

int ags_sort_passes(int num_tiles) {
    int tb = 0;
    while ((1 << tb) < num_tiles) ++tb;
    return (32 + tb + 7) / 8;
}

AgsIdList ags_sorted_ids(char* ws, const AgsLayout& L, int binning_mode) {
    if (binning_mode == AGS_BIN_DIRECT)  // sorted copies in keys1, at slot * tile_cap (ags_k_tile_sort_direct)
        return AgsIdList{(const uint32_t*)(ws + L.keys1), 2};
    if (binning_mode == AGS_BIN_RADIX)  // payloads end in buffer (passes & 1)
        return AgsIdList{(const uint32_t*)(ws + ((ags_sort_passes(L.num_tiles) & 1) ? L.vals1 : L.vals0)), 1};
    return AgsIdList{(const uint32_t*)(ws + L.keys0), 2}; // low word of (depth<<32 | id), little endian
}

void ags_launch_binning(const AgsFrame& F, const AgsGaussians& in, char* ws, const AgsLayout& L, hipStream_t s) {
    uint32_t* status = (uint32_t*)(ws + L.status);
    uint32_t* bsum = (uint32_t*)(ws + L.block_sums);
    uint64_t* keys[2] = {(uint64_t*)(ws + L.keys0), (uint64_t*)(ws + L.keys1)};
    uint32_t* vals[2] = {(uint32_t*)(ws + L.vals0), (uint32_t*)(ws + L.vals1)};
    uint32_t* hist = (uint32_t*)(ws + L.hist);
    uint32_t* totals = (uint32_t*)(ws + L.totals);
    const uint32_t cap = (uint32_t)L.cap;
    hipLaunchKernelGGL(ags_k_scan_blocks, dim3(1), dim3(1024), 0, s, bsum, L.n_blocks, status, cap,
                       (const uint32_t*)(ws + L.block_vis));
    hipLaunchKernelGGL(ags_k_duplicate, dim3(L.n_blocks), dim3(AGS_PRE_THREADS), 0, s, in.n, F.tiles_x,
                       (const uint32_t*)(ws + L.tiles), (const ushort4*)(ws + L.rect),
                       (const AgsGeom*)(ws + L.geom), bsum, keys[0], vals[0], cap);
    int tb = 0;
    while ((1 << tb) < L.num_tiles) ++tb;
    const int total_bits = 32 + tb;
    int cur = 0, pass = 0;
    for (int shift = 0; shift < total_bits; shift += 8, ++pass) {
        const int bits = (total_bits - shift) < 8 ? (total_bits - shift) : 8;
        const uint32_t mask = (1u << bits) - 1u;
        uint32_t* tot = totals + pass * 256;
        hipLaunchKernelGGL(ags_k_sort_hist, dim3(L.nb_cap), dim3(AGS_SORT_THREADS), 0, s, keys[cur], status, hist,
                           tot, L.nb_cap, shift, mask);
        hipLaunchKernelGGL(ags_k_sort_scan, dim3(256), dim3(256), 0, s, status, hist, tot, L.nb_cap);
        hipLaunchKernelGGL(ags_k_sort_scatter, dim3(L.nb_cap), dim3(AGS_SORT_THREADS), 0, s, keys[cur], vals[cur],
                           keys[cur ^ 1], vals[cur ^ 1], status, hist, L.nb_cap, shift, mask, bits);
        cur ^= 1;
    }
    const int rblocks = (int)((L.cap + 255) / 256);
    hipLaunchKernelGGL(ags_k_ranges, dim3(rblocks), dim3(256), 0, s, keys[cur], status,
                       (uint2*)(ws + L.ranges));
}

// =======================================================================================
// Tile-sort binning (default): launch-lean alternative to the global radix sort.
//   preprocess<COUNT_TILES>  tile_count[t] += 1 per touched tile            (atomics)
//   ags_k_scan_tiles         ranges[t] = exclusive scan of tile_count       (1 workgroup)
//   ags_k_bucket             key (depth_bits<<32 | id) -> slot ranges[t].x + fill[t]++
//   ags_k_tile_sort          per tile: bitonic sort of its keys in LDS (<= 4096 keys) or in
//                            place in HBM/L2 (larger), one 256-thread workgroup per tile
// Keys are unique (the id is in the low word), so the result does not depend on the order
// the atomics were served in: same per-tile order as the stable radix path.
__global__ __launch_bounds__(1024) void ags_k_scan_tiles(const uint32_t* __restrict__ tile_count, int T,
                                                         uint2* __restrict__ ranges, uint32_t* __restrict__ status,
                                                         uint32_t cap, const uint32_t* __restrict__ block_vis,
                                                         int nblk, AgsViewStride vs) {
    { // (offsets are 0 for a single view)
        const size_t wo = (size_t)blockIdx.y * (size_t)vs.ws;
        AGS_WS_SHIFT(tile_count, wo); AGS_WS_SHIFT(ranges, wo); AGS_WS_SHIFT(status, wo); AGS_WS_SHIFT(block_vis, wo);
    }
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nvis = ags_sum_block_vis(block_vis, nblk, wtot);
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < T; base += 1024) {
        const int i = base + tid;
        const uint32_t x = (i < T) ? tile_count[i] : 0u;
        const uint32_t inc = ags_wave_incl_scan_u32(x);
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (int k = 0; k < wave; ++k) woff += wtot[k];
        const uint32_t carry = carry_s;
        if (i < T) {
            const uint32_t b = carry + woff + inc - x, e = b + x;
            ranges[i] = make_uint2(b < cap ? b : cap, e < cap ? e : cap);
        }
        __syncthreads();
        if (tid == 1023) carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (tid == 0) {
        const uint32_t total = carry_s;
        status[0] = total;
        status[1] = total < cap ? total : cap;
        status[2] = total > cap ? 1u : 0u;
        status[3] = nvis;
        ags_status_sticky(status, total, cap);
    }
}

// SCAN = true (images of <= AGS_BUCKET_SCAN_TILES tiles): every workgroup first rebuilds the
// exclusive scan of the tile counts in its own LDS (16 consecutive tiles per lane, DPP wave scan)
// instead of waiting for a separate 1-workgroup scan launch: 13 KB of L2 reads per workgroup buy
// one kernel and one dependency edge less per forward (-5 us at 1200x680).  Workgroup 0 publishes
// the ranges and the status block for the kernels that follow.
#define AGS_BUCKET_SCAN_TILES 4096
template <bool SCAN, bool AGG>
__global__ __launch_bounds__(AGS_PRE_THREADS) void ags_k_bucket(
    int n, int tiles_x, const uint32_t* __restrict__ tiles, const ushort4* __restrict__ rect,
    const AgsGeom* __restrict__ geom, uint2* __restrict__ ranges, uint32_t* __restrict__ tile_fill,
    uint64_t* __restrict__ keys, const uint32_t* __restrict__ tile_count, int T, uint32_t cap,
    uint32_t* __restrict__ status, const uint32_t* __restrict__ block_vis, int nblk, AgsViewStride vs) {
    { // (offsets are 0 for a single view)
        const size_t wo = (size_t)blockIdx.y * (size_t)vs.ws;
        AGS_WS_SHIFT(tiles, wo); AGS_WS_SHIFT(rect, wo); AGS_WS_SHIFT(geom, wo); AGS_WS_SHIFT(ranges, wo);
        AGS_WS_SHIFT(tile_fill, wo); AGS_WS_SHIFT(keys, wo); AGS_WS_SHIFT(tile_count, wo); AGS_WS_SHIFT(status, wo);
        AGS_WS_SHIFT(block_vis, wo);
    }
    __shared__ AgsEmitRec emit[AGS_PRE_THREADS];
    __shared__ uint32_t depth_bits[AGS_PRE_THREADS];
    __shared__ uint32_t pre[SCAN ? AGS_BUCKET_SCAN_TILES + 1 : 1]; // pre[t] = instances of tiles < t
    __shared__ uint32_t wtot[AGS_PRE_THREADS / 64];
    [[maybe_unused]] const int tl_w = blockIdx.x * (AGS_PRE_THREADS / 64) + (threadIdx.x >> 6);
    AGS_TL(1, tl_w, 0);
    if (SCAN) {
        constexpr int PER = AGS_BUCKET_SCAN_TILES / AGS_PRE_THREADS; // 16 consecutive tiles per lane
        const int t0 = threadIdx.x * PER;
        uint32_t c[PER];
        uint32_t sum = 0;
#pragma unroll
        for (int k = 0; k < PER; k += 4) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (t0 + k + 3 < T) v = *reinterpret_cast<const uint4*>(tile_count + t0 + k);
            else {
                if (t0 + k < T) v.x = tile_count[t0 + k];
                if (t0 + k + 1 < T) v.y = tile_count[t0 + k + 1];
                if (t0 + k + 2 < T) v.z = tile_count[t0 + k + 2];
            }
            c[k] = v.x; c[k + 1] = v.y; c[k + 2] = v.z; c[k + 3] = v.w;
            sum += v.x + v.y + v.z + v.w;
        }
        const uint32_t inc = ags_wave_incl_scan_u32(sum);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        uint32_t run = inc - sum;
        for (int k = 0; k < wave; ++k) run += wtot[k];
        if (threadIdx.x == 0) pre[0] = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            run += c[k];
            pre[t0 + k + 1] = run;
        }
        __syncthreads();
        if (blockIdx.x == 0) {
            for (int t = threadIdx.x; t < T; t += AGS_PRE_THREADS) {
                const uint32_t b = pre[t], e = pre[t + 1];
                ranges[t] = make_uint2(b < cap ? b : cap, e < cap ? e : cap);
            }
            uint32_t v = 0;
            for (int k = threadIdx.x; k < nblk; k += AGS_PRE_THREADS) v += block_vis[k];
            v = ags_wave_sum_u32(v);
            __syncthreads();
            if (lane == 0) wtot[wave] = v;
            __syncthreads();
            if (threadIdx.x == 0) {
                const uint32_t total = pre[T];
                status[0] = total;
                status[1] = total < cap ? total : cap;
                status[2] = total > cap ? 1u : 0u;
                ags_status_sticky(status, total, cap);
                uint32_t nv = 0;
                for (int k = 0; k < AGS_PRE_THREADS / 64; ++k) nv += wtot[k];
                status[3] = nv;
            }
        }
    }
    AGS_TL(1, tl_w, 1);
    const int i = blockIdx.x * AGS_PRE_THREADS + threadIdx.x;
    const uint32_t cnt = (i < n) ? tiles[i] : 0u;
    uint32_t x0 = 0, y0 = 0, wd = 1;
    AgsGeom g;
    g.mx = g.my = g.ca = g.cb = g.cc = g.o = 0.f;
    if (cnt) {
        const ushort4 rc = rect[i];
        x0 = rc.x; y0 = rc.y; wd = (uint32_t)(rc.z - rc.x);
        const float4* src = reinterpret_cast<const float4*>(geom + i);
        const float4 r0 = src[0], r1 = src[1];
        g.mx = r0.x; g.my = r0.y; g.ca = r0.z; g.cb = r0.w; g.cc = r1.x; g.o = r1.y;
        depth_bits[threadIdx.x] = __float_as_uint(r1.z);
    }
    AGS_TL(1, tl_w, 2);
    // same predicate and same inputs as the counting pass in ags_k_preprocess<true>
    ags_emit_tiles_balanced(emit + (threadIdx.x & ~63), cnt, x0, y0, wd, (uint32_t)threadIdx.x, g, tiles_x,
                            [&](bool hit, uint32_t t, uint32_t owner_tid, int) {
        const uint32_t got = ags_wave_agg_inc<AGG, true>(tile_fill, t, hit);
        if (hit) {
            uint32_t b, e;
            if (SCAN) { b = pre[t]; e = pre[t + 1]; e = e < cap ? e : cap; }
            else { const uint2 rg = ranges[t]; b = rg.x; e = rg.y; }
            const uint32_t slot = b + got;
            if (slot < e)
                keys[slot] = ((uint64_t)depth_bits[owner_tid] << 32) | (uint32_t)(blockIdx.x * AGS_PRE_THREADS + owner_tid);
        }
    });
    AGS_TL(1, tl_w, 3);
}

// The per-tile sort (ags_sort_tile_keys, ags_internal.h) as its own launch: one 256-thread workgroup
// per tile, 16 KiB of LDS.  Running it at the head of the tile's render workgroup instead was
// measured: +1 % at C2, but -7 % / -19 % on the 1.5 M / 5 M-surfel configurations, where most tiles
// hold more than 64 keys and would sort with half the threads and a quarter of the LDS.
#define AGS_TSORT_LDS_KEYS 2048
__global__ __launch_bounds__(256) void ags_k_tile_sort(const uint2* __restrict__ ranges, uint64_t* keys, int num_tiles,
                                                       AgsViewStride vs) {
    { // (offsets are 0 for a single view)
        const size_t wo = (size_t)blockIdx.y * (size_t)vs.ws;
        AGS_WS_SHIFT(ranges, wo); AGS_WS_SHIFT(keys, wo);
    }
    __shared__ __attribute__((aligned(16))) uint64_t sk[AGS_TSORT_LDS_KEYS];   // ags_bitonic8_load/store move 16-byte granules
    if (threadIdx.x < 64) AGS_TL(5, blockIdx.x, 0);
    const int tile = ags_xcd_remap(blockIdx.x, num_tiles);
    const uint2 rg = ranges[tile];
    ags_sort_tile_keys<256, AGS_TSORT_LDS_KEYS, false>(keys + rg.x, rg.y - rg.x, sk, threadIdx.x, keys + rg.x);
    if (threadIdx.x < 64) { AGS_TL(5, blockIdx.x, 1); AGS_TL_VAL(5, blockIdx.x, 6, rg.y - rg.x); }
}

// AGS_BIN_DIRECT: the unsorted keys sit in the tile's own range [tile * tile_cap, + count) of keys0 (written by the
// per-Gaussian kernel).  This gives the tile its SLOT - its place in its XCD band's heaviest-first order - sorts the
// keys into keys1 at slot * tile_cap, leaves the header ranges[slot] = {tile, count} for the blend kernels
// (ags_block_slot) and adds the list length to the spread partial sums / maxima the forward blend kernel turns into
// the status block.
__global__ __launch_bounds__(256) void ags_k_tile_sort_direct(uint2* __restrict__ ranges, uint64_t* keys_in,
                                                              uint64_t* keys_out, const uint32_t* __restrict__ tile_count,
                                                              uint32_t tile_cap, uint32_t* __restrict__ partial, int num_tiles,
                                                              AgsViewStride vs, uint32_t tc_stride) {
    { // (offsets are 0 for a single view)
        const size_t wo = (size_t)blockIdx.y * (size_t)vs.ws;
        AGS_WS_SHIFT(ranges, wo); AGS_WS_SHIFT(keys_in, wo); AGS_WS_SHIFT(keys_out, wo); AGS_WS_SHIFT(tile_count, wo);
        AGS_WS_SHIFT(partial, wo);
    }
    __shared__ __attribute__((aligned(16))) uint64_t sk[AGS_TSORT_LDS_KEYS];   // ags_bitonic8_load/store move 16-byte granules
    __shared__ uint32_t rank_part[4];
    if (threadIdx.x < 64) AGS_TL(5, blockIdx.x, 0);
    const int tile = ags_xcd_remap(blockIdx.x, num_tiles);
    const uint32_t cnt = tile_count[(size_t)tile * tc_stride];
    const uint32_t K = cnt < tile_cap ? cnt : tile_cap;
    // the tile's slot: band start + number of band tiles with a longer list (ties: lower index first) - exact,
    // deterministic, no counters to reset; <= T/8 counts read per workgroup
    int band_size;
    const int band0 = ags_xcd_band(blockIdx.x & 7, num_tiles, band_size);
    uint32_t ahead = 0;
    for (int j = threadIdx.x; j < band_size; j += 256) {
        const uint32_t c = tile_count[(size_t)(band0 + j) * tc_stride];
        ahead += (c > cnt || (c == cnt && band0 + j < tile)) ? 1u : 0u;
    }
    ahead = ags_wave_sum_u32(ahead);
    if ((threadIdx.x & 63) == 0) rank_part[threadIdx.x >> 6] = ahead;
    __syncthreads();
    const uint32_t slot = (uint32_t)band0 + rank_part[0] + rank_part[1] + rank_part[2] + rank_part[3];
    if (threadIdx.x == 0) {
        ranges[slot] = make_uint2((uint32_t)tile, K);
        if (cnt) {
            atomicAdd(&partial[AGS_PART(blockIdx.x, AGS_PART_SUM)], cnt);
            atomicMax(&partial[AGS_PART(blockIdx.x, AGS_PART_MAX)], cnt);
        }
    }
    ags_sort_tile_keys<256, AGS_TSORT_LDS_KEYS, false>(keys_in + (size_t)tile * tile_cap, K, sk, threadIdx.x,
                                                       keys_out + (size_t)slot * tile_cap);
    if (threadIdx.x < 64) { AGS_TL(5, blockIdx.x, 1); AGS_TL_VAL(5, blockIdx.x, 6, K); AGS_TL_VAL(5, blockIdx.x, 7, (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)(__builtin_amdgcn_s_getreg(63508) & 15) << 32)); }
}

// The same for views whose tile lists cannot exceed 64 R keys (tile_cap <= 128: the 1200x680 view of 200 k surfels has
// 51 at most): ONE wave per tile, R keys per lane, four tiles per workgroup.  The general form holds 16 KiB of LDS and
// four wave slots per tile, so a CU takes 8 tiles at a time and the 12.6 tiles per CU of that view need two rounds (2048
// tiles start at once, the rest when those end: per-CU span 6.3 us for wave lives of 2.7 us,
// profiles/experiments/timeline.py); with a wave per tile all tiles are resident at once.
template <int R>
__global__ __launch_bounds__(256) void ags_k_tile_sort_direct_wave(uint2* __restrict__ ranges, const uint64_t* __restrict__ keys_in,
                                                                   uint64_t* __restrict__ keys_out,
                                                                   const uint32_t* __restrict__ tile_count, uint32_t tile_cap,
                                                                   uint32_t* __restrict__ partial, int num_tiles, AgsViewStride vs,
                                                                   uint32_t tc_stride) {
    { // (offsets are 0 for a single view)
        const size_t wo = (size_t)blockIdx.y * (size_t)vs.ws;
        AGS_WS_SHIFT(ranges, wo); AGS_WS_SHIFT(keys_in, wo); AGS_WS_SHIFT(keys_out, wo); AGS_WS_SHIFT(tile_count, wo);
        AGS_WS_SHIFT(partial, wo);
    }
    // workgroup b serves FOUR tiles of XCD band b % 8 (tiles 4 (b / 8) .. + 3 of the band), one per wave: the band's
    // counts are read once per workgroup (every thread holds <= 2 of them) and compared with the four tiles' counts
    __shared__ uint32_t part[4][4];      // [wave that counted][tile of the workgroup]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int band_size;
    const int band0 = ags_xcd_band(blockIdx.x & 7, num_tiles, band_size);
    const int j0 = 4 * (int)(blockIdx.x >> 3);
    if (j0 >= band_size) return;                                   // workgroup-uniform
    const int mine_j = j0 + wave;
    const bool live = mine_j < band_size;                          // wave-uniform
    const int tile = band0 + (live ? mine_j : j0);
    [[maybe_unused]] const int tl_w = tile;
    if (live) AGS_TL(5, tl_w, 0);
    // the tile's keys are requested before its count is here (key 64 r + lane in register r; stale keys beyond the
    // count are masked below)
    const uint64_t* src = keys_in + (size_t)tile * tile_cap;
    uint64_t mine[R];
#pragma unroll
    for (int r = 0; r < R; ++r) mine[r] = ((uint32_t)(64 * r + lane) < tile_cap) ? src[64 * r + lane] : ~0ull;
    uint32_t tc[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) tc[w] = (j0 + w < band_size) ? tile_count[(size_t)(band0 + j0 + w) * tc_stride] : 0u;
    uint32_t ahead[4] = {0u, 0u, 0u, 0u};
    for (int j = threadIdx.x; j < band_size; j += 256) {
        const uint32_t c = tile_count[(size_t)(band0 + j) * tc_stride];
#pragma unroll
        for (int w = 0; w < 4; ++w) ahead[w] += (c > tc[w] || (c == tc[w] && j < j0 + w)) ? 1u : 0u;
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t a = ags_wave_sum_u32(ahead[w]);
        if (lane == 0) part[wave][w] = a;
    }
    __syncthreads();
    if (!live) return;
    const uint32_t cnt = tc[wave];
    const uint32_t K = cnt < tile_cap ? cnt : tile_cap;
#ifdef AGS_EXP_IDENTITY_SLOTS   // experiment: tiles keep their natural order (no heaviest-first slots)
    const uint32_t slot = (uint32_t)tile;
#else
    const uint32_t slot = (uint32_t)band0 + part[0][wave] + part[1][wave] + part[2][wave] + part[3][wave];
#endif
    if (lane == 0) {
        ranges[slot] = make_uint2((uint32_t)tile, K);
        if (cnt) {
            atomicAdd(&partial[AGS_PART(tile, AGS_PART_SUM)], cnt);
            atomicMax(&partial[AGS_PART(tile, AGS_PART_MAX)], cnt);
        }
    }
    uint32_t lo[R], hi[R], rank[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if ((uint32_t)(64 * r + lane) >= K) mine[r] = ~0ull;
        lo[r] = (uint32_t)mine[r]; hi[r] = (uint32_t)(mine[r] >> 32); rank[r] = 0;
    }
    // keys are unique: rank = number of smaller keys; every key of the list is broadcast through SGPRs (v_readlane)
#pragma unroll
    for (int q = 0; q < R; ++q) {
        if (K <= (uint32_t)(64 * q)) break;                       // wave-uniform
        const uint32_t nq = (K - 64u * q) < 64u ? (K - 64u * q) : 64u;
        for (uint32_t j = 0; j < nq; ++j) {
            const uint64_t other = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hi[q], j) << 32) |
                                   (uint32_t)__builtin_amdgcn_readlane((int)lo[q], j);
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (r == 0 || K > (uint32_t)(64 * r)) rank[r] += (other < mine[r]) ? 1u : 0u;
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        if ((uint32_t)(64 * r + lane) < K) keys_out[(size_t)slot * tile_cap + rank[r]] = mine[r];
    AGS_TL(5, tl_w, 1); AGS_TL_VAL(5, tl_w, 6, K);
    AGS_TL_VAL(5, tl_w, 7, (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)(__builtin_amdgcn_s_getreg(63508) & 15) << 32));
}

// ags_workspace_init_batch: the counter regions of `views` consecutive workspaces cleared by ONE launch (a memset per view
// is ~7 us of host time each; a training batch re-binds eleven workspaces at every keyframe)
__global__ __launch_bounds__(256) void ags_k_clear_regions(char* __restrict__ base, size_t stride, size_t offset, size_t bytes) {
    uint4* p = reinterpret_cast<uint4*>(base + (size_t)blockIdx.y * stride + offset);
    const size_t n16 = bytes >> 4;                                    // (the region is a multiple of 256 bytes)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
void ags_launch_clear_regions(char* base, size_t stride, size_t offset, size_t bytes, int views, hipStream_t s) {
    const int blocks = (int)((((bytes >> 4) + 255) / 256) < 64 ? (((bytes >> 4) + 255) / 256) : 64);
    hipLaunchKernelGGL(ags_k_clear_regions, dim3(blocks < 1 ? 1 : blocks, views), dim3(256), 0, s, base, stride, offset, bytes);
}

// ags_zero_many: blockIdx.y = region; 16-byte stores where the size allows, the tail in words
struct AgsZeroMany { char* p[AGS_ZERO_MANY_MAX]; size_t bytes[AGS_ZERO_MANY_MAX]; };
__global__ __launch_bounds__(256) void ags_k_zero_many(AgsZeroMany z) {
    char* p = z.p[blockIdx.y];
    const size_t bytes = z.bytes[blockIdx.y], n16 = bytes >> 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        reinterpret_cast<uint4*>(p)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0) {
        const size_t words = (bytes & 15) >> 2;
        if (threadIdx.x < words) reinterpret_cast<uint32_t*>(p + (n16 << 4))[threadIdx.x] = 0u;
    }
}
void ags_launch_zero_many(int count, void* const* regions, const size_t* bytes, hipStream_t s) {
    AgsZeroMany z = {};
    size_t most = 0;
    for (int k = 0; k < count; ++k) { z.p[k] = (char*)regions[k]; z.bytes[k] = bytes[k]; most = bytes[k] > most ? bytes[k] : most; }
    size_t blocks = ((most >> 4) + 1023) / 1024;     // four 16-byte stores per thread for the largest region
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(ags_k_zero_many, dim3((unsigned)blocks, count), dim3(256), 0, s, z);
}

void ags_launch_direct_sort(char* ws, const AgsLayout& L, const AgsViewStride& vs, hipStream_t s) {
    const bool no_wave = L.tune.tile_sort_no_wave != 0;   // AgsTuning: always the 256-thread form
    const uint32_t tile_cap = ags_direct_tile_cap(L);
    if (tile_cap <= 128u && !no_wave) {
#define AGS_LAUNCH_TSORT_WAVE(R)                                                                                          \
    hipLaunchKernelGGL(ags_k_tile_sort_direct_wave<R>, dim3(8 * ((((L.num_tiles + 7) / 8) + 3) / 4), vs.views), dim3(256), 0, s, (uint2*)(ws + L.ranges), \
                       (const uint64_t*)(ws + L.keys0), (uint64_t*)(ws + L.keys1), (const uint32_t*)(ws + L.tile_count),    \
                       tile_cap, (uint32_t*)(ws + L.totals), L.num_tiles, vs, (uint32_t)L.tc_stride)
        if (tile_cap <= 64u) AGS_LAUNCH_TSORT_WAVE(1); else AGS_LAUNCH_TSORT_WAVE(2);
#undef AGS_LAUNCH_TSORT_WAVE
        return;
    }
    hipLaunchKernelGGL(ags_k_tile_sort_direct, dim3(L.num_tiles, vs.views), dim3(256), 0, s, (uint2*)(ws + L.ranges),
                       (uint64_t*)(ws + L.keys0), (uint64_t*)(ws + L.keys1), (const uint32_t*)(ws + L.tile_count),
                       ags_direct_tile_cap(L), (uint32_t*)(ws + L.totals), L.num_tiles, vs, (uint32_t)L.tc_stride);
}

void ags_launch_tile_binning(const AgsFrame& F, const AgsGaussians& in, char* ws, const AgsLayout& L,
                             const AgsViewStride& vs, hipStream_t s) {
    uint32_t* status = (uint32_t*)(ws + L.status);
    uint2* ranges = (uint2*)(ws + L.ranges);
    uint64_t* keys = (uint64_t*)(ws + L.keys0);
    const uint32_t* tile_count = (const uint32_t*)(ws + L.tile_count);
    const uint32_t* block_vis = (const uint32_t*)(ws + L.block_vis);
    const bool no_fuse = L.tune.bucket_no_scan != 0;   // AgsTuning: always the separate tile-scan launch
    if (L.num_tiles <= AGS_BUCKET_SCAN_TILES && !no_fuse) {
#define AGS_LAUNCH_BUCKET(SCAN, AGG)                                                                                     \
    hipLaunchKernelGGL((ags_k_bucket<SCAN, AGG>), dim3(L.n_blocks, vs.views), dim3(AGS_PRE_THREADS), 0, s, in.n, F.tiles_x, \
                       (const uint32_t*)(ws + L.tiles), (const ushort4*)(ws + L.rect), (const AgsGeom*)(ws + L.geom),       \
                       ranges, (uint32_t*)(ws + L.tile_fill), keys, tile_count, L.num_tiles, (uint32_t)L.cap, status,       \
                       block_vis, L.n_blocks, vs)
        if (L.num_tiles <= AGS_AGG_MAX_TILES) AGS_LAUNCH_BUCKET(true, true); else AGS_LAUNCH_BUCKET(true, false);
    } else {
        hipLaunchKernelGGL(ags_k_scan_tiles, dim3(1, vs.views), dim3(1024), 0, s, tile_count, L.num_tiles, ranges, status,
                           (uint32_t)L.cap, block_vis, L.n_blocks, vs);
        AGS_LAUNCH_BUCKET(false, false);
#undef AGS_LAUNCH_BUCKET
    }
    hipLaunchKernelGGL(ags_k_tile_sort, dim3(L.num_tiles, vs.views), dim3(256), 0, s, (const uint2*)ranges, keys,
                       L.num_tiles, vs);
}
