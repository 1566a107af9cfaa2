// Per-tile alpha compositing, forward (F6) and backward (B1).
//
// Wave64-native shape, not a 16x16-thread CUDA block: a wavefront owns SLOTS of the four
// 8x8-pixel quadrants of a 16x16 tile and every lane owns SLOTS pixels (one per quadrant; the
// code and the comments below still call a wave's slots its "strips" - they were 16x4 strips until
// the more compact 8x8 shape measured 5 % fewer (surfel, slot) evaluations), so
//   * a round stages 64 projected surfels (64 B each, one per lane) into the wave's own
//     4 KiB of LDS; a wave never waits for another wave (no workgroup barrier);
//   * each staged record is read from LDS once per wave (broadcast ds_read_b128) and serves
//     SLOTS pixel evaluations;
//   * the staging lane of a surfel also computes which strips it can reach at all
//     (ags_reaches_box, exact and conservative); the round's loop then walks only the set bits
//     of the 64-bit ballot of "reaches one of my strips" (s_ff1), and evaluates only the
//     reachable strips under scalar branches; `__any` skips the accumulate when no lane takes
//     the surfel and `__all(done)` ends the wave early;
//   * the backward (default: ags_k_render_bwd_mfma, one quadrant per wave) leaves the reduction of
//     each surfel's 15 partial gradients over the wave's pixels to the f32 matrix cores and adds
//     four whole gradient records per atomic instruction; the VALU form (ags_k_render_bwd<SLOTS>,
//     AGS_BWD_MFMA=0) reduces with a transposed permlane-swap/DPP reduction (ags_wave_reduce16) and
//     issues ONE 15-lane atomic per (surfel, wave) instead of 64 x SLOTS x 15 atomics.
// SLOTS = 4 is one wave per tile; SLOTS = 2 / 1 split a tile over 2 / 4 waves (fewer VGPRs,
// more waves in flight).  Measured best: 1 below ~12 k tiles in flight, 2 above (AGS_RENDER_SLOTS overrides).
// Blocks are mapped to tiles XCD-aware (ags_xcd_remap) so one XCD's L2 serves a contiguous
// band of tiles.  Counterpart of renderCUDA fwd/bwd in SURVEY.md §2.3; arithmetic in
// surfel_math.h.
#include "ags_internal.h"
#include "loss_pixel.h"

AGS_TL_DEFINE(render)
AGS_PROBE_DEFINE()

// Issue priority by phase.  The SIMD arbitrates VALU issue between its resident waves by priority, then AGE: a wave
// that has just started competes with older waves that sit in their blend loops and keep the vector pipe busy, and
// gets the leftover issue slots - its short prologue (a few dozen address and set-up instructions between the loads)
// crawls, its loads go out late, and the same happens to the few instructions in front of its final stores.
// Prologue and epilogue therefore run at raised priority and the blend loop at the default.
// (the macros live in ags_experiments.h: an experiment build can switch them off)

template <int N>
struct AgsWaveStageT {  // one per wave, in LDS
    AgsGeom sg[N];
    uint32_t sid[N];
};
typedef AgsWaveStageT<64> AgsWaveStage;

// stage surfel `gid` from lane `lane` and return its strip-reach mask for tile (bx0,by0): bit s is
// set when the surfel can reach strip s; only this wave's SLOTS strips (from strip0) are tested
template <int SLOTS, typename STAGE = AgsWaveStage>
__device__ __forceinline__ uint32_t ags_stage_one(STAGE& st, int lane, const AgsGeom* __restrict__ geom,
                                                  uint32_t gid, float bx0, float by0, int strip0) {
    const float4* src = reinterpret_cast<const float4*>(geom + gid);
    const float4 r0 = src[0], r1 = src[1], r2 = src[2], r3 = src[3];
    float4* dst = reinterpret_cast<float4*>(&st.sg[lane]);
    dst[0] = r0; dst[1] = r1; dst[2] = r2; dst[3] = r3;
    st.sid[lane] = gid;
    AgsGeom me;
    me.mx = r0.x; me.my = r0.y; me.ca = r0.z; me.cb = r0.w; me.cc = r1.x; me.o = r1.y;
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        const int s = strip0 + k;
        const float qx0 = bx0 + 8.f * (float)(s & 1), qy0 = by0 + 8.f * (float)(s >> 1); // 8x8 quadrant s
        m |= ags_reaches_box(me, qx0, qx0 + 7.f, qy0, qy0 + 7.f) ? (1u << s) : 0u;
    }
    return m;
}

// the same in two halves, so that independent work can sit between the gather and its use: issue = the four
// 16-byte loads of the record; commit = park it in the wave's LDS stage and compute the reach mask
struct AgsRec4 { float4 r0, r1, r2, r3; };
__device__ __forceinline__ AgsRec4 ags_stage_issue(const AgsGeom* __restrict__ geom, uint32_t gid) {
    const float4* src = reinterpret_cast<const float4*>(geom + gid);
    AgsRec4 r;
    r.r0 = src[0]; r.r1 = src[1]; r.r2 = src[2]; r.r3 = src[3];
    return r;
}
template <int SLOTS, typename STAGE>
__device__ __forceinline__ uint32_t ags_stage_commit(STAGE& st, int lane, const AgsRec4& r, uint32_t gid, float bx0,
                                                     float by0, int strip0) {
    float4* dst = reinterpret_cast<float4*>(&st.sg[lane]);
    dst[0] = r.r0; dst[1] = r.r1; dst[2] = r.r2; dst[3] = r.r3;
    st.sid[lane] = gid;
    AgsGeom me;
    me.mx = r.r0.x; me.my = r.r0.y; me.ca = r.r0.z; me.cb = r.r0.w; me.cc = r.r1.x; me.o = r.r1.y;
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        const int s = strip0 + k;
        const float qx0 = bx0 + 8.f * (float)(s & 1), qy0 = by0 + 8.f * (float)(s >> 1);
        m |= ags_reaches_box(me, qx0, qx0 + 7.f, qy0, qy0 + 7.f) ? (1u << s) : 0u;
    }
    return m;
}

// One whole 16-byte piece of a staged record per LDS instruction.  The LDS pipe is shared by the CU's four SIMDs and a
// wave-instruction occupies it by WIDTH CLASS (ds_read_b128: 4 cycles, b96: 8, read2_b32: 4, b64: 2 - MI355X_MICROARCH.md),
// so sixteen dwords cost 16 cycles as four b128 reads but 36 as the b96 / read2_b32 mix the compiler picks when it
// narrows the loads to the components each basic block uses - and with 32 resident waves per CU the blend loops are
// as close to the LDS limit as to the VALU one.  Volatile keeps the access whole.
typedef float ags_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ags_lds_read16(const float4* p) {
    // (explicit LDS address space: a volatile access through a generic pointer would become a flat load)
    const ags_f4 v = *(const volatile __attribute__((address_space(3))) ags_f4*)(p);
    return make_float4(v.x, v.y, v.z, v.w);
}

// What the forward and the matrix-core backward stage per surfel: the quadrant-local polynomial form of
// surfel_math.h (ags_quad_coeffs) - the payload of the 64-byte record plus one {E0, E1, E2, D0} per quadrant the
// wave owns.  64 B for one quadrant per wave, 80 B for two, 112 B for four.
template <int SLOTS>
struct AgsStagedRec {
    float4 a;              // E3, E4, E5, gx
    float4 b;              // gy, r, g, b
    float4 c;              // nx, ny, nz, conf
    float4 slot[SLOTS];    // E0, E1, E2, D0 of quadrant strip0 + s
};
template <int SLOTS, int N, bool SID>
struct AgsWaveStageQ {     // one per wave, in LDS (allocated in 1280-byte granules: profiles/experiments/lds_granule.cpp)
    AgsStagedRec<SLOTS> sg[N];
    uint32_t sid[N];       // surfel ids (forward: the importance / count atomics)
};
template <int SLOTS, int N>
struct AgsWaveStageQ<SLOTS, N, false> {
    AgsStagedRec<SLOTS> sg[N];
};
// park record `r` of surfel `gid` from lane `lane` in quadrant-local form; returns its strip-reach mask;
// oxy (optional): mean - centre of the wave's FIRST quadrant (the backward's moment shift)
template <int SLOTS, int N, bool SID>
__device__ __forceinline__ uint32_t ags_stage_commit_q(AgsWaveStageQ<SLOTS, N, SID>& st, int lane, const AgsRec4& r, uint32_t gid,
                                                       float bx0, float by0, int strip0, float2* oxy = nullptr) {
    AgsGeom g;
    g.mx = r.r0.x; g.my = r.r0.y; g.ca = r.r0.z; g.cb = r.r0.w; g.cc = r.r1.x; g.o = r.r1.y; g.dc = r.r1.z; g.gx = r.r1.w;
    g.gy = r.r2.x;
    AgsQuadShared sh;
    ags_quad_shared(g, sh);
    AgsStagedRec<SLOTS>& d = st.sg[lane];
    d.a = make_float4(sh.E3, sh.E4, sh.E5, g.gx);
    d.b = r.r2;
    d.c = r.r3;
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        const int s = strip0 + k;
        const float qx0 = bx0 + 8.f * (float)(s & 1), qy0 = by0 + 8.f * (float)(s >> 1); // 8x8 quadrant s
        AgsQuadCoef q;
        ags_quad_coeffs(g, qx0 + 3.5f, qy0 + 3.5f, q);
        d.slot[k] = make_float4(q.E0, q.E1, q.E2, q.D0);
        if (k == 0 && oxy) *oxy = make_float2(g.mx - (qx0 + 3.5f), g.my - (qy0 + 3.5f));
        m |= ags_reaches_box(g, qx0, qx0 + 7.f, qy0, qy0 + 7.f) ? (1u << s) : 0u;
    }
    if constexpr (SID) st.sid[lane] = gid;
    return m;
}

__device__ __forceinline__ void ags_wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// AGS_BIN_DIRECT: nobody has added up the view's instance count yet (there is no scan): the first wave of the
// forward blend kernel turns the spread partial sums / maxima (ags_k_tile_sort_direct, ags_k_preprocess<2>) into the
// status block and leaves them zeroed for the next pass.  status == nullptr: the other binning modes.
struct AgsFinalize { uint32_t* status; uint32_t* partial; uint32_t tile_cap; uint32_t tc_stride; };

__device__ __forceinline__ void ags_finalize_status(const AgsFinalize& fin, int num_tiles, int lane) {
    uint32_t s = fin.partial[AGS_PART(lane, AGS_PART_SUM)], m = fin.partial[AGS_PART(lane, AGS_PART_MAX)],
             v = fin.partial[AGS_PART(lane, AGS_PART_VIS)];
    fin.partial[AGS_PART(lane, AGS_PART_SUM)] = 0u; fin.partial[AGS_PART(lane, AGS_PART_MAX)] = 0u;
    fin.partial[AGS_PART(lane, AGS_PART_VIS)] = 0u;
    s = ags_wave_sum_u32(s); m = ags_wave_max_u32(m); v = ags_wave_sum_u32(v);
    if (lane == 0) {
        const unsigned long long need64 = (unsigned long long)m * (unsigned long long)num_tiles;
        const uint32_t need = need64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)need64;
        const bool over = m > fin.tile_cap;
        fin.status[0] = s; fin.status[1] = s; fin.status[2] = over ? 1u : 0u; fin.status[3] = v;
        if (need > fin.status[4]) fin.status[4] = need;
        if (over) fin.status[5] += 1u;
        fin.status[6] = m; fin.status[7] = need;
        fin.status[AGS_STATUS_EARLY] = 0u;   // the per-Gaussian kernel's early overflow note: consumed (AgsWorkspace.early_status_host)
    }
}

template <int SLOTS, bool STATS, bool LOSS = false>
__global__ __launch_bounds__(64) void ags_k_render_fwd(
    AgsFrame F, int normalize_depth, float weight_thres, const float* __restrict__ bgp,
    const float* __restrict__ mask, const uint2* __restrict__ ranges, const uint32_t* __restrict__ vals,
    int id_stride, const AgsGeom* __restrict__ geom, AgsImages out, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, float* __restrict__ importance, int* __restrict__ count, int num_tiles,
    uint32_t* __restrict__ tile_count, uint32_t* __restrict__ tile_fill, AgsFinalize fin,
    uint32_t tile_cap, AgsViewStride vs, int seen_only, AgsLossFuse lf) {
    // seen_only (AGS_STATS_SEEN, an explicit argument - not a null `importance`, which the batched prologue below would
    // shift into a non-null bogus pointer for the views y >= 1): count[i] = 1 for every surfel with a counted pixel
    { // batched forward: this workgroup's view // (offsets are 0 for a single view)
        const size_t wo = (size_t)blockIdx.y * (size_t)vs.ws, po = (size_t)blockIdx.y * (size_t)vs.px;
        AGS_WS_SHIFT(ranges, wo); AGS_WS_SHIFT(vals, wo); AGS_WS_SHIFT(geom, wo); AGS_WS_SHIFT(final_T, wo);
        AGS_WS_SHIFT(n_contrib, wo); AGS_WS_SHIFT(tile_count, wo); AGS_WS_SHIFT(tile_fill, wo);
        if (fin.status) { AGS_WS_SHIFT(fin.status, wo); AGS_WS_SHIFT(fin.partial, wo); }
        if (mask) mask += po;
        out.rgb += 3 * po; out.normal += 3 * po; out.depth += po; out.opacity += po; out.confidence += po;
        if (STATS) { if (importance) importance += (size_t)blockIdx.y * (size_t)vs.n; count += (size_t)blockIdx.y * (size_t)vs.n; }
        if constexpr (LOSS) {   // stage 1 of the loss head rides in the epilogue: this view's slices of its images
            const size_t go = lf.gt_index ? (size_t)lf.gt_index[blockIdx.y] * (size_t)vs.px : po;
            lf.gt_rgb += 3 * go; lf.gt_depth += go; lf.n_img += 3 * po; lf.d_rgb += 3 * po; lf.d_depth += po;
        }
    }
    __shared__ AgsWaveStageQ<SLOTS, 64, STATS> st;
    const int lane = threadIdx.x;
    int slot, wave;   // blockIdx-derived: SGPRs, strip masks become scalar tests
    if (!ags_wave_block(blockIdx.x, num_tiles, 4 / SLOTS, slot, wave)) return;
    // device-side configuration (AgsCamera.config): the STATS kernel is launched and config[3] says whether the
    // statistics are wanted at all (wave-uniform: scalar branches around the per-surfel reductions)
    normalize_depth = ags_cfg_flag(F.cfg, 1, normalize_depth);
    const bool stats_on = STATS && ags_cfg_flag(F.cfg, 3, 1) != 0;
    [[maybe_unused]] const int tl_w = slot * (4 / SLOTS) + wave;
    AGS_TL(2, tl_w, 0);
    AGS_PRIO_HIGH();
    if (fin.status && blockIdx.x == 0) ags_finalize_status(fin, num_tiles, lane);   // wave-uniform
    // direct binning: the first 64 ids of the block's slot are requested before the slot's header says which tile
    // this is and how long its list (ags_block_slot); lanes beyond the list hold a stale key and are masked below
    uint32_t spec_id = 0;
    if (tile_cap && (uint32_t)lane < tile_cap)
        spec_id = vals[((size_t)slot * tile_cap + lane) * id_stride];
    uint2 rg;
    const int tile = ags_slot_tile(ranges, slot, tile_cap, rg);
    const int tx = tile % F.tiles_x, ty = tile / F.tiles_x;
    const int strip0 = wave * SLOTS;                       // first of this wave's slots (8x8 quadrants of the tile)
    // quadrant q = strip0 + s sits at (q & 1, q >> 1); lane l is pixel (l & 7, l >> 3) of its quadrant
    const int pxl = tx * AGS_TILE + (lane & 7), pyl = ty * AGS_TILE + (lane >> 3);
#define AGS_PX(s) (pxl + 8 * ((strip0 + (s)) & 1))
#define AGS_PY(s) (pyl + 8 * ((strip0 + (s)) >> 1))
    const float bx0 = (float)(tx * AGS_TILE), by0 = (float)(ty * AGS_TILE);
    const uint32_t my_strips = ((1u << SLOTS) - 1u) << strip0;
    // last consumer of this tile's binning counters: leave them zero for the next forward pass
    if (wave == 0 && lane == 0) { tile_count[(size_t)tile * (tile_cap ? fin.tc_stride : 1u)] = 0u; tile_fill[tile] = 0u; }
    // the lane's pixel relative to the centre of its quadrant - the same in every quadrant the wave owns
    const float qx = (float)(lane & 7) - 3.5f, qy = (float)(lane >> 3) - 3.5f;
    AgsPix pix[SLOTS];
    float mk[SLOTS];
    int alldone = 1;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const bool inside = (AGS_PX(s) < F.W) && (AGS_PY(s) < F.H);
        ags_pix_init(pix[s], inside);
        mk[s] = inside ? 1.f : 0.f;
        if (STATS && stats_on && mask != nullptr && inside) mk[s] = mask[(size_t)AGS_PY(s) * F.W + AGS_PX(s)] > 0.f ? 1.f : 0.f;
        alldone &= pix[s].done;
    }
    AGS_TL(2, tl_w, 1);
    [[maybe_unused]] uint32_t tl_iters = 0;
    AGS_PROBE_VARS();
    for (uint32_t base = rg.x; base < rg.y; base += 64) {
        if (__all(alldone)) break;
        ags_wave_lds_sync();
        const uint32_t idx = base + lane;
        uint32_t m = 0;
        if (idx < rg.y) {
            const uint32_t gid = (tile_cap && base == rg.x) ? spec_id : vals[(size_t)idx * id_stride];
            m = ags_stage_commit_q<SLOTS, 64, STATS>(st, lane, ags_stage_issue(geom, gid), gid, bx0, by0, strip0);
        }
        ags_wave_lds_sync();
        unsigned long long act = __ballot((m & my_strips) != 0u); // staged surfels that reach my strips
        if (base == rg.x) AGS_TL(2, tl_w, 2);
        AGS_PRIO_LOOP();
        tl_iters += (uint32_t)__builtin_popcountll(act);
        while (act) {
            const int k = __ffsll((long long)act) - 1;
            act &= act - 1;
            const uint32_t mk_bits = ((uint32_t)__builtin_amdgcn_readlane((int)m, k)) >> strip0; // wave-uniform
            const AgsStagedRec<SLOTS>& g = st.sg[k];
            const float4 ga = ags_lds_read16(&g.a);                       // E3, E4, E5, gx
            const AgsQuadShared sh = {ga.x, ga.y, ga.z};
            // al[s] = the pixel's alpha if it takes the surfel, else 0 (alpha >= 1/255 > 0 when it does)
            float al[SLOTS], d0[SLOTS];
            bool any = false;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                al[s] = d0[s] = 0.f;
                if (SLOTS == 1 || (mk_bits & (1u << s))) { // one slot: the ballot above already says it is reached
                    const float4 gs = ags_lds_read16(&g.slot[s]);         // E0, E1, E2, D0
                    const AgsQuadCoef qc = {gs.x, gs.y, gs.z, gs.w};
                    const float a = ags_alpha_quad(sh, qc, qx, qy);
                    const bool take = a >= AGS_ALPHA_MIN && !pix[s].done;   // (alpha >= 1/255 > 0 when taken: the mask is the test)
                    al[s] = take ? a : 0.f;
                    d0[s] = gs.w;
                    any |= take;
                }
            }
            AGS_PROBE_PAIR(any);
            if (!__any(any)) continue;
            const uint32_t pos1 = base - rg.x + k + 1;
            const float4 gb = ags_lds_read16(&g.b), gc = ags_lds_read16(&g.c);   // gy, r, g, b | nx, ny, nz, conf
            const float dq = fmaf(gb.x, qy, ga.w * qx);                   // the surfel's depth slope at this pixel offset
            float wsum = 0.f;
            uint32_t wcnt = 0;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                if (SLOTS == 1 || __any(al[s] > 0.f)) { // wave-uniform; lanes that do not take the surfel blend alpha = 0
                    const float w = ags_blend_apply_q(pix[s], gb.y, gb.z, gb.w, gc.x, gc.y, gc.z, gc.w, d0[s] + dq, al[s], pos1);
                    if (STATS && stats_on) { const float wm = w * mk[s]; wsum += wm; wcnt += (wm > weight_thres) ? 1u : 0u; }
                }
            }
            if (STATS && stats_on && seen_only) {
                // AGS_STATS_SEEN: the caller only asks WHETHER a surfel has a counted pixel (post_processing's `counts >= 1`):
                // one vote and one plain store instead of two wave reductions and two atomics per surfel and wave
                if constexpr (STATS) {
                    if (__any(wcnt != 0u) && lane == 0) count[st.sid[k]] = 1;
                }
            } else if (STATS && stats_on) {
                const float ts = ags_wave_sum(wsum);
                const uint32_t tc = ags_wave_sum_u32(wcnt);
                if constexpr (STATS) {
                    if (lane == 0 && ts > 0.f) {
                        const uint32_t gid = st.sid[k];
                        atomicAdd(&importance[gid], ts);
                        if (tc) atomicAdd(&count[gid], (int)tc);
                    }
                }
            }
        }
        alldone = 1;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) alldone &= pix[s].done;
    }
    AGS_TL(2, tl_w, 3);
    AGS_PROBE_FLUSH(0);
    AGS_PRIO_HIGH();
    AGS_TL_VAL(2, tl_w, 5, rg.y - rg.x);
    AGS_TL_VAL(2, tl_w, 6, tl_iters);
    AGS_TL_VAL(2, tl_w, 7, (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32));
    const float bg0 = bgp[0], bg1 = bgp[1], bg2 = bgp[2];
    const size_t HW = (size_t)F.H * F.W;
    [[maybe_unused]] float s_rgb = 0.f, s_dep = 0.f;   // LOSS: this wave's L1 sums
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int px = AGS_PX(s), py = AGS_PY(s);
        if (px < F.W && py < F.H) {
            const size_t o = (size_t)py * F.W + px;
            const float T = pix[s].T, A = 1.f - T;
            const float c0 = pix[s].c0 + T * bg0, c1 = pix[s].c1 + T * bg1, c2 = pix[s].c2 + T * bg2;
            const float dep = normalize_depth ? pix[s].d / fmaxf(A, AGS_DEPTH_A_EPS) : pix[s].d;
            out.rgb[o] = c0; out.rgb[HW + o] = c1; out.rgb[2 * HW + o] = c2;
            out.normal[o] = pix[s].n0; out.normal[HW + o] = pix[s].n1; out.normal[2 * HW + o] = pix[s].n2;
            out.depth[o] = dep;
            out.opacity[o] = A;
            out.confidence[o] = pix[s].cf;
            final_T[o] = T;
            n_contrib[o] = pix[s].last;
            if constexpr (LOSS) {
                // ags_k_loss_stage1 (loss.hip) on the values just stored: the same per-pixel function (loss_pixel.h)
                const AgsStage1Pixel lp = ags_loss_stage1_pixel(A, pix[s].n0, pix[s].n1, pix[s].n2, c0, c1, c2, lf.gt_rgb[o],
                                                                lf.gt_rgb[HW + o], lf.gt_rgb[2 * HW + o], dep, lf.gt_depth[o],
                                                                lf.k_rgb, lf.k_depth);
                lf.n_img[o] = lp.n[0]; lf.n_img[HW + o] = lp.n[1]; lf.n_img[2 * HW + o] = lp.n[2];
                lf.d_rgb[o] = lp.d_rgb[0]; lf.d_rgb[HW + o] = lp.d_rgb[1]; lf.d_rgb[2 * HW + o] = lp.d_rgb[2];
                lf.d_depth[o] = lp.d_depth;
                s_rgb += lp.s_rgb; s_dep += lp.s_dep;
                if (lp.vis) atomicAdd(&lf.msum[o], 1);          // concurrent views into a pre-zeroed count
            }
        }
    }
    if constexpr (LOSS) {
        const float t_rgb = ags_wave_sum(s_rgb), t_dep = ags_wave_sum(s_dep);
        if (lane == 0) {
            float* row = lf.accum + (size_t)(blockIdx.x & (AGS_LOSS_ACCUM_ROWS - 1)) * lf.accum_stride;
            const int view = (int)blockIdx.y;
            atomicAdd(&row[0], t_rgb); atomicAdd(&row[1], t_dep);
            atomicAdd(&row[4 + 2 * view], t_rgb); atomicAdd(&row[5 + 2 * view], t_dep);
        }
    }
    AGS_TL(2, tl_w, 4);
}

#ifdef AGS_BWD_WAVES   // experiment knob: force a register budget for N resident waves per SIMD
#define AGS_BWD_ATTR __attribute__((amdgpu_waves_per_eu(AGS_BWD_WAVES, AGS_BWD_WAVES)))
#else
#define AGS_BWD_ATTR
#endif
template <int SLOTS>
__global__ __launch_bounds__(64) AGS_BWD_ATTR void ags_k_render_bwd(
    AgsFrame F, int normalize_depth, const float* __restrict__ bgp, const uint2* __restrict__ ranges,
    const uint32_t* __restrict__ vals, int id_stride, const AgsGeom* __restrict__ geom,
    const float* __restrict__ depth_out, const float* __restrict__ opac_out, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, AgsImageGrads dout, float* __restrict__ dgeom, int num_tiles,
    AgsTick tick, uint32_t tile_cap, AgsViewStride vs) {
    { // batched backward: this workgroup's view // (offsets are 0 for a single view)
        const size_t wo = (size_t)blockIdx.y * (size_t)vs.ws, po = (size_t)blockIdx.y * (size_t)vs.px;
        AGS_WS_SHIFT(ranges, wo); AGS_WS_SHIFT(vals, wo); AGS_WS_SHIFT(geom, wo); AGS_WS_SHIFT(final_T, wo);
        AGS_WS_SHIFT(n_contrib, wo); AGS_WS_SHIFT(dgeom, wo);
        depth_out += po; opac_out += po;
        if (dout.d_rgb) dout.d_rgb += 3 * po;
        if (dout.d_normal) dout.d_normal += 3 * po;
        if (dout.d_depth) dout.d_depth += po;
        if (dout.d_opacity) dout.d_opacity += po;
        if (dout.d_confidence) dout.d_confidence += po;
    }
    __shared__ AgsWaveStage st;
    const int lane = threadIdx.x;
    int slot, wave;
    if (!ags_wave_block(blockIdx.x, num_tiles, 4 / SLOTS, slot, wave)) return;
    normalize_depth = ags_cfg_flag(F.cfg, 1, normalize_depth);
    // side job of a step's last backward: advance the Adam device clock.  Nothing in this launch
    // reads it; the per-Gaussian kernel that follows (fused step) or ags_adam_step_device does.
    if (tick.clock && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        ags_adam_tick(tick.clock, tick.lr, tick.beta1, tick.beta2);
        if (tick.count_snap) *tick.count_snap = (uint32_t)*tick.rows_count;
    }
    uint2 rg;
    const int tile = ags_slot_tile(ranges, slot, tile_cap, rg);
    if (rg.y <= rg.x) return;
    const int tx = tile % F.tiles_x, ty = tile / F.tiles_x;
    const int strip0 = wave * SLOTS;
    const int pxl = tx * AGS_TILE + (lane & 7), pyl = ty * AGS_TILE + (lane >> 3);
    const float bx0 = (float)(tx * AGS_TILE), by0 = (float)(ty * AGS_TILE);
    const uint32_t my_strips = ((1u << SLOTS) - 1u) << strip0;
    const float bg[3] = {bgp[0], bgp[1], bgp[2]};
    const size_t HW = (size_t)F.H * F.W;
    AgsPixGrad pg[SLOTS];
    uint32_t mymax = 0;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int px = AGS_PX(s), py = AGS_PY(s);
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
    if (maxlast == 0) return; // wave-uniform; no workgroup barrier anywhere in this kernel
    // after the reduction every quad holds one field of the gradient record; its first lane adds it (15 is padding)
    const int my_field = ((lane & 3) == 0 && ags_reduce16_field(lane) < 15) ? ags_reduce16_field(lane) : -1;
    for (int r = (int)((maxlast - 1) >> 6); r >= 0; --r) {
        const uint32_t k0 = (uint32_t)r << 6;
        ags_wave_lds_sync();
        uint32_t m = 0;
        if (k0 + lane < maxlast) m = ags_stage_one<SLOTS>(st, lane, geom, vals[(size_t)(rg.x + k0 + lane) * id_stride], bx0, by0, strip0);
        ags_wave_lds_sync();
        unsigned long long act = __ballot((m & my_strips) != 0u);
        while (act) { // back to front: highest staged position first
            const int k = 63 - __clzll((long long)act);
            act &= ~(1ull << k);
            const uint32_t mk_bits = ((uint32_t)__builtin_amdgcn_readlane((int)m, k)) >> strip0;
            const AgsGeom g = st.sg[k];
            const uint32_t pos1 = k0 + k + 1;
            float dx[SLOTS], dy[SLOTS], al[SLOTS]; // al[s] = 0 for a pixel that does not take the surfel
            bool any = false;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                dx[s] = dy[s] = al[s] = 0.f;
                if (SLOTS == 1 || (mk_bits & (1u << s))) { // one slot: the ballot above already says it is reached
                    float a;
                    const bool ok = ags_alpha(g, (float)AGS_PX(s), (float)AGS_PY(s), dx[s], dy[s], a) && (pos1 <= pg[s].last);
                    al[s] = ok ? a : 0.f;
                }
                any |= al[s] > 0.f;
            }
            if (!__any(any)) continue;
            AgsGeomGrad acc;
            float* a = reinterpret_cast<float*>(&acc);
#pragma unroll
            for (int j = 0; j < 16; ++j) a[j] = 0.f;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s)
                if (SLOTS == 1 || __any(al[s] > 0.f)) // wave-uniform branch; inactive lanes contribute with alpha = 0
                    ags_blend_bwd_apply(pg[s], g, dx[s], dy[s], al[s], acc);
            const float mine = ags_wave_reduce16(a, lane); // every quad ends up with one field's total
            if (my_field >= 0) unsafeAtomicAdd(dgeom + (size_t)st.sid[k] * 16 + my_field, mine);
        }
    }
}

// ---------------------------------------------------------------------------------------
// Blend backward with the per-surfel reduction on the matrix cores (one 8x8 quadrant per wave).
//
// Every one of a surfel's 15 gradient sums is a contraction over the wave's 64 pixels of a per-pixel
// FACTOR the blend recurrence produces (gp = alpha * dL/dalpha, or the blend weight w) with a
// per-pixel FEATURE that does not depend on the surfel once the pixel offsets are taken about the
// quadrant centre (qx, qy) instead of the surfel's mean:
//     fields 0-5  (m1x m1y m2xx m2xy m2yy m0):  sum_p gp[s][p] * {qx, qy, qx^2, qx qy, qy^2, 1}[p]
//     fields 6-14 (ddc dgx dgy dr dg db dn*):   sum_p  w[s][p] * {dDn, dDn qx, dDn qy, dC0..2, dN0..2}[p]
// The wave parks gp and w of the surfels it blends in LDS (row 2 i: gp of slot i, row 2 i + 1: its w;
// [row][pixel], conflict-free both ways) and every 8 surfels runs 16
// v_mfma_f32_16x16x4_f32 (exact f32, an fmaf chain per output) over them: D[row][field] =
// sum_p LDS[row][p] * FEAT[p][field], lane l supplying A = LDS[l & 15][t + 16 (l >> 4)] and B =
// FEAT[t + 16 (l >> 4)][l & 15] (16 registers, fixed for the tile).  The gp rows are meaningful in
// columns 0-5, the w rows in columns 6-14, and the accumulator leaves lane l with column (l & 15)
// of rows 4 (l >> 4) .. +3 = (gp, w) of two slots: 16 consecutive lanes hold the 16 consecutive
// fields of one surfel's gradient record, so after shifting the moments from the quadrant centre
// to the surfel's mean ((qx - ox)^2 = qx^2 - 2 ox qx + ox^2: four in-row lane permutes per slot) ONE
// atomic instruction adds the whole records of four surfels.  This replaces, per surfel and wave, 15 multiply-adds per pixel and the
// 38-instruction transposed wave reduction of ags_k_render_bwd<1>; it pays where tile lists are long.
typedef float ags_f32x4 __attribute__((ext_vector_type(4)));

#ifndef AGS_MFMA_STAGE
#define AGS_MFMA_STAGE 32   // records staged per round: 32 keeps a wave at 6.4 KB of LDS = 6 waves per SIMD
#endif
// The per-surfel reduction of the blend backward in exact f32 (default, BF16 = false) or on the bf16 matrix pipe
// (AgsTuning.bwd_reduce = AGS_BWD_BF16_SPLIT, BF16 = true).
// An f32 matrix instruction (v_mfma_f32_16x16x4_f32, 1024 multiply-adds in ~37 cycles) runs at the vector ALUs' rate and
// does not overlap with other waves' vector instructions on its SIMD (profiles/experiments/mfma_valu_overlap.cpp: a
// VALU stream and an f32-MFMA stream on one SIMD take the SUM of their times): the sixteen of a flush cost as much as
// 32 vector instructions per pixel and surfel, 38 % of the kernel on config 5.  v_mfma_f32_16x16x32_bf16 does 8192 in
// ~28 cycles.  Both operands are split x = hi + lo (bf16 each, by truncation: hi = the upper 16 bits, lo = the upper 16
// bits of x - hi) and the product is formed as hi.hi + lo.hi + hi.lo with f32 accumulation (the dropped lo.lo term is
// 2^-16 relative; every term's truncation error is below 2^-16 of the term): six matrix instructions per flush instead
// of sixteen, four more vector instructions per pixel and surfel for the split.  Gradients move by ~1e-5 relative
// (tolerance 1e-3); the accumulator leaves the matrix pipe in the same layout, so the rest of the flush is unchanged.
// The reference's arithmetic is fp32, so the exact form is what a caller gets unless it asks for the split.
struct AgsWaveBatch {       // one per wave, in LDS: 2048 + 4352 = 6400 B = five 1280-byte granules -> 25 waves per CU
    AgsStagedRec<1> sg[AGS_MFMA_STAGE];
    union {
        struct {
            unsigned short gh[16][68];   // bf16 hi parts: row 2 i = gp of slot i, row 2 i + 1 = its w; [row][pixel], 136-byte rows
            unsigned short gl[16][68];   // bf16 lo parts
        };
        float gw[16][68];   // f32 form: row 2 i: gp of slot i, row 2 i + 1: its w; 68 floats: 16-byte row reads of 16 lanes hit 64 banks
    };
};
// the 8 bytes behind the 64 pixels of batch row `row` (hi rows of the bf16 form / the f32 rows): where a slot's
// (mean - quadrant centre) [even row] and surfel id [odd row] ride along
#define AGS_BWD_PAD(row) (BF16 ? (void*)&wb.gh[(row)][64] : (void*)&wb.gw[(row)][64])
// RED (AgsTuning.bwd_reduce): AGS_BWD_F32, AGS_BWD_BF16_SPLIT, or AGS_BWD_BF16X3 - the THREE-way split x = hi + mid + lo
// (bf16 each by truncation: 8 + 8 + 8 significand bits, so the split of an f32 is EXACT) with the six products of
// relative size >= 2^-16 formed on the bf16 pipe and accumulated in f32: hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi.
// The dropped terms (mid.lo, lo.mid, lo.lo) are below 2^-24 of the product - the size of ONE f32 rounding, of which the
// exact-f32 form makes one per multiply-add.  The factors are parked as f32 (the f32 form's buffer and two stores per
// pair) and split when the flush reads them as the A operand - 16 values per lane and flush either way, but no extra
// LDS stores and no larger buffer.  Twelve matrix instructions per flush (f32 form: sixteen, each ~1.3x as long).
typedef __bf16 ags_bf8 __attribute__((ext_vector_type(8)));
typedef unsigned int ags_u4 __attribute__((ext_vector_type(4)));
// two floats -> one register of two truncated bf16 (lo half = a, hi half = b)
__device__ __forceinline__ uint32_t ags_pack_bf16_trunc(float a, float b) {
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ float ags_bf16_residual(float x) { return x - __uint_as_float(__float_as_uint(x) & 0xFFFF0000u); }

#ifndef AGS_MFMA_WAVES
#define AGS_MFMA_WAVES 6      // register budget: 512 / 6 -> 80 VGPRs
#endif
#ifndef AGS_MFMA_WAVES_X3
#define AGS_MFMA_WAVES_X3 5   // the three-way split keeps 24 registers of B operands (f32 / two-way: 16): 512 / 5 -> 96 VGPRs
#endif
#define AGS_MFMA_WAVES_OF(RED) ((RED) == AGS_BWD_BF16X3 ? AGS_MFMA_WAVES_X3 : AGS_MFMA_WAVES)
// eight consecutive f32 -> their bf16 hi parts, and the residuals in place
__device__ __forceinline__ ags_u4 ags_split8_bf16(float x[8]) {
    const ags_u4 h = {ags_pack_bf16_trunc(x[0], x[1]), ags_pack_bf16_trunc(x[2], x[3]), ags_pack_bf16_trunc(x[4], x[5]),
                      ags_pack_bf16_trunc(x[6], x[7])};
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = ags_bf16_residual(x[k]);
    return h;
}

template <int RED>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(AGS_MFMA_WAVES_OF(RED), AGS_MFMA_WAVES_OF(RED)))) void ags_k_render_bwd_mfma(
    AgsFrame F, int normalize_depth, const float* __restrict__ bgp, const uint2* __restrict__ ranges,
    const uint32_t* __restrict__ vals, int id_stride, const AgsGeom* __restrict__ geom,
    const float* __restrict__ depth_out, const float* __restrict__ opac_out, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, AgsImageGrads dout, float* __restrict__ dgeom, int num_tiles,
    AgsTick tick, uint32_t tile_cap, AgsViewStride vs) {
    {
        const size_t wo = (size_t)blockIdx.y * (size_t)vs.ws, po = (size_t)blockIdx.y * (size_t)vs.px;
        AGS_WS_SHIFT(ranges, wo); AGS_WS_SHIFT(vals, wo); AGS_WS_SHIFT(geom, wo); AGS_WS_SHIFT(final_T, wo);
        AGS_WS_SHIFT(n_contrib, wo); AGS_WS_SHIFT(dgeom, wo);
        depth_out += po; opac_out += po;
        if (dout.d_rgb) dout.d_rgb += 3 * po;
        if (dout.d_normal) dout.d_normal += 3 * po;
        if (dout.d_depth) dout.d_depth += po;
        if (dout.d_opacity) dout.d_opacity += po;
        if (dout.d_confidence) dout.d_confidence += po;
    }
    constexpr int SLOTS = 1;
    constexpr bool BF16 = RED == AGS_BWD_BF16_SPLIT, X3 = RED == AGS_BWD_BF16X3;
    __shared__ AgsWaveBatch wb;
    const int lane = threadIdx.x;
    int slot, wave;
    if (!ags_wave_block(blockIdx.x, num_tiles, 4, slot, wave)) return;
    normalize_depth = ags_cfg_flag(F.cfg, 1, normalize_depth);
    AgsWaveStageQ<1, AGS_MFMA_STAGE, false>& st = *reinterpret_cast<AgsWaveStageQ<1, AGS_MFMA_STAGE, false>*>(&wb.sg[0]);
    static_assert(sizeof(AgsWaveBatch) <= 6400, "the blend backward's LDS per wave: five 1280-byte granules");
    if (tick.clock && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        ags_adam_tick(tick.clock, tick.lr, tick.beta1, tick.beta2);
        if (tick.count_snap) *tick.count_snap = (uint32_t)*tick.rows_count;
    }
    [[maybe_unused]] const int tl_w = slot * 4 + wave;
    AGS_TL(3, tl_w, 0);
    AGS_PRIO_HIGH();
    // direct binning: the slot's first ids are requested before its header (which tile, how long a list) is here
    uint32_t gid_early = 0;
    if (tile_cap && lane < AGS_MFMA_STAGE && (uint32_t)lane < tile_cap)
        gid_early = vals[((size_t)slot * tile_cap + lane) * id_stride];
    uint2 rg;
    const int tile = ags_slot_tile(ranges, slot, tile_cap, rg);
    if (rg.y <= rg.x) return;
    const int tx = tile % F.tiles_x, ty = tile / F.tiles_x;
    const int strip0 = wave;
    const int pxl = tx * AGS_TILE + (lane & 7), pyl = ty * AGS_TILE + (lane >> 3);
    const float bx0 = (float)(tx * AGS_TILE), by0 = (float)(ty * AGS_TILE);
    const uint32_t my_strips = 1u << strip0;
    const float bg[3] = {bgp[0], bgp[1], bgp[2]};
    const size_t HW = (size_t)F.H * F.W;
    // Load order = dependency depth.  A wave's prologue is a chain of dependent loads (ids -> records; n_contrib ->
    // the pixel's gradients) and at short lists the prologue is a third of the wave's life, so everything that can be
    // requested at once is: when the whole list fits ONE staging round (most tiles of a 1200x680 view) the ids are
    // requested before the pixel's inputs and the records as soon as the ids are here - the pixel set-up and the
    // feature exchange below then run while the gather is in flight.  The pixel's inputs are requested together with
    // n_contrib, not behind it.
    const uint32_t list_len = rg.y - rg.x;
    const bool early = AGS_EARLY_GATHER && list_len <= (uint32_t)AGS_MFMA_STAGE;   // wave-uniform
    if (!tile_cap && early && lane < (int)list_len) gid_early = vals[(size_t)(rg.x + lane) * id_stride];
    AgsPixGrad pg;
    AgsRec4 rec_early = {};
    {
        const int px = AGS_PX(0), py = AGS_PY(0);
        float dC[3] = {0, 0, 0}, dN[3] = {0, 0, 0}, dD = 0, dO = 0, dCf = 0, dep = 0, opa = 0, Tf = 1.f;
        uint32_t last = 0;
        if (px < F.W && py < F.H) {
            const size_t o = (size_t)py * F.W + px;
            last = n_contrib[o];
            if (dout.d_rgb) { dC[0] = dout.d_rgb[o]; dC[1] = dout.d_rgb[HW + o]; dC[2] = dout.d_rgb[2 * HW + o]; }
            if (dout.d_normal) { dN[0] = dout.d_normal[o]; dN[1] = dout.d_normal[HW + o]; dN[2] = dout.d_normal[2 * HW + o]; }
            if (dout.d_depth) dD = dout.d_depth[o];
            if (dout.d_opacity) dO = dout.d_opacity[o];
            if (dout.d_confidence) dCf = dout.d_confidence[o];
            dep = depth_out[o]; opa = opac_out[o]; Tf = final_T[o];
        }
        if (early && lane < (int)list_len) rec_early = ags_stage_issue(geom, gid_early);   // needs only the ids
        if (!last) { dC[0] = dC[1] = dC[2] = dN[0] = dN[1] = dN[2] = dD = dO = dCf = dep = opa = 0.f; Tf = 1.f; }
        ags_pixgrad_init(pg, dC, dN, dD, dO, dCf, dep, opa, Tf, last, bg, normalize_depth);
    }
    // park the early records now (not across the feature exchange: sixteen more live registers there spill)
    // the staging lane keeps its surfel's id and (mean - quadrant centre): the batch slots take them with v_readlane
    uint32_t m_early = 0, my_gid = gid_early;
    float2 my_oxy = make_float2(0.f, 0.f);
    if (early && lane < (int)list_len)
        m_early = ags_stage_commit_q<1, AGS_MFMA_STAGE, false>(st, lane, rec_early, gid_early, bx0, by0, strip0, &my_oxy);
    const uint32_t maxlast = ags_wave_max_u32(pg.last);
    AGS_TL(3, tl_w, 1);
    if (maxlast == 0) return; // wave-uniform; no workgroup barrier anywhere in this kernel

    // ---- B operands: field (lane & 15) of the 16 pixels t + 16 (lane >> 4) --------------------------
    const int fld = lane & 15, kgrp = lane >> 4;
    const float qx = (float)(lane & 7) - 3.5f, qy = (float)(lane >> 3) - 3.5f;   // the lane's pixel about the quadrant centre
    // B operands of v_mfma_f32_16x16x32_bf16: lane = (feature fld, k-group kgrp) holds the feature of pixels
    // 32 b + 8 kgrp .. + 7 for the two K blocks b, as bf16 hi and lo parts: 2 x 4 + 2 x 4 registers
    // (f32 form: FE = the feature of pixels t + 16 kgrp, 16 registers; only one of the two sets is live)
    ags_u4 BH[2] = {}, BL[2] = {}, BM[2] = {};
    float FE[16] = {};
    if constexpr (X3) {
        float* ex = &wb.gw[0][0];
        const float feat[16] = {qx, qy, qx * qx, qx * qy, qy * qy, 1.f, pg.dDn, pg.dDn * qx, pg.dDn * qy,
                                pg.dC0, pg.dC1, pg.dC2, pg.dN0, pg.dN1, pg.dN2, 0.f};
#pragma unroll
        for (int f = 0; f < 15; ++f) ex[f * 68 + lane] = feat[f];
        ags_wave_lds_sync();
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float4* row = reinterpret_cast<const float4*>(ex + fld * 68 + 32 * b + 8 * kgrp);
            float4 u = make_float4(0.f, 0.f, 0.f, 0.f), v = u;
            if (fld < 15) { u = row[0]; v = row[1]; }
            float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
            BH[b] = ags_split8_bf16(x);
            BM[b] = ags_split8_bf16(x);
            BL[b] = ags_u4{ags_pack_bf16_trunc(x[0], x[1]), ags_pack_bf16_trunc(x[2], x[3]), ags_pack_bf16_trunc(x[4], x[5]),
                           ags_pack_bf16_trunc(x[6], x[7])};      // (eight significand bits are left: exact)
        }
        ags_wave_lds_sync();
    } else if constexpr (BF16) {
        float* ex = reinterpret_cast<float*>(&wb.gh[0][0]);       // [feature][pixel] f32, rows of 68 floats: 4080 B of the 4352
        const float feat[16] = {qx, qy, qx * qx, qx * qy, qy * qy, 1.f, pg.dDn, pg.dDn * qx, pg.dDn * qy,
                                pg.dC0, pg.dC1, pg.dC2, pg.dN0, pg.dN1, pg.dN2, 0.f};
        static_assert(sizeof(wb.gh) + sizeof(wb.gl) >= (14 * 68 + 64) * 4, "feature exchange fits the batch buffer");
#pragma unroll
        for (int f = 0; f < 15; ++f) ex[f * 68 + lane] = feat[f];
        ags_wave_lds_sync();
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float4* row = reinterpret_cast<const float4*>(ex + fld * 68 + 32 * b + 8 * kgrp);
            float4 u = make_float4(0.f, 0.f, 0.f, 0.f), v = u;
            if (fld < 15) { u = row[0]; v = row[1]; }
            BH[b] = ags_u4{ags_pack_bf16_trunc(u.x, u.y), ags_pack_bf16_trunc(u.z, u.w), ags_pack_bf16_trunc(v.x, v.y),
                           ags_pack_bf16_trunc(v.z, v.w)};
            BL[b] = ags_u4{ags_pack_bf16_trunc(ags_bf16_residual(u.x), ags_bf16_residual(u.y)),
                           ags_pack_bf16_trunc(ags_bf16_residual(u.z), ags_bf16_residual(u.w)),
                           ags_pack_bf16_trunc(ags_bf16_residual(v.x), ags_bf16_residual(v.y)),
                           ags_pack_bf16_trunc(ags_bf16_residual(v.z), ags_bf16_residual(v.w))};
        }
        ags_wave_lds_sync();
    } else {
        // every pixel lane publishes its 16 feature values ([field][pixel], rows of 68 floats: the
        // 16-byte row reads below are conflict-free); the batch buffer is still unused
        float* ex = &wb.gw[0][0];
        const float feat[16] = {qx, qy, qx * qx, qx * qy, qy * qy, 1.f, pg.dDn, pg.dDn * qx, pg.dDn * qy,
                                pg.dC0, pg.dC1, pg.dC2, pg.dN0, pg.dN1, pg.dN2, 0.f};
        static_assert(sizeof(wb.gw) >= (14 * 68 + 64) * 4, "feature exchange fits the batch buffer");
#pragma unroll
        for (int f = 0; f < 15; ++f) ex[f * 68 + lane] = feat[f];
        ags_wave_lds_sync();
        const float4* row = reinterpret_cast<const float4*>(ex + fld * 68 + 16 * kgrp);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (fld < 15) v = row[q];
            FE[4 * q] = v.x; FE[4 * q + 1] = v.y; FE[4 * q + 2] = v.z; FE[4 * q + 3] = v.w;
        }
        ags_wave_lds_sync();
    }
    // per-lane constants of the shift to the surfel's mean (see flush): which of ox, oy and their products
    // this lane's field takes
    const float gx1 = fld == 0 ? 1.f : 0.f, gy1 = fld == 1 ? 1.f : 0.f;                     // gp rows: m1x, m1y
    const float kxx = fld == 2 ? 1.f : 0.f, kxy = fld == 3 ? 1.f : 0.f, kyy = fld == 4 ? 1.f : 0.f;
    const float mYx = fld == 2 ? 2.f : 0.f, mYy = fld == 3 ? 1.f : 0.f, mZx = fld == 3 ? 1.f : 0.f, mZy = fld == 4 ? 2.f : 0.f;
    const float wx1 = fld == 7 ? 1.f : 0.f, wy1 = fld == 8 ? 1.f : 0.f;                     // w rows: dgx, dgy
    int nb = 0; // filled slots (wave-uniform)
    // id and (mean - quadrant centre) of the surfel in batch slot j travel with the slot's parked factors: the staging
    // lane itself writes them into the 8 unused bytes at the end of the slot's two hi rows (one exec-masked 8-byte and
    // one 4-byte LDS write - two vector instructions per pair where copying them into "slot lanes" through v_readlane
    // and selects took ten), and the flush reads them back with the lane's row address
    AGS_TL(3, tl_w, 2);
    [[maybe_unused]] uint32_t tl_iters = 0, tl_flush = 0;
    AGS_PROBE_VARS();

    auto flush = [&]() {
        ags_wave_lds_sync();
        // the lane's two slots (2 kgrp, 2 kgrp + 1): offsets to the mean and ids, requested ahead of the matrix instructions
        float2 slot_oxy[2];
        uint32_t slot_sid[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            slot_oxy[h] = *reinterpret_cast<const float2*>(AGS_BWD_PAD(4 * kgrp + 2 * h));
            slot_sid[h] = *reinterpret_cast<const uint32_t*>(AGS_BWD_PAD(4 * kgrp + 2 * h + 1));
        }
        ags_f32x4 d = {0.f, 0.f, 0.f, 0.f}, d_odd = {0.f, 0.f, 0.f, 0.f}; // two chains: a dependent MFMA waits 40 cycles, an independent one 32
        if constexpr (X3) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            // A operand: row fld, pixels 32 b + 8 kgrp .. + 7 as f32 (two 16-byte reads), split three ways in registers
            const float4* pa = reinterpret_cast<const float4*>(&wb.gw[fld][32 * b + 8 * kgrp]);
            const float4 u = pa[0], v = pa[1];
            float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
            const ags_bf8 ah = __builtin_bit_cast(ags_bf8, ags_split8_bf16(x));
            const ags_bf8 am = __builtin_bit_cast(ags_bf8, ags_split8_bf16(x));
            const ags_bf8 al = __builtin_bit_cast(ags_bf8, ags_u4{ags_pack_bf16_trunc(x[0], x[1]), ags_pack_bf16_trunc(x[2], x[3]),
                                                                  ags_pack_bf16_trunc(x[4], x[5]), ags_pack_bf16_trunc(x[6], x[7])});
            const ags_bf8 bh = __builtin_bit_cast(ags_bf8, BH[b]), bm = __builtin_bit_cast(ags_bf8, BM[b]),
                          bl = __builtin_bit_cast(ags_bf8, BL[b]);
            // the leading products in one chain, the five corrections (<= 2^-8 of them) in the other
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, d, 0, 0, 0);
            d_odd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, d_odd, 0, 0, 0);
            d_odd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, d_odd, 0, 0, 0);
            d_odd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, d_odd, 0, 0, 0);
            d_odd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, d_odd, 0, 0, 0);
            d_odd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, d_odd, 0, 0, 0);
        }
        } else if constexpr (BF16) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            // A operand: row fld, pixels 32 b + 8 kgrp .. + 7 as eight bf16 = two 8-byte reads (136-byte rows are 8-byte aligned)
            const uint2* ph = reinterpret_cast<const uint2*>(&wb.gh[fld][32 * b + 8 * kgrp]);
            const uint2* pl = reinterpret_cast<const uint2*>(&wb.gl[fld][32 * b + 8 * kgrp]);
            const uint2 h0 = ph[0], h1 = ph[1], l0 = pl[0], l1 = pl[1];
            const ags_bf8 ah = __builtin_bit_cast(ags_bf8, ags_u4{h0.x, h0.y, h1.x, h1.y});
            const ags_bf8 al = __builtin_bit_cast(ags_bf8, ags_u4{l0.x, l0.y, l1.x, l1.y});
            const ags_bf8 bh = __builtin_bit_cast(ags_bf8, BH[b]), bl = __builtin_bit_cast(ags_bf8, BL[b]);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, d, 0, 0, 0);
            d_odd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, d_odd, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, d, 0, 0, 0);
        }
        } else {
        const float4* col = reinterpret_cast<const float4*>(&wb.gw[fld][16 * kgrp]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a4 = col[q];
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, FE[4 * q], d, 0, 0, 0);
            d_odd = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, FE[4 * q + 1], d_odd, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, FE[4 * q + 2], d, 0, 0, 0);
            d_odd = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, FE[4 * q + 3], d_odd, 0, 0, 0);
        }
        }
        d += d_odd;
        // lane: field fld of rows 4 kgrp + r = (gp, w) of slots 2 kgrp and 2 kgrp + 1
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int slot = 2 * kgrp + h;
            const float ox = slot_oxy[h].x, oy = slot_oxy[h].y;
            const uint32_t sid = slot_sid[h];
            float* rec = dgeom + (size_t)sid * 16 + fld;
            // gp row: raw moments about the quadrant centre -> about the surfel's mean
            const float gpv = d[2 * h];
            // lane 5 / 0 / 1 of every 16-lane row to the whole row: DPP row_newbcast (one VALU move each; the
            // ds_bpermute they replace occupies the LDS crossbar for 24 cycles per wave-instruction)
            const float R0 = ags_dpp_f<0x150 + 5>(gpv);
            const float R1x = ags_dpp_f<0x150 + 0>(gpv);
            const float R1y = ags_dpp_f<0x150 + 1>(gpv);
            const float c0 = gx1 * ox + gy1 * oy - (kxx * ox * ox + kxy * ox * oy + kyy * oy * oy);
            const float outg = gpv - c0 * R0 - (mYx * ox + mYy * oy) * R1x - (mZx * ox + mZy * oy) * R1y;
            // w row: the depth-slope sums move the same way
            const float wv = d[2 * h + 1];
            const float Q0 = ags_dpp_f<0x150 + 6>(wv);
            const float outw = wv - (wx1 * ox + wy1 * oy) * Q0;
            if (slot < nb && fld < 15) unsafeAtomicAdd(rec, fld < 6 ? outg : outw);
        }
        ags_wave_lds_sync();
        nb = 0;
    };

    constexpr int RSH = AGS_MFMA_STAGE == 64 ? 6 : (AGS_MFMA_STAGE == 32 ? 5 : 4);
    static_assert((1 << RSH) == AGS_MFMA_STAGE, "records staged per round: 16, 32 or 64");
    for (int r = (int)((maxlast - 1) >> RSH); r >= 0; --r) {
        const uint32_t k0 = (uint32_t)r << RSH;
        ags_wave_lds_sync();
        uint32_t m = 0;
        if (early) {                       // the only round (r == 0): the records were parked in the prologue
            if (lane < (int)maxlast) m = m_early;
        } else if (lane < AGS_MFMA_STAGE && k0 + lane < maxlast) {
            my_gid = vals[(size_t)(rg.x + k0 + lane) * id_stride];
            m = ags_stage_commit_q<1, AGS_MFMA_STAGE, false>(st, lane, ags_stage_issue(geom, my_gid), my_gid, bx0, by0, strip0, &my_oxy);
        }
        ags_wave_lds_sync();
        unsigned long long act = __ballot((m & my_strips) != 0u);
        if (r == (int)((maxlast - 1) >> RSH)) AGS_TL(3, tl_w, 3);
        AGS_PRIO_LOOP();
        tl_iters += (uint32_t)__builtin_popcountll(act);
        while (act) { // back to front: highest staged position first
            const int k = 63 - __clzll((long long)act);
            act &= ~(1ull << k);
            const AgsStagedRec<1>& g = st.sg[k];
            const uint32_t pos1 = k0 + k + 1;
            const float4 ga = ags_lds_read16(&g.a), gs = ags_lds_read16(&g.slot[0]);   // E3, E4, E5, gx | E0, E1, E2, D0
            const AgsQuadShared sh = {ga.x, ga.y, ga.z};
            const AgsQuadCoef qc = {gs.x, gs.y, gs.z, gs.w};
            const float a = ags_alpha_quad(sh, qc, qx, qy);
            const bool take = a >= AGS_ALPHA_MIN && pos1 <= pg.last;   // (alpha >= 1/255 > 0 when taken: the mask is the test)
            AGS_PROBE_PAIR(take);
            if (!__any(take)) continue;
            const float alpha = take ? a : 0.f;
            // the blend recurrence (ags_blend_bwd_apply without its accumulation)
            const float iom = ags_rcp(1.f - alpha);
            pg.T = pg.T * iom;
            const float w = alpha * pg.T;
            const float4 gb = ags_lds_read16(&g.b), gc = ags_lds_read16(&g.c);         // gy, r, g, b | nx, ny, nz, conf
            const float dpix = fmaf(gb.x, qy, fmaf(ga.w, qx, gs.w));
            const float gsum = pg.dC0 * gb.y + pg.dC1 * gb.z + pg.dC2 * gb.w + pg.dN0 * gc.x + pg.dN1 * gc.y + pg.dN2 * gc.z
                             + pg.dDn * dpix + pg.dCf * gc.w + pg.dA;
            const float dalpha = pg.T * gsum - pg.S * iom;
            pg.S += w * gsum;
            const float gp = (alpha < AGS_ALPHA_MAX) ? alpha * dalpha : 0.f;
            if constexpr (BF16) {
                // park the two factors as bf16 hi + lo (truncations: the stores take the registers' upper halves)
                wb.gh[2 * nb][lane] = (unsigned short)(__float_as_uint(gp) >> 16);
                wb.gh[2 * nb + 1][lane] = (unsigned short)(__float_as_uint(w) >> 16);
                wb.gl[2 * nb][lane] = (unsigned short)(__float_as_uint(ags_bf16_residual(gp)) >> 16);
                wb.gl[2 * nb + 1][lane] = (unsigned short)(__float_as_uint(ags_bf16_residual(w)) >> 16);
            } else {
                wb.gw[2 * nb][lane] = gp;
                wb.gw[2 * nb + 1][lane] = w;
            }
            if (lane == k) {   // the surfel's staging lane (k is wave-uniform: one exec-masked pair of LDS writes)
                *reinterpret_cast<float2*>(AGS_BWD_PAD(2 * nb)) = my_oxy;
                *reinterpret_cast<uint32_t*>(AGS_BWD_PAD(2 * nb + 1)) = my_gid;
            }
            if (++nb == 8) { flush(); ++tl_flush; }
        }
    }
    AGS_TL(3, tl_w, 4);
    if (nb) { flush(); ++tl_flush; }
    AGS_PROBE_FLUSH(1);
    AGS_TL(3, tl_w, 5);
    AGS_TL_VAL(3, tl_w, 6, tl_iters | ((unsigned long long)tl_flush << 32));
    AGS_TL_VAL(3, tl_w, 7, (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)(__builtin_amdgcn_s_getreg(63508) & 15) << 32));
}

// How many strips per wave: one wave per tile when the image has enough tiles to fill the
// 1024 SIMDs several times over, otherwise split tiles over more waves.
static int ags_pick_slots(int num_tiles, const AgsTuning& tune) {
    if (tune.render_slots) return tune.render_slots;   // the caller's choice (AgsTuning.render_slots: 1 / 2 / 4)
    // measured (re-measured after the r01-k..o slimming of the per-wave overhead): one slot per
    // wave (four waves per tile) wins up to ~11 k tiles in flight (1200x680 = 3225, a training batch of 11 views
    // at 512x512 = 11 264); two slots per wave from 2048x2048 (16 384 tiles, ~500 surfels per tile) up
    if (num_tiles >= 12288) return 2;
    return 1;
}

template <int SLOTS>
static void launch_fwd(const AgsFrame& F, const AgsCamera& cam, char* ws, const AgsLayout& L, AgsIdList ids,
                       const AgsImages& out, const AgsPerGaussian& pg, const AgsViewStride& vs, bool direct, hipStream_t s,
                       const AgsLossFuse* loss) {
    AgsFinalize fin = {nullptr, nullptr, 0u, 1u};
    if (direct) fin = AgsFinalize{(uint32_t*)(ws + L.status), (uint32_t*)(ws + L.totals), ags_direct_tile_cap(L), (uint32_t)L.tc_stride};
    const uint32_t tile_cap = direct ? ags_direct_tile_cap(L) : 0u;
    const uint2* ranges = (const uint2*)(ws + L.ranges);
    const AgsGeom* geom = (const AgsGeom*)(ws + L.geom);
    float* fT = (float*)(ws + L.final_T);
    uint32_t* nc = (uint32_t*)(ws + L.n_contrib);
    const dim3 block(64), grid(8 * ags_wave_blocks_per_xcd(L.num_tiles, 4 / SLOTS), vs.views);
    const int seen_only = (cam.want_stats == AGS_STATS_SEEN && !cam.config) ? 1 : 0;
    if (cam.want_stats || cam.config)   // (device-side configuration: config[3] decides inside the kernel)
        hipLaunchKernelGGL((ags_k_render_fwd<SLOTS, true>), grid, block, 0, s, F, cam.normalize_depth,
                           cam.weight_thres, cam.bg, cam.render_mask, ranges, ids.ids, ids.stride, geom, out, fT, nc,
                           seen_only ? nullptr : pg.importance, pg.count, L.num_tiles, (uint32_t*)(ws + L.tile_count),
                           (uint32_t*)(ws + L.tile_fill), fin, tile_cap, vs, seen_only, AgsLossFuse{});
    else if (loss)
        hipLaunchKernelGGL((ags_k_render_fwd<SLOTS, false, true>), grid, block, 0, s, F, cam.normalize_depth,
                           cam.weight_thres, cam.bg, cam.render_mask, ranges, ids.ids, ids.stride, geom, out, fT, nc,
                           pg.importance, pg.count, L.num_tiles, (uint32_t*)(ws + L.tile_count),
                           (uint32_t*)(ws + L.tile_fill), fin, tile_cap, vs, 0, *loss);
    else
        hipLaunchKernelGGL((ags_k_render_fwd<SLOTS, false>), grid, block, 0, s, F, cam.normalize_depth,
                           cam.weight_thres, cam.bg, cam.render_mask, ranges, ids.ids, ids.stride, geom, out, fT, nc,
                           pg.importance, pg.count, L.num_tiles, (uint32_t*)(ws + L.tile_count),
                           (uint32_t*)(ws + L.tile_fill), fin, tile_cap, vs, 0, AgsLossFuse{});
}

template <int SLOTS>
static void launch_bwd(const AgsFrame& F, const AgsCamera& cam, char* ws, const AgsLayout& L, AgsIdList ids,
                       const AgsImages& fwd, const AgsImageGrads& dout, const AgsTick& tick, uint32_t tile_cap,
                       const AgsViewStride& vs, hipStream_t s) {
    hipLaunchKernelGGL((ags_k_render_bwd<SLOTS>), dim3(8 * ags_wave_blocks_per_xcd(L.num_tiles, 4 / SLOTS), vs.views), dim3(64), 0, s, F,
                       cam.normalize_depth, cam.bg, (const uint2*)(ws + L.ranges), ids.ids, ids.stride,
                       (const AgsGeom*)(ws + L.geom), fwd.depth, fwd.opacity, (const float*)(ws + L.final_T),
                       (const uint32_t*)(ws + L.n_contrib), dout, (float*)(ws + L.dgeom), L.num_tiles, tick, tile_cap, vs);
}

void ags_launch_render_fwd(const AgsFrame& F, const AgsCamera& cam, char* ws, const AgsLayout& L,
                           AgsIdList ids, const AgsImages& out, const AgsPerGaussian& pg, const AgsViewStride& vs,
                           bool direct, hipStream_t s, const AgsLossFuse* loss) {
    // strips per wave by the number of tiles in flight: a batch of views fills the GPU like one big image
    switch (ags_pick_slots(L.num_tiles * vs.views, L.tune)) {
        case 1: launch_fwd<1>(F, cam, ws, L, ids, out, pg, vs, direct, s, loss); break;
        case 2: launch_fwd<2>(F, cam, ws, L, ids, out, pg, vs, direct, s, loss); break;
        default: launch_fwd<4>(F, cam, ws, L, ids, out, pg, vs, direct, s, loss); break;
    }
}

void ags_launch_render_bwd(const AgsFrame& F, const AgsCamera& cam, char* ws, const AgsLayout& L,
                           AgsIdList ids, const AgsImages& fwd, const AgsImageGrads& dout, const AgsTick& tick,
                           const AgsViewStride& vs, bool direct, hipStream_t s) {
    const uint32_t tile_cap = direct ? ags_direct_tile_cap(L) : 0u;
    // AgsTuning.bwd_reduce: exact f32 matrix instructions (default), the bf16 hi/lo split, or no matrix instructions
    if (L.tune.bwd_reduce != AGS_BWD_VALU) {
#define AGS_LAUNCH_BWD_MFMA(RED)                                                                                          \
    hipLaunchKernelGGL(ags_k_render_bwd_mfma<RED>, dim3(8 * ags_wave_blocks_per_xcd(L.num_tiles, 4), vs.views), dim3(64), 0, s, F, \
                       cam.normalize_depth, cam.bg, (const uint2*)(ws + L.ranges), ids.ids, ids.stride,                    \
                       (const AgsGeom*)(ws + L.geom), fwd.depth, fwd.opacity, (const float*)(ws + L.final_T),             \
                       (const uint32_t*)(ws + L.n_contrib), dout, (float*)(ws + L.dgeom), L.num_tiles, tick, tile_cap, vs)
        if (L.tune.bwd_reduce == AGS_BWD_BF16_SPLIT) AGS_LAUNCH_BWD_MFMA(AGS_BWD_BF16_SPLIT);
        else if (L.tune.bwd_reduce == AGS_BWD_BF16X3) AGS_LAUNCH_BWD_MFMA(AGS_BWD_BF16X3);
        else AGS_LAUNCH_BWD_MFMA(AGS_BWD_F32);
#undef AGS_LAUNCH_BWD_MFMA
        return;
    }
    const int slots = ags_pick_slots(L.num_tiles * vs.views, L.tune);
    switch (slots) {
        case 1: launch_bwd<1>(F, cam, ws, L, ids, fwd, dout, tick, tile_cap, vs, s); break;
        case 2: launch_bwd<2>(F, cam, ws, L, ids, fwd, dout, tick, tile_cap, vs, s); break;
        default: launch_bwd<4>(F, cam, ws, L, ids, fwd, dout, tick, tile_cap, vs, s); break;
    }
}
