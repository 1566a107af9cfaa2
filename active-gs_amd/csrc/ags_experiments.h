// Instrumentation for EXPERIMENT builds of libags_raster.so (profiles/experiments/build_exp.py passes the -D flags).
// The product build defines none of them: every macro below then expands to nothing (AGS_TL*), to the shipped setting
// (AGS_PRIO_*, AGS_EARLY_GATHER), and no experiment code is compiled into the product's translation units.  Builds that
// computed wrong results on purpose to price a part of a kernel (the blend backward without its flush / matrix
// instructions / atomics, the per-Gaussian kernels without their emission) are not kept in the sources: what they
// measured is in DESIGN.md sections 5 and 9.
#pragma once

// ---- experiment builds only (-DAGS_TIMELINE, profiles/experiments/timeline.py): every wave notes the shader clock
// (s_memtime) at a few phase boundaries into a caller-provided buffer [kernel][wave][8]; compiled out otherwise.
#ifndef AGS_TL_WAVES
#define AGS_TL_WAVES 16384
#endif
// -DAGS_TL_REALTIME: stamps from the chip-wide 100 MHz reference counter (s_memrealtime: 10 ns steps, the same on
// every CU) instead of the shader clock, which every CU counts on its own: for launch ramps and kernel-to-kernel gaps
#ifdef AGS_TL_REALTIME
#define AGS_TL_CLOCK() __builtin_amdgcn_s_memrealtime()
#else
#define AGS_TL_CLOCK() __builtin_readcyclecounter()
#endif
#if defined(AGS_TIMELINE) && defined(__HIPCC__)
#define AGS_TL_DEFINE(tu)                                                                                          \
    static __device__ unsigned long long* ags_tl_buf = nullptr;                                                    \
    void ags_tl_set_##tu(void* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(ags_tl_buf), &p, sizeof(p)); }
#define AGS_TL(kid, wave_id, phase)                                                                                \
    do {                                                                                                           \
        if ((threadIdx.x & 63) == 0 && ags_tl_buf && (unsigned)(wave_id) < AGS_TL_WAVES)                           \
            ags_tl_buf[(((size_t)(kid) * AGS_TL_WAVES) + (wave_id)) * 8 + (phase)] = AGS_TL_CLOCK();               \
    } while (0)
#define AGS_TL_VAL(kid, wave_id, phase, v)                                                                         \
    do {                                                                                                           \
        if ((threadIdx.x & 63) == 0 && ags_tl_buf && (unsigned)(wave_id) < AGS_TL_WAVES)                           \
            ags_tl_buf[(((size_t)(kid) * AGS_TL_WAVES) + (wave_id)) * 8 + (phase)] = (unsigned long long)(v);      \
    } while (0)
#else
#define AGS_TL_DEFINE(tu)
#define AGS_TL(kid, wave_id, phase) do { } while (0)
#define AGS_TL_VAL(kid, wave_id, phase, v) do { } while (0)
#endif
void ags_tl_set_preprocess(void*); void ags_tl_set_binning(void*); void ags_tl_set_render(void*);

// ---- experiment builds only (-DAGS_PROBE_LANES, profiles/experiments/lane_occupancy.py): how full the blend loops' waves
// are.  Per kernel (0 = forward, 1 = backward) eight 64-bit sums over all waves of a launch: [0] (surfel, wave) pairs that
// entered the loop body, [1] pairs with at least one pixel taken, [2] pixels taken, [3] 4x4 pixel blocks (of the wave's four)
// with a pixel taken, [4] 8x2 row pairs (16 consecutive lanes, of four) with a pixel taken, [5] waves.
#if defined(AGS_PROBE_LANES) && defined(__HIPCC__)
#define AGS_PROBE_DEFINE()                                                                                         \
    static __device__ unsigned long long ags_probe_sums[2][8];                                                     \
    extern "C" int ags_probe_read(unsigned long long* out16, int reset) {                                           \
        if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(ags_probe_sums), sizeof(unsigned long long) * 16) != hipSuccess) return -1; \
        if (reset) { unsigned long long z[16] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(ags_probe_sums), z, sizeof(z)); }    \
        return 0;                                                                                                  \
    }
#define AGS_PROBE_VARS() unsigned int pr_pairs = 0, pr_any = 0, pr_px = 0, pr_b44 = 0, pr_b82 = 0
#define AGS_PROBE_PAIR(take)                                                                                       \
    do {                                                                                                           \
        const unsigned long long bm = __ballot(take);                                                              \
        ++pr_pairs; pr_any += bm != 0ull; pr_px += (unsigned)__builtin_popcountll(bm);                             \
        /* lane = 8 y + x: the 4x4 block (y >> 2, x >> 2) */                                                       \
        const unsigned long long left = 0x0F0F0F0Full, top = 0xFFFFFFFFull;                                       \
        pr_b44 += ((bm & left) != 0) + ((bm & (left << 4)) != 0) + ((bm & (left << 32)) != 0) + ((bm & (left << 36)) != 0); \
        (void)top;                                                                                                 \
        pr_b82 += ((bm & 0xFFFFull) != 0) + ((bm & (0xFFFFull << 16)) != 0) + ((bm & (0xFFFFull << 32)) != 0) + ((bm & (0xFFFFull << 48)) != 0); \
    } while (0)
#define AGS_PROBE_FLUSH(kid)                                                                                       \
    do {                                                                                                           \
        if ((threadIdx.x & 63) == 0) {                                                                             \
            atomicAdd(&ags_probe_sums[kid][0], (unsigned long long)pr_pairs); atomicAdd(&ags_probe_sums[kid][1], (unsigned long long)pr_any); \
            atomicAdd(&ags_probe_sums[kid][2], (unsigned long long)pr_px); atomicAdd(&ags_probe_sums[kid][3], (unsigned long long)pr_b44);   \
            atomicAdd(&ags_probe_sums[kid][4], (unsigned long long)pr_b82); atomicAdd(&ags_probe_sums[kid][5], 1ull);                        \
        }                                                                                                          \
    } while (0)
#else
#define AGS_PROBE_DEFINE()
#define AGS_PROBE_VARS() do { } while (0)
#define AGS_PROBE_PAIR(take) do { } while (0)
#define AGS_PROBE_FLUSH(kid) do { } while (0)
#endif

// Issue priority by phase of the blend kernels (render.hip).  The SIMD arbitrates VALU issue between its resident waves
// by priority, then AGE: a wave that has just started competes with older waves that sit in their blend loops and keep
// the vector pipe busy - its short prologue crawls, its loads go out late, and the same happens to the few instructions
// in front of its final stores.  Prologue and epilogue run at raised priority, the blend loop at the default.
// -DAGS_EXP_NO_PRIO: without (measured neutral to +0.7 % step).
#ifdef AGS_EXP_NO_PRIO
#define AGS_PRIO_HIGH() do { } while (0)
#define AGS_PRIO_LOOP() do { } while (0)
#else
#define AGS_PRIO_HIGH() __builtin_amdgcn_s_setprio(3)
#define AGS_PRIO_LOOP() __builtin_amdgcn_s_setprio(0)
#endif
// -DAGS_EXP_NO_EARLY: the blend backward's prologue as it was (ids and records requested behind the pixel set-up)
#ifdef AGS_EXP_NO_EARLY
#define AGS_EARLY_GATHER false
#else
#define AGS_EARLY_GATHER true
#endif
