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
