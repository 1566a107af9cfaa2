// VALU issue-rate calibration on gfx950: how many wave64 VALU instructions per cycle does ONE SIMD retire with
// 1..8 resident waves, for plain fp32 FMA, v_exp_f32 / v_rcp_f32 (transcendental), DPP moves, v_readlane and
// ds_bpermute?  Answers what "VALU busy" means for the blend kernels (bench.py roofline_valu).
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 valu_rate.cpp -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITER 4096
template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, float seed, long long* cyc) {
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
            if (KIND == 1) a[i] = __builtin_amdgcn_exp2f(a[i]);
            if (KIND == 2) a[i] = __builtin_amdgcn_rcpf(a[i]);
            if (KIND == 3) a[i] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0xB1, 0xF, 0xF, true)) + 1.f;
            if (KIND == 4) a[i] += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a[i]), 5));
            if (KIND == 5) a[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((threadIdx.x ^ 1) << 2, __builtin_bit_cast(int, a[i])));
            if (KIND == 6) { asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*(double*)&a[i & ~1])); }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND>
void run(const char* name, int insts_per_elem) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 4 * 8 * 64 * 4); hipMalloc(&cyc, 256 * 4 * 8 * 8);
    printf("%-22s", name);
    for (int wps : {1, 2, 4, 6, 8}) {               // waves per SIMD: blocks of one wave, 4 * wps per CU
        const int blocks = 256 * 4 * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<KIND><<<blocks, 64>>>(out, 1.f, cyc);      // warm
        hipEventRecord(e0); k<KIND><<<blocks, 64>>>(out, 1.f, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto c : h) avg += c; avg /= blocks;
        const double insts = (double)N_ITER * 8 * insts_per_elem;       // per wave
        // s_memtime / readcyclecounter ticks at 100 MHz on gfx9: report wall-based rate instead
        const double per_simd_per_us = insts * wps / (ms * 1e3);          // wave-insts per SIMD per us
        printf("  wps=%d: %.0f inst/us/SIMD (%.2f cyc/inst @2.4GHz)", wps, per_simd_per_us, 2400.0 / per_simd_per_us);
    }
    printf("\n");
}
int main() {
    run<0>("v_fma_f32", 1); run<6>("v_pk_fma_f32", 1); run<1>("v_exp_f32", 1); run<2>("v_rcp_f32", 1);
    run<3>("v_mov_dpp + v_add", 2); run<4>("v_readlane + v_add", 2); run<5>("ds_bpermute", 1);
    return 0;
}
