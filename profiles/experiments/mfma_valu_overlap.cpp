// Do matrix instructions and vector instructions of DIFFERENT waves on one SIMD overlap?  Half the waves of every SIMD
// run a VALU FMA stream, the other half a matrix stream (f32 16x16x4, or bf16 16x16x32); each stream is also timed alone.
// If the pipes are separate the mixed run takes max(alone_a, alone_b); if the matrix instruction occupies the vector
// ALUs it takes the sum.   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.cpp -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
#define N_ITER 2048
template <int MODE>   // 0: all VALU, 1: all f32 MFMA, 2: all bf16 MFMA, 3: even waves VALU / odd f32 MFMA, 4: even VALU / odd bf16 MFMA
__global__ __launch_bounds__(64) void k(float* out) {
    const bool valu = MODE == 0 || ((MODE == 3 || MODE == 4) && (blockIdx.x & 1) == 0);
    const bool f32m = MODE == 1 || (MODE == 3 && (blockIdx.x & 1));
    float a[8]; f4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0}, d2 = {0, 0, 0, 0}, d3 = {0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) a[i] = 1.f + i + threadIdx.x;
    bf8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(float)(threadIdx.x + i); y[i] = (__bf16)(float)(i + 1); }
    if (valu) {
#pragma unroll 1
        for (int it = 0; it < N_ITER; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);      // 8 VALU per iteration
    } else if (f32m) {
#pragma unroll 1
        for (int it = 0; it < N_ITER / 8; ++it) {                                          // 4 MFMA (32 cyc each) per iteration
            d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], a[1], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], a[3], d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4], a[5], d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[6], a[7], d3, 0, 0, 0);
        }
    } else {
#pragma unroll 1
        for (int it = 0; it < N_ITER / 8; ++it) {
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, d3, 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s + d0[0] + d1[1] + d2[2] + d3[3];
}
template <int MODE>
float run(int blocks) {
    float* out; hipMalloc(&out, (size_t)blocks * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 64>>>(out); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks, 64>>>(out); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); hipFree(out); return ms * 1e3f;
}
int main() {
    const int blocks = 256 * 4 * 4;   // 4 waves per SIMD (mixed modes: 2 + 2)
    printf("4 waves/SIMD, us:  all VALU %.1f | all f32-MFMA %.1f | all bf16-MFMA %.1f | VALU+f32-MFMA mixed %.1f | VALU+bf16-MFMA mixed %.1f\n",
           run<0>(blocks), run<1>(blocks), run<2>(blocks), run<3>(blocks), run<4>(blocks));
    printf("half the waves alone (2 waves/SIMD), us:  VALU %.1f | f32-MFMA %.1f | bf16-MFMA %.1f\n", run<0>(blocks / 2), run<1>(blocks / 2), run<2>(blocks / 2));
    return 0;
}
