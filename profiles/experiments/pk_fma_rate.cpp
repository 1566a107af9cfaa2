// Issue rate of v_pk_fma_f32 against v_fma_f32 on gfx950: N waves per SIMD run a dependent-free stream of one or the other;
// reports cycles per wave-instruction per SIMD.  hipcc --offload-arch=gfx950 -O2 pk_fma_rate.cpp -o pk_fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int PK>
__global__ void k(float* out, int iters) {
    f2 a0 = {1.f, 2.f}, a1 = {3.f, 4.f}, a2 = {5.f, 6.f}, a3 = {7.f, 8.f}, a4 = {1.5f, 2.5f}, a5 = {3.5f, 4.5f}, a6 = {5.5f, 6.5f}, a7 = {7.5f, 8.5f};
    const f2 m = {1.0000001f, 0.9999999f}, c = {1e-9f, -1e-9f};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (PK) {
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x), "v"(c.x));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + (float)(t1 - t0) * 0.f;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}
int main() {
    float* d; hipMalloc(&d, 1 << 24);
    const int iters = 20000;
    for (int waves = 1; waves <= 8; waves *= 2) {
        for (int pk = 0; pk < 2; ++pk) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            // 256 CUs x 4 SIMDs x `waves` waves: one workgroup of 64 x waves x 4 threads per CU
            dim3 grid(256), block(64 * 4 * waves);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                if (pk) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, d, iters); else hipLaunchKernelGGL(k<0>, grid, block, 0, 0, d, iters);
                hipEventRecord(b); hipEventSynchronize(b);
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            // wave-instructions per SIMD = waves * iters * 8
            printf("%s waves/SIMD %d: %.3f ms -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ",
                   waves, ms, ms * 1e6 / (waves * (double)iters * 8), ms * 1e6 / (waves * (double)iters * 8) * 2.4);
        }
    }
    return 0;
}
