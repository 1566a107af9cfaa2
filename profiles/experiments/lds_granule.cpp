// How many 64-thread workgroups does a CU hold as a function of the LDS each one allocates?  (occupancy API + a
// census kernel: the API is known to be off by one near edges - MI355X_MICROARCH.md)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(int* census, int* maxres, int spin) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = 1.f;
    if (threadIdx.x == 0) {
        const int cu = blockIdx.x;  // not the real CU; census via atomic high-water mark per XCC/SE/CU
        unsigned hw = __builtin_amdgcn_s_getreg(63492), xcc = __builtin_amdgcn_s_getreg(63508) & 15;
        const int key = (((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 15));
        const int now = atomicAdd(&census[key], 1) + 1;
        atomicMax(&maxres[key], now);
        for (volatile int i = 0; i < spin; ++i) { }
        atomicAdd(&census[key], -1);
        (void)cu;
    }
}
int main() {
    int *census, *maxres; hipMalloc(&census, 4096 * 4); hipMalloc(&maxres, 4096 * 4);
    for (int lds = 5120; lds <= 7424; lds += 128) {
        int api = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, k, 64, lds);
        hipMemset(census, 0, 4096 * 4); hipMemset(maxres, 0, 4096 * 4);
        k<<<256 * 40, 64, lds>>>(census, maxres, 20000);
        hipDeviceSynchronize();
        int h[4096]; hipMemcpy(h, maxres, sizeof(h), hipMemcpyDeviceToHost);
        int mx = 0; for (int i = 0; i < 4096; ++i) mx = h[i] > mx ? h[i] : mx;
        printf("lds %5d B: API %2d blocks/CU, census max %2d resident per CU\n", lds, api, mx);
    }
    return 0;
}
