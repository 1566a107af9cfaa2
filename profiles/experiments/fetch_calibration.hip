// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of this library's kernels.
// Every kernel moves a KNOWN number of bytes over a working set far beyond the 256 MiB Infinity Cache (sources of
// 1 GiB), so that cache hits cannot hide requests:
//   read_stream16     64 lanes x 16 B coalesced (the per-Gaussian kernels' parameter rows, the images)
//   read_stream4      64 lanes x 4 B coalesced (id lists, radii, one image channel row by row)
//   read_stream8      64 lanes x 8 B coalesced (the (depth | id) keys)
//   read_rows32       a wave reads an 8x8-pixel quadrant of a 2048-wide f32 image: eight 32-byte row segments (the blend
//                     kernels' image reads: n_contrib, final_T, the image gradients)
//   read_gather64     one 64-byte record per 4 lanes at a RANDOM record index (the blend kernels' AgsGeom gather)
//   read_gather64_1l  one 64-byte record per LANE (four 16-byte loads: ags_k_preprocess_bwd_rows' record reads)
//   read_gather16     16 B per lane at a random index
//   read_gather4      4 B per lane at a random index (radii / member flags by row id)
//   write_stream16    64 lanes x 16 B coalesced
//   write_scatter64   one 64-byte record per 4 lanes at a random index (geometry records of visible rows)
//   write_scatter8    8 B per lane at a random index (the (depth | id) keys of one-pass binning)
//   write_scatter4    4 B per lane at a random index
//   atomic_noret4     one float atomicAdd without return per lane at a random index (gradient records: 16 lanes = 64 B)
//   atomic_noret64    16 consecutive lanes add into ONE 64-byte record at a random index (the blend backward's flush)
//   atomic_ret4       one returning uint atomicAdd per lane at a random counter (the slot atomics of one-pass binning)
// Build: hipcc --offload-arch=gfx950 -O3 fetch_calibration.hip -o fetch_calibration
// Run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- ./fetch_calibration     (and again with WRITE_SIZE)
// The program prints the algorithmic bytes of every kernel as JSON; fetch_calibration_summary.py divides.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

__global__ void read_stream16(const float4* __restrict__ src, size_t n16, float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) sink[0] = acc;
}

__global__ void read_stream4(const float* __restrict__ src, size_t n4, float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
    if (acc == 123.456f) sink[0] = acc;
}

__global__ void read_stream8(const float2* __restrict__ src, size_t n8, float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const float2 v = src[i];
        acc += v.x + v.y;
    }
    if (acc == 123.456f) sink[0] = acc;
}

// the image as rows of 2048 floats; wave q reads quadrant q: pixel (8 (q % 256) + (lane & 7), 8 (q / 256) + (lane >> 3))
__global__ void read_rows32(const float* __restrict__ src, size_t quadrants, float* __restrict__ sink) {
    float acc = 0.f;
    const int lane = threadIdx.x & 63;
    const size_t wave0 = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t q = wave0; q < quadrants; q += nw) {
        const size_t x = 8 * (q % 256) + (lane & 7), y = 8 * (q / 256) + (lane >> 3);
        acc += src[y * 2048 + x];
    }
    if (acc == 123.456f) sink[0] = acc;
}

__global__ void read_gather64(const float4* __restrict__ src, uint32_t records, size_t touches, float* __restrict__ sink) {
    float acc = 0.f;   // 4 lanes share a record: lane sub reads its 16 bytes
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < touches * 4; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t rec = hash32((uint32_t)(i >> 2)) % records;
        const float4 v = src[(size_t)rec * 4 + (i & 3)];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) sink[0] = acc;
}

__global__ void read_gather64_1l(const float4* __restrict__ src, uint32_t records, size_t touches, float* __restrict__ sink) {
    float acc = 0.f;   // one lane reads a whole record (four 16-byte loads)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < touches; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t rec = hash32((uint32_t)i) % records;
        const float4* p = src + (size_t)rec * 4;
        const float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 123.456f) sink[0] = acc;
}

__global__ void read_gather16(const float4* __restrict__ src, uint32_t n16, size_t touches, float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < touches; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = src[hash32((uint32_t)i) % n16];
        acc += v.x + v.w;
    }
    if (acc == 123.456f) sink[0] = acc;
}

__global__ void read_gather4(const float* __restrict__ src, uint32_t n4, size_t touches, float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < touches; i += (size_t)gridDim.x * blockDim.x)
        acc += src[hash32((uint32_t)i) % n4];
    if (acc == 123.456f) sink[0] = acc;
}

__global__ void write_stream16(float4* __restrict__ dst, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}

__global__ void write_scatter64(float4* __restrict__ dst, uint32_t records, size_t touches) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < touches * 4; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t rec = hash32((uint32_t)(i >> 2)) % records;
        dst[(size_t)rec * 4 + (i & 3)] = make_float4(1.f, 2.f, 3.f, (float)i);
    }
}

__global__ void write_scatter8(uint64_t* __restrict__ dst, uint32_t n8, size_t touches) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < touches; i += (size_t)gridDim.x * blockDim.x)
        dst[hash32((uint32_t)i) % n8] = i;
}

__global__ void write_scatter4(uint32_t* __restrict__ dst, uint32_t n4, size_t touches) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < touches; i += (size_t)gridDim.x * blockDim.x)
        dst[hash32((uint32_t)i) % n4] = (uint32_t)i;
}

__global__ void atomic_noret4(float* __restrict__ dst, uint32_t n4, size_t touches) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < touches; i += (size_t)gridDim.x * blockDim.x)
        unsafeAtomicAdd(dst + hash32((uint32_t)i) % n4, 1.0f);
}

__global__ void atomic_noret64(float* __restrict__ dst, uint32_t records, size_t touches) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < touches * 16; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t rec = hash32((uint32_t)(i >> 4)) % records;
        unsafeAtomicAdd(dst + (size_t)rec * 16 + (i & 15), 1.0f);
    }
}

__global__ void atomic_ret4(uint32_t* __restrict__ dst, uint32_t n4, size_t touches, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < touches; i += (size_t)gridDim.x * blockDim.x)
        acc += atomicAdd(dst + hash32((uint32_t)i) % n4, 1u);
    if (acc == 0xFFFFFFF3u) sink[0] = acc;
}

int main() {
    const size_t BYTES = (size_t)1 << 30;            // 1 GiB source / destination: 4x the Infinity Cache
    const size_t TOUCH = (size_t)1 << 24;            // random touches per gather / scatter kernel (16.8 M)
    char *a = nullptr, *b = nullptr;
    float* sink = nullptr;
    CHECK(hipMalloc(&a, BYTES)); CHECK(hipMalloc(&b, BYTES)); CHECK(hipMalloc(&sink, 256));
    CHECK(hipMemset(a, 0, BYTES)); CHECK(hipMemset(b, 0, BYTES));
    CHECK(hipDeviceSynchronize());
    const dim3 grid(256 * 16), block(256);
    const int REP = 3;
    for (int r = 0; r < REP; ++r) {
        hipLaunchKernelGGL(read_stream16, grid, block, 0, 0, (const float4*)a, BYTES / 16, sink);
        hipLaunchKernelGGL(read_stream4, grid, block, 0, 0, (const float*)a, BYTES / 4, sink);
        hipLaunchKernelGGL(read_stream8, grid, block, 0, 0, (const float2*)a, BYTES / 8, sink);
        hipLaunchKernelGGL(read_rows32, grid, block, 0, 0, (const float*)a, BYTES / 256, sink);
        hipLaunchKernelGGL(read_gather64, grid, block, 0, 0, (const float4*)a, (uint32_t)(BYTES / 64), TOUCH, sink);
        hipLaunchKernelGGL(read_gather64_1l, grid, block, 0, 0, (const float4*)a, (uint32_t)(BYTES / 64), TOUCH, sink);
        hipLaunchKernelGGL(read_gather16, grid, block, 0, 0, (const float4*)a, (uint32_t)(BYTES / 16), TOUCH, sink);
        hipLaunchKernelGGL(read_gather4, grid, block, 0, 0, (const float*)a, (uint32_t)(BYTES / 4), TOUCH, sink);
        hipLaunchKernelGGL(write_stream16, grid, block, 0, 0, (float4*)b, BYTES / 16);
        hipLaunchKernelGGL(write_scatter64, grid, block, 0, 0, (float4*)b, (uint32_t)(BYTES / 64), TOUCH);
        hipLaunchKernelGGL(write_scatter8, grid, block, 0, 0, (uint64_t*)b, (uint32_t)(BYTES / 8), TOUCH);
        hipLaunchKernelGGL(write_scatter4, grid, block, 0, 0, (uint32_t*)b, (uint32_t)(BYTES / 4), TOUCH);
        hipLaunchKernelGGL(atomic_noret4, grid, block, 0, 0, (float*)b, (uint32_t)(BYTES / 4), TOUCH);
        hipLaunchKernelGGL(atomic_noret64, grid, block, 0, 0, (float*)b, (uint32_t)(BYTES / 64), TOUCH);
        hipLaunchKernelGGL(atomic_ret4, grid, block, 0, 0, (uint32_t*)b, (uint32_t)(BYTES / 4), TOUCH, (uint32_t*)sink);
        CHECK(hipDeviceSynchronize());
    }
    // algorithmic bytes per launch: {read, write}
    printf("{\"read_stream4\": [%zu, 0], \"read_stream8\": [%zu, 0], \"read_rows32\": [%zu, 0], ", BYTES, BYTES, BYTES);
    printf("\"read_stream16\": [%zu, 0], \"read_gather64\": [%zu, 0], \"read_gather64_1l\": [%zu, 0], \"read_gather16\": [%zu, 0], "
           "\"read_gather4\": [%zu, 0], \"write_stream16\": [0, %zu], \"write_scatter64\": [0, %zu], \"write_scatter8\": [0, %zu], "
           "\"write_scatter4\": [0, %zu], \"atomic_noret4\": [%zu, %zu], \"atomic_noret64\": [%zu, %zu], \"atomic_ret4\": [%zu, %zu], "
           "\"_touches\": %zu, \"_bytes\": %zu}\n",
           BYTES, TOUCH * 64, TOUCH * 64, TOUCH * 16, TOUCH * 4, BYTES, TOUCH * 64, TOUCH * 8, TOUCH * 4,
           TOUCH * 4, TOUCH * 4, TOUCH * 64, TOUCH * 64, TOUCH * 4, TOUCH * 4, TOUCH, BYTES);
    return 0;
}
