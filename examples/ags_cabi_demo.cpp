// Torch-free use of the C ABI (include/ags_raster.h): synthetic surfels on a wall, one view,
// forward + backward + fused Adam captured ONCE into a hipGraph and replayed.
// Build: hipcc --offload-arch=gfx950 -O2 -Iinclude examples/ags_cabi_demo.cpp -Lactive-gs_amd/lib -lags_raster
// Prints: instances, ms per replayed step, a checksum of the image and of the parameters.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ags_raster.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define CHECK_AGS(x) do { int c_ = (x); if (c_ != AGS_OK) { fprintf(stderr, "AGS error %s at %s:%d\n", ags_error_string(c_), __FILE__, __LINE__); exit(3); } } while (0)

template <typename T>
static T* dev_copy(const std::vector<T>& h) {
    T* d = nullptr;
    CHECK_HIP(hipMalloc(&d, h.size() * sizeof(T)));
    CHECK_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}
template <typename T>
static T* dev_zero(size_t n) {
    T* d = nullptr;
    CHECK_HIP(hipMalloc(&d, n * sizeof(T)));
    CHECK_HIP(hipMemset(d, 0, n * sizeof(T)));
    return d;
}
static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); }

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 50000, H = 340, W = 600, steps = argc > 2 ? atoi(argv[2]) : 200;
    unsigned seed = 12345u;
    std::vector<float> means(3 * n), scales(3 * n), rots(4 * n), opac(n), col(3 * n), conf(n);
    for (int i = 0; i < n; ++i) {  // RAW parameters of surfels on the wall z = 2, facing the camera
        means[3 * i] = (frand(seed) - 0.5f) * 4.f; means[3 * i + 1] = (frand(seed) - 0.5f) * 2.4f; means[3 * i + 2] = 2.f;
        scales[3 * i] = logf(0.5f + 2.5f * frand(seed)); scales[3 * i + 1] = logf(0.5f + 2.5f * frand(seed)); scales[3 * i + 2] = -1e10f;
        rots[4 * i] = 0.f; rots[4 * i + 1] = 1.f; rots[4 * i + 2] = 0.05f * (frand(seed) - 0.5f); rots[4 * i + 3] = 0.f; // R_x(180): normal -z
        opac[i] = 3.f * (frand(seed) - 0.5f);
        for (int k = 0; k < 3; ++k) col[3 * i + k] = frand(seed);
        conf[i] = frand(seed);
    }
    const float tanx = 1.0f, tany = tanx * H / W, nearp = 0.001f, farp = 10.f;
    std::vector<float> view = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};     // camera at the origin, +z forward
    std::vector<float> proj = {1 / tanx, 0, 0, 0, 0, 1 / tany, 0, 0, 0, 0, farp / (farp - nearp), 1, 0, 0, -farp * nearp / (farp - nearp), 0};
    std::vector<float> bg = {0.f, 0.f, 0.f, 0.f};
    float *d_means = dev_copy(means), *d_scales = dev_copy(scales), *d_rots = dev_copy(rots), *d_opac = dev_copy(opac),
          *d_col = dev_copy(col), *d_conf = dev_copy(conf), *d_view = dev_copy(view), *d_proj = dev_copy(proj), *d_bg = dev_copy(bg);
    const size_t P = (size_t)H * W;
    AgsImages img = {dev_zero<float>(3 * P), dev_zero<float>(3 * P), dev_zero<float>(P), dev_zero<float>(P), dev_zero<float>(P)};
    AgsPerGaussian pg = {dev_zero<float>(n), dev_zero<int32_t>(n), dev_zero<int32_t>(n)};
    std::vector<float> hd(3 * P);
    for (auto& x : hd) x = (frand(seed) - 0.5f) / (float)P;
    float* d_drgb = dev_copy(hd);
    AgsImageGrads dimg = {d_drgb, nullptr, nullptr, nullptr, nullptr};
    AgsGaussianGrads grads = {};
    grads.d_means3D = dev_zero<float>(3 * n); grads.d_scales = dev_zero<float>(3 * n); grads.d_rotations = dev_zero<float>(4 * n);
    grads.d_opacities = dev_zero<float>(n); grads.d_colors = dev_zero<float>(3 * n); grads.d_means2D = nullptr; grads.accumulate = 0;
    AgsAdamTensors adam = {};
    float* params[5] = {d_means, d_scales, d_rots, d_opac, d_col};
    const float* gptr[5] = {grads.d_means3D, grads.d_scales, grads.d_rotations, grads.d_opacities, grads.d_colors};
    const int64_t numel[5] = {3LL * n, 3LL * n, 4LL * n, n, 3LL * n};
    const float lrs[5] = {5e-4f, 1e-2f, 5e-4f, 1e-2f, 1e-4f};
    for (int k = 0; k < 5; ++k) {
        adam.param[k] = params[k]; adam.grad[k] = gptr[k]; adam.numel[k] = numel[k]; adam.lr[k] = lrs[k];
        adam.exp_avg[k] = dev_zero<float>(numel[k]); adam.exp_avg_sq[k] = dev_zero<float>(numel[k]);
    }
    void* clock = dev_zero<char>(64);
    grads.adam_clock = clock; grads.adam_beta1 = 0.9f; grads.adam_beta2 = 0.999f;
    for (int k = 0; k < 5; ++k) grads.adam_lr[k] = lrs[k];

    AgsCamera cam = {H, W, tanx, tany, 1.0f, 0.03f, 1, 1, 0, 0, d_view, d_proj, d_bg, nullptr};
    AgsGaussians g = {n, d_means, d_scales, d_rots, d_opac, d_col, d_conf, /*raw_params*/ 1, 0.01f, 0.05f};
    AgsWorkspace ws = {nullptr, 0, 4LL * n + 65536, AGS_BIN_TILE_SORT};
    ws.bytes = ags_workspace_bytes(n, H, W, ws.max_instances);
    CHECK_HIP(hipMalloc(&ws.ptr, ws.bytes));
    hipStream_t s;
    CHECK_HIP(hipStreamCreate(&s));
    CHECK_AGS(ags_workspace_init(&ws, n, H, W, s));

    auto step = [&]() {
        CHECK_AGS(ags_forward(&cam, &g, &img, &pg, &ws, s));
        CHECK_AGS(ags_backward(&cam, &g, &img, &pg, &dimg, &grads, &ws, s));
        CHECK_AGS(ags_adam_step_device(&adam, 0.9f, 0.999f, 1e-15f, clock, /*pre_ticked*/ 1, s));
    };
    step();  // eager once: shows the library works without capture
    AgsStatus st;
    CHECK_AGS(ags_read_status(&ws, &st, s));
    if (st.overflow) { fprintf(stderr, "workspace overflow: need %u instances\n", st.num_instances); return 4; }

    hipGraph_t graph; hipGraphExec_t exec;
    CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    step();
    CHECK_HIP(hipStreamEndCapture(s, &graph));
    CHECK_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int i = 0; i < 20; ++i) CHECK_HIP(hipGraphLaunch(exec, s));
    CHECK_HIP(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    CHECK_HIP(hipEventCreate(&e0)); CHECK_HIP(hipEventCreate(&e1));
    CHECK_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < steps; ++i) CHECK_HIP(hipGraphLaunch(exec, s));
    CHECK_HIP(hipEventRecord(e1, s));
    CHECK_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));

    std::vector<float> rgb(3 * P), m2(3 * n);
    int32_t clk[16];
    CHECK_HIP(hipMemcpy(rgb.data(), img.rgb, rgb.size() * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(m2.data(), d_means, m2.size() * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(clk, clock, 64, hipMemcpyDeviceToHost));
    double sum_rgb = 0, moved = 0;
    for (float x : rgb) sum_rgb += x;
    for (size_t i = 0; i < m2.size(); ++i) moved += fabs((double)m2[i] - means[i]);
    printf("ags_version=%d n=%d image=%dx%d visible=%u instances=%u adam_steps=%d ms_per_step=%.4f gaussians_per_s=%.3e mean_rgb=%.6f mean_abs_move=%.3e\n",
           ags_version(), n, W, H, st.num_visible, st.num_instances, clk[0], ms / steps, n / (ms / steps * 1e-3), sum_rgb / rgb.size(), moved / m2.size());
    const bool ok = st.num_visible > 0 && std::isfinite(sum_rgb) && sum_rgb > 0 && moved > 0 && clk[0] == steps + 21;
    printf(ok ? "OK\n" : "FAILED\n");
    return ok ? 0 : 1;
}
