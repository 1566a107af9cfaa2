// Host side of the drop-in module `diff_gaussian_rasterization_2d` as native code: the autograd node that sits between
// an unmodified caller (/root/reference/utils/operations.py:682-713: GaussianRasterizer(settings)(**9 kwargs), then
// total_loss.backward(), mapping/gaussian_map.py:125) and the C ABI of libags_raster.so (include/ags_raster.h).
//
// Why native: per view the GPU work is 60-90 us; a Python autograd.Function spent ~90 us per forward and, because the
// autograd engine runs backward nodes on its own device thread, ~90 us per backward waiting for the interpreter lock -
// the GPU idled behind the host (profiles/r03_a_prof_dropin_*.txt).  Here a forward is a few allocations, one struct
// fill and three launches, a backward needs no interpreter at all.  torch types appear ONLY in this file (device
// memory and autograd plumbing); the library is bound through its C ABI, resolved with dlsym from the path the Python
// side loaded (so AGS_LIB_PATH builds are honoured).
//
// What a call does (same contract as the CUDA extension's _RasterizeGaussians):
//   * inputs are used in place when they are contiguous float32 on the GPU (the reference's always are);
//   * `config` (operations.py:697-699) stays on the device: the kernels read its four flags there (AgsCamera.config);
//   * outputs are fresh tensors; the view's workspace (projected records, keys, per-pixel blend state, gradient
//     records) comes from a pool keyed by (device, image size, binning mode) - a slab laid out for another map size is
//     re-initialised if it is large enough - and goes back when the autograd node is destroyed (the graph is dropped);
//   * the workspace check.  DEFAULT (round 4): every call reads the status block back before it returns and repairs an
//     overflow on the spot - what the CUDA extension's num_rendered read-back does on every call; an unmodified caller
//     (mapping/mapper.py:98-104 has no retry anywhere above the rasterizer) can never see truncated tile lists or an
//     exception for a legitimately larger view.  OPT-IN (`deferred`: the per-call flag of rasterize(), which
//     facade.SurfelRenderer sets around its own loops and settles per batch, or AGS_DROPIN_STATUS=deferred for callers that
//     call check_overflow() every iteration): a call that finds a pooled workspace (2x the largest need seen so far)
//     copies the status block to page-locked memory without waiting and a LATER call / settle() / check_overflow() looks
//     at it - no stream synchronisation per view.  Passes without grad are always checked at once unless the per-call flag
//     says otherwise (their callers read the result on the host anyway).
//   * one-pass binning needs tiles x the LONGEST tile list of key slots; a view size whose lists are badly skewed is
//     moved to the scan-based binning (same images), which needs the instance total.
#include <torch/custom_class.h>
#include <torch/extension.h>

#include <c10/hip/HIPStream.h>
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <deque>
#include <map>
#include <mutex>
#include <sstream>
#include <tuple>
#include <vector>

#include "ags_raster.h"

namespace {

// ---- the C ABI, resolved at init() from the library the Python side loaded
struct Abi {
    void* handle = nullptr;
    decltype(&ags_workspace_bytes) workspace_bytes = nullptr;
    decltype(&ags_workspace_init) workspace_init = nullptr;
    decltype(&ags_forward) forward = nullptr;
    decltype(&ags_backward) backward = nullptr;
    decltype(&ags_read_status) read_status = nullptr;
    decltype(&ags_read_status_async) read_status_async = nullptr;
    decltype(&ags_error_string) error_string = nullptr;
} abi;

void check_rc(int rc, const char* what) {
    TORCH_CHECK(rc == AGS_OK, what, " failed: ", abi.error_string ? abi.error_string(rc) : "?", " (", rc, ")");
}
#define HIP_OK(expr) TORCH_CHECK((expr) == hipSuccess, "HIP call failed: " #expr)

// ---- options / counters / sizing state (process-wide, guarded by one mutex: forward runs on the caller's thread,
// the node's release on the autograd engine's)
struct Options {
    int binning_mode = AGS_BIN_DIRECT;
    double skew_factor = 8.0;            // direct binning is left when it needs this many times the instance total ...
    double direct_budget_bytes = double(1ull << 30);   // ... AND more than this much workspace for its key slots (24 B each)
    bool always_check = true;            // status read-back after every forward pass (the CUDA extension's behaviour): safe for
                                         // callers that never look at check_overflow(); false = deferred checks (opt-in)
    double headroom = 2.0;               // a new workspace holds this many times the largest need seen so far
    double min_headroom = 1.25;          // a pooled workspace is reused while it holds at least this many times that need
    int max_pending = 64;
    int pool_max_per_key = 16;
} opt;
struct Counters { int64_t forward_calls = 0, status_syncs = 0, deferred_checks = 0, overflows = 0, mode_switches = 0, early_waits = 0,
                  repaired = 0; } cnt;

using SizeKey = std::tuple<int, int, int>;           // device, h, w
using NeedKey = std::tuple<int, int, int, int>;      // device, h, w, mode
using PoolKey = std::tuple<int, int, int, int>;      // device, h, w, mode
// a pooled slab remembers the layout it was initialised for (surfels, key slots): a map that has grown or shrunk since
// re-initialises a slab that is large enough instead of leaving one set of workspaces per map size behind
struct Pooled { at::Tensor ws; int64_t cap; int n; };
std::mutex mu;
AgsTuning tuning{};                  // kernel selection handed to the library with every workspace (set_tuning: the Python side
                                     // fills it from AGS_* environment variables; the library itself reads none)
std::map<NeedKey, int64_t> need_seen;
std::map<SizeKey, int> mode_for;
std::map<PoolKey, std::vector<Pooled>> pool;
constexpr int64_t kU32 = 0xFFFFFFFFll;

struct Slot { AgsStatus* host = nullptr; hipEvent_t ev = nullptr; };
struct Pending { Slot slot; SizeKey key; int mode; int n; int64_t cap; };
std::vector<Slot> free_slots;
std::deque<Pending> pending;
std::vector<std::string> overflow_reports;

Slot take_slot() {
    if (!free_slots.empty()) { Slot s = free_slots.back(); free_slots.pop_back(); return s; }
    Slot s;
    HIP_OK(hipHostMalloc((void**)&s.host, sizeof(AgsStatus), hipHostMallocDefault));
    HIP_OK(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
    return s;
}

// fold one view's status into the sizing state; returns a description if the view overflowed (mu held)
std::string note_need(const SizeKey& key, int mode, int n, const AgsStatus& st) {
    const int64_t instances = st.num_instances, needed = st.needed_instances;
    const NeedKey nk{std::get<0>(key), std::get<1>(key), std::get<2>(key), mode};
    int64_t& seen = need_seen[nk];
    if (needed > seen) seen = needed;
    if (mode == AGS_BIN_DIRECT && double(needed) > opt.skew_factor * double(std::max<int64_t>(instances, 1 << 16)) &&
        double(needed) * 24.0 > opt.direct_budget_bytes) {
        // skewed tile lists: this view size goes on with the scan-based binning, sized by the instance total
        mode_for[key] = AGS_BIN_TILE_SORT;
        cnt.mode_switches++;
        int64_t& ts = need_seen[NeedKey{std::get<0>(key), std::get<1>(key), std::get<2>(key), AGS_BIN_TILE_SORT}];
        if (instances > ts) ts = instances;
    }
    if (!st.overflow) return {};
    std::ostringstream os;
    os << "a " << std::get<2>(key) << "x" << std::get<1>(key) << " view of " << n << " surfels needed " << needed
       << " tile-instance slots (" << instances << " instances, binning mode " << mode << ")";
    return os.str();
}

// look at the status copies that have landed (all of them if `block`); returns the overflow reports gathered so far
// (and forgets them): empty = every deferred check so far was fine
std::vector<std::string> collect_pending(bool block, bool take = true) {
    std::vector<std::string> out;
    std::lock_guard<std::mutex> g(mu);
    while (!pending.empty()) {
        Pending& p = pending.front();
        hipError_t q = hipEventQuery(p.slot.ev);
        if (q == hipErrorNotReady) {
            if (!block && (int)pending.size() <= opt.max_pending) break;
            HIP_OK(hipEventSynchronize(p.slot.ev));
        } else {
            TORCH_CHECK(q == hipSuccess, "hipEventQuery failed");
        }
        cnt.deferred_checks++;
        const std::string what = note_need(p.key, p.mode, p.n, *p.slot.host);
        if (!what.empty()) {
            cnt.overflows++;
            overflow_reports.push_back(what + " but its workspace held " + std::to_string(p.cap));
        }
        free_slots.push_back(p.slot);
        pending.pop_front();
    }
    if (take) out.swap(overflow_reports);
    return out;
}
// ... and throws if one of them reports an overflow
void poll_pending(bool block) {
    const std::vector<std::string> reports = collect_pending(block);
    std::string msg;
    for (size_t i = 0; i < reports.size(); ++i) msg += (i ? "; " : "") + reports[i];
    TORCH_CHECK(msg.empty(), "diff_gaussian_rasterization_2d: ", msg,
                ": the tile lists of that DEFERRED-check call were truncated, its images and gradients are invalid.  The "
                "workspace size has been raised - repeat the iteration (deferred checks are opt-in: the default checks "
                "and repairs every call before it returns).");
}

// a pooled workspace laid out for n surfels with at least min_cap key slots; else a pooled slab that is large enough,
// re-initialised for (n, cap); else a new one of cap; -> (tensor, slots, newly made or re-initialised)
std::tuple<at::Tensor, int64_t, bool> take_workspace(const PoolKey& pk, int n, int h, int w, int64_t min_cap, int64_t cap,
                                                    const at::TensorOptions& bytes_opt, int mode, hipStream_t stream) {
    const size_t bytes = abi.workspace_bytes(n, h, w, cap);
    at::Tensor reuse;
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = pool.find(pk);
        if (it != pool.end()) {
            auto& v = it->second;
            for (size_t k = v.size(); k-- > 0;) {          // newest first
                if (v[k].n == n && v[k].cap >= min_cap) {
                    Pooled p = std::move(v[k]);
                    v.erase(v.begin() + k);
                    return {p.ws, p.cap, false};
                }
            }
            for (size_t k = v.size(); k-- > 0;) {
                if ((size_t)v[k].ws.numel() >= bytes) { reuse = std::move(v[k].ws); v.erase(v.begin() + k); break; }
            }
            if (!reuse.defined()) v.clear();   // every pooled slab is too small for this map: the need only grows, let them go
        }
    }
    at::Tensor ws = reuse.defined() ? reuse : at::empty({(int64_t)bytes}, bytes_opt);
    AgsWorkspace wss{ws.data_ptr(), (size_t)ws.numel(), cap, mode, &tuning};
    check_rc(abi.workspace_init(&wss, n, h, w, stream), "ags_workspace_init");
    return {ws, cap, true};
}

// returns its workspace to the pool when the autograd node lets go of its saved state
struct Lease : torch::CustomClassHolder {
    PoolKey key; at::Tensor ws; int64_t cap; int n;
    Lease(PoolKey k, at::Tensor t, int64_t c, int n_) : key(k), ws(std::move(t)), cap(c), n(n_) {}
    ~Lease() override {
        std::lock_guard<std::mutex> g(mu);
        auto& v = pool[key];
        if ((int)v.size() >= opt.pool_max_per_key) v.erase(v.begin());
        v.push_back(Pooled{std::move(ws), cap, n});
    }
};

// (an IValue can only carry a REGISTERED custom class)
static const auto kLeaseRegistration = torch::class_<Lease>("ags_raster", "WorkspaceLease");

at::Tensor f32c(const at::Tensor& t, const c10::Device& dev, const char* name) {
    at::Tensor x = t;
    if (x.device() != dev) {
        TORCH_CHECK(!x.is_cuda() && x.numel() <= 16, "diff_gaussian_rasterization_2d (MI355X build): ", name,
                    " must live on the GPU of means3D (", dev, "), got ", x.device(), "; there is no CPU fallback");
        x = x.to(dev);   // tiny host-side settings tensors (bg, ...) are moved; big ones are an error
    }
    if (x.scalar_type() != at::kFloat) x = x.to(at::kFloat);
    return x.is_contiguous() ? x : x.contiguous();
}

struct RasterizeFn : public torch::autograd::Function<RasterizeFn> {
    static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, const at::Tensor& means3D,
                                                  const at::Tensor& means2D, const at::Tensor& opacities,
                                                  const at::Tensor& confidences, const at::Tensor& colors,
                                                  const at::Tensor& scales, const at::Tensor& rotations, const at::Tensor& bg_in,
                                                  const at::Tensor& view_in, const at::Tensor& proj_in,
                                                  const at::Tensor& mask_in, const at::Tensor& config_in, int64_t h, int64_t w,
                                                  double tanx, double tany, double scale_mod, double weight_thres,
                                                  int64_t grad_flags /* bit 0: a backward may follow; bit 1: means2D wants its gradient;
                                                                        bit 2: the caller settles deferred checks itself (settle()) */) {
        TORCH_CHECK(means3D.is_cuda(), "diff_gaussian_rasterization_2d (MI355X build): tensors must be on the GPU; "
                                       "there is no CPU fallback");
        TORCH_CHECK(abi.forward, "the rasterizer library is not loaded (torch_binding.init)");
        const c10::Device dev = means3D.device();
        const c10::DeviceGuard guard(dev);
        // non-blocking look at earlier calls' status copies (sizes learnt); an overflow among them is raised here unless
        // this caller settles its passes itself (bit 2: the reports wait for settle())
        if (grad_flags & 4) collect_pending(false, false); else poll_pending(false);
        hipStream_t stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
        const at::Tensor V = f32c(view_in, dev, "viewmatrix"), P = f32c(proj_in, dev, "projmatrix"), bg = f32c(bg_in, dev, "bg");
        at::Tensor mask, cfg;
        if (mask_in.defined() && mask_in.numel() > 0) {
            mask = f32c(mask_in, dev, "render_mask");
            TORCH_CHECK(mask.numel() == h * w, "render_mask must hold image_height*image_width values");
        }
        int flags[4] = {1, 1, 0, 0};               // normalize_depth, perpix_depth, want_stats, front_only
        if (config_in.defined() && config_in.numel() > 0) {
            TORCH_CHECK(config_in.numel() >= 5, "config must hold 5 values");
            if (config_in.is_cuda()) {
                cfg = f32c(config_in, dev, "config");      // stays on the device: the kernels read the flags there
            } else {
                const at::Tensor c = config_in.to(at::kFloat).contiguous();
                const float* p = c.data_ptr<float>();
                for (int k = 0; k < 4; ++k) flags[k] = p[k + 1] > 0.f ? 1 : 0;
            }
        }
        const at::Tensor m3 = f32c(means3D, dev, "means3D"), sc = f32c(scales, dev, "scales"), rot = f32c(rotations, dev, "rotations"),
                         op = f32c(opacities, dev, "opacities").reshape({-1}), col = f32c(colors, dev, "colors_precomp"),
                         conf = f32c(confidences, dev, "confidences").reshape({-1});
        const int64_t n = m3.size(0);
        TORCH_CHECK(m3.dim() == 2 && m3.size(1) == 3 && sc.numel() == 3 * n && rot.numel() == 4 * n && op.numel() == n &&
                    col.numel() == 3 * n && conf.numel() == n, "diff_gaussian_rasterization_2d: inconsistent input shapes");
        TORCH_CHECK(n <= 0x7FFFFFFF && h > 0 && w > 0, "diff_gaussian_rasterization_2d: bad sizes");
        AgsCamera cs{(int32_t)h, (int32_t)w, (float)tanx, (float)tany, (float)scale_mod, (float)weight_thres, flags[0], flags[1],
                     flags[2], flags[3], V.data_ptr<float>(), P.data_ptr<float>(), bg.data_ptr<float>(),
                     mask.defined() ? mask.data_ptr<float>() : nullptr, cfg.defined() ? cfg.data_ptr<float>() : nullptr};
        AgsGaussians gs{(int32_t)n, n ? m3.data_ptr<float>() : nullptr, n ? sc.data_ptr<float>() : nullptr,
                        n ? rot.data_ptr<float>() : nullptr, n ? op.data_ptr<float>() : nullptr,
                        n ? col.data_ptr<float>() : nullptr, n ? conf.data_ptr<float>() : nullptr, 0, 0.01f, 0.05f};
        const auto fo = m3.options();
        const auto io = fo.dtype(at::kInt);
        at::Tensor rgb = at::empty({3, h, w}, fo), normal = at::empty({3, h, w}, fo), depth = at::empty({1, h, w}, fo),
                   opacity = at::empty({1, h, w}, fo), confidence = at::empty({1, h, w}, fo), radii = at::empty({n}, io);
        // device-side flags: the per-Gaussian kernel clears the statistics itself; host flags: zero-filled here
        at::Tensor importance = cfg.defined() ? at::empty({n}, fo) : at::zeros({n}, fo);
        at::Tensor count = cfg.defined() ? at::empty({n}, io) : at::zeros({n}, io);
        AgsImages im{rgb.data_ptr<float>(), normal.data_ptr<float>(), depth.data_ptr<float>(), opacity.data_ptr<float>(),
                     confidence.data_ptr<float>()};
        AgsPerGaussian pg{n ? importance.data_ptr<float>() : nullptr, n ? count.data_ptr<int32_t>() : nullptr,
                          n ? radii.data_ptr<int32_t>() : nullptr, AgsRowSet{nullptr, nullptr, nullptr}};
        const SizeKey key{dev.index(), (int)h, (int)w};
        bool must_sync;
        // checked before it returns: by default every call; with deferred checks switched on (option / per-call flag)
        // still every pass without grad unless the caller itself asked to defer (it settles the batch)
        { std::lock_guard<std::mutex> g(mu); cnt.forward_calls++;
          must_sync = (grad_flags & 4) ? false : (opt.always_check || !(grad_flags & 1)); }
        at::Tensor ws;
        int64_t ws_cap = 0;
        int mode = 0;
        PoolKey pk;
        for (int attempts = 0;; ++attempts) {
            int64_t seen;
            {
                std::lock_guard<std::mutex> g(mu);
                auto mf = mode_for.find(key);
                mode = mf == mode_for.end() ? opt.binning_mode : mf->second;
                auto ns = need_seen.find(NeedKey{dev.index(), (int)h, (int)w, mode});
                seen = ns == need_seen.end() ? 0 : ns->second;
            }
            const int64_t floor_cap = std::max<int64_t>(1 << 16, 2 * n);
            const int64_t min_cap = std::min(std::max<int64_t>((int64_t)(seen * opt.min_headroom) + 1024, floor_cap), kU32);
            const int64_t new_cap = std::min(std::max<int64_t>((int64_t)(seen * opt.headroom) + 1024, floor_cap), kU32);
            pk = PoolKey{dev.index(), (int)h, (int)w, mode};
            bool fresh;
            std::tie(ws, ws_cap, fresh) = take_workspace(pk, (int)n, (int)h, (int)w, min_cap, new_cap, fo.dtype(at::kByte), mode, stream);
            AgsWorkspace wss{ws.data_ptr(), (size_t)ws.numel(), ws_cap, mode, &tuning, nullptr, nullptr};
            // A call that must be checked before it returns does not wait for the whole pass where it need not: with
            // one-pass binning the per-Gaussian kernel takes the key slots, so whether a tile's list outgrew its range is
            // known behind THAT kernel.  ags_forward copies the status block to page-locked memory and records an event
            // right there (AgsWorkspace.early_status_*); the tile sort and the blend are already queued when this thread
            // waits for the event - it waits for one kernel, not for three, and the stream never drains.
            const bool early = must_sync && mode == AGS_BIN_DIRECT && n > 0;
            Slot eslot;
            if (early) {
                { std::lock_guard<std::mutex> g(mu); eslot = take_slot(); }
                wss.early_status_host = eslot.host; wss.early_status_event = eslot.ev;
            }
            // host-side flags asking for statistics: importance / count were zero-filled once above, and an abandoned
            // (truncated) pass still ADDS into them - a repeat starts from zeros again, ordered behind that pass on the stream
            // (with device-side flags the per-Gaussian kernel clears them itself)
            const bool host_stats = !cfg.defined() && flags[2] && n > 0;
            if (host_stats && attempts > 0) { importance.zero_(); count.zero_(); }
            const int rc_fwd = abi.forward(&cs, &gs, &im, &pg, &wss, stream);
            if (rc_fwd != 0 && early) { std::lock_guard<std::mutex> g(mu); free_slots.push_back(eslot); }   // the slot goes back before the throw
            check_rc(rc_fwd, "ags_forward");
            if (early) {
                HIP_OK(hipEventSynchronize(eslot.ev));
                const uint32_t longest = eslot.host->early_tile_need;
                const int64_t tiles = ((h + 15) / 16) * ((w + 15) / 16);
                std::string over;
                bool mode_left = false;
                {
                    std::lock_guard<std::mutex> g(mu);
                    cnt.early_waits++;
                    free_slots.push_back(eslot);
                    if (longest) {       // a tile's list outgrew its slots: tiles x the longest list would have held the pass
                        AgsStatus st{};
                        st.overflow = 1;
                        st.needed_instances = (uint32_t)std::min<int64_t>((int64_t)longest * tiles, kU32);
                        st.max_tile_instances = longest;
                        over = note_need(key, mode, (int)n, st);
                        cnt.repaired++;
                        auto mf = mode_for.find(key);
                        mode_left = (mf == mode_for.end() ? opt.binning_mode : mf->second) != mode;
                    }
                }
                if (longest) {
                    // repaired here: the pass that is still running is abandoned (its outputs are overwritten by the
                    // repeat, which the stream orders behind it) and re-run in a workspace of the size just learnt
                    TORCH_CHECK(attempts < 4 && (ws_cap < kU32 || mode_left), "diff_gaussian_rasterization_2d: ", over,
                                " - more than a workspace can hold");
                    continue;
                }
                // fits.  What the pass needed in full (for sizing the next workspaces) arrives with the deferred copy below.
            } else if (fresh || must_sync) {
                // (scan-based binning modes, or a workspace that had to be made while checks are deferred): read the need
                // back behind the whole pass, like upstream's num_rendered read-back
                AgsStatus st;
                check_rc(abi.read_status(&wss, &st, stream), "ags_read_status");
                std::string over;
                bool mode_left;
                {
                    std::lock_guard<std::mutex> g(mu);
                    cnt.status_syncs++;
                    over = note_need(key, mode, (int)n, st);
                    auto mf = mode_for.find(key);
                    mode_left = (mf == mode_for.end() ? opt.binning_mode : mf->second) != mode;
                }
                if (over.empty()) break;
                // repaired here: re-run (checked again) in a workspace of the size just learnt; this one is dropped
                must_sync = true;
                TORCH_CHECK(attempts < 4 && (ws_cap < kU32 || mode_left), "diff_gaussian_rasterization_2d: ", over,
                            " - more than a workspace can hold");
                continue;
            }
            Slot slot;
            { std::lock_guard<std::mutex> g(mu); slot = take_slot(); }
            check_rc(abi.read_status_async(&wss, slot.host, stream), "ags_read_status_async");
            HIP_OK(hipEventRecord(slot.ev, stream));
            { std::lock_guard<std::mutex> g(mu); pending.push_back(Pending{slot, key, mode, (int)n, ws_cap}); }
            break;
        }
        if (grad_flags & 1) {
            // everything the raw pointers of the backward call point into is saved with the node; the lease hands the
            // workspace back to the pool when the node releases its saved state
            ctx->save_for_backward({depth, opacity, radii, m3, sc, rot, op, col, conf, V, P, bg,
                                    mask.defined() ? mask : at::Tensor(), cfg.defined() ? cfg : at::Tensor()});
            ctx->saved_data["lease"] = c10::make_intrusive<Lease>(pk, ws, ws_cap, (int)n);
            ctx->saved_data["scalars"] = std::vector<double>{double(h), double(w), tanx, tany, scale_mod, weight_thres,
                                                             double(flags[0]), double(flags[1]), double(flags[2]), double(flags[3]),
                                                             double(ws_cap), double(mode)};
            ctx->saved_data["need_m2d"] = (grad_flags & 2) != 0;
            ctx->saved_data["opac_shape"] = opacities.sizes().vec();
        } else {
            std::lock_guard<std::mutex> g(mu);
            auto& v = pool[pk];
            if ((int)v.size() >= opt.pool_max_per_key) v.erase(v.begin());
            v.push_back(Pooled{ws, ws_cap, (int)n});
        }
        ctx->set_materialize_grads(false);
        ctx->mark_non_differentiable({importance, count, radii});
        return {rgb, normal, depth, opacity, confidence, importance, count, radii};
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list g) {
        const auto saved = ctx->get_saved_variables();
        const at::Tensor &depth = saved[0], &opacity = saved[1], &radii = saved[2], &m3 = saved[3], &sc = saved[4], &rot = saved[5],
                         &op = saved[6], &col = saved[7], &conf = saved[8], &V = saved[9], &P = saved[10], &bg = saved[11],
                         &mask = saved[12], &cfg = saved[13];
        const std::vector<double> s = ctx->saved_data["scalars"].toDoubleVector();
        const auto lease = ctx->saved_data["lease"].toCustomClass<Lease>();
        const bool need_m2d = ctx->saved_data["need_m2d"].toBool();
        const std::vector<int64_t> opac_shape = ctx->saved_data["opac_shape"].toIntVector();
        const c10::Device dev = m3.device();
        const c10::DeviceGuard guard(dev);
        hipStream_t stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
        const int64_t n = m3.size(0);
        AgsCamera cs{(int32_t)s[0], (int32_t)s[1], (float)s[2], (float)s[3], (float)s[4], (float)s[5], (int32_t)s[6], (int32_t)s[7],
                     (int32_t)s[8], (int32_t)s[9], V.data_ptr<float>(), P.data_ptr<float>(), bg.data_ptr<float>(),
                     mask.defined() ? mask.data_ptr<float>() : nullptr, cfg.defined() ? cfg.data_ptr<float>() : nullptr};
        AgsGaussians gs{(int32_t)n, n ? m3.data_ptr<float>() : nullptr, n ? sc.data_ptr<float>() : nullptr,
                        n ? rot.data_ptr<float>() : nullptr, n ? op.data_ptr<float>() : nullptr,
                        n ? col.data_ptr<float>() : nullptr, n ? conf.data_ptr<float>() : nullptr, 0, 0.01f, 0.05f};
        const auto fo = m3.options();
        at::Tensor g_m = at::empty({n, 3}, fo), g_s = at::empty({n, 3}, fo), g_r = at::empty({n, 4}, fo), g_o = at::empty({n}, fo),
                   g_c = at::empty({n, 3}, fo), g_m2;
        if (need_m2d) g_m2 = at::empty({n, 3}, fo);
        at::Tensor d[5];
        for (int k = 0; k < 5; ++k)
            if (g[k].defined()) d[k] = f32c(g[k], dev, "image gradient");
        auto fp = [](const at::Tensor& t) -> const float* { return t.defined() && t.numel() ? t.data_ptr<float>() : nullptr; };
        AgsImageGrads dout{fp(d[0]), fp(d[1]), fp(d[2]), fp(d[3]), fp(d[4])};
        AgsGaussianGrads din{};
        din.d_means3D = n ? g_m.data_ptr<float>() : nullptr; din.d_scales = n ? g_s.data_ptr<float>() : nullptr;
        din.d_rotations = n ? g_r.data_ptr<float>() : nullptr; din.d_opacities = n ? g_o.data_ptr<float>() : nullptr;
        din.d_colors = n ? g_c.data_ptr<float>() : nullptr; din.d_means2D = (need_m2d && n) ? g_m2.data_ptr<float>() : nullptr;
        din.accumulate = 0;
        // (the blend backward reads the forward's depth and opacity images, its per-pixel state in the workspace and radii)
        AgsImages im{nullptr, nullptr, depth.data_ptr<float>(), opacity.data_ptr<float>(), nullptr};
        AgsPerGaussian pg{nullptr, nullptr, n ? radii.data_ptr<int32_t>() : nullptr, AgsRowSet{nullptr, nullptr, nullptr}};
        AgsWorkspace wss{lease->ws.data_ptr(), (size_t)lease->ws.numel(), (int64_t)s[10], (int32_t)s[11], &tuning};
        check_rc(abi.backward(&cs, &gs, &im, &pg, &dout, &din, &wss, stream), "ags_backward");
        at::Tensor none;
        return {g_m, need_m2d ? g_m2 : none, g_o.reshape(opac_shape), none, g_c, g_s, g_r,
                none, none, none, none, none, none, none, none, none, none, none, none};
    }
};

std::vector<at::Tensor> rasterize(const at::Tensor& means3D, const at::Tensor& means2D, const at::Tensor& opacities,
                                  const at::Tensor& confidences, const at::Tensor& colors, const at::Tensor& scales,
                                  const at::Tensor& rotations, const at::Tensor& bg, const at::Tensor& viewmatrix,
                                  const at::Tensor& projmatrix, const at::Tensor& render_mask, const at::Tensor& config, int64_t h,
                                  int64_t w, double tanx, double tany, double scale_mod, double weight_thres, bool deferred) {
    // (inside apply() grad mode is off and the node may not exist: what the forward needs to know is decided here)
    int64_t grad_flags = 0;
    if (at::GradMode::is_enabled()) {
        const bool any = means3D.requires_grad() || (means2D.defined() && means2D.requires_grad()) || opacities.requires_grad() ||
                         colors.requires_grad() || scales.requires_grad() || rotations.requires_grad();
        grad_flags = (any ? 1 : 0) | ((means2D.defined() && means2D.numel() && means2D.requires_grad()) ? 2 : 0);
    }
    if (deferred) grad_flags |= 4;
    return RasterizeFn::apply(means3D, means2D, opacities, confidences, colors, scales, rotations, bg, viewmatrix, projmatrix,
                              render_mask, config, h, w, tanx, tany, scale_mod, weight_thres, grad_flags);
}

void init(const std::string& path) {
    void* hnd = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
    TORCH_CHECK(hnd, "cannot load ", path, ": ", dlerror(), " - there is no CPU fallback for the rasterizer");
    abi.handle = hnd;
#define AGS_SYM(field, name)                                                                      \
    abi.field = reinterpret_cast<decltype(abi.field)>(dlsym(hnd, #name));                         \
    TORCH_CHECK(abi.field, path, " lacks the symbol " #name)
    AGS_SYM(workspace_bytes, ags_workspace_bytes); AGS_SYM(workspace_init, ags_workspace_init); AGS_SYM(forward, ags_forward);
    AGS_SYM(backward, ags_backward); AGS_SYM(read_status, ags_read_status); AGS_SYM(read_status_async, ags_read_status_async);
    AGS_SYM(error_string, ags_error_string);
#undef AGS_SYM
}

void set_option(const std::string& name, double v) {
    std::lock_guard<std::mutex> g(mu);
    if (name == "binning_mode") opt.binning_mode = (int)v;
    else if (name == "skew_factor") opt.skew_factor = v;
    else if (name == "direct_budget_bytes") opt.direct_budget_bytes = v;
    else if (name == "always_check") opt.always_check = v != 0.0;
    else if (name == "headroom") opt.headroom = v;
    else if (name == "min_headroom") opt.min_headroom = v;
    else if (name == "max_pending") opt.max_pending = (int)v;
    else TORCH_CHECK(false, "unknown option ", name);
}

void set_tuning(int64_t bwd_reduce, int64_t render_slots, int64_t cull_first_min_n, int64_t tile_sort_no_wave, int64_t bucket_no_scan) {
    std::lock_guard<std::mutex> g(mu);
    tuning = AgsTuning{(int32_t)bwd_reduce, (int32_t)render_slots, (int32_t)cull_first_min_n, (int32_t)tile_sort_no_wave,
                       (int32_t)bucket_no_scan, 0, {0, 0}};
}

double get_option(const std::string& name) {
    std::lock_guard<std::mutex> g(mu);
    if (name == "binning_mode") return opt.binning_mode;
    if (name == "skew_factor") return opt.skew_factor;
    if (name == "direct_budget_bytes") return opt.direct_budget_bytes;
    if (name == "always_check") return opt.always_check ? 1.0 : 0.0;
    if (name == "headroom") return opt.headroom;
    if (name == "min_headroom") return opt.min_headroom;
    if (name == "max_pending") return opt.max_pending;
    TORCH_CHECK(false, "unknown option ", name);
}

std::map<std::string, int64_t> counters() {
    std::lock_guard<std::mutex> g(mu);
    return {{"forward_calls", cnt.forward_calls}, {"status_syncs", cnt.status_syncs}, {"deferred_checks", cnt.deferred_checks},
            {"overflows", cnt.overflows}, {"mode_switches", cnt.mode_switches}, {"pending", (int64_t)pending.size()},
            {"early_waits", cnt.early_waits}, {"repaired", cnt.repaired}};
}

// what the module has learnt: [(device, h, w, mode, largest need seen)], [(device, h, w, mode in use)], pooled workspaces
// [(device, h, w, mode, count, bytes)]
std::tuple<std::vector<std::vector<int64_t>>, std::vector<std::vector<int64_t>>, std::vector<std::vector<int64_t>>> state() {
    std::lock_guard<std::mutex> g(mu);
    std::vector<std::vector<int64_t>> a, b, c;
    for (auto& kv : need_seen) a.push_back({std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first), kv.second});
    for (auto& kv : mode_for) b.push_back({std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), kv.second});
    for (auto& kv : pool) {
        int64_t bytes = 0;
        for (auto& p : kv.second) bytes += p.ws.numel();
        c.push_back({std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first),
                     (int64_t)kv.second.size(), bytes});
    }
    return {a, b, c};
}

// forget everything learnt (sizes, modes, pooled workspaces); pending checks are settled first (errors swallowed)
void reset_state() {
    try { poll_pending(true); } catch (...) {}
    std::lock_guard<std::mutex> g(mu);
    need_seen.clear(); mode_for.clear(); pool.clear(); overflow_reports.clear();
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "native host side of diff_gaussian_rasterization_2d (MI355X build): autograd node over the C ABI of libags_raster.so";
    m.def("init", &init, "load the rasterizer library (path) and resolve the C ABI");
    m.def("rasterize", &rasterize);
    m.def("check_overflow", []() { poll_pending(true); },
          "wait for the status copies of all forward passes issued so far; raises if one of them outgrew its workspace");
    m.def("settle", []() { return collect_pending(true); },
          "wait for the status copies of all deferred-check passes issued so far; returns the overflow reports (empty: none) "
          "and forgets them - the sizes have been raised, the caller repeats the passes");
    m.def("set_option", &set_option);
    m.def("set_tuning", &set_tuning, "AgsTuning fields handed to the library with every workspace");
    m.def("get_option", &get_option);
    m.def("counters", &counters);
    m.def("state", &state);
    m.def("reset_state", &reset_state);
}
