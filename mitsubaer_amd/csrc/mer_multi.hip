// mer_multi.hip -- several GPUs behind the C-ABI, in one process (include/mer.h: mer_multi_*).
//
// The reference renders with N local worker threads that pull image blocks from a queue and merge them into the film under a mutex
// (src/librender/renderproc.cpp:79,142-149; src/mitsuba/mitsuba.cpp:281).  Here a "worker" is a GPU: one context per device, one host
// thread per context while a render runs (mer_render's host loop blocks on its own read-backs), volumes replicated, the sample space cut
// into one mer_shard per context, and ONE exchange at the end -- the films are sum-reduced onto the first device:
//   * distinct devices: ncclReduce over xGMI, all ranks issued from this thread inside ncclGroupStart / ncclGroupEnd on a communicator made
//     by ncclCommInitAll.  librccl.so is loaded at run time (dlopen), so libmer.so itself carries no RCCL dependency and a one-GPU box
//     never touches it;
//   * otherwise (a device listed twice -- RCCL refuses that --, RCCL missing, or rccl = 0): the other films are copied to the first
//     device (hipMemcpyPeerAsync) and added by a kernel, in context order (deterministic summation order).
// Paths are independent: there is no other collective on this path.
#include "mer_internal.hpp"
#include <dlfcn.h>
#include <rccl/rccl.h>          // types and prototypes only: the symbols are resolved with dlsym
#include <chrono>
#include <thread>

namespace {

std::string g_multi_error;

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string why;            // why it is unavailable
    bool load() {
        if (handle) return true;
        if (!why.empty()) return false;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
        }
        if (!handle) { why = std::string("librccl.so not loadable: ") + (dlerror() ? dlerror() : "?"); return false; }
#define MER_SYM(field, sym) field = (decltype(field)) dlsym(handle, sym); if (!field) { why = std::string("librccl.so lacks ") + sym; dlclose(handle); handle = nullptr; return false; }
        MER_SYM(CommInitAll, "ncclCommInitAll") MER_SYM(CommDestroy, "ncclCommDestroy") MER_SYM(GroupStart, "ncclGroupStart")
        MER_SYM(GroupEnd, "ncclGroupEnd") MER_SYM(Reduce, "ncclReduce") MER_SYM(GetErrorString, "ncclGetErrorString")
#undef MER_SYM
        return true;
    }
};

__global__ void film_add_kernel(float *dst, const float *src, size_t n) {
    const size_t i = ((size_t) blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        float4 a = *(float4 *) (dst + i); const float4 b = *(const float4 *) (src + i);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        *(float4 *) (dst + i) = a;
    } else for (size_t k = i; k < n; k++) dst[k] += src[k];
}

}  // namespace

struct mer_multi {
    std::vector<mer_context *> ctx;
    std::vector<int> devices;
    std::vector<hipStream_t> streams;         // one non-blocking stream per context: film zeroing, render, reduction
    std::string error;
    bool distinct = true;
    Rccl rccl;
    std::vector<ncclComm_t> comms;            // made on first use, for exactly this device list
    // per-context film + the staging buffer of the peer-copy reduction (on the first device); they grow, never shrink
    std::vector<float *> film; size_t film_floats = 0;
    float *staging = nullptr; size_t staging_floats = 0;
    int last_reduce = MER_REDUCE_NONE; std::vector<float> last_render_ms; float last_reduce_ms = 0;
    uint64_t last_counters[MER_C_COUNT] = {};
};

namespace {
int mfail(mer_multi *m, const std::string &msg) { m->error = msg; return 1; }
#define MULTI_HIP(m, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return mfail((m), std::string(#call) + " failed: " + hipGetErrorString(e_)); } while (0)
}

extern "C" {

int mer_multi_create(const int32_t *device_ids, int32_t n, mer_multi **out) {
    if (!device_ids || n < 1 || n > 64 || !out) { g_multi_error = "mer_multi_create: need 1..64 device ids"; return 1; }
    mer_multi *m = new mer_multi();
    for (int i = 0; i < n; i++) {
        mer_context *c = nullptr;
        if (mer_context_create(device_ids[i], &c)) {
            g_multi_error = std::string("mer_multi_create: device ") + std::to_string(device_ids[i]) + ": " + mer_last_error(nullptr);
            mer_multi_destroy(m); return 1;
        }
        m->ctx.push_back(c); m->devices.push_back(device_ids[i]);
        for (int k = 0; k < i; k++) if (device_ids[k] == device_ids[i]) m->distinct = false;
        hipStream_t s = nullptr;
        if (hipSetDevice(device_ids[i]) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess || mer_context_set_stream(c, s)) {
            g_multi_error = "mer_multi_create: stream creation failed"; mer_multi_destroy(m); return 1;
        }
        m->streams.push_back(s);
    }
    // the peer-copy reduction reads the other devices' films from the first device
    for (int i = 1; i < n && m->distinct; i++) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, m->devices[0], m->devices[i]) == hipSuccess && can) { (void) hipSetDevice(m->devices[0]); (void) hipDeviceEnablePeerAccess(m->devices[i], 0); (void) hipGetLastError(); }
    }
    m->film.assign(n, nullptr); m->last_render_ms.assign(n, 0.0f);
    *out = m;
    return 0;
}

void mer_multi_destroy(mer_multi *m) {
    if (!m) return;
    if (m->rccl.handle) for (ncclComm_t c : m->comms) if (c) (void) m->rccl.CommDestroy(c);
    for (size_t i = 0; i < m->ctx.size(); i++) {
        (void) hipSetDevice(m->devices[i]);
        if (i < m->film.size() && m->film[i]) (void) hipFree(m->film[i]);
        if (i == 0 && m->staging) (void) hipFree(m->staging);
        if (m->ctx[i]) { (void) mer_context_set_stream(m->ctx[i], nullptr); mer_context_destroy(m->ctx[i]); }
        if (i < m->streams.size() && m->streams[i]) (void) hipStreamDestroy(m->streams[i]);
    }
    delete m;
}

const char *mer_multi_last_error(mer_multi *m) { return m ? m->error.c_str() : g_multi_error.c_str(); }
int32_t mer_multi_size(mer_multi *m) { return m ? (int32_t) m->ctx.size() : 0; }
mer_context *mer_multi_context(mer_multi *m, int32_t i) { return (m && i >= 0 && i < (int32_t) m->ctx.size()) ? m->ctx[i] : nullptr; }

int mer_multi_set_option(mer_multi *m, const char *name, int64_t value) {
    if (!m) return 1;
    for (mer_context *c : m->ctx) if (mer_context_set_option(c, name, value)) return mfail(m, mer_last_error(c));
    return 0;
}

int mer_multi_volume_upload(mer_multi *m, const mer_grid_desc *desc, const void *host_data, int32_t layout, mer_volume *out) {
    if (!m || !out) return 1;
    mer_volume h0 = 0;
    for (size_t i = 0; i < m->ctx.size(); i++) {
        mer_volume h = 0;
        if (mer_volume_upload(m->ctx[i], desc, host_data, layout, &h)) return mfail(m, mer_last_error(m->ctx[i]));
        if (i == 0) h0 = h;
        else if (h != h0) return mfail(m, "mer_multi_volume_upload: the contexts' volume handles differ (volumes created through mer_multi_context()?)");
    }
    *out = h0;
    return 0;
}
int mer_multi_volume_build_spline(mer_multi *m, mer_volume v) {
    if (!m) return 1;
    for (mer_context *c : m->ctx) if (mer_volume_build_spline(c, v)) return mfail(m, mer_last_error(c));
    return 0;
}
int mer_multi_volume_destroy(mer_multi *m, mer_volume v) {
    if (!m) return 1;
    for (mer_context *c : m->ctx) if (mer_volume_destroy(c, v)) return mfail(m, mer_last_error(c));
    return 0;
}

int mer_multi_render(mer_multi *m, const mer_scene_desc *scene, int32_t shard_mode, int32_t spp_begin, int32_t spp_count, uint64_t seed,
                     int32_t rccl, float *film_host) {
    if (!m || !scene || !film_host) return 1;
    if (shard_mode != MER_SHARD_SAMPLES && shard_mode != MER_SHARD_TILES) return mfail(m, "mer_multi_render: unknown shard mode");
    if (spp_begin < 0 || spp_count < 0) return mfail(m, "mer_multi_render: invalid sample range");
    const int n = (int) m->ctx.size();
    int32_t ch = 5;
    if (mer_film_channels(m->ctx[0], scene, &ch)) return mfail(m, mer_last_error(m->ctx[0]));
    const size_t floats = (size_t) scene->width * scene->height * (size_t) ch;
    if (floats > m->film_floats) {
        for (int i = 0; i < n; i++) {
            MULTI_HIP(m, hipSetDevice(m->devices[i]));
            if (m->film[i]) (void) hipFree(m->film[i]);
            m->film[i] = nullptr;
            MULTI_HIP(m, hipMalloc((void **) &m->film[i], floats * sizeof(float)));
        }
        m->film_floats = floats;
    }
    // ---- render: one host thread per context (each blocks in its own wavefront loop)
    std::vector<int> rc(n, 0);
    std::vector<std::thread> workers;
    for (int i = 0; i < n; i++) {
        workers.emplace_back([&, i]() {
            const auto t0 = std::chrono::steady_clock::now();
            mer_context *c = m->ctx[i];
            mer_shard sh;
            if (shard_mode == MER_SHARD_SAMPLES) { sh.spp_begin = spp_begin + i; sh.spp_count = std::max(0, (spp_count - i + n - 1) / n); sh.spp_stride = n; sh.tile_rank = 0; sh.tile_count = 1; }
            else { sh.spp_begin = spp_begin; sh.spp_count = spp_count; sh.spp_stride = 1; sh.tile_rank = i; sh.tile_count = n; }
            rc[i] = (hipSetDevice(m->devices[i]) != hipSuccess) || mer_counters_reset(c) || mer_film_zero_n(c, m->film[i], scene->width, scene->height, ch) ||
                    mer_render(c, scene, &sh, seed, m->film[i]) || mer_synchronize(c);
            m->last_render_ms[i] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        });
    }
    for (std::thread &t : workers) t.join();
    for (int i = 0; i < n; i++) if (rc[i]) return mfail(m, std::string("context ") + std::to_string(i) + " (device " + std::to_string(m->devices[i]) + "): " + mer_last_error(m->ctx[i]));
    // ---- reduce onto the first device
    const auto t0 = std::chrono::steady_clock::now();
    const bool want_rccl = (rccl == 2) || (rccl == 1 && n > 1 && m->distinct);
    m->last_reduce = MER_REDUCE_NONE;
    bool use_rccl = want_rccl && m->distinct && m->rccl.load();
    if (use_rccl && m->comms.empty()) {
        Rccl &R = m->rccl;
        m->comms.assign(n, nullptr);
        const ncclResult_t r = R.CommInitAll(m->comms.data(), n, m->devices.data());
        if (r != ncclSuccess) {
            m->comms.clear();
            m->rccl.why = std::string("ncclCommInitAll failed: ") + R.GetErrorString(r);
            if (rccl == 2) return mfail(m, m->rccl.why);
            use_rccl = false;                   // rccl = 1 is "RCCL where it works": the films still get reduced, by peer copies (mer_multi_last_stats says which)
        }
    }
    if (use_rccl) {
        Rccl &R = m->rccl;
        ncclResult_t r = R.GroupStart();
        for (int i = 0; i < n && r == ncclSuccess; i++) {
            MULTI_HIP(m, hipSetDevice(m->devices[i]));
            r = R.Reduce(m->film[i], m->film[i], floats, ncclFloat, ncclSum, 0, m->comms[i], m->streams[i]);
        }
        const ncclResult_t r2 = R.GroupEnd();
        if (r != ncclSuccess || r2 != ncclSuccess) return mfail(m, std::string("ncclReduce failed: ") + R.GetErrorString(r != ncclSuccess ? r : r2));
        for (int i = 0; i < n; i++) { MULTI_HIP(m, hipSetDevice(m->devices[i])); MULTI_HIP(m, hipStreamSynchronize(m->streams[i])); }
        m->last_reduce = MER_REDUCE_RCCL;
    } else if (rccl == 2) {
        return mfail(m, "mer_multi_render: RCCL requested but unavailable: " + (m->distinct ? m->rccl.why : std::string("a device is listed twice")));      // rccl = 2 demands it
    } else if (n > 1) {
        MULTI_HIP(m, hipSetDevice(m->devices[0]));
        if (m->staging_floats < floats) {
            if (m->staging) (void) hipFree(m->staging);
            m->staging = nullptr;
            MULTI_HIP(m, hipMalloc((void **) &m->staging, floats * sizeof(float)));
            m->staging_floats = floats;
        }
        for (int i = 1; i < n; i++) {         // context order: the summation order does not depend on which render finished first
            if (m->devices[i] == m->devices[0]) MULTI_HIP(m, hipMemcpyAsync(m->staging, m->film[i], floats * sizeof(float), hipMemcpyDeviceToDevice, m->streams[0]));
            else MULTI_HIP(m, hipMemcpyPeerAsync(m->staging, m->devices[0], m->film[i], m->devices[i], floats * sizeof(float), m->streams[0]));
            hipLaunchKernelGGL(film_add_kernel, dim3((unsigned) ((floats / 4 + 255) / 256 + 1)), dim3(256), 0, m->streams[0], m->film[0], m->staging, floats);
        }
        MULTI_HIP(m, hipGetLastError());
        MULTI_HIP(m, hipStreamSynchronize(m->streams[0]));
        m->last_reduce = MER_REDUCE_PEER_COPY;
    }
    if (mer_film_download_n(m->ctx[0], m->film[0], scene->width, scene->height, ch, film_host)) return mfail(m, mer_last_error(m->ctx[0]));
    m->last_reduce_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (int k = 0; k < MER_C_COUNT; k++) m->last_counters[k] = 0;
    for (int i = 0; i < n; i++) {
        uint64_t c[MER_C_COUNT];
        if (mer_counters_read(m->ctx[i], c)) return mfail(m, mer_last_error(m->ctx[i]));
        for (int k = 0; k < MER_C_COUNT; k++) m->last_counters[k] += c[k];
    }
    return 0;
}

int mer_multi_last_stats(mer_multi *m, int32_t *reduce_path, float *render_ms, float *reduce_ms, uint64_t counters[MER_C_COUNT]) {
    if (!m) return 1;
    if (reduce_path) *reduce_path = m->last_reduce;
    if (render_ms) for (size_t i = 0; i < m->ctx.size(); i++) render_ms[i] = m->last_render_ms[i];
    if (reduce_ms) *reduce_ms = m->last_reduce_ms;
    if (counters) for (int k = 0; k < MER_C_COUNT; k++) counters[k] = m->last_counters[k];
    return 0;
}

}  // extern "C"
