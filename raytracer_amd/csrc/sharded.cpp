// rt_sharded_*: one scene replicated on several GPUs of one process, every GPU rendering its own range of the
// samples of every pixel, the float accumulation buffers summed by one RCCL all-reduce over xGMI. It stands in for
// the worker threads of the reference's driver (src/main.rs:189-230: 8 threads, each owning a band of image
// columns, results merged by the caller): sample ranges balance better than columns, and because the random
// numbers of a path are keyed by (pixel, sample, bounce) the picture does not depend on the partition except for
// the order of the f32 additions.
//
// One host thread and one stream per device; the threads are started once per scene and wait for renders (a render
// call hands each its share and waits for all), and the picture crosses PCIe through a pinned staging buffer of the
// scene. RCCL is loaded at run time (dlopen), so the library has no link-time dependency on it and single-GPU users
// never touch it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt/abi.h"
#include "host/errors.hpp"

namespace {

#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return rt::set_error(e_ == hipErrorOutOfMemory ? RT_E_NOMEM : RT_E_DEVICE,                     \
                                 std::string(#expr) + ": " + hipGetErrorString(e_));                       \
    } while (0)

// The few RCCL entry points used, with the ABI of rccl.h (ncclComm_t is a pointer, the enums are ints).
struct Rccl {
    void* handle = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*AllReduce)(const void* send, void* recv, size_t count, int datatype, int op, void* comm, hipStream_t stream) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    static constexpr int kFloat32 = 7;  // ncclFloat32
    static constexpr int kSum = 0;      // ncclSum
};

std::mutex g_rccl_mutex;
Rccl g_rccl;
bool g_rccl_tried = false;

const Rccl* rccl() {  // null when RCCL cannot be loaded
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (!g_rccl_tried) {
        g_rccl_tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            g_rccl.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (g_rccl.handle) break;
        }
        if (g_rccl.handle) {
            auto sym = [&](const char* n) { return dlsym(g_rccl.handle, n); };
            g_rccl.CommInitAll = reinterpret_cast<decltype(g_rccl.CommInitAll)>(sym("ncclCommInitAll"));
            g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
            g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(sym("ncclAllReduce"));
            g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
            g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
            g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
            if (!g_rccl.CommInitAll || !g_rccl.CommDestroy || !g_rccl.AllReduce || !g_rccl.GroupStart || !g_rccl.GroupEnd) {
                dlclose(g_rccl.handle);
                g_rccl = Rccl();
            }
        }
    }
    return g_rccl.handle ? &g_rccl : nullptr;
}

int rccl_fail(const Rccl* r, const char* what, int code) {
    return rt::set_error(RT_E_DEVICE, std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(code) : "RCCL error " + std::to_string(code)));
}

struct Shard {
    int device = -1;
    RtScene* scene = nullptr;
    hipStream_t stream = nullptr;
    float* accum = nullptr;
    size_t accum_floats = 0;
    void* comm = nullptr;
    // the worker thread's report of the current render
    int rc = RT_OK;
    std::string err;
    RtRenderStats stats{};
};

}  // namespace

struct RtShardedScene {
    std::vector<Shard> shards;
    bool have_comm = false;
    // pinned host memory the picture is staged in on its way to and from device 0
    float* pinned = nullptr;
    size_t pinned_floats = 0;
    // the worker threads (one per shard, started by rt_sharded_create) and the render they are working on
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    uint64_t generation = 0;       // bumped by every render call
    uint32_t remaining = 0;        // workers that have not finished the current generation
    bool quit = false;
    const RtCamera* cam = nullptr;
    const RtRenderParams* params = nullptr;
    size_t floats = 0;
};

namespace {

// What shard i does for one render: its accumulation buffer starts from the caller's picture (shard 0, staged in the
// pinned buffer) or from zero, then its share of the samples is rendered into it.
int shard_render(RtShardedScene* s, uint32_t i) {
    Shard& sh = s->shards[i];
    const size_t floats = s->floats;
    const uint32_t world = (uint32_t)s->shards.size();
    if (sh.accum_floats < floats) {
        if (sh.accum) { HIP_TRY(hipFree(sh.accum)); sh.accum = nullptr; sh.accum_floats = 0; }
        HIP_TRY(hipMalloc((void**)&sh.accum, floats * sizeof(float)));
        sh.accum_floats = floats;
    }
    // the sum over devices must come out as caller's buffer + all samples: the first device starts from the caller's
    // values, the others from zero
    if (i == 0u) HIP_TRY(hipMemcpyAsync(sh.accum, s->pinned, floats * sizeof(float), hipMemcpyHostToDevice, sh.stream));
    else HIP_TRY(hipMemsetAsync(sh.accum, 0, floats * sizeof(float), sh.stream));
    RtRenderParams q = *s->params;
    rt_shard_samples(s->params->sample_begin, s->params->sample_end, i, world, &q.sample_begin, &q.sample_end);
    std::memset(&sh.stats, 0, sizeof sh.stats);
    if (q.sample_end > q.sample_begin)
        if (int rc = rt_render_device(sh.scene, s->cam, &q, sh.accum, sh.stream, &sh.stats)) return rc;
    HIP_TRY(hipStreamSynchronize(sh.stream));
    return RT_OK;
}

void worker_main(RtShardedScene* s, uint32_t i) {
    (void)hipSetDevice(s->shards[i].device);  // once: the current device is a property of the thread
    uint64_t seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lock(s->m);
            s->cv_job.wait(lock, [&] { return s->quit || s->generation != seen; });
            if (s->quit) return;
            seen = s->generation;
        }
        Shard& sh = s->shards[i];
        sh.rc = shard_render(s, i);
        sh.err = sh.rc ? rt_last_error() : "";  // the message is thread-local: carry it to the caller's thread
        {
            std::lock_guard<std::mutex> lock(s->m);
            if (--s->remaining == 0) s->cv_done.notify_all();
        }
    }
}

}  // namespace

extern "C" {

void rt_shard_samples(uint32_t sample_begin, uint32_t sample_end, uint32_t rank, uint32_t world, uint32_t* lo, uint32_t* hi) {
    // contiguous, disjoint, covering; sizes differ by at most one, the first (total % world) ranks hold the larger
    const uint32_t total = sample_end > sample_begin ? sample_end - sample_begin : 0u;
    if (world == 0u || rank >= world) { if (lo) *lo = sample_begin; if (hi) *hi = sample_begin; return; }
    const uint32_t base = total / world, rem = total % world;
    const uint32_t a = sample_begin + rank * base + std::min(rank, rem);
    if (lo) *lo = a;
    if (hi) *hi = a + base + (rank < rem ? 1u : 0u);
}

void rt_sharded_destroy(RtShardedScene* s) {
    if (!s) return;
    {
        std::lock_guard<std::mutex> lock(s->m);
        s->quit = true;
    }
    s->cv_job.notify_all();
    for (auto& t : s->workers) t.join();
    if (s->pinned) (void)hipHostFree(s->pinned);
    const Rccl* r = s->have_comm ? rccl() : nullptr;
    for (Shard& sh : s->shards) {
        if (sh.device < 0) continue;
        (void)hipSetDevice(sh.device);
        if (sh.comm && r) (void)r->CommDestroy(sh.comm);
        if (sh.scene) rt_scene_destroy(sh.scene);
        if (sh.accum) (void)hipFree(sh.accum);
        if (sh.stream) (void)hipStreamDestroy(sh.stream);
    }
    delete s;
}

int rt_sharded_device_count(const RtShardedScene* s) { return s ? (int)s->shards.size() : 0; }

int rt_sharded_create(const RtSceneDesc* desc, const int* devices, int n_devices, RtShardedScene** out) {
    if (!out) return rt::set_error(RT_E_INVALID, "null output");
    *out = nullptr;
    if (!desc) return rt::set_error(RT_E_INVALID, "null scene description");
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1) {
        (void)hipGetLastError();
        return rt::set_error(RT_E_DEVICE, "no HIP device");
    }
    std::vector<int> devs;
    if (devices) {
        if (n_devices < 1) return rt::set_error(RT_E_INVALID, "empty device list");
        devs.assign(devices, devices + n_devices);
    } else {  // all of them, or the first n_devices
        const int n = n_devices > 0 ? n_devices : visible;
        for (int i = 0; i < n; ++i) devs.push_back(i);
    }
    for (size_t i = 0; i < devs.size(); ++i) {
        if (devs[i] < 0 || devs[i] >= visible) return rt::set_error(RT_E_INVALID, "device ordinal out of range");
        for (size_t j = 0; j < i; ++j)
            if (devs[j] == devs[i]) return rt::set_error(RT_E_INVALID, "device listed twice");
    }
    int before = 0;
    (void)hipGetDevice(&before);
    RtShardedScene* s = new (std::nothrow) RtShardedScene();
    if (!s) return rt::set_error(RT_E_NOMEM, "out of host memory");
    s->shards.resize(devs.size());
    auto body = [&]() -> int {
        for (size_t i = 0; i < devs.size(); ++i) {
            Shard& sh = s->shards[i];
            sh.device = devs[i];
            HIP_TRY(hipSetDevice(sh.device));
            if (int rc = rt_scene_create(desc, &sh.scene)) return rc;  // a replica in this device's HBM
            HIP_TRY(hipStreamCreateWithFlags(&sh.stream, hipStreamNonBlocking));
        }
        const Rccl* r = rccl();
        if (!r) {
            if (devs.size() > 1) return rt::set_error(RT_E_UNSUPPORTED, "RCCL (librccl.so) could not be loaded: needed to sum the buffers of several GPUs");
            return RT_OK;  // one device: nothing to sum
        }
        std::vector<void*> comms(devs.size(), nullptr);
        if (int e = r->CommInitAll(comms.data(), (int)devs.size(), devs.data())) return rccl_fail(r, "ncclCommInitAll", e);
        for (size_t i = 0; i < devs.size(); ++i) s->shards[i].comm = comms[i];
        s->have_comm = true;
        return RT_OK;
    };
    const int rc = body();
    (void)hipSetDevice(before);
    if (rc) { rt_sharded_destroy(s); return rc; }
    for (uint32_t i = 0; i < (uint32_t)devs.size(); ++i) s->workers.emplace_back(worker_main, s, i);
    *out = s;
    return RT_OK;
}

int rt_sharded_render(RtShardedScene* s, const RtCamera* cam, const RtRenderParams* p, float* accum_rgb, RtRenderStats* stats) {
    if (!s || !cam || !p || !accum_rgb) return rt::set_error(RT_E_INVALID, "null argument");
    if (p->width < 2 || p->height < 2 || p->sample_end < p->sample_begin) return rt::set_error(RT_E_INVALID, "bad render parameters");
    const size_t floats = (size_t)p->width * p->height * 3u;
    const uint32_t world = (uint32_t)s->shards.size();
    int before = 0;
    (void)hipGetDevice(&before);
    if (s->pinned_floats < floats) {
        if (s->pinned) { HIP_TRY(hipHostFree(s->pinned)); s->pinned = nullptr; s->pinned_floats = 0; }
        HIP_TRY(hipHostMalloc((void**)&s->pinned, floats * sizeof(float), hipHostMallocPortable));
        s->pinned_floats = floats;
    }
    std::memcpy(s->pinned, accum_rgb, floats * sizeof(float));
    {
        std::unique_lock<std::mutex> lock(s->m);
        s->cam = cam; s->params = p; s->floats = floats;
        s->remaining = world;
        ++s->generation;
        s->cv_job.notify_all();
        s->cv_done.wait(lock, [&] { return s->remaining == 0; });
    }
    auto finish = [&]() -> int {
        for (uint32_t i = 0; i < world; ++i)
            if (s->shards[i].rc) return rt::set_error(s->shards[i].rc, "device " + std::to_string(s->shards[i].device) + ": " + s->shards[i].err);
        if (s->have_comm) {
            const Rccl* r = rccl();
            if (int e = r->GroupStart()) return rccl_fail(r, "ncclGroupStart", e);
            for (uint32_t i = 0; i < world; ++i) {
                Shard& sh = s->shards[i];
                if (int e = r->AllReduce(sh.accum, sh.accum, floats, Rccl::kFloat32, Rccl::kSum, sh.comm, sh.stream)) {
                    (void)r->GroupEnd();
                    return rccl_fail(r, "ncclAllReduce", e);
                }
            }
            if (int e = r->GroupEnd()) return rccl_fail(r, "ncclGroupEnd", e);
        }
        // the sum lands in every device's buffer; device 0's goes back to the caller, behind the all-reduce on its stream
        HIP_TRY(hipSetDevice(s->shards[0].device));
        HIP_TRY(hipMemcpyAsync(s->pinned, s->shards[0].accum, floats * sizeof(float), hipMemcpyDeviceToHost, s->shards[0].stream));
        for (uint32_t i = 0; i < world; ++i) {
            HIP_TRY(hipSetDevice(s->shards[i].device));
            HIP_TRY(hipStreamSynchronize(s->shards[i].stream));
        }
        std::memcpy(accum_rgb, s->pinned, floats * sizeof(float));
        if (stats) {
            std::memset(stats, 0, sizeof *stats);
            for (uint32_t i = 0; i < world; ++i) {
                const RtRenderStats& st = s->shards[i].stats;
                stats->paths += st.paths;
                stats->rays += st.rays;
                stats->batches = std::max(stats->batches, st.batches);
                stats->ms_total = std::max(stats->ms_total, st.ms_total);    // the devices run side by side
                stats->ms_extend = std::max(stats->ms_extend, st.ms_extend);
                stats->ms_shade = std::max(stats->ms_shade, st.ms_shade);
            }
        }
        return RT_OK;
    };
    const int rc = finish();
    (void)hipSetDevice(before);
    return rc;
}

}  // extern "C"
