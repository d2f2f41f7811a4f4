// libpvhip runtime plumbing: device selection, the compute streams (one by default; the scheduler may fork
// independent branches of a graph onto up to 8), a size-bucketed device-memory
// pool (the scheduler keeps every activation alive between infer() calls and re-produces the same
// shapes every call, so exact-size reuse makes steady-state allocation free), copies, events and
// hipGraph capture of a whole forward pass.
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "pvhip_common.h"

namespace pvhip {

State& state() {
    static State s;
    return s;
}

static Settings g_settings;
const Settings& settings() {
    if (g_settings.generation == 0) load_settings();     // a *_supported query before pvhip_init (no device needed)
    return g_settings;
}

void load_settings() {
    Settings s;
    s.generation = g_settings.generation + 1;
    auto env   = [](const char* name) -> const char* { const char* e = getenv(name); return (e != nullptr && e[0] != 0) ? e : nullptr; };
    auto is0   = [&](const char* name) { const char* e = env(name); return e != nullptr && e[0] == '0'; };
    auto num   = [&](const char* name, int dflt) { const char* e = env(name); return e != nullptr ? atoi(e) : dflt; };
    auto tri   = [&](const char* name) { const char* e = env(name); return e == nullptr ? 1 : (e[0] == '0' ? 0 : (e[0] == 'f' ? 2 : 1)); };
    s.conv_winograd  = !is0("PVHIP_CONV_WINOGRAD");
    s.conv_winograd4 = tri("PVHIP_CONV_WINOGRAD4");
    s.conv_winograd5 = tri("PVHIP_CONV_WINOGRAD5");
    s.conv_pointwise = !is0("PVHIP_CONV_POINTWISE");
    s.conv_stem      = !is0("PVHIP_CONV_STEM");
    { const char* e = env("PVHIP_CONV_STEM_WINO"); s.conv_stem_wino = e != nullptr && e[0] == '1'; }
    for (int i = 0; i < 8; ++i) {
        char name[16];
        snprintf(name, sizeof(name), "PVHIP_TUNE%d", i);
        s.tune[i] = num(name, 0);
    }
    if (const char* e = env("PVHIP_FUSE_POOLCONV")) s.fuse_poolconv = e[0] == '0' ? 0 : (e[0] == '4' ? 4 : (e[0] == '1' ? 1 : 2));
    s.pool3 = !(env("PVHIP_POOL3") != nullptr && num("PVHIP_POOL3", 1) == 0);
    s.stream_nt = num("PVHIP_STREAM_NT", 1);
    s.stream_wg = num("PVHIP_STREAM_WG", 16);
    if (s.stream_wg < 1) s.stream_wg = 1;
    if (s.stream_wg > 64) s.stream_wg = 64;
    s.conv_lds_pad_kb = num("PVHIP_CONV_LDS_PAD_KB", 0);
    s.conv_nopw = env("PVHIP_CONV_NOPW") != nullptr;
    s.conv_novalid = env("PVHIP_CONV_NOVALID") != nullptr;
    s.dwconv_cols = (int)num("PVHIP_DWCONV_COLS", 1);
    { const int b = num("PVHIP_CONV_F16_BM", 0); s.f16_bm = (b == 32 || b == 64 || b == 128) ? b : 0; }
    s.f16_c8_wgs = num("PVHIP_CONV_F16_C8_WGS", 0);
    s.f16_c8_prod = num("PVHIP_CONV_F16_C8_PROD", 0);
    s.f16_c8_reg = num("PVHIP_CONV_F16_C8_REG", 0);
    s.f16_c8_tpw = num("PVHIP_CONV_F16_C8_TPW", 0);
    s.lrnpool_wave = num("PVHIP_LRNPOOL_WAVE", 0) != 0;
    { const int v = num("PVHIP_CONV_MULTI_BM", 32); s.multi_bm = (v == 64 || v == 128) ? v : 32; }
    s.pw_stagger_pct = num("PVHIP_PW_STAGGER", 0);
    s.pw_tn = num("PVHIP_PW_TN", 0);
    s.pool3_kb    = num("PVHIP_POOL3_KB", 16);
    s.pool3_stage = num("PVHIP_POOL3_STAGE", 1) != 0;
    s.pool3_wg    = num("PVHIP_POOL3_WG", 0);
    if (const char* e = env("PVHIP_POOL3_CFG")) {
        int g = 0, sg = 0, band = 0;
        if (sscanf(e, "%d,%d,%d", &g, &sg, &band) == 3 && g > 0 && sg > 0 && band > 0) { s.pool3_g = g; s.pool3_s = sg; s.pool3_band = band; }
    }
    s.pool3_verbose = env("PVHIP_POOL3_VERBOSE") != nullptr;
    s.pool_lds_kb   = num("PVHIP_POOL_LDS_KB", 16);
    { const int v = num("PVHIP_WINO_KB", 0); s.wino_kb = (v == 32 || v == 64) ? v : 0; }
    s.wino_small = !is0("PVHIP_WINO_SMALL");
    s.wino_waves = num("PVHIP_WINO_WAVES", 8) == 4 ? 4 : 8;
    s.wino_balance = num("PVHIP_WINO_BALANCE", 1) != 0;
    s.wino_ragged = num("PVHIP_WINO_RAGGED", 1) != 0;
    s.wino_shared = num("PVHIP_WINO_SHARED", 1);
    s.wino_shared_min_tiles = num("PVHIP_WINO_SHARED_MIN_TILES", 2048);
    s.wino_shared_lag = num("PVHIP_WINO_SHARED_LAG", 0);
    s.wino_shared_prio = num("PVHIP_WINO_SHARED_PRIO", 1);
    s.wino_shared_odd = num("PVHIP_WINO_SHARED_ODD", 1) != 0;
    s.poolconv_prio = num("PVHIP_POOLCONV_PRIO", -1);
    s.wino_shared_old = num("PVHIP_WINO_SHARED_OLD", 1);
#ifdef PVHIP_DIAG
    // the predecessor convolution kernels and their tile overrides exist in the diagnostic build only
    if (const char* e = env("PVHIP_CONV_KERNEL")) s.conv_kernel = strcmp(e, "lds") == 0 ? 1 : (strcmp(e, "wave") == 0 ? 2 : 0);
    if (const char* e = env("PVHIP_CONV_TILE")) {
        int bm = 0, bn = 0;
        if (sscanf(e, "%dx%d", &bm, &bn) == 2 && (bm == 32 || bm == 64 || bm == 128) && (bn == 128 || bn == 256)) { s.tile_bm = bm; s.tile_bn = bn; }
    }
    if (const char* e = env("PVHIP_CONV_WTILE")) sscanf(e, "%dx%d", &s.wtile_m, &s.wtile_n);
    s.conv_pw16 = env("PVHIP_CONV_PW") != nullptr;
    s.conv_ablate  = num("PVHIP_CONV_ABLATE", 0);
    s.wino4_ablate = num("PVHIP_WINO4_ABLATE", 0);
    s.pw_ablate    = num("PVHIP_PW_ABLATE", 0);
    s.stem_ablate  = num("PVHIP_STEM_ABLATE", 0);
#endif
    g_settings = s;
}

static thread_local char g_err[512] = "no error";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

namespace {
struct Pool {
    std::mutex                                  mu;
    struct Live { size_t bytes; int epoch; bool captured; };   // captured: allocated while a pass was being captured into a graph
    std::unordered_map<void*, Live>             live;    // ptr -> rounded bytes, allocation epoch
    std::map<size_t, std::vector<void*>>        cached;  // rounded bytes -> free blocks
    std::vector<std::pair<void*, size_t>>       deferred; // epoch-0 blocks freed while streams were forked: reusable after a full sync
    // Allocation epochs: one per forward pass that runs on several streams (or asynchronously).  A block freed while
    // the pass that allocated it may still be executing waits in `parked` until that pass is declared finished;
    // blocks of finished passes (the previous outputs a new pass replaces) are reusable at once.
    int                                                     next_epoch = 1, current_epoch = 0;
    std::unordered_map<int, std::vector<std::pair<void*, size_t>>> parked;   // open epoch -> freed blocks
    size_t                                      bytes_live = 0, bytes_cached = 0;
    // Blocks freed while a forward pass is being CAPTURED into a hipGraph (workspaces, temporaries): the graph's kernels hold
    // their addresses for every replay, so they are not handed out again until the graph is destroyed.
    std::vector<std::pair<void*, size_t>>       capture_keep;
};
Pool& pool() {
    static Pool p;
    return p;
}
inline size_t round_up(size_t b) {
    const size_t q = 512;
    return b == 0 ? q : (b + q - 1) / q * q;
}

// Host-side wait for every stream, after which blocks freed while the streams were forked may be reused.
int sync_all_streams() {
    State& s = state();
    for (int i = 0; i < kMaxStreams; ++i)
        if (s.streams[i] != nullptr) {
            hipError_t e = hipStreamSynchronize(s.streams[i]);
            if (e != hipSuccess) return fail(PVHIP_EHIP, "hipStreamSynchronize(stream %d) -> %s", i, hipGetErrorString(e));
        }
    Pool& p = pool();
    std::lock_guard<std::mutex> g(p.mu);
    for (auto& blk : p.deferred) {
        p.cached[blk.second].push_back(blk.first);
        p.bytes_cached += blk.second;
    }
    p.deferred.clear();
    s.forked = s.current != 0;
    return PVHIP_OK;
}
}  // namespace
}  // namespace pvhip

using namespace pvhip;

extern "C" {

int pvhip_abi_version(void) { return PVHIP_ABI_VERSION; }

const char* pvhip_last_error(void) { return g_err; }

int pvhip_device_count(int* count) {
    PVHIP_CHECK_ARG(count != nullptr);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(PVHIP_EHIP, "hipGetDeviceCount -> %s", hipGetErrorString(e));
    }
    *count = n;
    return PVHIP_OK;
}

int pvhip_init(int device) {
    State& s = state();
    if (s.ready) {
        if (s.device == device) return PVHIP_OK;
        return fail(PVHIP_EINVAL, "pvhip_init: already initialised on device %d", s.device);
    }
    int n = 0;
    PVHIP_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n)
        return fail(PVHIP_EINVAL, "pvhip_init: device %d out of range (%d visible)", device, n);
    PVHIP_HIP(hipSetDevice(device));
    PVHIP_HIP(hipStreamCreateWithFlags(&s.streams[0], hipStreamNonBlocking));
    s.stream  = s.streams[0];
    s.current = 0;
    s.forked  = false;
    s.device = device;
    s.ready  = true;
    load_settings();
    return PVHIP_OK;
}

int pvhip_settings_reload(void) {
    load_settings();
    return PVHIP_OK;
}

int pvhip_pool_release(void) {
    PVHIP_REQUIRE_INIT();
    if (int rc = sync_all_streams()) return rc;
    Pool& p = pool();
    std::lock_guard<std::mutex> g(p.mu);
    for (auto& kv : p.cached)
        for (void* ptr : kv.second) (void)hipFree(ptr);
    p.cached.clear();
    p.bytes_cached = 0;
    return PVHIP_OK;
}

int pvhip_shutdown(void) {
    State& s = state();
    if (!s.ready) return PVHIP_OK;
    (void)pvhip_comm_destroy();
    (void)pvhip_pool_release();
    {
        Pool& p = pool();
        std::lock_guard<std::mutex> g(p.mu);
        for (auto& kv : p.live) (void)hipFree(kv.first);
        p.live.clear();
        for (auto& kv : p.parked)
            for (auto& blk : kv.second) (void)hipFree(blk.first);
        p.parked.clear();
        for (auto& blk : p.deferred) (void)hipFree(blk.first);
        p.deferred.clear();
        p.current_epoch = 0;
        p.bytes_live = 0;
    }
    for (int i = 0; i < kMaxStreams; ++i)
        if (s.streams[i] != nullptr) {
            (void)hipStreamDestroy(s.streams[i]);
            s.streams[i] = nullptr;
        }
    s.stream  = nullptr;
    s.current = 0;
    s.forked  = false;
    s.ready  = false;
    s.device = -1;
    return PVHIP_OK;
}

int pvhip_device_name(char* buf, size_t buflen) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(buf != nullptr && buflen > 0);
    hipDeviceProp_t prop;
    PVHIP_HIP(hipGetDeviceProperties(&prop, state().device));
    // some driver stacks report an empty marketing name: say what the architecture is then
    const char* name = prop.name[0] != 0 ? prop.name : (strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? "AMD Instinct MI355X-class (gfx950)" : "AMD GPU");
    snprintf(buf, buflen, "%s (%s, %d CUs)", name, prop.gcnArchName, prop.multiProcessorCount);
    return PVHIP_OK;
}

int pvhip_malloc(void** ptr, size_t bytes) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(ptr != nullptr);
    const size_t rb = round_up(bytes);
    Pool&        p  = pool();
    std::lock_guard<std::mutex> g(p.mu);
    auto it = p.cached.find(rb);
    if (it != p.cached.end() && !it->second.empty()) {
        *ptr = it->second.back();
        it->second.pop_back();
        p.bytes_cached -= rb;
    } else {
        if (state().capturing)
            return fail(PVHIP_EHIP, "pvhip_malloc(%zu bytes): the pool has no such block while a pass is being captured into a graph "
                                    "(run the pass eagerly first: steady-state inference allocates nothing new)", rb);
        void*      d = nullptr;
        hipError_t e = hipMalloc(&d, rb);
        if (e != hipSuccess) {
            // one retry after dropping the cache
            (void)hipGetLastError();
            for (int i = 0; i < kMaxStreams; ++i)
                if (state().streams[i] != nullptr) (void)hipStreamSynchronize(state().streams[i]);
            for (auto& blk : p.deferred) {
                p.cached[blk.second].push_back(blk.first);
                p.bytes_cached += blk.second;
            }
            p.deferred.clear();
            for (auto& kv : p.cached)
                for (void* c : kv.second) (void)hipFree(c);
            p.cached.clear();
            p.bytes_cached = 0;
            e = hipMalloc(&d, rb);
            if (e != hipSuccess)
                return fail(PVHIP_EHIP, "pvhip_malloc(%zu bytes) -> %s", rb, hipGetErrorString(e));
        }
        *ptr = d;
    }
    p.live[*ptr] = Pool::Live{rb, p.current_epoch, state().capturing};
    p.bytes_live += rb;
    return PVHIP_OK;
}

int pvhip_free(void* ptr) {
    if (ptr == nullptr) return PVHIP_OK;
    if (!state().ready) return PVHIP_OK;  // process teardown: memory already released
    Pool& p = pool();
    std::lock_guard<std::mutex> g(p.mu);
    auto it = p.live.find(ptr);
    if (it == p.live.end()) return fail(PVHIP_EINVAL, "pvhip_free: %p is not a live pvhip block", ptr);
    const size_t rb    = it->second.bytes;
    const int    epoch = it->second.epoch;
    const bool   captured = it->second.captured;
    p.live.erase(it);
    p.bytes_live -= rb;
    if (state().capturing && captured) {
        // allocated AND freed inside the capture (a workspace, a temporary): the captured kernels use it on every replay, so
        // it stays out of circulation until the graph is destroyed.  (A block of an EARLIER pass freed now -- an output this
        // pass replaces -- is in no captured kernel: it goes back to the pool as always and may serve this very pass.)
        p.capture_keep.emplace_back(ptr, rb);
        return PVHIP_OK;
    }
    if (epoch != 0) {
        auto open = p.parked.find(epoch);
        if (open != p.parked.end()) {       // its pass may still be running on some stream
            open->second.emplace_back(ptr, rb);
            return PVHIP_OK;
        }
    } else if (state().forked) {
        // consumers may still be queued on another stream: reusable only after the next full sync
        p.deferred.emplace_back(ptr, rb);
        return PVHIP_OK;
    }
    // Stream-ordered reuse: every consumer of this block was enqueued on the single compute
    // stream before the free, and the next owner's work is enqueued after it.
    p.cached[rb].push_back(ptr);
    p.bytes_cached += rb;
    return PVHIP_OK;
}

int pvhip_pool_epoch_begin(int* epoch) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(epoch != nullptr);
    Pool& p = pool();
    std::lock_guard<std::mutex> g(p.mu);
    *epoch          = p.next_epoch++;
    p.current_epoch = *epoch;
    p.parked[*epoch];            // open
    return PVHIP_OK;
}

int pvhip_pool_epoch_dispatched(void) {
    PVHIP_REQUIRE_INIT();
    Pool& p = pool();
    std::lock_guard<std::mutex> g(p.mu);
    p.current_epoch = 0;
    return PVHIP_OK;
}

int pvhip_pool_epoch_end(int epoch) {
    PVHIP_REQUIRE_INIT();
    Pool& p = pool();
    std::lock_guard<std::mutex> g(p.mu);
    auto it = p.parked.find(epoch);
    if (it == p.parked.end()) return fail(PVHIP_EINVAL, "pvhip_pool_epoch_end: epoch %d is not open", epoch);
    for (auto& blk : it->second) {
        p.cached[blk.second].push_back(blk.first);
        p.bytes_cached += blk.second;
    }
    p.parked.erase(it);
    if (p.current_epoch == epoch) p.current_epoch = 0;
    return PVHIP_OK;
}

int pvhip_pool_stats(size_t* bytes_in_use, size_t* bytes_cached) {
    Pool& p = pool();
    std::lock_guard<std::mutex> g(p.mu);
    if (bytes_in_use) *bytes_in_use = p.bytes_live;
    if (bytes_cached) *bytes_cached = p.bytes_cached;
    return PVHIP_OK;
}

int pvhip_memcpy_h2d(void* dst, const void* src, size_t bytes) {
    PVHIP_REQUIRE_INIT();
    if (bytes == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(dst != nullptr && src != nullptr);
    PVHIP_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, state().stream));
    // the host buffer may be pageable and is free to change once we return
    PVHIP_HIP(hipStreamSynchronize(state().stream));
    return PVHIP_OK;
}

int pvhip_memcpy_d2h(void* dst, const void* src, size_t bytes) {
    PVHIP_REQUIRE_INIT();
    if (bytes == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(dst != nullptr && src != nullptr);
    PVHIP_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, state().stream));
    PVHIP_HIP(hipStreamSynchronize(state().stream));
    if (state().forked && state().current == 0) return sync_all_streams();
    return PVHIP_OK;
}

/* Page-locked host memory for read-backs: a device-to-host copy into pageable memory goes through the runtime's staging buffer (~100 us for the
 * 1 MB Result of a batch of 256), into pinned memory it is one DMA (device.py keeps a small pool for DeviceTensor.numpy()). */
void* pvhip_host_alloc(size_t bytes) {
    if (!state().ready || bytes == 0) return nullptr;
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

int pvhip_host_free(void* p) {
    PVHIP_REQUIRE_INIT();
    if (p == nullptr) return PVHIP_OK;
    PVHIP_HIP(hipHostFree(p));
    return PVHIP_OK;
}

int pvhip_memcpy_d2d(void* dst, const void* src, size_t bytes) {
    PVHIP_REQUIRE_INIT();
    if (bytes == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(dst != nullptr && src != nullptr);
    PVHIP_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, state().stream));
    return PVHIP_OK;
}

int pvhip_memset(void* dst, int byte, size_t bytes) {
    PVHIP_REQUIRE_INIT();
    if (bytes == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(dst != nullptr);
    PVHIP_HIP(hipMemsetAsync(dst, byte, bytes, state().stream));
    return PVHIP_OK;
}

int pvhip_sync(void) {
    PVHIP_REQUIRE_INIT();
    if (state().capturing) return fail(PVHIP_EINVAL, "pvhip_sync: a pass is being captured into a graph (nothing executes until it is replayed)");
    return sync_all_streams();
}

int pvhip_stream_select(int index) {
    PVHIP_REQUIRE_INIT();
    State& s = state();
    if (index < 0 || index >= kMaxStreams)
        return fail(PVHIP_EINVAL, "pvhip_stream_select: stream %d outside 0..%d", index, kMaxStreams - 1);
    if (s.streams[index] == nullptr) PVHIP_HIP(hipStreamCreateWithFlags(&s.streams[index], hipStreamNonBlocking));
    s.stream  = s.streams[index];
    s.current = index;
    if (index != 0) s.forked = true;
    return PVHIP_OK;
}

int pvhip_stream_wait_event(void* ev) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(ev != nullptr);
    PVHIP_HIP(hipStreamWaitEvent(state().stream, (hipEvent_t)ev, 0));
    return PVHIP_OK;
}

int pvhip_event_create_untimed(void** ev) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(ev != nullptr);
    hipEvent_t e;
    PVHIP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *ev = (void*)e;
    return PVHIP_OK;
}

int pvhip_event_create(void** ev) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(ev != nullptr);
    hipEvent_t e;
    PVHIP_HIP(hipEventCreate(&e));
    *ev = (void*)e;
    return PVHIP_OK;
}

int pvhip_event_destroy(void* ev) {
    if (ev == nullptr || !state().ready) return PVHIP_OK;
    PVHIP_HIP(hipEventDestroy((hipEvent_t)ev));
    return PVHIP_OK;
}

int pvhip_event_record(void* ev) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(ev != nullptr);
    PVHIP_HIP(hipEventRecord((hipEvent_t)ev, state().stream));
    return PVHIP_OK;
}

int pvhip_event_sync(void* ev) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(ev != nullptr);
    PVHIP_HIP(hipEventSynchronize((hipEvent_t)ev));
    return PVHIP_OK;
}

int pvhip_event_elapsed_ms(void* start, void* stop, float* ms) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(start != nullptr && stop != nullptr && ms != nullptr);
    PVHIP_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return PVHIP_OK;
}

// hipGraph capture of a whole forward pass (the run_tasks loop, inference_engine.py:259-292): every launch of the pass -- on the
// current stream and on the streams that join it through events -- becomes a node of one graph that is replayed with ONE call.
struct GraphExec {
    hipGraphExec_t                        exec = nullptr;
    std::vector<std::pair<void*, size_t>> keep;      // blocks freed during the capture: pinned until the graph is destroyed
};

int pvhip_graph_begin_capture(void) {
    PVHIP_REQUIRE_INIT();
    if (state().capturing) return fail(PVHIP_EINVAL, "pvhip_graph_begin_capture: a capture is already open");
    // relaxed: event creation / destruction (Python's garbage collector) and pool bookkeeping go on beside the capture
    PVHIP_HIP(hipStreamBeginCapture(state().stream, hipStreamCaptureModeRelaxed));
    state().capturing = true;
    return PVHIP_OK;
}

int pvhip_graph_capture_status(int* status) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(status != nullptr);
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    PVHIP_HIP(hipStreamIsCapturing(state().stream, &st));
    *status = st == hipStreamCaptureStatusActive ? 1 : (st == hipStreamCaptureStatusInvalidated ? 2 : 0);
    return PVHIP_OK;
}

int pvhip_graph_end_capture(void** graph_exec) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(graph_exec != nullptr);
    if (!state().capturing) return fail(PVHIP_EINVAL, "pvhip_graph_end_capture: no capture is open");
    state().capturing = false;
    GraphExec* ge = new GraphExec;
    {
        Pool& p = pool();
        std::lock_guard<std::mutex> g(p.mu);
        ge->keep.swap(p.capture_keep);
    }
    auto release = [&]() {
        Pool& p = pool();
        std::lock_guard<std::mutex> g(p.mu);
        for (auto& blk : ge->keep) { p.cached[blk.second].push_back(blk.first); p.bytes_cached += blk.second; }
        delete ge;
    };
    hipGraph_t g = nullptr;
    // a capture that something invalidated on the way (a synchronisation, work on a stream that never joined it) must not reach
    // hipGraphInstantiate: say so instead
    hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
    const hipError_t qe = hipStreamIsCapturing(state().stream, &cst);
    if (getenv("PVHIP_GRAPH_VERBOSE") != nullptr) fprintf(stderr, "pvhip_graph_end_capture: hipStreamIsCapturing -> %d, status %d\n", (int)qe, (int)cst);
    hipError_t e = hipStreamEndCapture(state().stream, &g);
    if (getenv("PVHIP_GRAPH_VERBOSE") != nullptr) fprintf(stderr, "pvhip_graph_end_capture: hipStreamEndCapture -> %d, graph %p\n", (int)e, (void*)g);
    if (e == hipSuccess && (g == nullptr || cst != hipStreamCaptureStatusActive)) e = hipErrorStreamCaptureInvalidated;
    if (e != hipSuccess) {
        if (g != nullptr) (void)hipGraphDestroy(g);
        release();
        (void)hipGetLastError();
        return fail(PVHIP_EHIP, "hipStreamEndCapture -> %s (capture status %d)", hipGetErrorString(e), (int)cst);
    }
    if (getenv("PVHIP_GRAPH_VERBOSE") != nullptr) {
        size_t n_nodes = 0;
        (void)hipGraphGetNodes(g, nullptr, &n_nodes);
        fprintf(stderr, "pvhip_graph_end_capture: %zu nodes\n", n_nodes);
    }
    e = hipGraphInstantiate(&ge->exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) { release(); return fail(PVHIP_EHIP, "hipGraphInstantiate -> %s", hipGetErrorString(e)); }
    *graph_exec = (void*)ge;
    return PVHIP_OK;
}

int pvhip_graph_launch(void* graph_exec) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(graph_exec != nullptr);
    PVHIP_HIP(hipGraphLaunch(static_cast<GraphExec*>(graph_exec)->exec, state().stream));
    return PVHIP_OK;
}

int pvhip_graph_destroy(void* graph_exec) {
    if (graph_exec == nullptr) return PVHIP_OK;
    GraphExec* ge = static_cast<GraphExec*>(graph_exec);
    if (state().ready) {
        (void)sync_all_streams();
        if (ge->exec != nullptr) (void)hipGraphExecDestroy(ge->exec);
        Pool& p = pool();
        std::lock_guard<std::mutex> g(p.mu);
        for (auto& blk : ge->keep) { p.cached[blk.second].push_back(blk.first); p.bytes_cached += blk.second; }
    }
    delete ge;
    return PVHIP_OK;
}

}  // extern "C"
