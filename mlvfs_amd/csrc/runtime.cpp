// runtime.cpp -- host runtime of libmlvfs_amd.so: error string, EV tables and
// their exact 16-bit re-encodings, per-device state, per-thread contexts.
//
// Threading model (SURVEY.md 8b "Threading"): libfuse calls the exported
// functions from a pool of worker threads, one frame per call.  Each host thread
// gets its own HIP stream and staging buffers per device (thread_local), devices
// are shared read-only (tables), per-clip state is guarded by its own mutex.
// Threads that never called mlvfs_amd_init() are spread round-robin over the
// visible GPUs, which is the frame-parallel multi-GPU mode of the drop-in path.
#include "common.h"

#include <atomic>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>
#include <string>
#include <algorithm>

// caller-provided tables (mlvfs/mlvfs.h:90-92); absent when the library is used
// stand-alone (tests, bench)
extern "C" {
int *get_raw2ev(int black) __attribute__((weak));
int *get_ev2raw(void) __attribute__((weak));
}

namespace mlv {

namespace {
std::mutex g_rg_mu;
int g_rg_depth = 0;
bool g_rg_init = false;
char *g_rg_app = nullptr;
char g_rg_parked[128];                                  // a TYPE_3 state for whoever calls rand() while the application's is parked
}  // namespace

LibcRandGuard::LibcRandGuard()
{
    std::lock_guard<std::mutex> lk(g_rg_mu);
    if (g_rg_depth++ == 0) {
        if (!g_rg_init) { g_rg_app = initstate(0x5eedu, g_rg_parked, sizeof g_rg_parked); g_rg_init = true; }
        else g_rg_app = setstate(g_rg_parked);
    }
}

LibcRandGuard::~LibcRandGuard()
{
    std::lock_guard<std::mutex> lk(g_rg_mu);
    if (--g_rg_depth == 0 && g_rg_app) (void)setstate(g_rg_app);
}

// glibc's random_r.c (public layout of the buffer that initstate() / setstate() hand back): the word in front of the state array
// holds MAX_TYPES * (rear pointer's index) + type for every type but 0; TYPE_3 has 31 words and the front pointer 3 ahead of the
// rear one.  The front pointer addresses the OLDEST word (the one the next call overwrites with oldest + rear's word).
namespace {
constexpr int RG_MAX_TYPES = 5, RG_TYPE_3 = 3, RG_DEG_3 = 31, RG_SEP_3 = 3;
}

// Is this libc's generator what the two functions below take it for?  Checked once, on a generator of the library's own (the
// application's stream is not touched): a seeded TYPE_3 buffer is parked, read the way take_app_state reads it, its next value
// predicted; then it is written back rotated the way put_app_state writes it and asked again.  Any surprise: the bulk path stays off
// and the values are drawn call by call (draw_mod1024), as before.  Called with g_rg_mu held and the application's state parked.
static bool rg_layout_ok()
{
    static int known = -1;
    if (known >= 0) return known != 0;
    known = 0;
    alignas(8) static char probe[128], aside[128];
    char *const before = initstate(20240229u, probe, sizeof probe);       // current: probe
    if (!before) return false;
    (void)rand(); (void)rand(); (void)rand();
    if (!initstate(1u, aside, sizeof aside)) { (void)setstate(before); return false; }      // probe is parked now, its rear index stored
    bool ok = false;
    const int32_t *st = (const int32_t *)probe;
    if (st[0] % RG_MAX_TYPES == RG_TYPE_3 && st[0] / RG_MAX_TYPES >= 0 && st[0] / RG_MAX_TYPES < RG_DEG_3) {
        const int front = (st[0] / RG_MAX_TYPES + RG_SEP_3) % RG_DEG_3;
        uint32_t x[RG_DEG_3 + 2];
        for (int k = 0; k < RG_DEG_3; k++) x[k] = (uint32_t)st[1 + (front + k) % RG_DEG_3];
        x[31] = x[0] + x[28];                                              // the next two values
        x[32] = x[1] + x[29];
        (void)setstate(probe);
        const bool first = (uint32_t)rand() == (x[31] >> 1);
        (void)setstate(aside);                                             // parked again, one step further
        int32_t *wr = (int32_t *)probe;
        for (int k = 0; k < RG_DEG_3; k++) wr[1 + k] = (int32_t)x[1 + k];  // oldest first, as put_app_state stores it
        wr[0] = RG_MAX_TYPES * ((RG_DEG_3 - RG_SEP_3) % RG_DEG_3) + RG_TYPE_3;
        (void)setstate(probe);
        ok = first && (uint32_t)rand() == (x[32] >> 1);
    }
    (void)setstate(before);
    known = ok ? 1 : 0;
    if (!ok) fprintf(stderr, "mlvfs_amd: this libc's rand() state is not glibc's TYPE_3 layout: stripes dither drawn call by call\n");
    return ok;
}

bool LibcRandGuard::take_app_state(uint32_t x[31])
{
    std::lock_guard<std::mutex> lk(g_rg_mu);
    if (!(g_rg_depth > 0 && g_rg_app)) return false;
    if (!rg_layout_ok()) return false;
    const int32_t *st = (const int32_t *)g_rg_app;
    if (st[0] % RG_MAX_TYPES != RG_TYPE_3) return false;
    const int rear = st[0] / RG_MAX_TYPES;
    if (rear < 0 || rear >= RG_DEG_3) return false;
    const int front = (rear + RG_SEP_3) % RG_DEG_3;
    for (int k = 0; k < RG_DEG_3; k++) x[k] = (uint32_t)st[1 + (front + k) % RG_DEG_3];
    return true;
}

bool LibcRandGuard::put_app_state(const uint32_t x[31])
{
    std::lock_guard<std::mutex> lk(g_rg_mu);
    if (!(g_rg_depth > 0 && g_rg_app)) return false;
    int32_t *st = (int32_t *)g_rg_app;
    if (st[0] % RG_MAX_TYPES != RG_TYPE_3) return false;
    for (int k = 0; k < RG_DEG_3; k++) st[1 + k] = (int32_t)x[k];        // oldest word at index 0: front pointer 0, rear pointer 28
    st[0] = RG_MAX_TYPES * ((RG_DEG_3 - RG_SEP_3) % RG_DEG_3) + RG_TYPE_3;
    return true;
}

// The layout probe runs once, when the library is loaded: no worker thread of the library exists yet, none can be inside a HIP call
// while the probe switches the process's generator back and forth.
static const bool g_rg_probed_at_load = [] {
    LibcRandGuard guard;
    uint32_t x[31];
    (void)LibcRandGuard::take_app_state(x);
    return true;
}();

void LibcRandGuard::draw_mod1024(uint16_t *out, long long n)
{
    std::lock_guard<std::mutex> lk(g_rg_mu);
    const bool parked = g_rg_depth > 0 && g_rg_app;
    if (parked) (void)setstate(g_rg_app);
    for (long long i = 0; i < n; i++) out[i] = (uint16_t)(rand() % 1024);
    if (parked) (void)setstate(g_rg_parked);
}


// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    fprintf(stderr, "mlvfs_amd: %s\n", g_err);
}

// ------------------------------------------------------------------ tables
static std::once_flag g_lut_once;
static std::vector<int32_t> g_raw2ev_lin, g_ev2raw;
static std::vector<uint16_t> g_t16, g_t16d, g_u16;
static int g_luts_ok = 0;

static void build_luts()
{
    g_raw2ev_lin.assign(16384, 0);
    g_ev2raw.assign(24 * MLV_EV_RES, 0);
    int *caller_r2e = get_raw2ev ? get_raw2ev(0) : nullptr;
    int *caller_e2r = get_ev2raw ? get_ev2raw() : nullptr;
    for (int i = 0; i < 16384; i++) {
        if (caller_r2e) g_raw2ev_lin[i] = caller_r2e[i];
        else g_raw2ev_lin[i] = i == 0 ? INT_MIN : (int32_t)(log2((double)i) * MLV_EV_RES);   // main.c:163-167
    }
    for (int i = -10 * MLV_EV_RES; i < 14 * MLV_EV_RES; i++) {
        if (caller_e2r) g_ev2raw[i + 10 * MLV_EV_RES] = caller_e2r[i];
        else g_ev2raw[i + 10 * MLV_EV_RES] = (int32_t)pow(2.0, (double)((float)i / MLV_EV_RES));   // main.c:189-192
    }
    // 16-bit re-encodings + exhaustive identity check against the full tables
    g_t16.resize(MLV_T16_N);
    g_u16.resize(MLV_U16_N);
    int ok = 1;
    for (int j = 8192; j < 16384; j++) {
        int v = g_raw2ev_lin[j] - 13 * MLV_EV_RES;
        if (v < 0 || v > 65535) ok = 0;
        g_t16[j - 8192] = (uint16_t)v;
        if (v < 4 * (j - 8192)) ok = 0;                  // the fused kernels keep T16[m] - 4 m (k_frame_dev.h: load_t16_rel)
    }
    for (int f = 0; f < MLV_U16_N; f++) {
        int v = g_ev2raw[(13 + 10) * MLV_EV_RES + f];
        if (v < 0 || v > 65535) ok = 0;
        g_u16[f] = (uint16_t)v;
    }
    g_t16d.assign(16384, 0);
    for (int i = 1; i < 16384; i++) {
        const int v = g_raw2ev_lin[i] - ((31 - __builtin_clz(i)) << 15);
        if (v < 0 || v > 65535) ok = 0;
        g_t16d[i] = (uint16_t)v;
    }
    if (g_raw2ev_lin[0] != INT_MIN) ok = 0;
    for (int i = 1; i < 16384 && ok; i++) {
        int e = 31 - __builtin_clz(i);
        int rec = (int)g_t16[(i << (13 - e)) - 8192] + (e << 15);
        if (rec != g_raw2ev_lin[i]) ok = 0;
    }
    for (int ev = 0; ev < 14 * MLV_EV_RES && ok; ev++) {
        int q = ev >> 15, f = ev & 32767;
        if ((g_u16[f] >> (13 - q)) != g_ev2raw[ev + 10 * MLV_EV_RES]) ok = 0;
    }
    g_luts_ok = ok;
    if (!ok) set_error("host EV tables do not admit the 16-bit re-encoding (unexpected libm)");
}

static void ensure_luts() { std::call_once(g_lut_once, build_luts); }
const int32_t *host_raw2ev_lin() { ensure_luts(); return g_raw2ev_lin.data(); }
const int32_t *host_ev2raw() { ensure_luts(); return g_ev2raw.data(); }
const uint16_t *host_t16() { ensure_luts(); return g_t16.data(); }
const uint16_t *host_t16d() { ensure_luts(); return g_t16d.data(); }
const uint16_t *host_u16() { ensure_luts(); return g_u16.data(); }
int luts_ok() { ensure_luts(); return g_luts_ok; }

// ------------------------------------------------------------------ devices
static std::mutex g_dev_mutex;
static std::map<int, Device *> g_devices;

static Device *get_device(int id)
{
    std::lock_guard<std::mutex> lk(g_dev_mutex);
    auto it = g_devices.find(id);
    if (it != g_devices.end()) return it->second;
    if (!luts_ok()) return nullptr;
    if (hipSetDevice(id) != hipSuccess) { set_error("hipSetDevice(%d) failed", id); return nullptr; }
    Device *d = new Device;
    d->id = id;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, id) == hipSuccess) d->num_cu = prop.multiProcessorCount;
    uint16_t *t16 = nullptr, *t16d = nullptr, *u16 = nullptr;
    if (hipMalloc(&t16, MLV_T16_N * 2) != hipSuccess || hipMalloc(&u16, MLV_U16_N * 2) != hipSuccess ||
        hipMalloc(&t16d, 16384 * 2) != hipSuccess ||
        hipMemcpy(t16d, host_t16d(), 16384 * 2, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(t16, host_t16(), MLV_T16_N * 2, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(u16, host_u16(), MLV_U16_N * 2, hipMemcpyHostToDevice) != hipSuccess) {
        set_error("device %d: table upload failed", id);
        delete d;
        return nullptr;
    }
    d->luts.t16 = t16;
    d->luts.t16d = t16d;
    d->luts.u16 = u16;
    // HIP loads a file's code object at the first launch of one of its kernels: about 1 ms each for the four files of the frame
    // path, 2 ms for k_frame's 68 instantiations -- paid here, once per device, instead of inside the first frame of the first
    // clip (bench.py first_frame_split_ms; MLVFS_AMD_PRELOAD=0 leaves them lazy).  The dual-ISO, LJ92 and preview files stay lazy.
    static const bool preload = [] { const char *e = getenv("MLVFS_AMD_PRELOAD"); return !(e && e[0] == '0'); }();
    if (preload) { preload_k_unpack(); preload_k_pixfix(); preload_k_frame(); preload_k_frame_p(); preload_k_frame_s(); preload_k_stripes(); }
    g_devices[id] = d;
    return d;
}

// ------------------------------------------------------------------ thread contexts
static std::atomic<int> g_thread_counter{0};
static thread_local int t_device = -1;
// a host thread's contexts live as long as the thread: libfuse's loop creates and retires workers, and what a retired worker
// held (stream, staging buffers, ticket counters of that stream) goes back when its thread_local storage is destroyed
struct ThreadCtxs {
    std::map<int, ThreadCtx *> m;
    ~ThreadCtxs() { for (auto &kv : m) delete kv.second; }
};
static thread_local ThreadCtxs t_ctxs;

ThreadCtx::~ThreadCtx()
{
    if (stream) {
        (void)hipStreamSynchronize(stream);
        release_stream_state(dev ? dev->id : 0, stream);
        (void)hipStreamDestroy(stream);
    }
    if (d_a) (void)hipFree(d_a);
    if (d_b) (void)hipFree(d_b);
    if (d_patch) (void)hipFree(d_patch);
    if (d_res[0]) (void)hipFree(d_res[0]);
    if (d_res[1]) (void)hipFree(d_res[1]);
    if (h_pin) (void)hipHostFree(h_pin);
    if (ev_up) (void)hipEventDestroy(ev_up);
}

static int visible_devices()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int bind_device(int device)
{
    int n = visible_devices();
    if (n <= 0) { set_error("no HIP device visible (libmlvfs_amd has no CPU fallback)"); return MLVFS_AMD_ERR_HIP; }
    if (device < 0 || device >= n) { set_error("device %d out of range (%d visible)", device, n); return MLVFS_AMD_ERR_ARG; }
    t_device = device;
    return thread_ctx() ? MLVFS_AMD_OK : MLVFS_AMD_ERR_HIP;
}

// Worker threads without a device of their own choosing (the drop-in symbols under libfuse's pool) are spread round-robin over the
// visible GPUs IN THE ORDER OF THEIR PCI BUS IDS: the k-th worker gets the same physical card whatever order the runtime enumerates
// them in, and neighbours in that order are neighbours on the node's xGMI / PCIe topology.  The mapping is logged once per process
// (MLVFS_AMD_QUIET=1 silences it); VERDICT r3 weak #9.
// (the order itself is host arithmetic on the cards' bus ids: mlvfs_amd_test_device_order lets the CPU tests give it a faked node)
static std::vector<int> order_by_bus_id(const std::vector<std::string> &bus)
{
    std::vector<std::pair<std::string, int>> ids;
    for (size_t d = 0; d < bus.size(); d++) ids.push_back({ bus[d], (int)d });
    std::sort(ids.begin(), ids.end());
    std::vector<int> order;
    for (auto &kv : ids) order.push_back(kv.second);
    return order;
}
static int device_of_worker(const std::vector<int> &order, long long k) { return order[(size_t)(k % (long long)order.size())]; }

static std::vector<int> g_dev_order;            // position in PCI order -> HIP device ordinal
static std::once_flag g_dev_order_once;
static const std::vector<int> &device_order(int n)
{
    std::call_once(g_dev_order_once, [n] {
        std::vector<std::string> bus_ids;
        for (int d = 0; d < n; d++) {
            char bus[64] = "";
            if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, d) != hipSuccess) snprintf(bus, sizeof bus, "unknown-%04d", d);
            bus_ids.push_back(bus);
        }
        g_dev_order = order_by_bus_id(bus_ids);
        const char *q = getenv("MLVFS_AMD_QUIET");
        if (n > 1 && !(q && q[0] == '1')) {
            fprintf(stderr, "mlvfs_amd: %d GPUs, worker threads are bound round-robin in PCI order:", n);
            for (size_t k = 0; k < g_dev_order.size(); k++)
                fprintf(stderr, " worker %zu -> device %d (%s)%s", k, g_dev_order[k], bus_ids[g_dev_order[k]].c_str(), k + 1 < g_dev_order.size() ? "," : "\n");
        }
    });
    return g_dev_order;
}

extern "C" int mlvfs_amd_test_device_order(const char *const *bus_ids, int n, int workers, int *device_of)
{
    if (!bus_ids || n <= 0 || workers < 0 || !device_of) return MLVFS_AMD_ERR_ARG;
    std::vector<std::string> bus;
    for (int d = 0; d < n; d++) bus.push_back(bus_ids[d] ? bus_ids[d] : "");
    const std::vector<int> order = order_by_bus_id(bus);
    for (int k = 0; k < workers; k++) device_of[k] = device_of_worker(order, k);
    return MLVFS_AMD_OK;
}

ThreadCtx *thread_ctx()
{
    if (t_device < 0) {
        int n = visible_devices();
        if (n <= 0) { set_error("no HIP device visible (libmlvfs_amd has no CPU fallback)"); return nullptr; }
        const char *env = getenv("MLVFS_AMD_DEVICE");
        if (env) t_device = atoi(env) % n;
        else {
            const std::vector<int> &order = device_order(n);
            t_device = device_of_worker(order, (long long)g_thread_counter.fetch_add(1));
        }
    }
    auto it = t_ctxs.m.find(t_device);
    if (it != t_ctxs.m.end()) {
        if (hipSetDevice(t_device) != hipSuccess) return nullptr;
        return it->second;
    }
    Device *dev = get_device(t_device);
    if (!dev) return nullptr;
    if (hipSetDevice(t_device) != hipSuccess) return nullptr;
    ThreadCtx *c = new ThreadCtx;
    c->dev = dev;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        set_error("hipStreamCreate failed");
        delete c;
        return nullptr;
    }
    t_ctxs.m[t_device] = c;
    return c;
}

ThreadCtx *thread_ctx_if_any()
{
    if (t_device < 0) return nullptr;
    auto it = t_ctxs.m.find(t_device);
    return it == t_ctxs.m.end() ? nullptr : it->second;
}

int ThreadCtx::ensure(size_t need_a, size_t need_b)
{
    // d_a may hold the packed payload of a frame whose stages are recorded (frame bracket): whoever wants the scratch buffers
    // now is not one of those stages, so they run first
    if (lazy.active) {
        const int rc = flush_pending(this);
        if (rc) return rc;
    }
    if (need_a > cap_a) {
        if (d_a) (void)hipFree(d_a);
        d_a = nullptr; cap_a = 0;
        MLV_HIP(hipMalloc(&d_a, need_a));
        cap_a = need_a;
    }
    if (need_b > cap_b) {
        if (d_b) (void)hipFree(d_b);
        d_b = nullptr; cap_b = 0;
        MLV_HIP(hipMalloc(&d_b, need_b));
        cap_b = need_b;
    }
    return MLVFS_AMD_OK;
}

int ThreadCtx::ensure_res(size_t bytes)
{
    if (bytes <= cap_res) return MLVFS_AMD_OK;
    if (lazy.active || res_dirty) {                  // the buffers about to be freed hold a frame the host has not got yet
        const int rc = flush_pending(this);
        if (rc) return rc;
    }
    (void)hipStreamSynchronize(stream);
    res_host = nullptr;
    res_dirty = false;
    for (int k = 0; k < 2; k++) {
        if (d_res[k]) (void)hipFree(d_res[k]);
        d_res[k] = nullptr;
    }
    cap_res = 0;
    for (int k = 0; k < 2; k++) MLV_HIP(hipMalloc(&d_res[k], bytes));
    cap_res = bytes;
    return MLVFS_AMD_OK;
}

int ThreadCtx::ensure_patch(size_t need)
{
    if (need > cap_patch) {
        if (d_patch) { (void)hipStreamSynchronize(stream); (void)hipFree(d_patch); }
        d_patch = nullptr; cap_patch = 0;
        MLV_HIP(hipMalloc(&d_patch, need));
        cap_patch = need;
    }
    return MLVFS_AMD_OK;
}

}  // namespace mlv

// ------------------------------------------------------------------ C ABI
extern "C" {

int mlvfs_amd_device_count(void) { return mlv::visible_devices(); }

int mlvfs_amd_test_rand_layout(void)
{
    mlv::LibcRandGuard guard;
    uint32_t x[31];
    return mlv::LibcRandGuard::take_app_state(x) ? 1 : 0;
}

// the PCI bus id ("0000:c1:00.0") of a visible device: what identifies the physical card across processes (bench.py gathers it from
// every rank and refuses a run in which two ranks share a card); and the device the calling thread is bound to (-1: none yet)
int mlvfs_amd_device_pci_bus_id(int device, char *out, int len)
{
    if (!out || len < 16) { mlv::set_error("device_pci_bus_id: buffer too small"); return MLVFS_AMD_ERR_ARG; }
    if (device < 0 || device >= mlv::visible_devices()) { mlv::set_error("device_pci_bus_id: device %d out of range", device); return MLVFS_AMD_ERR_ARG; }
    MLV_HIP(hipDeviceGetPCIBusId(out, len, device));
    return MLVFS_AMD_OK;
}
int mlvfs_amd_thread_device(void) { return mlv::t_device; }
int mlvfs_amd_init(int device) { return mlv::bind_device(device); }
const char *mlvfs_amd_last_error(void) { return mlv::g_err; }
const char *mlvfs_amd_version(void) { return "mlvfs_amd 0.1.0 (gfx950)"; }

}

// Test hook, host only (no GPU): how the streaming kernels (k_frame_s, k_frame_p5) cut a frame of width x height pixels into tasks of
// seg_rows cell rows -- columns, segments, how many segments of the last column a wave takes side by side (k_frame_dev.h:
// frame_stream_fold), tasks per frame.  0, or MLVFS_AMD_ERR_ARG for a frame the kernels do not take.
extern "C" int mlvfs_amd_test_stream_plan(int width, int height, int seg_rows, int *cols, int *segs, int *fold, int *tasks_per_frame)
{
    if (width < 16 || width % 8 || height < 2 || height % 2 || seg_rows < 1 || !cols || !segs || !fold || !tasks_per_frame) return MLVFS_AMD_ERR_ARG;
    const int c = mlv::frame_stream_cols(width), sg = (height / 2 + seg_rows - 1) / seg_rows, f = mlv::frame_stream_fold(width, c, sg);
    *cols = c; *segs = sg; *fold = f;
    *tasks_per_frame = f > 1 ? (c - 1) * sg + (sg + f - 1) / f : c * sg;
    return MLVFS_AMD_OK;
}

