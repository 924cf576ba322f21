// dualiso.cpp -- host side of the full dual-ISO conversion (cr2hdr 20-bit,
// mlvfs/hdr.c:230-1957): the scalar decisions between the kernels of k_dualiso.hip,
// the libm-built tables, the drop-in cr2hdr20_convert_data and its device-resident form.
//
// Implemented: interp_method 0 (AMaZE + edge-directed interpolation, k_amaze.hip) and 1 (mean23),
// full-res on/off, alias map on/off, chroma smoothing 0/2/3/5.
//
// Reference quirks reproduced on purpose:
//  * active_area.x1 = 0 empties the noise loops: dark noise is the default 8.0;
//  * the dither cache is never initialised: the 20 -> 16 bit step is deterministic;
//  * the 20-bit EV tables are cached per black level, in one cache per consumer
//    (AMaZE interpolation / mean23 interpolation / mix / final blend), and keep the white
//    level they were first built with (hdr.c:1080,1240,1575,1672).  Process-global here too.
#include "clip.h"
#include "dualiso.h"

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

namespace mlv {

constexpr int EVR = 32768, N20 = 1 << 20;

// ------------------------------------------------------------------ host tables
struct HostLut {                 // one per consumer function of the reference
    int black = -1;
    int built_white = -1;                // the white level the table was built with (not part of its key: hdr.c:833)
    unsigned version = 0;
    std::vector<int> raw2ev, ev2raw;     // ev2raw[0] is EV index -10*32768
};

// Independent entries over [0, n) on a few host threads: the 2^20-entry tables of a black level are 1.8 M calls of log2 / pow per
// consumer (a conversion's first call in a process: 64 ms on one thread, and again whenever a clip brings another black level).
template <typename F> static void table_rows(int n, F fn)
{
    const unsigned hw = std::thread::hardware_concurrency();
    const int nt = (int)std::min<unsigned>(8, hw ? hw : 1);
    if (nt <= 1 || n < (1 << 14)) { fn(0, n); return; }
    std::vector<std::thread> th;
    const int per = (n + nt - 1) / nt;
    for (int t = 1; t < nt; t++) { const int a = t * per, b = std::min(n, a + per); if (a < b) th.emplace_back([=] { fn(a, b); }); }
    fn(0, std::min(n, per));
    for (auto &x : th) x.join();
}

static void host_lut_build(HostLut &L, int black, int white)          // hdr.c:839-874
{
    L.raw2ev.resize(N20);
    L.ev2raw.resize(24 * EVR);
    int *raw2ev = L.raw2ev.data();
    int *ev2raw = L.ev2raw.data() + 10 * EVR;
    table_rows(N20, [=](int a, int b) {
        for (int i = a; i < b; i++) {
            double signal = i / 64.0 - black / 64.0;
            if (signal < -1023) signal = -1023;
            raw2ev[i] = signal > 0 ? (int)round(log2(1 + signal) * EVR) : -(int)round(log2(1 - signal) * EVR);
        }
    });
    table_rows(10 * EVR, [=](int a, int b) {
        for (int k = a; k < b; k++) {
            const int i = k - 10 * EVR;
            const double v = black + 64 - round(64 * pow(2, (double)-i / EVR));
            ev2raw[i] = (int)(v < 0 ? 0 : (v > black ? black : v));
        }
    });
    const int ev_white = raw2ev[white];
    table_rows(14 * EVR, [=](int a, int b) {
        for (int i = a; i < b; i++) {
            const double v = black - 64 + round(64 * pow(2, (double)i / EVR));
            ev2raw[i] = (int)(v < black ? black : (v > N20 - 1 ? N20 - 1 : v));
            if (i >= ev_white) ev2raw[i] = std::max(ev2raw[i], white);
        }
    });
    ev2raw[raw2ev[0]] = 0;
    L.black = black;
    L.built_white = white;
    L.version++;
}

struct HostCurves {              // build_fullres_curve (hdr.c:890-913) + the log2 part of the mix curve (hdr.c:1566)
    int black = -1;
    int thr = N20;                   // fullres[i] > 0.8 <=> i >= thr
    int fr_lo = 0, fr_hi = N20;      // fullres[i] == fullres[0] below fr_lo, == fullres[N20 - 1] from fr_hi on
    unsigned version = 0;
    std::vector<double> fullres, log2sig;
};

static std::mutex g_di_mutex;
static HostLut g_lut_interp, g_lut_mix, g_lut_blend, g_lut_amaze;
static HostCurves g_curves;
static std::vector<double> g_evf;        // raw2evf_base: log2(i) * 32768, i = 0 gives -inf (main.c:136-148)

// Device copies are IMMUTABLE once uploaded: conversions of other threads that took the pointers (and released the lock) may
// still have kernels queued on their own streams, so a table whose host version changed (another clip's black level) goes to
// a fresh allocation and the old one is retired; retired sets are freed, after a device-wide synchronisation, once more than
// a handful have piled up (two clips with different black levels served alternately) or on mlvfs_amd_dualiso_reset().
struct DeviceTables {            // per device copies + the host version they mirror
    int *raw2ev[4] = { nullptr, nullptr, nullptr, nullptr }, *ev2raw[4] = { nullptr, nullptr, nullptr, nullptr };
    unsigned ver[4] = { 0, 0, 0, 0 };
    double *fullres = nullptr, *log2sig = nullptr, *evf = nullptr;
    unsigned curves_ver = 0;
    int2 *mix_pair = nullptr;
    unsigned packed_curves_ver = 0, packed_mix_ver = 0, same_mix_ver = 0, same_blend_ver = 0;
    int blend_is_mix = 0;
    std::vector<void *> retired;
    void retire(void *p)
    {
        if (!p) return;
        retired.push_back(p);
        if (retired.size() > 16) {
            (void)hipDeviceSynchronize();
            for (void *q : retired) (void)hipFree(q);
            retired.clear();
        }
    }
};
static std::map<int, DeviceTables> g_dev_tables;

static int upload_lut(DeviceTables &T, int k, const HostLut &H)
{
    if (T.ver[k] == H.version && T.raw2ev[k]) return MLVFS_AMD_OK;
    T.retire(T.raw2ev[k]); T.retire(T.ev2raw[k]);
    T.raw2ev[k] = T.ev2raw[k] = nullptr;
    MLV_HIP(hipMalloc(&T.raw2ev[k], sizeof(int) * N20));
    MLV_HIP(hipMalloc(&T.ev2raw[k], sizeof(int) * 24 * EVR));
    MLV_HIP(hipMemcpy(T.raw2ev[k], H.raw2ev.data(), sizeof(int) * N20, hipMemcpyHostToDevice));
    MLV_HIP(hipMemcpy(T.ev2raw[k], H.ev2raw.data(), sizeof(int) * 24 * EVR, hipMemcpyHostToDevice));
    T.ver[k] = H.version;
    return MLVFS_AMD_OK;
}

// tables for one conversion; caller holds no lock
static int prepare_tables(int device, int black, int white, int interp_method, DiLuts *L, const double **d_evf)
{
    std::lock_guard<std::mutex> lk(g_di_mutex);
    DeviceTables &T = g_dev_tables[device];
    if (g_evf.empty()) {
        g_evf.resize(16384);
        for (int i = 0; i < 16384; i++) g_evf[i] = log2((double)i) * EVR;
    }
    if (!T.evf) {
        MLV_HIP(hipMalloc(&T.evf, sizeof(double) * 16384));
        MLV_HIP(hipMemcpy(T.evf, g_evf.data(), sizeof(double) * 16384, hipMemcpyHostToDevice));
    }
    *d_evf = T.evf;
    if (!L) return MLVFS_AMD_OK;
    // each consumer function of the reference owns a cache; only the interpolator that runs touches its own
    HostLut *H[4] = { &g_lut_interp, &g_lut_mix, &g_lut_blend, &g_lut_amaze };
    const int ik = interp_method == 0 ? 3 : 0;
    for (int k = 0; k < 4; k++) {
        if ((k == 0 || k == 3) && k != ik) continue;
        if (H[k]->black != black) {                                          // white is not part of the key
            const HostLut *same = nullptr;                                   // (a table another consumer built from the same two levels)
            for (int j = 0; j < 4; j++)
                if (j != k && H[j]->black == black && H[j]->built_white == white) same = H[j];
            if (same) { H[k]->raw2ev = same->raw2ev; H[k]->ev2raw = same->ev2raw; H[k]->black = black; H[k]->built_white = white; H[k]->version++; }
            else host_lut_build(*H[k], black, white);
        }
        int rc = upload_lut(T, k, *H[k]);
        if (rc) return rc;
    }
    if (g_curves.black != black) {
        g_curves.fullres.resize(N20);
        g_curves.log2sig.resize(N20);
        double *const c_log2sig = g_curves.log2sig.data(), *const c_fullres = g_curves.fullres.data();
        table_rows(N20, [=](int a, int b) {
            for (int i = a; i < b; i++) {
                const double sig = i / 64.0 - black / 64.0;
                const double ev2 = log2(sig > 1 ? sig : 1);
                c_log2sig[i] = ev2;
                double t = ev2 - 4;
                t = t < 0 ? 0 : (t > 4 ? 4 : t);
                c_fullres[i] = (-cos(t * M_PI / 4) + 1) / 2;
            }
        });
        int thr = N20;
        for (int i = N20 - 1; i >= 0 && g_curves.fullres[i] > 0.8; i--) thr = i;
        for (int i = 0; i < thr; i++)
            if (g_curves.fullres[i] > 0.8) { set_error("dual ISO: the full-res curve is not monotone"); return MLVFS_AMD_ERR_ARG; }
        g_curves.thr = thr;
        int lo = 0, hi = N20;
        while (lo < N20 && g_curves.fullres[lo] == g_curves.fullres[0]) lo++;
        while (hi > 0 && g_curves.fullres[hi - 1] == g_curves.fullres[N20 - 1]) hi--;
        g_curves.fr_lo = lo; g_curves.fr_hi = hi > lo ? hi : lo;
        g_curves.black = black;
        g_curves.version++;
    }
    if (T.curves_ver != g_curves.version || !T.fullres) {
        T.retire(T.fullres); T.retire(T.log2sig);
        T.fullres = T.log2sig = nullptr;
        MLV_HIP(hipMalloc(&T.fullres, sizeof(double) * N20));
        MLV_HIP(hipMalloc(&T.log2sig, sizeof(double) * N20));
        MLV_HIP(hipMemcpy(T.fullres, g_curves.fullres.data(), sizeof(double) * N20, hipMemcpyHostToDevice));
        MLV_HIP(hipMemcpy(T.log2sig, g_curves.log2sig.data(), sizeof(double) * N20, hipMemcpyHostToDevice));
        T.curves_ver = g_curves.version;
    }
    if (T.packed_curves_ver != g_curves.version || T.packed_mix_ver != g_lut_mix.version || !T.mix_pair) {
        T.retire(T.mix_pair);
        T.mix_pair = nullptr;
        std::vector<int2> mp((size_t)24 * EVR);
        for (int i = 0; i < 24 * EVR; i++) { const int r = g_lut_mix.ev2raw[i]; mp[i] = int2{ r, g_lut_mix.raw2ev[r] }; }
        MLV_HIP(hipMalloc(&T.mix_pair, sizeof(int2) * 24 * EVR));
        MLV_HIP(hipMemcpy(T.mix_pair, mp.data(), sizeof(int2) * 24 * EVR, hipMemcpyHostToDevice));
        T.packed_curves_ver = g_curves.version; T.packed_mix_ver = g_lut_mix.version;
    }
    L->mix_pair = T.mix_pair + 10 * EVR;
    L->fullres_thr = g_curves.thr;
    L->fr_lo = g_curves.fr_lo; L->fr_hi = g_curves.fr_hi; L->fr_lo_val = g_curves.fullres[0]; L->fr_hi_val = g_curves.fullres[N20 - 1];
    if (T.same_mix_ver != g_lut_mix.version || T.same_blend_ver != g_lut_blend.version) {
        T.blend_is_mix = g_lut_mix.raw2ev == g_lut_blend.raw2ev;
        T.same_mix_ver = g_lut_mix.version; T.same_blend_ver = g_lut_blend.version;
    }
    L->blend_is_mix = T.blend_is_mix;
    L->interp_raw2ev = T.raw2ev[ik]; L->interp_ev2raw = T.ev2raw[ik] + 10 * EVR;
    L->mix_raw2ev = T.raw2ev[1];    L->mix_ev2raw = T.ev2raw[1] + 10 * EVR;
    L->blend_raw2ev = T.raw2ev[2];  L->blend_ev2raw = T.ev2raw[2] + 10 * EVR;
    L->fullres_curve = T.fullres;
    L->log2sig = T.log2sig;
    return MLVFS_AMD_OK;
}

static thread_local double t_last_scalars[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };

// ------------------------------------------------------------------ per-thread work buffers
struct DiWork {
    void *base = nullptr;
    size_t cap = 0;
    DiWork() = default;
    DiWork(const DiWork &) = delete;
    ~DiWork() { if (base) (void)hipFree(base); }             // with the host thread that owned it
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return MLVFS_AMD_OK;
        if (base) (void)hipFree(base);
        base = nullptr; cap = 0;
        void *grown = nullptr;
        MLV_HIP(hipMalloc(&grown, bytes));
        base = grown;
        cap = bytes;
        return MLVFS_AMD_OK;
    }
};
static thread_local std::map<int, DiWork> t_work;
struct PinnedWork {              // page-locked host landing zone for the summaries the host decisions read
    void *base = nullptr;
    size_t cap = 0;
    PinnedWork() = default;
    PinnedWork(const PinnedWork &) = delete;
    ~PinnedWork() { if (base) (void)hipHostFree(base); }
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return MLVFS_AMD_OK;
        if (base) (void)hipHostFree(base);
        base = nullptr; cap = 0;
        MLV_HIP(hipHostMalloc(&base, bytes, hipHostMallocDefault));
        cap = bytes;
        return MLVFS_AMD_OK;
    }
};
static thread_local std::map<int, PinnedWork> t_pinned;
// AMaZE tile planes: zeroed when (re)allocated or when the plane geometry changes, like the reference's calloc per call
// (for one geometry every tile rewrites exactly what it wrote before, so what must read as zero stays zero)
struct AmazeWork : DiWork { int w = 0, h = 0; };
static thread_local std::map<int, AmazeWork> t_amaze;
static int amaze_scratch_for(int device, int w, int h, hipStream_t s, float **out)
{
    AmazeWork &aw = t_amaze[device];
    const size_t need = amaze_scratch_bytes(w, h);
    if (need > aw.cap || aw.w != w || aw.h != h) {
        int rc = aw.ensure(need);
        if (rc) return rc;
        MLV_HIP(hipMemsetAsync(aw.base, 0, aw.cap, s));
        aw.w = w; aw.h = h;
    }
    *out = (float *)aw.base;
    return MLVFS_AMD_OK;
}

// ------------------------------------------------------------------ candidate slopes of match_exposures
// test_a = pow(2, -ev) for ev = 0, 0.002, ... < 6 with ev ACCUMULATED like the reference's loop (hdr.c:752-755): the same for
// every frame, so built once with the node's libm and kept on each device
static std::vector<double> g_ta;
static std::map<int, double *> g_dev_ta;
static int candidate_slopes(int device, const double **d_ta, const double **h_ta, int *ncand)
{
    std::lock_guard<std::mutex> lk(g_di_mutex);
    if (g_ta.empty())
        for (double ev = 0; ev < 6; ev += 0.002) {
            if (g_ta.size() >= 3100) break;
            g_ta.push_back(pow(2, -ev));
        }
    double *&d = g_dev_ta[device];
    if (!d) {
        MLV_HIP(hipMalloc(&d, sizeof(double) * g_ta.size()));
        MLV_HIP(hipMemcpy(d, g_ta.data(), sizeof(double) * g_ta.size(), hipMemcpyHostToDevice));
    }
    *d_ta = d; *h_ta = g_ta.data(); *ncand = (int)g_ta.size();
    return MLVFS_AMD_OK;
}

// MLVFS_AMD_DI_TIMING=1: wall-clock of the host-visible phases of one conversion on stderr (tuning aid)
struct PhaseTimer {
    bool on;
    std::chrono::steady_clock::time_point t0;
    PhaseTimer() : on(getenv("MLVFS_AMD_DI_TIMING") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void mark(const char *what)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[di] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

// ------------------------------------------------------------------ from a frame's integer decisions to its parameters
// What the reference decides after its analysis passes, in its order, with its progress lines: the checks that make it give up
// (return 0: the frame is left alone), the exposure fit's scalars and the libm values derived from them.  `d` comes from the
// device (k_di_decide_*), `ta` are the candidate slopes.
struct DiOptions { int interp_method, use_fullres, use_alias_map, chroma_smooth_method; };

// The weight of the half-res mix (hdr.c:1562-1575) is a function of the bright value b alone: t = log2(max(b / 64 - black / 64, 1)) +
// corr_ev - (max_ev - overlap), clamped to [0, overlap].  log2 is monotone, so the clamps cut the 20-bit range in three: t = 0 below
// mix_lo, t = overlap from mix_hi on, and only the band between needs the table and the cosine.  The two bounds by bisection, with
// the expressions of the kernel (and of the table: prepare_tables).
static void mix_band(DiParams &p)
{
    const int black = p.black20;
    auto t_of = [&](int i) {
        const double sig = i / 64.0 - black / 64.0;
        const double ev = log2(sig > 1 ? sig : 1) + p.corr_ev;
        return ev - (p.max_ev - p.overlap);
    };
    auto first = [&](auto pred) {                              // first i in [0, 2^20] with pred(i), pred monotone false -> true
        int lo = 0, hi = 1 << 20;
        while (lo < hi) { const int mid = (lo + hi) / 2; if (pred(mid)) hi = mid; else lo = mid + 1; }
        return lo;
    };
    p.mix_lo = first([&](int i) { return t_of(i) > 0; });
    p.mix_hi = first([&](int i) { return !(t_of(i) < p.overlap); });
}

static int finish_decisions(const DiDecide &d, const double *ta, int w, int H, int black14, const DiOptions &o, DiParams *out, double scalars[8])
{
    DiParams p{};
    *out = p;                                                             // h = 0: leave the frame alone
    if (!d.check_ok) return 0;                                          // hdr_check, hdr.c:432-438
    const bool amaze = o.interp_method == 0;
    const bool rggb = d.rggb != 0;
    const int ay1 = rggb ? 0 : 1, h = rggb ? H : H - 1;
    const int *is_bright = d.is_bright;
    printf("ISO pattern     : %c%c%c%c %s\n", is_bright[0] ? 'B' : 'd', is_bright[1] ? 'B' : 'd', is_bright[2] ? 'B' : 'd',
           is_bright[3] ? 'B' : 'd', "RGGB");
    if (is_bright[0] + is_bright[1] + is_bright[2] + is_bright[3] != 2) { printf("Bright/dark detection error\n"); return 0; }
    if (is_bright[0] == is_bright[2] || is_bright[1] == is_bright[3]) { printf("Interlacing method not supported\n"); return 0; }
    printf("White levels    : %d %d\n", d.white_dark, d.white_bright);
    const int black = black14 * 64, white = d.white_dark * 64, white_bright = d.white_bright * 64;
    printf("Noise levels    : %.02f %.02f %.02f %.02f (14-bit)\n", 8.0, 8.0, 8.0, 8.0);
    const int dark_noise = 8 * 64;
    const double dark_noise_ev = 3.0 + 6, bright_noise_ev0 = 3.0 + 6;
    const bool cs = o.chroma_smooth_method == 2 || o.chroma_smooth_method == 3 || o.chroma_smooth_method == 5;
    p.w = w; p.h = h; p.ay1 = ay1;
    p.is_bright_bits = is_bright[0] | (is_bright[1] << 1) | (is_bright[2] << 2) | (is_bright[3] << 3);
    p.black20 = black; p.white20 = white;
    p.match_white20 = std::min(white, white_bright);
    p.dark_noise = dark_noise;
    p.use_fullres = o.use_fullres; p.use_alias_map = o.use_alias_map; p.chroma_smooth = cs ? o.chroma_smooth_method : 0;
    // ---- match_exposures (hdr.c:638-823)
    if (d.n <= 0) { printf("Doesn't look like interlaced ISO\n"); return 0; }
    double a = 0, b = 0;
    if (d.hi_n > 0 && d.best >= 0) { a = ta[d.best]; b = d.dmed - d.bmed * a; }
    const double b20 = b * 16;
    p.a = a; p.b20 = b20;
    p.white_darkened = (int)((p.match_white20 - black + b20) * a + black);
    const double factor = 1 / a;
    if (factor < 1.2 || !std::isfinite(factor)) { printf("Doesn't look like interlaced ISO\n"); return 0; }
    const double corr_ev = log2(factor);
    printf("ISO difference  : %.2f EV (%d)\n", log2(factor), (int)round(factor * 100));
    printf("Black delta     : %.2f\n", b / 4);
    const double lowiso_dr = log2(white - black) - dark_noise_ev, highiso_dr = log2(white_bright - black) - bright_noise_ev0;
    printf("Dynamic range   : %.02f (+) %.02f => %.02f EV (in theory)\n", lowiso_dr, highiso_dr, highiso_dr + corr_ev);
    printf("Interpolation   : %s\n", amaze ? "amaze-edge" : "mean23");
    if (o.use_fullres) printf("Full-res reconstruction...\n");
    // ---- mix_images preconditions (hdr.c:1539-1556)
    double overlap = lowiso_dr - corr_ev;
    overlap -= std::min(3.0, overlap - 3);
    printf("ISO overlap     : %.1f EV (approx)\n", overlap);
    if (overlap < 0.5) { printf("Overlap error\n"); return 0; }
    if (overlap < 2) printf("Overlap too small, use a smaller ISO difference for better results.\n");
    printf("Half-res blending...\n");
    p.corr_ev = corr_ev; p.overlap = overlap;
    p.max_ev = log2(white / 64 - black / 64);
    mix_band(p);
    scalars[0] = rggb; scalars[1] = is_bright[0] * 8 + is_bright[1] * 4 + is_bright[2] * 2 + is_bright[3];
    scalars[2] = white; scalars[3] = white_bright; scalars[4] = a; scalars[5] = b; scalars[6] = corr_ev; scalars[7] = p.white_darkened;
    if (o.chroma_smooth_method) printf("Chroma smoothing...\n");
    if (o.chroma_smooth_method && !cs) fprintf(stderr, "Unsupported chroma smooth method\nUnsupported chroma smooth method\n");
    if (o.use_alias_map) printf("Building alias map...\nFiltering alias map...\nSmoothing alias map...\n");
    printf("Final blending...\n");
    *out = p;
    return 1;
}

// squeezed row map of the AMaZE path (hdr.c:977-1026): dark rows from the top, bright rows from h/4*2; rows that do not fit are
// dropped.  sq[0..h): squeezed row an image row is written to (-1 none); sq[hs..hs+h): the row it is looked up at (0 if none)
static void squeezed_rows(const DiParams &p, int *sq, int hs)
{
    const int h = p.h;
    auto is_bright = [&](int y) { return (p.is_bright_bits >> (y & 3)) & 1; };
    for (int y = 0; y < hs; y++) { sq[y] = -1; sq[hs + y] = 0; }
    for (int pass = 0; pass < 2; pass++) {
        int yh = -1;
        for (int y = 0; y < h; y++) {
            if (is_bright(y) != pass) continue;
            if (yh < 0) yh = pass ? h / 4 * 2 + y : y;
            sq[y] = yh; sq[hs + y] = yh;
            yh++;
            if (pass && yh >= h) break;
        }
    }
    // the two exposures may claim the same squeezed row (odd geometries): the later writer, a bright row, wins
    std::vector<int> owner((size_t)h, -1);
    for (int pass = 0; pass < 2; pass++)
        for (int y = 0; y < h; y++)
            if (is_bright(y) == pass && sq[y] >= 0) owner[sq[y]] = y;
    for (int y = 0; y < h; y++)
        if (sq[y] >= 0 && owner[sq[y]] != y) sq[y] = -1;
    for (int y = 0; y < hs; y++) sq[2 * hs + y] = y < h ? owner[y] : 0;         // squeezed row -> its source row, -1: no exposure lands on it (stays zero)
}

// AMaZE tile planes of a batch: frame f's blocks at f * stride; a slot is zeroed when it is (re)allocated or when the rows of the
// frame in it change (RGGB and GBRG frames of one clip differ by a row), like the reference's calloc per call
struct AmazeSlots : DiWork { int w = 0, H = 0; std::vector<int> slot_h; size_t stride = 0; };
static thread_local std::map<int, AmazeSlots> t_amaze_slots;

// ------------------------------------------------------------------ the conversion of a batch of device frames
// d_frames: nframes frames of w x H uint16 in HBM, img_stride bytes apart, converted in place.  results[f] = 1 converted, 0 not dual
// ISO / failed like the reference (the frame is untouched).  Returns 0, or < 0 on an error of the library.
// Host synchronisations: ONE between the analysis kernels and the conversion kernels (the decisions of all frames come back in
// one copy: the libm scalars and the reference's progress lines are the host's), one at the end (the interpolation's statistics,
// which the reference prints).  With `fh` (the drop-in symbol: one frame) one more in front: the focus / bad pixel repairs that
// hdr.c:1943-1947 makes between the check and the analysis need the check's verdict.
int cr2hdr20_batch(ThreadCtx *c, struct frame_headers *fh, void *d_frames, size_t img_stride, int nframes, int w, int H, int black14,
                   int white14, const DiOptions &o, int bad_pixels_mode, hipStream_t stream, int *results, bool *frame_touched)
{
    if (frame_touched) *frame_touched = false;
    for (int f = 0; f < nframes; f++) results[f] = 0;
    PhaseTimer pt;
    if (w <= 0 || H <= 8 || nframes <= 0) return 0;
    if (fh && nframes != 1) { set_error("cr2hdr20: frame headers go with a single frame"); return MLVFS_AMD_ERR_ARG; }
    const size_t N = (size_t)w * H, S = (N + 63) / 64 * 64, NF = (size_t)nframes;
    const double *d_evf = nullptr;
    int rc = prepare_tables(c->dev->id, 0, 0, 1, nullptr, &d_evf);
    if (rc) return rc;
    const double *d_ta = nullptr, *h_ta = nullptr;
    int ncand = 0;
    rc = candidate_slopes(c->dev->id, &d_ta, &h_ta, &ncand);
    if (rc) return rc;

    // ---- work buffer layout (per-frame blocks, frame f at f * its stride)
    const int nsx = (w + 2) / 3, nsy_max = H / 3 + 2;
    const size_t ns = ((size_t)nsx * nsy_max + 63) / 64 * 64;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o2 = off; off += up(bytes); return o2; };
    const size_t hist_stride = up(sizeof(unsigned) * DI_D_WORDS) / 4, derived_stride = up(sizeof(unsigned) * di_derived_words()) / 4;
    const size_t hi_stride = up(ns / 25 * 4 + 65536) / 4;
    const size_t o_hist = take(NF * hist_stride * 4), o_check = take(NF * 2 * 8), o_derived = take(NF * derived_stride * 4),
                 o_ds = take(NF * ns * 4), o_bs = take(NF * ns * 4), o_hbd = take(NF * 2 * sizeof(unsigned) * DI_HIST_N),
                 o_hi = take(NF * 2 * hi_stride * 4), o_score = take(NF * 4 * 3104), o_rows = take(NF * (size_t)nsy_max * 12),
                 o_dd = take(NF * sizeof(DiDecide)), o_pp = take(NF * sizeof(DiParams));
    const size_t o_raw = take(NF * S * 4), o_dark = take(NF * S * 4), o_bright = take(NF * S * 4), o_full = take(NF * S * 4),
                 o_half = take(NF * S * 4), o_over = take(NF * S * 2), o_amap = take(NF * S * 2), o_aux = take(NF * S * 2),
                 o_amap2 = take(NF * S * 2);
    // an unknown method only logs in the reference (hdr.c:1518) and leaves the "smoothed" copies unsmoothed
    const bool cs = o.chroma_smooth_method == 2 || o.chroma_smooth_method == 3 || o.chroma_smooth_method == 5;
    const size_t cells_stride = ((size_t)3 * (H / 2) * (w / 2) + 63) / 64 * 64;
    const size_t o_full_s = take(cs ? NF * S * 4 : 0), o_half_s = take(cs ? NF * S * 4 : 0), o_cells = take(cs ? NF * cells_stride * 4 : 0);
    const bool amaze = o.interp_method == 0;
    const size_t o_cfa = take(amaze ? NF * S * 4 : 0), o_red = take(amaze ? NF * S * 4 : 0), o_green = take(amaze ? NF * S * 4 : 0),
                 o_blue = take(amaze ? NF * S * 4 : 0), o_ev = take(amaze ? NF * S * 12 : NF * S * 4), o_gray = take(amaze ? NF * S * 4 : 0), o_dir = take(amaze ? NF * S : 0),
                 o_sq = take(amaze ? NF * (size_t)H * 12 : 0), o_stats = take(NF * 16 * DI_STAT_SLOTS);
    DiWork &wk = t_work[c->dev->id];
    rc = wk.ensure(off);
    if (rc) return rc;
    uint8_t *B = (uint8_t *)wk.base;

    // page-locked landing zone: every copy of this path starts or ends here (a copy from or to pageable memory waits inside the
    // runtime for the stream to reach it, and the calls of the other host threads wait with it)
    const size_t ph_dd = 0, ph_pp = ph_dd + up(NF * sizeof(DiDecide)), ph_sq = ph_pp + up(NF * sizeof(DiParams)),
                 ph_st = ph_sq + up(NF * (size_t)H * 12), ph_check = ph_st + up(NF * 16 * DI_STAT_SLOTS), ph_end = ph_check + up(NF * 16);
    PinnedWork &pw = t_pinned[c->dev->id];
    rc = pw.ensure(ph_end);
    if (rc) return rc;
    uint8_t *PH = (uint8_t *)pw.base;
    DiDecide *dd = (DiDecide *)(PH + ph_dd);
    DiParams *pp = (DiParams *)(PH + ph_pp);

    DiBatch bt{};
    bt.pp = (const DiParams *)(B + o_pp);
    bt.p0.w = w; bt.p0.h = H; bt.p0.black20 = black14 * 64;       // (the black level is the call's, the same for every frame)
    bt.p0.use_fullres = o.use_fullres; bt.p0.use_alias_map = o.use_alias_map; bt.p0.chroma_smooth = cs ? o.chroma_smooth_method : 0;
    bt.S = S; bt.img_stride = img_stride; bt.nframes = nframes;
    DiDecideBuffers D{};
    D.hist = (const unsigned *)(B + o_hist); D.hist_stride = hist_stride;
    D.check = (const double *)(B + o_check); D.check_stride = 2;
    D.derived = (unsigned *)(B + o_derived); D.derived_stride = derived_stride;
    D.hist_bd = (const unsigned *)(B + o_hbd);
    D.rows = (int *)(B + o_rows); D.nsy_max = nsy_max;
    D.score = (const int *)(B + o_score); D.score_stride = 3104; D.ncand = ncand;
    D.dd = (DiDecide *)(B + o_dd); D.pp = (DiParams *)(B + o_pp);

    // ---- hdr_check + all histograms in one pass over every frame
    auto analyse = [&]() {
        return di_launch_analyse(d_frames, w, H, black14, white14, d_evf, (unsigned *)(B + o_hist), (double *)(B + o_check), stream, nframes,
                                 img_stride, hist_stride, 2);
    };
    rc = analyse();
    if (rc) return rc;
    if (fh) {
        // the drop-in symbol: focus / bad pixels are repaired between the check and the analysis (hdr.c:1943-1947), only on frames
        // the check accepts
        double *check = (double *)(PH + ph_check);
        MLV_HIP(hipMemcpyAsync(check, B + o_check, 16, hipMemcpyDeviceToHost, stream));
        MLV_HIP(hipStreamSynchronize(stream));
        pt.mark("analyse (kernel + D2H)");
        if (!(check[0] / check[1] > 0.5)) return 0;                     // hdr_check, hdr.c:432-438
        bool ch1 = false, ch2 = false;
        rc = focus_pixels_device(fh, c, d_frames, 1, &ch1);
        if (rc) return rc;
        if (bad_pixels_mode) {
            rc = bad_pixels_device(fh, c, d_frames, bad_pixels_mode == 2, 1, &ch2);
            if (rc) return rc;
        }
        if (ch1 || ch2) {
            if (frame_touched) *frame_touched = true;       // the reference repairs in place before it can still bail out
            rc = analyse();
            if (rc) return rc;
        }
        D.check_passed = 1;                                 // (the reference checks the frame as it was before the repairs)
    }
    if (o.interp_method != 0 && o.interp_method != 1) { set_error("cr2hdr20_convert_data: unknown interpolation method"); return 0; }
    if (amaze && ((w & 3) || w < 36 || H < 37)) {
        // the reference's SSE2 AMaZE leaves green columns unwritten when w % 4 != 0 and mirrors from row/column 35
        set_error("cr2hdr20_convert_data: the AMaZE interpolation needs a width that is a multiple of 4 and a frame of at least 36x37; frame not converted");
        return 0;
    }

    // ---- the decisions of every frame, on the device: pattern, fields, whites; samples, order statistics; highlight rows; fit
    rc = di_launch_decide_pattern(d_frames, bt, H, black14, D, stream);
    if (!rc) rc = di_launch_subsample(d_frames, bt, nsx, nsy_max, ns, (int *)(B + o_ds), (int *)(B + o_bs), (unsigned *)(B + o_hbd), stream);
    if (!rc) rc = di_launch_decide_quantiles(bt, D, stream);
    if (!rc) rc = di_launch_hi_count((const int *)(B + o_bs), nsx, nsy_max, ns, 0, 0, D.dd, bt, (int *)(B + o_rows), stream);
    if (!rc) rc = di_launch_decide_rows(bt, D, stream);
    if (!rc) rc = di_launch_hi_compact((const int *)(B + o_ds), (const int *)(B + o_bs), nsx, nsy_max, ns, 0, 0, D.dd, bt, (const int *)(B + o_rows),
                                       (int *)(B + o_hi), hi_stride, stream);
    if (!rc) rc = di_launch_score((const int *)(B + o_hi), hi_stride, 0, d_ta, ncand, 0, 0, D.dd, nframes, (int *)(B + o_score), 3104, stream);
    if (!rc) rc = di_launch_decide_fit(bt, D, stream);
    if (rc) return rc;
    MLV_HIP(hipMemcpyAsync(dd, B + o_dd, NF * sizeof(DiDecide), hipMemcpyDeviceToHost, stream));
    MLV_HIP(hipStreamSynchronize(stream));                  // <- the round trip of the batch
    pt.mark("decisions (device) + D2H");

    if (getenv("MLVFS_AMD_DI_DEBUG"))
        for (int f = 0; f < nframes; f++) {
            const DiDecide &d = dd[f];
            fprintf(stderr, "[di] frame %d: check %.6g/%.6g ok %d rggb %d bright %d%d%d%d raw %d %d %d %d whites %d %d n %lld bmed %d b_lo %d b_hi %d dmed %d hi_n %d best %d (%d)\n",
                    f, d.check_sum, d.check_n, d.check_ok, d.rggb, d.is_bright[0], d.is_bright[1], d.is_bright[2], d.is_bright[3], d.bd_raw[0], d.bd_raw[1],
                    d.bd_raw[2], d.bd_raw[3], d.white_dark, d.white_bright, d.n, d.bmed, d.b_lo, d.b_hi, d.dmed, d.hi_n, d.best, d.best_score);
        }
    // ---- host: checks, libm scalars, progress lines; the tables of the first frame that converts
    int nconv = 0;
    bt.nheights = 0;
    DiLuts L{};
    for (int f = 0; f < nframes; f++) {
        double sc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        results[f] = finish_decisions(dd[f], h_ta, w, H, black14, o, &pp[f], sc);
        if (results[f] != 1) continue;
        memcpy(t_last_scalars, sc, sizeof sc);
        if (!nconv++) {
            rc = prepare_tables(c->dev->id, pp[f].black20, pp[f].white20, o.interp_method, &L, &d_evf);
            if (rc) return rc;
        }
        bool seen = false;
        for (int k = 0; k < bt.nheights; k++) seen = seen || bt.heights[k] == pp[f].h;
        if (!seen && bt.nheights < 2) bt.heights[bt.nheights++] = pp[f].h;
    }
    pt.mark("checks + scalars + tables (host)");
    if (nconv == 0) return 0;
    MLV_HIP(hipMemcpyAsync(B + o_pp, pp, NF * sizeof(DiParams), hipMemcpyHostToDevice, stream));

    DiPlanes P{ (uint32_t *)(B + o_raw), (uint32_t *)(B + o_dark), (uint32_t *)(B + o_bright), (uint32_t *)(B + o_full),
                (uint32_t *)(B + o_half), (uint32_t *)(B + o_full_s), (uint32_t *)(B + o_half_s), (uint16_t *)(B + o_over),
                (uint16_t *)(B + o_amap), (uint16_t *)(B + o_aux), (uint16_t *)(B + o_amap2), (int *)(B + o_cells) };
    P.cells_stride = cells_stride;
    if (!amaze) {
        P.ev_red = (int *)(B + o_ev);                        // mean23: raw2ev of the matched frame, written with it
        rc = di_launch_match(d_frames, bt, H, L, P, stream);
        if (rc) return rc;
    }
    if (amaze) {
        int *sq = (int *)(PH + ph_sq);                         // per frame: sq_dst | sq_row | source row of a squeezed row, H entries each
        for (int f = 0; f < nframes; f++) {
            if (results[f] == 1) squeezed_rows(pp[f], sq + (size_t)f * 3 * H, H);
            else memset(sq + (size_t)f * 3 * H, 0, (size_t)3 * H * 4);
        }
        MLV_HIP(hipMemcpyAsync(B + o_sq, sq, NF * (size_t)3 * H * 4, hipMemcpyHostToDevice, stream));
        // AMaZE's tile planes
        AmazeSlots &as = t_amaze_slots[c->dev->id];
        const size_t a_stride = amaze_scratch_bytes(w, H) / sizeof(float);
        if (as.w != w || as.H != H || as.slot_h.size() < NF) {
            rc = as.ensure(a_stride * sizeof(float) * NF);
            if (rc) return rc;
            MLV_HIP(hipMemsetAsync(as.base, 0, a_stride * sizeof(float) * NF, stream));
            as.w = w; as.H = H; as.stride = a_stride;
            as.slot_h.assign(NF, 0);
        }
        for (int f = 0; f < nframes; f++) {
            if (results[f] != 1) continue;
            if (as.slot_h[f] != 0 && as.slot_h[f] != pp[f].h)
                MLV_HIP(hipMemsetAsync((float *)as.base + (size_t)f * a_stride, 0, a_stride * sizeof(float), stream));
            as.slot_h[f] = pp[f].h;
        }
        P.cfa = (float *)(B + o_cfa); P.red = (float *)(B + o_red); P.green = (float *)(B + o_green); P.blue = (float *)(B + o_blue);
        P.ev_red = (int *)(B + o_ev); P.ev_green = P.ev_red + NF * S; P.ev_blue = P.ev_green + NF * S;
        P.gray_ev = (int *)(B + o_gray); P.dir = (uint8_t *)(B + o_dir);
        P.sq_dst = (const int *)(B + o_sq); P.sq_row = P.sq_dst + H;
        P.stats = (unsigned *)(B + o_stats); P.amaze_scratch = (float *)as.base; P.amaze_scratch_stride = a_stride;
        // A batch of 8 or more goes out in parts of 4 frames: the parts' AMaZE one after the other on the caller's stream, everything
        // behind a part's AMaZE (interpolation, alias map, blend: table- and bandwidth-bound) on a second stream, where it runs under
        // the next part's AMaZE, which is bound by instruction issue -- what two host threads with a batch each get, for one.
        // (The first form alternated the PARTS between the two streams.  Same dependencies for two parts, yet its speed depended on
        // the order in which the process had created its streams -- 8.5 or 9.2 ms per batch of 8, each reproducible: the tail then
        // competes with the NEXT part's AMaZE from whichever queue it happens to share, and loses where that queue is served last,
        // exactly as when the tail's stream is given low priority.  tools/hwq_probe.sh.)
        static const bool split_on = [] { const char *e = getenv("MLVFS_AMD_DI_SPLIT"); return !e || atoi(e) != 0; }();
        if (split_on && nframes >= 8) {
            struct Half {
                hipStream_t st = nullptr; hipEvent_t fork = nullptr, amaze_a = nullptr, join = nullptr;
                ~Half() { if (fork) (void)hipEventDestroy(fork); if (amaze_a) (void)hipEventDestroy(amaze_a); if (join) (void)hipEventDestroy(join); if (st) (void)hipStreamDestroy(st); }
            };
            static thread_local std::map<int, Half> t_half;
            Half &hf = t_half[c->dev->id];
            if (!hf.st) {
                // (priority: a knob for experiments.  Which of the queues is served first while AMaZE workgroups wait for a free CU decides
                // whether the tail overlaps at all)
                static const int prio = [] { const char *e = getenv("MLVFS_AMD_DI_TAIL_PRIO"); return e ? atoi(e) : 0; }();
                MLV_HIP(hipStreamCreateWithPriority(&hf.st, hipStreamNonBlocking, prio));
                MLV_HIP(hipEventCreateWithFlags(&hf.fork, hipEventDisableTiming));
                MLV_HIP(hipEventCreateWithFlags(&hf.amaze_a, hipEventDisableTiming));
                MLV_HIP(hipEventCreateWithFlags(&hf.join, hipEventDisableTiming));
            }
            // frames per part: enough complete AMaZE tiles for about four rounds of k_amaze_rows' 256 workgroups (3584x1320: 282 per
            // frame -> 4 frames; 1736x976: 84 -> 13, a batch of 8 stays whole -- in parts of 4 it took 4.4 instead of 3.9 ms)
            const int part_env = [] { const char *e = getenv("MLVFS_AMD_DI_PART"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();   // (read per call: tests)
            int nfx = 0, nfy = 0;
            amaze_rows_extent(w, H, &nfx, &nfy);
            const int per_frame = nfx * nfy > 0 ? nfx * nfy : 1;
            const int part = part_env ? part_env : std::max(4, (1024 + per_frame - 1) / per_frame);
            const int nparts = nframes / part > 1 ? nframes / part : 1;
            MLV_HIP(hipEventRecord(hf.fork, stream));
            MLV_HIP(hipStreamWaitEvent(hf.st, hf.fork, 0));
            // The last part is smaller than the others (3/4 of an even share: 8 frames = 5 + 3): what follows its AMaZE -- interpolation,
            // alias map, blend -- runs with the chip to itself, and the less of that the better (1 037 -> 1 074 conversions/s,
            // profiles/r05/di_experiments.log).  MLVFS_AMD_DI_FIRST=n (two parts): n frames in the first one (experiments).
            static const int first_env = [] { const char *e = getenv("MLVFS_AMD_DI_FIRST"); return e ? atoi(e) : 0; }();
            const int even = nframes / nparts, last = nparts > 1 ? std::max(1, (3 * even + 2) / 4) : nframes, front = nframes - last;
            for (int k = 0; k < nparts; k++) {
                DiBatch bk = bt;
                if (nparts == 1) { bk.f0 = 0; bk.nframes = nframes; }
                else if (k == nparts - 1) { bk.f0 = front; bk.nframes = last; }
                else { bk.f0 = (int)((long long)front * k / (nparts - 1)); bk.nframes = (int)((long long)front * (k + 1) / (nparts - 1)) - bk.f0; }
                if (first_env > 0 && first_env < nframes && nparts == 2) { bk.f0 = k ? first_env : 0; bk.nframes = k ? nframes - first_env : first_env; }
                rc = di_launch_amaze_interp(d_frames, bk, H, L, P, stream, hf.amaze_a, hf.st);
                if (!rc) rc = di_launch_convert(bk, H, L, P, amaze, d_frames, hf.st);
                if (rc) break;
            }
            // the second stream joins the caller's whether or not every part went out: what it has queued reads and writes the batch's
            // planes, and the caller may free or reuse them as soon as this call has returned an error
            const hipError_t ej = hipEventRecord(hf.join, hf.st);
            const hipError_t ew = ej == hipSuccess ? hipStreamWaitEvent(stream, hf.join, 0) : ej;
            if (ew != hipSuccess) {
                (void)hipStreamSynchronize(hf.st);
                if (!rc) { set_error("cr2hdr20_batch: joining the second stream failed: %s", hipGetErrorString(ew)); rc = MLVFS_AMD_ERR_HIP; }
            }
            if (rc) return rc;
        } else {
            rc = di_launch_amaze_interp(d_frames, bt, H, L, P, stream);
            if (rc) return rc;
            rc = di_launch_convert(bt, H, L, P, amaze, d_frames, stream);
            if (rc) return rc;
        }
    } else {
        rc = di_launch_convert(bt, H, L, P, amaze, d_frames, stream);
        if (rc) return rc;
    }
    if (amaze) {
        unsigned *st = (unsigned *)(PH + ph_st);
        MLV_HIP(hipMemcpyAsync(st, B + o_stats, NF * 16 * DI_STAT_SLOTS, hipMemcpyDeviceToHost, stream));
        MLV_HIP(hipStreamSynchronize(stream));               // (also keeps the pinned parameter blocks alive until their uploads have happened)
        for (int f = 0; f < nframes; f++) {
            if (results[f] != 1) continue;
            unsigned long long s4[4] = { 0, 0, 0, 0 };                          // the workgroups count into DI_STAT_SLOTS slots per frame
            for (int k = 0; k < DI_STAT_SLOTS; k++)
                for (int c4 = 0; c4 < 4; c4++) s4[c4] += st[((size_t)f * DI_STAT_SLOTS + k) * 4 + c4];
            printf("AMaZE interpolation ...\nEdge-directed interpolation...\n");
            printf("Semi-overexposed: %.02f%%\n", s4[0] * 100.0 / (s4[0] + s4[1]));
            printf("Deep shadows    : %.02f%%\n", s4[2] * 100.0 / (s4[2] + s4[3]));
        }
    } else MLV_HIP(hipStreamSynchronize(stream));
    pt.mark("match + interpolation + blend");
    for (int f = 0; f < nframes; f++) {
        if (results[f] != 1) continue;
        printf("Noise level     : %.02f (20-bit), ideally %.02f\n", 8.0, 8.0);
        printf("Dynamic range   : %.02f EV (cooked)\n", log2(pp[f].white20 - pp[f].black20) - log2(8.0));
    }
    return 0;
}

// one frame (the drop-in symbol, mlvfs_amd_cr2hdr20_dev): 1 converted, 0 not dual ISO / failed like the reference, < 0 library error
int cr2hdr20_device(ThreadCtx *c, struct frame_headers *fh, void *d_frame, int w, int H, int black14, int white14,
                    int interp_method, int use_fullres, int use_alias_map, int chroma_smooth_method, int bad_pixels_mode,
                    hipStream_t stream, bool *frame_touched)
{
    int result = 0;
    const DiOptions o{ interp_method, use_fullres, use_alias_map, chroma_smooth_method };
    const int rc = cr2hdr20_batch(c, fh, d_frame, (size_t)w * H * 2, 1, w, H, black14, white14, o, bad_pixels_mode, stream, &result, frame_touched);
    return rc < 0 ? rc : result;
}

}  // namespace mlv

using namespace mlv;

extern "C" {

int cr2hdr20_convert_data(struct frame_headers *fh, uint16_t *image_data, int interp_method, int fullres, int use_alias_map,
                          int chroma_smooth, int fix_bad_pixels_mode)
{
    LibcRandGuard rand_guard;                      // HIP code may run: keep the caller's rand() stream out of its reach
    const int w = fh->rawi_hdr.xRes, h = fh->rawi_hdr.yRes;
    ThreadCtx *c = thread_ctx();
    if (!c) return 0;
    // a stage of the drop-in sequence, right behind the unpack (dropin.cpp): on the device copy the unpack left, or on an upload;
    // inside a frame bracket the result stays on the device
    const size_t bytes = (size_t)w * h * 2;
    void *d_frame = nullptr;
    int which = 0;
    bool was_dirty = false;
    if (inplace_stage_begin(c, STAGE_DUALISO, image_data, bytes, &d_frame, &which, &was_dirty)) return 0;
    bool touched = false;
    const int r = cr2hdr20_device(c, fh, d_frame, w, h, fh->rawi_hdr.raw_info.black_level, fh->rawi_hdr.raw_info.white_level,
                                  interp_method, fullres, use_alias_map, chroma_smooth, fix_bad_pixels_mode, c->stream, &touched);
    // not converted: the frame only carries the pixel repairs the reference would have made by now.  A library error (r < 0: a launch
    // may have failed after the blend had begun to rewrite the frame in place) abandons the device copy like every other stage does
    inplace_stage_end(c, STAGE_DUALISO, image_data, bytes, which, was_dirty, r >= 0, r == 1 || touched);
    if (r != 1) return 0;
    fh->rawi_hdr.raw_info.black_level *= 4;                            // hdr.c:1951-1952
    fh->rawi_hdr.raw_info.white_level *= 4;
    return 1;
}

int mlvfs_amd_cr2hdr20_dev(const mlvfs_amd_geom_t *geom, void *d_frame, int interp_method, int fullres, int use_alias_map,
                           int chroma_smooth, void *stream)
{
    ThreadCtx *c = thread_ctx();
    if (!c) return MLVFS_AMD_ERR_HIP;
    return cr2hdr20_device(c, nullptr, d_frame, geom->width, geom->height, geom->black, geom->white, interp_method, fullres,
                           use_alias_map, chroma_smooth, 0, pick_stream(stream, c), nullptr);
}

int mlvfs_amd_cr2hdr20_batch_dev(const mlvfs_amd_geom_t *geom, void *d_frames, size_t stride, int nframes, int interp_method,
                                 int fullres, int use_alias_map, int chroma_smooth, int *results, void *stream)
{
    ThreadCtx *c = thread_ctx();
    if (!c) return MLVFS_AMD_ERR_HIP;
    if (!geom || !d_frames || !results || nframes < 0) { set_error("cr2hdr20_batch: null argument"); return MLVFS_AMD_ERR_ARG; }
    if (nframes > 1 && stride < (size_t)geom->width * geom->height * 2) { set_error("cr2hdr20_batch: stride smaller than a frame"); return MLVFS_AMD_ERR_ARG; }
    const DiOptions o{ interp_method, fullres, use_alias_map, chroma_smooth };
    return cr2hdr20_batch(c, nullptr, d_frames, stride, nframes, geom->width, geom->height, geom->black, geom->white, o, 0,
                          pick_stream(stream, c), results, nullptr);
}

int mlvfs_amd_amaze_demosaic_dev(const float *d_raw, int width, int height, float *d_red, float *d_green, float *d_blue, void *stream)
{
    ThreadCtx *c = thread_ctx();
    if (!c) return MLVFS_AMD_ERR_HIP;
    if ((width & 3) || width < 36 || height < 36) { set_error("amaze_demosaic: width must be a multiple of 4 and the plane at least 36x36"); return MLVFS_AMD_ERR_ARG; }
    hipStream_t s = pick_stream(stream, c);
    float *scratch = nullptr;
    const int rc = amaze_scratch_for(c->dev->id, width, height, s, &scratch);
    if (rc) return rc;
    return amaze_launch(d_raw, width, height, d_red, d_green, d_blue, scratch, s);
}

// How amaze_launch splits a width x height plane (no GPU needed; tests): the first nfx x nfy tiles of the 128-pixel tile grid are
// complete and go through k_amaze_rows.hip, every other tile through k_amaze.hip.
void mlvfs_amd_amaze_rows_extent(int width, int height, int *nfx, int *nfy)
{
    const int before = g_amaze_rows_mode;
    g_amaze_rows_mode = 1;
    int fx = 0, fy = 0;
    amaze_rows_extent(width, height, &fx, &fy);
    g_amaze_rows_mode = before;
    if (nfx) *nfx = fx;
    if (nfy) *nfy = fy;
}

// Test hook: the heads of output-less chains through k_amaze_rows.hip as well (k_amaze_rows.hip: amaze_rows_extra).  mode -1: the
// environment decides (default off), 0 / 1: forced; returns the mode before.  *count (may be null): how many such tiles a
// width x height plane has when the mode is on.
int mlvfs_amd_amaze_rows_extra_mode(int mode, int width, int height, int *count)
{
    const int before = g_amaze_rows_extra_mode;
    if (count) {
        const int rows_before = g_amaze_rows_mode;
        g_amaze_rows_mode = 1; g_amaze_rows_extra_mode = 1;
        *count = amaze_rows_extra(width, height, 0);
        g_amaze_rows_mode = rows_before;
    }
    g_amaze_rows_extra_mode = mode < 0 ? -1 : (mode ? 1 : 0);
    return before;
}

// Debug: the AMaZE demosaic with its tile planes copied out.  mode 0: every tile through k_amaze.hip, d_planes gets the blocks in
// tile order (ty * tiles_x + tx); mode 1: the complete tiles through k_amaze_rows.hip, d_planes gets THEIR planes in the same
// layout, numbered ty * nfx + tx (nfx x nfy complete tiles, returned).  tests/ compare the two plane by plane.
int mlvfs_amd_amaze_debug(const float *d_raw, int width, int height, float *d_red, float *d_green, float *d_blue, int mode,
                          float *d_planes, size_t planes_floats, int *nfx, int *nfy)
{
    ThreadCtx *c = thread_ctx();
    if (!c) return MLVFS_AMD_ERR_HIP;
    if ((width & 3) || width < 36 || height < 36) { set_error("amaze_demosaic: width must be a multiple of 4 and the plane at least 36x36"); return MLVFS_AMD_ERR_ARG; }
    hipStream_t s = c->stream;
    float *scratch = nullptr;
    int rc = amaze_scratch_for(c->dev->id, width, height, s, &scratch);
    if (rc) return rc;
    const int before = g_amaze_rows_mode;
    g_amaze_rows_mode = mode ? 1 : 0;
    int fx = 0, fy = 0;
    amaze_rows_extent(width, height, &fx, &fy);
    if (nfx) *nfx = fx;
    if (nfy) *nfy = fy;
    if (mode && (size_t)fx * fy * AMAZE_TILE_FLOATS > planes_floats) { g_amaze_rows_mode = before; set_error("amaze_debug: plane buffer too small"); return MLVFS_AMD_ERR_ARG; }
    rc = amaze_launch(d_raw, width, height, d_red, d_green, d_blue, scratch, s, 1, 0, 0, nullptr, 0, mode ? d_planes : nullptr);
    g_amaze_rows_mode = before;
    if (rc) return rc;
    if (!mode) {
        const size_t n = std::min(planes_floats, amaze_scratch_bytes(width, height) / sizeof(float));
        MLV_HIP(hipMemcpyAsync(d_planes, scratch, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    MLV_HIP(hipStreamSynchronize(s));
    return MLVFS_AMD_OK;
}

// Debug getter: the global decisions of the calling thread's last full dual-ISO conversion that got as far as the exposure
// match -- {pattern is RGGB, is_bright[0..3] as bits 3..0, white (20 bit), white of the bright rows, a, b of the exposure fit
// (hdr.c:638-823), ISO difference in EV, darkened white} -- so that tests can compare them one by one with the checker's
// instead of only through the pixels they shape.
void mlvfs_amd_dualiso_last_scalars(double out[8]) { for (int i = 0; i < 8; i++) out[i] = t_last_scalars[i]; }

// gives back the work memory the CALLING thread holds for its dual-ISO conversions on every device (plane blocks, AMaZE tile planes,
// page-locked landing zone: ~0.93 GB per frame of its largest batch at 3584x1320); the next conversion allocates again
void mlvfs_amd_dualiso_trim(void)
{
    for (auto &kv : t_work) { if (kv.second.base) { (void)hipDeviceSynchronize(); (void)hipFree(kv.second.base); } kv.second.base = nullptr; kv.second.cap = 0; }
    for (auto &kv : t_amaze_slots) { if (kv.second.base) (void)hipFree(kv.second.base); kv.second.base = nullptr; kv.second.cap = 0; kv.second.w = kv.second.H = 0; kv.second.slot_h.clear(); }
    for (auto &kv : t_amaze) { if (kv.second.base) (void)hipFree(kv.second.base); kv.second.base = nullptr; kv.second.cap = 0; kv.second.w = kv.second.h = 0; }
    for (auto &kv : t_pinned) { if (kv.second.base) (void)hipHostFree(kv.second.base); kv.second.base = nullptr; kv.second.cap = 0; }
}

// test hook: forget the per-black table caches, as a fresh process would
void mlvfs_amd_dualiso_reset(void)
{
    std::lock_guard<std::mutex> lk(g_di_mutex);
    g_lut_interp.black = g_lut_mix.black = g_lut_blend.black = g_lut_amaze.black = -1;
    g_curves.black = -1;
}

}  // extern "C"
