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
#include <vector>

namespace mlv {

constexpr int EVR = 32768, N20 = 1 << 20;

// ------------------------------------------------------------------ host tables
struct HostLut {                 // one per consumer function of the reference
    int black = -1;
    unsigned version = 0;
    std::vector<int> raw2ev, ev2raw;     // ev2raw[0] is EV index -10*32768
};

static void host_lut_build(HostLut &L, int black, int white)          // hdr.c:839-874
{
    L.raw2ev.resize(N20);
    L.ev2raw.resize(24 * EVR);
    int *ev2raw = L.ev2raw.data() + 10 * EVR;
    for (int i = 0; i < N20; i++) {
        double signal = i / 64.0 - black / 64.0;
        if (signal < -1023) signal = -1023;
        L.raw2ev[i] = signal > 0 ? (int)round(log2(1 + signal) * EVR) : -(int)round(log2(1 - signal) * EVR);
    }
    for (int i = -10 * EVR; i < 0; i++) {
        const double v = black + 64 - round(64 * pow(2, (double)-i / EVR));
        ev2raw[i] = (int)(v < 0 ? 0 : (v > black ? black : v));
    }
    for (int i = 0; i < 14 * EVR; i++) {
        const double v = black - 64 + round(64 * pow(2, (double)i / EVR));
        ev2raw[i] = (int)(v < black ? black : (v > N20 - 1 ? N20 - 1 : v));
        if (i >= L.raw2ev[white]) ev2raw[i] = std::max(ev2raw[i], white);
    }
    ev2raw[L.raw2ev[0]] = 0;
    L.black = black;
    L.version++;
}

struct HostCurves {              // build_fullres_curve (hdr.c:890-913) + the log2 part of the mix curve (hdr.c:1566)
    int black = -1;
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
        if (H[k]->black != black) host_lut_build(*H[k], black, white);       // white is not part of the key
        int rc = upload_lut(T, k, *H[k]);
        if (rc) return rc;
    }
    if (g_curves.black != black) {
        g_curves.fullres.resize(N20);
        g_curves.log2sig.resize(N20);
        for (int i = 0; i < N20; i++) {
            const double sig = i / 64.0 - black / 64.0;
            const double ev2 = log2(sig > 1 ? sig : 1);
            g_curves.log2sig[i] = ev2;
            double t = ev2 - 4;
            t = t < 0 ? 0 : (t > 4 ? 4 : t);
            g_curves.fullres[i] = (-cos(t * M_PI / 4) + 1) / 2;
        }
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
        MLV_HIP(hipMalloc(&base, bytes));
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

// ------------------------------------------------------------------ host analysis helpers
static int kth_from_hist(const unsigned *hist, int bins, long long k)      // k-th smallest (0-based) bin index
{
    long long acc = 0;
    for (int i = 0; i < bins; i++) { acc += hist[i]; if (acc > k) return i; }
    return bins - 1;
}

static bool is_rggb_from_hist(const unsigned *hb)                             // hdr.c:463-494
{
    double d_rggb = 0, d_gbrg = 0;
    long long acc[4] = { 0, 0, 0, 0 };
    for (int i = 0; i < 16384; i++) {
        for (int k = 0; k < 4; k++) acc[k] += hb[k * 16384 + i];
        d_rggb += (double)std::llabs(acc[1] - acc[2]);
        d_gbrg += (double)std::llabs(acc[0] - acc[3]);
    }
    return d_rggb < d_gbrg;
}

static int bright_dark_from_hist(const unsigned *hg, int black, int is_bright[4])   // hdr.c:540-636
{
    const int white = 10000;
    long long total = 0;
    for (int i = 0; i < 16384; i++) total += hg[i];
    long long acc[4] = { 0, 0, 0, 0 };
    int raw[4] = { 0, 0, 0, 0 }, off[4] = { 0, 0, 0, 0 };
    const int ref_max = (int)(total * 0.998), ref_off = (int)(total * 0.05);
    // The reference walks ref = 0, 1, 2, ... (hdr.c:563-589); between two values of ref at which some class advances
    // nothing changes, so only those values are visited: same assignments, same exit.
    for (long long ref = 0; ref < ref_max;) {
        for (int i = 0; i < 4; i++)
            while (acc[i] < ref && raw[i] < 16384) { acc[i] += hg[i * 16384 + raw[i]]; raw[i]++; }
        if (ref < ref_off && std::max(std::max(raw[0], raw[1]), std::max(raw[2], raw[3])) < black + (white - black) / 4)
            memcpy(off, raw, sizeof off);
        if (raw[0] >= white || raw[1] >= white || raw[2] >= white || raw[3] >= white) break;
        long long next = ref_max;
        for (int i = 0; i < 4; i++)
            if (raw[i] < 16384) next = std::min(next, acc[i] + 1);     // first ref at which class i moves again
        ref = std::max(next, ref + 1);
    }
    for (int i = 0; i < 4; i++) raw[i] -= off[i];
    int s[4];
    memcpy(s, raw, sizeof s);
    std::sort(s, s + 4);
    const double med = (s[1] + s[2]) / 2;
    for (int i = 0; i < 4; i++) is_bright[i] = raw[i] > med;
    printf("ISO pattern     : %c%c%c%c %s\n", is_bright[0] ? 'B' : 'd', is_bright[1] ? 'B' : 'd', is_bright[2] ? 'B' : 'd',
           is_bright[3] ? 'B' : 'd', "RGGB");
    if (is_bright[0] + is_bright[1] + is_bright[2] + is_bright[3] != 2) { printf("Bright/dark detection error\n"); return 0; }
    if (is_bright[0] == is_bright[2] || is_bright[1] == is_bright[3]) { printf("Interlacing method not supported\n"); return 0; }
    return 1;
}

// white_detect (hdr.c:250-300) from the per-row-phase histograms of every 3rd pixel.  The
// reference caps each class list at max_pix entries by overwriting its last slot; `tail`
// holds the last `tail_rows` image rows so the overwritten samples can be taken out again.
static void whites_from_hist(const unsigned *hw, const int is_bright[4], int w, int h, int ay1, const uint16_t *tail,
                             int tail_rows, int *white_dark, int *white_bright)
{
    const int spr = (w + 2) / 3;
    const long long max_pix = (long long)w * h / 2 / 9;
    std::vector<unsigned> hist[2] = { std::vector<unsigned>(32768, 0), std::vector<unsigned>(32768, 0) };
    for (int ph = 0; ph < 4; ph++)
        for (int i = 0; i < 32768; i++) hist[is_bright[ph]][i] += hw[ph * 32768 + i];
    long long total[2] = { 0, 0 };
    for (int y = ay1; y < h; y += 3) total[is_bright[y % 4]] += spr;
    long long kept[2] = { total[0], total[1] };
    // remove the samples the cap overwrote: class indices max_pix-1 .. total-2
    long long idx[2] = { 0, 0 };
    for (int y = ay1; y < h; y += 3) {
        const int c = is_bright[y % 4];
        if (total[c] > max_pix && idx[c] + spr > max_pix - 1 && y >= h - tail_rows) {
            const uint16_t *row = tail + (size_t)(y - (h - tail_rows)) * w;
            for (int sx = 0; sx < spr; sx++) {
                const long long k = idx[c] + sx;
                if (k >= max_pix - 1 && k <= total[c] - 2) {
                    const int v = row[3 * sx] < 32767 ? row[3 * sx] : 32767;
                    hist[c][v]--;
                    kept[c]--;
                }
            }
        }
        idx[c] += spr;
    }
    // k-th smallest of the negated values = (k+1)-th largest value
    auto kth_largest = [&](int c, long long k) {
        if (kept[c] <= 0) return 0;
        if (k > kept[c] - 1) k = kept[c] - 1;
        long long acc = 0;
        for (int i = 32767; i >= 0; i--) { acc += hist[c][i]; if (acc > k) return i; }
        return 0;
    };
    const int w0 = kth_largest(0, 10) - 100, w1 = kth_largest(1, 50) - 1500;
    *white_dark = w0 < 10000 ? 10000 : (w0 > 16383 ? 16383 : w0);
    *white_bright = w1 < 5000 ? 5000 : (w1 > 16383 ? 16383 : w1);
    printf("White levels    : %d %d\n", *white_dark, *white_bright);
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

// ------------------------------------------------------------------ the conversion on a device frame
// d_frame: w x H uint16 frame in HBM, converted in place.  Returns 1 converted, 0 not dual ISO / failed
// like the reference, < 0 on an error of the library.
int cr2hdr20_device(ThreadCtx *c, struct frame_headers *fh, void *d_frame, int w, int H, int black14, int white14,
                    int interp_method, int use_fullres, int use_alias_map, int chroma_smooth_method, int bad_pixels_mode,
                    hipStream_t stream, bool *frame_touched)
{
    if (frame_touched) *frame_touched = false;
    PhaseTimer pt;
    if (w <= 0 || H <= 8) return 0;
    const size_t N = (size_t)w * H;
    const double *d_evf = nullptr;
    int rc = prepare_tables(c->dev->id, 0, 0, 1, nullptr, &d_evf);
    if (rc) return rc;

    // ---- work buffer layout
    const int nsx = (w + 2) / 3, nsy_max = H / 3 + 2;
    const size_t ns = (size_t)nsx * nsy_max;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += up(bytes); return o; };
    const size_t o_hist = take(sizeof(unsigned) * DI_D_WORDS), o_check = take(16), o_ds = take(ns * 4), o_bs = take(ns * 4),
                 o_hb = take(sizeof(unsigned) * DI_HIST_N), o_hd = take(sizeof(unsigned) * DI_HIST_N), o_hi_d = take(ns / 25 * 4 + 65536),
                 o_hi_b = take(ns / 25 * 4 + 65536), o_cand = take(8 * 2 * 3100), o_score = take(4 * 3100),
                 o_rows = take((size_t)nsy_max * 12);
    const size_t o_raw = take(N * 4), o_dark = take(N * 4), o_bright = take(N * 4), o_full = take(N * 4), o_half = take(N * 4),
                 o_over = take(N * 2), o_amap = take(N * 2), o_aux = take(N * 2), o_amap2 = take(N * 2);
    // an unknown method only logs in the reference (hdr.c:1518) and leaves the "smoothed" copies unsmoothed
    const bool cs = chroma_smooth_method == 2 || chroma_smooth_method == 3 || chroma_smooth_method == 5;
    const size_t o_full_s = take(cs ? N * 4 : 0), o_half_s = take(cs ? N * 4 : 0), o_cells = take(cs ? N * 3 : 0);
    const bool amaze = interp_method == 0;
    const size_t o_cfa = take(amaze ? N * 4 : 0), o_red = take(amaze ? N * 4 : 0), o_green = take(amaze ? N * 4 : 0),
                 o_blue = take(amaze ? N * 4 : 0), o_gray = take(amaze ? N * 4 : 0), o_dir = take(amaze ? N : 0),
                 o_sq = take(amaze ? (size_t)H * 8 : 0), o_stats = take(16);
    DiWork &wk = t_work[c->dev->id];
    rc = wk.ensure(off);
    if (rc) return rc;
    uint8_t *B = (uint8_t *)wk.base;

    // ---- hdr_check + all histograms in one pass
    const size_t ph_hist = 0, ph_ds = ph_hist + sizeof(unsigned) * DI_HIST_WORDS, ph_bs = ph_ds + ns * 4, ph_hb = ph_bs + ns * 4,
                 ph_hd = ph_hb + sizeof(unsigned) * DI_HIST_N, ph_tail = ph_hd + sizeof(unsigned) * DI_HIST_N,
                 ph_dev = ph_tail + (size_t)32 * w * 2, ph_edge = ph_dev + (o_check - o_hist) + 256,       // (the check sums travel behind the block)
                 ph_rows = ph_edge + (size_t)13 * w * 2,
                 // (every copy of this path starts or ends in this page-locked block: a copy from or to pageable memory waits inside
                 // the runtime for the stream to reach it, and the calls of the other host threads wait with it)
                 ph_cand = (ph_rows + (size_t)nsy_max * 12 + 63) / 64 * 64, ph_score = ph_cand + 8 * 2 * 3100, ph_sq = ph_score + 4 * 3100,
                 ph_st = ph_sq + (size_t)H * 8, ph_end = ph_st + 64;
    PinnedWork &pw = t_pinned[c->dev->id];
    rc = pw.ensure(ph_end);
    if (rc) return rc;
    uint8_t *PH = (uint8_t *)pw.base;
    unsigned *hist_p = (unsigned *)(PH + ph_hist);
    struct { unsigned *p; unsigned *data() const { return p; } } hist{ hist_p };
    double check[2];
    // device block + the first 5 and last 8 rows of the frame (the row ranges of the three derived histograms differ
    // from "all rows" only there: hdr.c:453 / :519 stop at a multiple of 4, the GBRG variant starts at row 5)
    const int top_rows = std::min(5, H), bot_rows = std::min(8, H - top_rows);
    auto analyse = [&]() -> int {
        int r = di_launch_analyse(d_frame, w, H, black14, white14, d_evf, (unsigned *)(B + o_hist), (double *)(B + o_check), stream);
        if (r) return r;
        unsigned *dev = (unsigned *)(PH + ph_dev);
        uint16_t *edge = (uint16_t *)(PH + ph_edge);
        MLV_HIP(hipMemcpyAsync(dev, B + o_hist, (o_check - o_hist) + 16, hipMemcpyDeviceToHost, stream));      // block + check sums: one copy
        MLV_HIP(hipMemcpyAsync(edge, d_frame, (size_t)top_rows * w * 2, hipMemcpyDeviceToHost, stream));
        MLV_HIP(hipMemcpyAsync(edge + (size_t)top_rows * w, (const uint16_t *)d_frame + (size_t)(H - bot_rows) * w, (size_t)bot_rows * w * 2,
                               hipMemcpyDeviceToHost, stream));
        MLV_HIP(hipStreamSynchronize(stream));
        memcpy(check, (const uint8_t *)dev + (o_check - o_hist), 16);
        unsigned *hb = hist.data() + DI_H_BAYER, *g0 = hist.data() + DI_H_GREEN0, *g1 = hist.data() + DI_H_GREEN1;
        memset(hb, 0, sizeof(unsigned) * 4 * 16384);
        for (int q = 0; q < 4; q++)
            for (int px = 0; px < 2; px++) {
                const unsigned *cl = dev + DI_D_CLASS + (size_t)(q * 2 + px) * 16384;
                unsigned *bay = hb + (size_t)((q & 1) * 2 + px) * 16384;
                for (int v = 0; v < 16384; v++) bay[v] += cl[v];
                if (px == 1 - (q & 1)) memcpy(g0 + (size_t)q * 16384, cl, sizeof(unsigned) * 16384);              // greens: x & 1 != y & 1
                if (px == (q & 1)) memcpy(g1 + (size_t)((q + 3) & 3) * 16384, cl, sizeof(unsigned) * 16384);     // greens of the frame one row lower
            }
        const int R0 = H / 4 * 4, h1 = H - 1, R1 = h1 / 4 * 4;
        auto take_out = [&](int y, const uint16_t *row) {
            const int y1 = y - 1;
            const bool in0 = y < R0, in1 = y1 >= 4 && y1 < R1;
            if (in0 && in1) return;
            for (int x = 0; x < w; x++) {
                const int v = row[x] & 16383;
                if (!in0) { hb[(size_t)((y & 1) * 2 + (x & 1)) * 16384 + v]--; if ((x & 1) != (y & 1)) g0[(size_t)(y & 3) * 16384 + v]--; }
                if (!in1 && (x & 1) == (y & 1)) g1[(size_t)(y1 & 3) * 16384 + v]--;
            }
        };
        for (int y = 0; y < top_rows; y++) take_out(y, edge + (size_t)y * w);
        for (int k = 0; k < bot_rows; k++) take_out(H - bot_rows + k, edge + (size_t)(top_rows + k) * w);
        memcpy(hist.data() + DI_H_WHITE0, dev + DI_D_WHITE0, sizeof(unsigned) * 8 * 32768);
        return MLVFS_AMD_OK;
    };
    rc = analyse();
    if (rc) return rc;
    pt.mark("analyse (kernel + D2H)");
    if (!(check[0] / check[1] > 0.5)) return 0;                         // hdr_check, hdr.c:432-438

    // ---- focus / bad pixels are repaired between the check and the analysis (hdr.c:1943-1947)
    if (fh) {
        bool ch1 = false, ch2 = false;
        rc = focus_pixels_device(fh, c, d_frame, 1, &ch1);
        if (rc) return rc;
        if (bad_pixels_mode) {
            rc = bad_pixels_device(fh, c, d_frame, bad_pixels_mode == 2, 1, &ch2);
            if (rc) return rc;
        }
        if (ch1 || ch2) {
            if (frame_touched) *frame_touched = true;       // the reference repairs in place before it can still bail out
            rc = analyse();
            if (rc) return rc;
        }
    }
    if (interp_method != 0 && interp_method != 1) { set_error("cr2hdr20_convert_data: unknown interpolation method"); return 0; }
    if (amaze && ((w & 3) || w < 36 || H < 37)) {
        // the reference's SSE2 AMaZE leaves green columns unwritten when w % 4 != 0 and mirrors from row/column 35
        set_error("cr2hdr20_convert_data: the AMaZE interpolation needs a width that is a multiple of 4 and a frame of at least 36x37; frame not converted");
        return 0;
    }

    // ---- pattern (hdr.c:1783-1795)
    const bool rggb = is_rggb_from_hist(hist.data() + DI_H_BAYER);
    const int ay1 = rggb ? 0 : 1;
    const int h = rggb ? H : H - 1;
    uint16_t *img = (uint16_t *)d_frame + (rggb ? 0 : w);
    int is_bright[4];
    if (!bright_dark_from_hist(hist.data() + (rggb ? DI_H_GREEN0 : DI_H_GREEN1), black14, is_bright)) return 0;

    // ---- white levels (hdr.c:1806-1810)
    const int tail_rows = std::min(h, 32);
    struct { uint16_t *p; size_t n; uint16_t *data() const { return p; } size_t size() const { return n; } } tail{ (uint16_t *)(PH + ph_tail), (size_t)tail_rows * w };
    MLV_HIP(hipMemcpyAsync(tail.data(), img + (size_t)(h - tail_rows) * w, tail.size() * 2, hipMemcpyDeviceToHost, stream));
    MLV_HIP(hipStreamSynchronize(stream));
    int wd, wb;
    whites_from_hist(hist.data() + (rggb ? DI_H_WHITE0 : DI_H_WHITE1), is_bright, w, h, ay1, tail.data(), tail_rows, &wd, &wb);
    const int black = black14 * 64, white = wd * 64, white_bright = wb * 64;
    printf("Noise levels    : %.02f %.02f %.02f %.02f (14-bit)\n", 8.0, 8.0, 8.0, 8.0);
    const int dark_noise = 8 * 64;
    const double dark_noise_ev = 3.0 + 6, bright_noise_ev0 = 3.0 + 6;

    DiParams p{};
    p.w = w; p.h = h; p.ay1 = ay1;
    p.is_bright_bits = is_bright[0] | (is_bright[1] << 1) | (is_bright[2] << 2) | (is_bright[3] << 3);
    p.black20 = black; p.white20 = white;
    p.match_white20 = std::min(white, white_bright);
    p.dark_noise = dark_noise;
    p.use_fullres = use_fullres; p.use_alias_map = use_alias_map; p.chroma_smooth = cs ? chroma_smooth_method : 0;

    // ---- match_exposures (hdr.c:638-823)
    const int y0 = ay1 + 2;
    const int nsy = (h - 2 > y0) ? (h - 2 - y0 + 2) / 3 : 0;
    pt.mark("pattern + whites (host)");
    rc = di_launch_subsample(img, p, nsx, nsy, (int *)(B + o_ds), (int *)(B + o_bs), (unsigned *)(B + o_hb), (unsigned *)(B + o_hd), stream);
    if (rc) return rc;
    struct UintSpan { unsigned *p; size_t n; unsigned *data() const { return p; } const unsigned *begin() const { return p; } const unsigned *end() const { return p + n; } };
    const UintSpan hb{ (unsigned *)(PH + ph_hb), (size_t)DI_HIST_N }, hd{ (unsigned *)(PH + ph_hd), (size_t)DI_HIST_N };
    static_assert(sizeof(unsigned) * DI_HIST_N % 256 == 0, "the two histograms lie back to back on the device and in the pinned block");
    MLV_HIP(hipMemcpyAsync(hb.data(), B + o_hb, 2 * sizeof(unsigned) * DI_HIST_N, hipMemcpyDeviceToHost, stream));      // hb and hd in one copy
    MLV_HIP(hipStreamSynchronize(stream));
    long long n = 0;
    for (unsigned v : hb) n += v;
    if (n <= 0) { printf("Doesn't look like interlaced ISO\n"); return 0; }
    auto med_k = [](long long m) { return (m & 1) ? m / 2 : m / 2 - 1; };
    const int bmed = kth_from_hist(hb.data(), DI_HIST_N, med_k(n)) - DI_HIST_OFF;
    const int b_lo = kth_from_hist(hb.data(), DI_HIST_N, n * 98 / 100) - DI_HIST_OFF;
    const int b_hi = kth_from_hist(hb.data(), DI_HIST_N, (long long)(n * 99.9 / 100)) - DI_HIST_OFF;
    const int dmed = kth_from_hist(hd.data(), DI_HIST_N, med_k(n)) - DI_HIST_OFF;
    const int nmax = (w + 2) * (h + 2) / 9, hi_nmax = nmax / 50;
    // highlight pairs stay on the device: per-row counts -> what each row contributes.  A row appends its qualifying
    // samples until the list has hi_nmax entries; the reference's `break` only leaves the inner loop (hdr.c:744), so
    // once the cap is reached every further row still appends its first qualifying sample.
    int *rowinfo = (int *)(PH + ph_rows);                       // counts | take | offset
    rc = di_launch_hi_count((const int *)(B + o_bs), nsx, nsy, b_lo, b_hi, (int *)(B + o_rows), stream);
    if (rc) return rc;
    MLV_HIP(hipMemcpyAsync(rowinfo, B + o_rows, (size_t)nsy * 4, hipMemcpyDeviceToHost, stream));
    MLV_HIP(hipStreamSynchronize(stream));
    int hi_n = 0;
    for (int sy = 0; sy < nsy; sy++) {
        const int take = std::min(rowinfo[sy], std::max(hi_nmax - hi_n, 1));
        rowinfo[nsy + sy] = take;
        rowinfo[2 * nsy + sy] = hi_n;
        hi_n += take;
    }
    MLV_HIP(hipMemcpyAsync(B + o_rows + (size_t)nsy * 4, rowinfo + nsy, (size_t)nsy * 8, hipMemcpyHostToDevice, stream));
    rc = di_launch_hi_compact((const int *)(B + o_ds), (const int *)(B + o_bs), nsx, nsy, b_lo, b_hi, (const int *)(B + o_rows) + nsy,
                              (const int *)(B + o_rows) + 2 * nsy, (int *)(B + o_hi_d), (int *)(B + o_hi_b), stream);
    if (rc) return rc;
    pt.mark("subsample + D2H + quantiles");
    double *cand = (double *)(PH + ph_cand);
    int ncand = 0;
    for (double ev = 0; ev < 6; ev += 0.002) {
        const double ta = pow(2, -ev);
        if (ncand >= 3100) break;                             // (3000 candidates: 6 / 0.002)
        cand[2 * ncand] = ta;
        cand[2 * ncand + 1] = dmed - bmed * ta;
        ncand++;
    }
    double a = 0, b = 0;
    if (hi_n > 0) {
        MLV_HIP(hipMemcpyAsync(B + o_cand, cand, (size_t)ncand * 16, hipMemcpyHostToDevice, stream));
        rc = di_launch_score((const int *)(B + o_hi_d), (const int *)(B + o_hi_b), hi_n, (const double *)(B + o_cand), ncand,
                             (int *)(B + o_score), stream);
        if (rc) return rc;
        int *score = (int *)(PH + ph_score);
        MLV_HIP(hipMemcpyAsync(score, B + o_score, (size_t)ncand * 4, hipMemcpyDeviceToHost, stream));
        MLV_HIP(hipStreamSynchronize(stream));
        int best = 0;
        for (int k = 0; k < ncand; k++)
            if (score[k] > best) { best = score[k]; a = cand[2 * k]; b = cand[2 * k + 1]; }
    }
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
    if (use_fullres) printf("Full-res reconstruction...\n");

    // ---- mix_images preconditions (hdr.c:1539-1556)
    double overlap = lowiso_dr - corr_ev;
    overlap -= std::min(3.0, overlap - 3);
    printf("ISO overlap     : %.1f EV (approx)\n", overlap);
    if (overlap < 0.5) { printf("Overlap error\n"); return 0; }
    if (overlap < 2) printf("Overlap too small, use a smaller ISO difference for better results.\n");
    printf("Half-res blending...\n");
    p.corr_ev = corr_ev; p.overlap = overlap;
    p.max_ev = log2(white / 64 - black / 64);
    {   // the host decisions of this conversion, for mlvfs_amd_dualiso_last_scalars
        double *sc = t_last_scalars;
        sc[0] = rggb; sc[1] = is_bright[0] * 8 + is_bright[1] * 4 + is_bright[2] * 2 + is_bright[3];
        sc[2] = white; sc[3] = white_bright; sc[4] = a; sc[5] = b; sc[6] = corr_ev; sc[7] = p.white_darkened;
    }

    DiLuts L{};
    pt.mark("score candidates");
    rc = prepare_tables(c->dev->id, black, white, interp_method, &L, &d_evf);
    if (rc) return rc;
    if (chroma_smooth_method) printf("Chroma smoothing...\n");
    if (chroma_smooth_method && !cs) fprintf(stderr, "Unsupported chroma smooth method\nUnsupported chroma smooth method\n");
    if (use_alias_map) printf("Building alias map...\nFiltering alias map...\nSmoothing alias map...\n");
    printf("Final blending...\n");
    DiPlanes P{ (uint32_t *)(B + o_raw), (uint32_t *)(B + o_dark), (uint32_t *)(B + o_bright), (uint32_t *)(B + o_full),
                (uint32_t *)(B + o_half), (uint32_t *)(B + o_full_s), (uint32_t *)(B + o_half_s), (uint16_t *)(B + o_over),
                (uint16_t *)(B + o_amap), (uint16_t *)(B + o_aux), (uint16_t *)(B + o_amap2), (int *)(B + o_cells) };
    pt.mark("tables");
    rc = di_launch_match(img, p, P, stream);
    if (rc) return rc;
    if (amaze) {
        // squeezed row map (hdr.c:977-1026): dark rows from the top, bright rows from h/4*2; rows that do not fit are dropped
        int *sq = (int *)(PH + ph_sq);                         // 2 h entries
        for (int y = 0; y < h; y++) { sq[y] = -1; sq[h + y] = 0; }
        for (int pass = 0; pass < 2; pass++) {
            int yh = -1;
            for (int y = 0; y < h; y++) {
                if (is_bright[y % 4] != pass) continue;
                if (yh < 0) yh = pass ? h / 4 * 2 + y : y;
                sq[y] = yh; sq[h + y] = yh;
                yh++;
                if (pass && yh >= h) break;
            }
        }
        {   // the two exposures may claim the same squeezed row (odd geometries): the later writer, a bright row, wins
            std::vector<int> owner((size_t)h, -1);
            for (int pass = 0; pass < 2; pass++)
                for (int y = 0; y < h; y++)
                    if (is_bright[y % 4] == pass && sq[y] >= 0) owner[sq[y]] = y;
            for (int y = 0; y < h; y++)
                if (sq[y] >= 0 && owner[sq[y]] != y) sq[y] = -1;
        }
        MLV_HIP(hipMemcpyAsync(B + o_sq, sq, (size_t)2 * h * 4, hipMemcpyHostToDevice, stream));
        float *amaze_scratch = nullptr;
        rc = amaze_scratch_for(c->dev->id, w, h, stream, &amaze_scratch);
        if (rc) return rc;
        P.cfa = (float *)(B + o_cfa); P.red = (float *)(B + o_red); P.green = (float *)(B + o_green); P.blue = (float *)(B + o_blue);
        P.gray_ev = (int *)(B + o_gray); P.dir = (uint8_t *)(B + o_dir);
        P.sq_dst = (const int *)(B + o_sq); P.sq_row = P.sq_dst + h;
        P.stats = (unsigned *)(B + o_stats); P.amaze_scratch = amaze_scratch;
        printf("AMaZE interpolation ...\n");
        rc = di_launch_amaze_interp(p, L, P, stream);
        if (rc) return rc;
        unsigned *st = (unsigned *)(PH + ph_st);
        MLV_HIP(hipMemcpyAsync(st, P.stats, 4 * sizeof(unsigned), hipMemcpyDeviceToHost, stream));
        MLV_HIP(hipStreamSynchronize(stream));               // also keeps `sq` alive until the upload has happened
        printf("Edge-directed interpolation...\n");
        printf("Semi-overexposed: %.02f%%\n", st[0] * 100.0 / (st[0] + st[1]));
        printf("Deep shadows    : %.02f%%\n", st[2] * 100.0 / (st[2] + st[3]));
    }
    pt.mark("amaze + edge directions");
    rc = di_launch_convert(p, L, P, amaze, img, stream);
    if (rc) return rc;
    if (pt.on) { (void)hipStreamSynchronize(stream); pt.mark("interp + mix + blend"); }
    printf("Noise level     : %.02f (20-bit), ideally %.02f\n", 8.0, 8.0);
    printf("Dynamic range   : %.02f EV (cooked)\n", log2(white - black) - log2(8.0));
    return 1;
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
    if (drop_resident(c, image_data)) return 0;             // this call rewrites the host frame: no resident copy of it (dropin.cpp)
    const size_t bytes = (size_t)w * h * 2;
    if (c->ensure(bytes, 0)) return 0;
    if (hipMemcpyAsync(c->d_a, image_data, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) { set_error("cr2hdr20: upload failed"); return 0; }
    bool touched = false;
    const int r = cr2hdr20_device(c, fh, c->d_a, w, h, fh->rawi_hdr.raw_info.black_level, fh->rawi_hdr.raw_info.white_level,
                                  interp_method, fullres, use_alias_map, chroma_smooth, fix_bad_pixels_mode, c->stream, &touched);
    if (r != 1) {
        // not converted: the frame only carries the pixel repairs the reference would have made by now
        if (touched) (void)hipMemcpyAsync(image_data, c->d_a, bytes, hipMemcpyDeviceToHost, c->stream);
        (void)hipStreamSynchronize(c->stream);
        return 0;
    }
    if (hipMemcpyAsync(image_data, c->d_a, bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) { set_error("cr2hdr20: download failed"); return 0; }
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

// Debug getter: the global decisions of the calling thread's last full dual-ISO conversion that got as far as the exposure
// match -- {pattern is RGGB, is_bright[0..3] as bits 3..0, white (20 bit), white of the bright rows, a, b of the exposure fit
// (hdr.c:638-823), ISO difference in EV, darkened white} -- so that tests can compare them one by one with the checker's
// instead of only through the pixels they shape.
void mlvfs_amd_dualiso_last_scalars(double out[8]) { for (int i = 0; i < 8; i++) out[i] = t_last_scalars[i]; }

// test hook: forget the per-black table caches, as a fresh process would
void mlvfs_amd_dualiso_reset(void)
{
    std::lock_guard<std::mutex> lk(g_di_mutex);
    g_lut_interp.black = g_lut_mix.black = g_lut_blend.black = g_lut_amaze.black = -1;
    g_curves.black = -1;
}

}  // extern "C"
