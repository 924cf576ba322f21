// clip.cpp -- per-clip state and the device-resident C API (PART 2 of
// include/mlvfs_amd.h): pixel-map dependency analysis, stripe-coefficient
// computation around the histogram kernels, fused pipeline entry point.
#include "clip.h"
#include <map>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <unordered_map>

namespace mlv {

KernelTimer &kernel_timer()
{
    static thread_local KernelTimer t;
    return t;
}

// ------------------------------------------------------------------ glibc rand()
// TYPE_3 additive feedback generator of glibc's random_r.c (public algorithm; the
// reference calls libc rand() in stripes.c:129-130 and never seeds it, so a fresh
// process starts from srand(1)).  x[i] = x[i-31] + x[i-3] mod 2^32 after a
// Lehmer-seeded warm-up of 310 discarded words; output = x >> 1.
void glibc_rand_stream(uint16_t *out, size_t n, uint64_t skip, unsigned seed)
{
    uint32_t ring[31];
    {
        uint32_t x[34 + 310];
        int32_t word = (int32_t)(seed ? seed : 1);
        x[0] = (uint32_t)word;
        for (int i = 1; i < 31; i++) {
            const int64_t hi = word / 127773, lo = word % 127773;
            int64_t nx = 16807 * lo - 2836 * hi;
            if (nx < 0) nx += 2147483647;
            word = (int32_t)nx;
            x[i] = (uint32_t)word;
        }
        for (int i = 31; i < 34; i++) x[i] = x[i - 31];
        for (int i = 34; i < 34 + 310; i++) x[i] = x[i - 31] + x[i - 3];
        for (int i = 0; i < 31; i++) ring[i] = x[34 + 310 - 31 + i];
    }
    int pos = 0;
    for (uint64_t k = 0; k < skip + n; k++) {
        const uint32_t v = ring[pos] + ring[(pos + 28) % 31];
        ring[pos] = v;
        pos = (pos + 1) % 31;
        if (k >= skip) out[k - skip] = (uint16_t)((v >> 1) % 1024);
    }
}

// ------------------------------------------------------------------ stripes solve
int stripes_solve(const int32_t *hist, const int32_t num[8], int frame_size, int32_t coeffs[8])
{
    for (int j = 0; j < 8; j++) {                                   // stripes.c:218-234
        if (num[j] < frame_size / 128) continue;
        int t = 0;
        for (int k = 0; k < 65536; k++) {
            t += hist[j * 65536 + k];
            if (t >= num[j] / 2) {
                coeffs[j] = (int32_t)(pow(2.0, (double)(k - 32768) / 32768) * 65536);
                break;
            }
        }
    }
    coeffs[0] = coeffs[1] = 65536;
    int needed = 0;
    for (int j = 0; j < 8; j++) {
        const double c = (double)coeffs[j] / 65536;
        if (c < 0.998 || c > 1.002) needed = 1;
    }
    return needed;
}

// ------------------------------------------------------------------ pixel map
static int entry_kind(int x, int y, int w, int h, int rules, int dual_iso)
{
    if (x > 2 && x < w - 3 && y > 2 && y < h - 3) return dual_iso ? 2 : 1;      // cs.c:320-329 / 468-478
    if (rules == 0) return 0;
    const long long i = x + (long long)y * w;
    if (!(i > 0 && i < (long long)w * h)) return 0;                              // cs.c:479
    const bool h_edge = (x >= w - 3 && x < w) || (x >= 0 && x <= 3);
    const bool v_edge = (y >= h - 3 && y < h) || (y >= 0 && y <= 3);
    if (h_edge && !v_edge && !dual_iso) return 3;
    if (v_edge && !h_edge) return 2;
    if (x >= 0 && x <= 3) return 4;
    if (x >= w - 3 && x < w) return 5;
    return 0;
}

static void taps_of_kind(int kind, bool (&used)[12])
{
    for (int t = 0; t < 12; t++) used[t] = false;
    if (kind == 1) for (int t = 0; t < 12; t++) used[t] = true;
    if (kind == 2) for (int t = 0; t < 6; t++) used[t] = true;
    if (kind == 3) for (int t = 6; t < 12; t++) used[t] = true;
    if (kind == 4) used[4] = true;
    if (kind == 5) used[1] = true;
}

static int tap_offset_host(int t, int w)
{
    const int d = (t % 6) < 3 ? (t % 6) - 3 : (t % 6) - 2;
    return t < 6 ? d : d * w;
}

int Clip::set_pixel_map(const int32_t *xy_in, size_t count, int rules_in, int dual_iso_in)
{
    std::lock_guard<std::mutex> lk(mu);
    xy.assign(xy_in, xy_in + 2 * count);
    rules = rules_in;
    dual_iso = dual_iso_in;
    n_entries = 0;
    n_levels = 0;
    if (count == 0) return MLVFS_AMD_OK;

    const int w = g.w, h = g.h;
    const int crop_x = (pan_x + 7) & ~7, crop_y = pan_y & ~1;                   // cs.c:224-225
    std::vector<PixEntry> ent(count);
    std::vector<int> level(count, 0);
    std::unordered_map<long long, int> last;                                     // position -> latest earlier entry
    last.reserve(count * 2);
    int max_level = 0;
    for (size_t m = 0; m < count; m++) {
        const int x = xy[2 * m] - crop_x, y = xy[2 * m + 1] - crop_y;
        PixEntry &e = ent[m];
        e.kind = entry_kind(x, y, w, h, rules, dual_iso);
        e.pos = (e.kind != 0) ? x + y * w : -1;
        e.emit = 0;
        for (int t = 0; t < 12; t++) e.dep[t] = -1;
        if (e.kind == 0) continue;
        bool used[12];
        taps_of_kind(e.kind, used);
        int lv = 0;
        for (int t = 0; t < 12; t++) {
            if (!used[t]) continue;
            auto it = last.find((long long)e.pos + tap_offset_host(t, w));
            if (it != last.end()) {
                e.dep[t] = it->second;
                lv = std::max(lv, level[it->second] + 1);
            }
        }
        level[m] = lv;
        max_level = std::max(max_level, lv);
        last[e.pos] = (int)m;
    }
    for (auto &kv : last) ent[kv.second].emit = 1;                               // the last writer of a position wins

    // stable order by level; remap dependency indices
    std::vector<int> order(count), where(count);
    for (size_t m = 0; m < count; m++) order[m] = (int)m;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return level[a] < level[b]; });
    for (size_t k = 0; k < count; k++) where[order[k]] = (int)k;
    std::vector<PixEntry> sorted(count);
    std::vector<int> off(max_level + 2, 0);
    for (size_t k = 0; k < count; k++) {
        sorted[k] = ent[order[k]];
        for (int t = 0; t < 12; t++)
            if (sorted[k].dep[t] >= 0) sorted[k].dep[t] = where[sorted[k].dep[t]];
        off[level[order[k]] + 1]++;
    }
    for (int l = 0; l <= max_level; l++) off[l + 1] += off[l];

    // per-tile lists for the fused kernel: an entry belongs to every tile whose plane
    // rectangle (tile + 2-cell halo) contains its cell
    const int tx_n = frame_tiles_x(w), ty_n = frame_tiles_y(h);
    std::vector<std::vector<int>> per_tile((size_t)tx_n * ty_n);
    for (size_t k = 0; k < count; k++) {
        const PixEntry &e = sorted[k];
        if (e.kind == 0 || !e.emit) continue;
        const int cx = (e.pos % w) >> 1, cy = (e.pos / w) >> 1;
        for (int ty = 0; ty < ty_n; ty++) {
            if (cy < ty * FRAME_TCH - FRAME_HC || cy >= (ty + 1) * FRAME_TCH + FRAME_HC) continue;
            for (int tx = 0; tx < tx_n; tx++) {
                if (cx < tx * FRAME_TCW - FRAME_HC || cx >= (tx + 1) * FRAME_TCW + FRAME_HC) continue;
                per_tile[(size_t)ty * tx_n + tx].push_back((int)k);
            }
        }
    }
    std::vector<int> tile_off(per_tile.size() + 1, 0), tile_ent;
    for (size_t i = 0; i < per_tile.size(); i++) {
        tile_off[i + 1] = tile_off[i] + (int)per_tile[i].size();
        tile_ent.insert(tile_ent.end(), per_tile[i].begin(), per_tile[i].end());
    }
    if (tile_ent.empty()) tile_ent.push_back(0);

    ThreadCtx *c = thread_ctx();
    if (!c) return MLVFS_AMD_ERR_HIP;
    if (d_entries) { (void)hipFree(d_entries); d_entries = nullptr; }
    if (d_level_off) { (void)hipFree(d_level_off); d_level_off = nullptr; }
    if (d_tile_off) { (void)hipFree(d_tile_off); d_tile_off = nullptr; }
    if (d_tile_ent) { (void)hipFree(d_tile_ent); d_tile_ent = nullptr; }
    MLV_HIP(hipMalloc(&d_tile_off, sizeof(int) * tile_off.size()));
    MLV_HIP(hipMalloc(&d_tile_ent, sizeof(int) * tile_ent.size()));
    MLV_HIP(hipMemcpy(d_tile_off, tile_off.data(), sizeof(int) * tile_off.size(), hipMemcpyHostToDevice));
    MLV_HIP(hipMemcpy(d_tile_ent, tile_ent.data(), sizeof(int) * tile_ent.size(), hipMemcpyHostToDevice));
    MLV_HIP(hipMalloc(&d_entries, sizeof(PixEntry) * count));
    MLV_HIP(hipMalloc(&d_level_off, sizeof(int) * off.size()));
    MLV_HIP(hipMemcpy(d_entries, sorted.data(), sizeof(PixEntry) * count, hipMemcpyHostToDevice));
    MLV_HIP(hipMemcpy(d_level_off, off.data(), sizeof(int) * off.size(), hipMemcpyHostToDevice));
    n_entries = (int)count;
    n_levels = max_level + 1;
    return MLVFS_AMD_OK;
}

int Clip::ensure_patches(int nframes)
{
    const size_t need = (size_t)nframes * (size_t)std::max(n_entries, 1) * sizeof(int2);
    if (need > patch_bytes) {
        if (d_patches) (void)hipFree(d_patches);
        d_patches = nullptr; patch_bytes = 0;
        MLV_HIP(hipMalloc(&d_patches, need));
        patch_bytes = need;
    }
    return MLVFS_AMD_OK;
}

Clip::~Clip()
{
    if (d_entries) (void)hipFree(d_entries);
    if (d_level_off) (void)hipFree(d_level_off);
    if (d_tile_off) (void)hipFree(d_tile_off);
    if (d_tile_ent) (void)hipFree(d_tile_ent);
    if (d_patches) (void)hipFree(d_patches);
    if (d_scratch) (void)hipFree(d_scratch);
}

int Clip::ensure_scratch(size_t bytes)
{
    if (bytes > scratch_bytes) {
        if (d_scratch) (void)hipFree(d_scratch);
        d_scratch = nullptr; scratch_bytes = 0;
        MLV_HIP(hipMalloc(&d_scratch, bytes));
        scratch_bytes = bytes;
    }
    return MLVFS_AMD_OK;
}

// ------------------------------------------------------------------ detection
int Clip::detect_bad_pixels(const void *d_frame, int aggressive, int dual_iso_in, hipStream_t stream)
{
    ThreadCtx *c = thread_ctx();
    if (!c) return MLVFS_AMD_ERR_HIP;
    const int w = g.w, h = g.h;
    const int wpr = (w + 63) / 64;
    const int crop_x = (pan_x + 7) & ~7, crop_y = pan_y & ~1;
    int cap = 1 << 16;
    std::vector<int32_t> host_list;
    for (;;) {
        const size_t mask_b = (size_t)h * wpr * 8, cnt_b = ((size_t)h * 4 + 15) / 16 * 16, list_b = (size_t)cap * 8;
        int rc = ensure_scratch(mask_b + cnt_b + list_b);
        if (rc) return rc;
        uint8_t *base = (uint8_t *)d_scratch;
        int *d_cnt = (int *)(base + mask_b);
        void *d_list = base + mask_b + cnt_b;
        rc = launch_badpix_detect(d_frame, w, h, g.black, aggressive, crop_x, crop_y, base, wpr, d_cnt, d_list, cap,
                                  c->dev->luts, stream);
        if (rc) return rc;
        std::vector<int> cnt(h);
        MLV_HIP(hipMemcpyAsync(cnt.data(), d_cnt, sizeof(int) * h, hipMemcpyDeviceToHost, stream));
        MLV_HIP(hipStreamSynchronize(stream));
        long long total = 0;
        for (int v : cnt) total += v;
        if (total > cap) { cap = (int)total; continue; }
        host_list.resize(2 * (size_t)total);
        if (total) {
            MLV_HIP(hipMemcpyAsync(host_list.data(), d_list, (size_t)total * 8, hipMemcpyDeviceToHost, stream));
            MLV_HIP(hipStreamSynchronize(stream));
        }
        break;
    }
    return set_pixel_map(host_list.data(), host_list.size() / 2, 0, dual_iso_in);
}

int Clip::fix_pixels(void *d_frames, size_t stride, int nframes, hipStream_t stream)
{
    if (n_entries == 0) return MLVFS_AMD_OK;
    ThreadCtx *c = thread_ctx();
    if (!c) return MLVFS_AMD_ERR_HIP;
    int rc = ensure_patches(nframes);
    if (rc) return rc;
    return launch_pixfix(false, d_frames, stride, g.w, g.black, d_entries, d_level_off, n_levels, n_entries, d_patches,
                         d_frames, stride, nframes, c->dev->luts, stream);
}

// ------------------------------------------------------------------ stripes
struct RecheckHost { int hist, a, b, r1, r2; };

int Clip::stripes_compute(const void *d_frame, int frame_size, int rand_mode, hipStream_t stream)
{
    ThreadCtx *c = thread_ctx();
    if (!c) return MLVFS_AMD_ERR_HIP;
    StripesWork wk;
    int rc = wk.init(this, g, 0, g.h);
    if (rc) return rc;
    long long accepted = 0;
    rc = wk.count(d_frame, &accepted, stream);
    if (rc) return rc;

    // the dither stream: two values per accepted call, in raster order
    std::vector<uint16_t> rnd((size_t)(2 * accepted) + 2);
    if (rand_mode == 0) {
        for (long long i = 0; i < 2 * accepted; i++) rnd[i] = (uint16_t)(rand() % 1024);   // the process-global stream
    } else {
        glibc_rand_stream(rnd.data(), (size_t)(2 * accepted), 0, 1);
    }
    void *d_rand = nullptr;
    MLV_HIP(hipMalloc(&d_rand, rnd.size() * 2));
    hipError_t e = hipMemcpyAsync(d_rand, rnd.data(), rnd.size() * 2, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) { (void)hipFree(d_rand); set_error("rand upload failed"); return MLVFS_AMD_ERR_HIP; }

    std::vector<int32_t> hist(8 * 65536);
    int32_t num[8];
    rc = wk.hist_to_host(d_frame, d_rand, 2 * accepted, hist.data(), num, stream);
    (void)hipFree(d_rand);
    if (rc) return rc;

    std::lock_guard<std::mutex> lk(mu);
    needed = stripes_solve(hist.data(), num, frame_size, coef);
    return MLVFS_AMD_OK;
}

// scratch layout for one shard of rows
int StripesWork::init(Clip *owner_, const Geom &g_, int row0_, int row1_)
{
    owner = owner_; g = g_; row0 = row0_; row1 = row1_;
    gpr = stripes_groups_per_row(g.w);
    n_groups = gpr * (row1 - row0);
    nblk = (n_groups + 255) / 256;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    o_counts = 0;
    o_bsum = o_counts + up((size_t)std::max(n_groups, 1));
    o_boff = o_bsum + up(sizeof(int) * (size_t)std::max(nblk, 1));
    o_total = o_boff + up(sizeof(long long) * (size_t)std::max(nblk, 1));
    o_hist = o_total + 256;
    o_num = o_hist + (size_t)8 * 65536 * 4;
    o_nre = o_num + 256;
    o_re = o_nre + 256;
    bytes = o_re + sizeof(RecheckHost) * RECHECK_CAP;
    int rc = owner->ensure_scratch(bytes);
    if (rc) return rc;
    base = (uint8_t *)owner->d_scratch;
    return MLVFS_AMD_OK;
}

int StripesWork::count(const void *d_frame, long long *accepted, hipStream_t stream)
{
    int rc = launch_stripes_count(d_frame, g.w, row0, row1, g.black, g.white, base + o_counts, (int *)(base + o_bsum),
                                  (long long *)(base + o_boff), (long long *)(base + o_total), stream);
    if (rc) return rc;
    MLV_HIP(hipMemcpyAsync(accepted, base + o_total, sizeof(long long), hipMemcpyDeviceToHost, stream));
    MLV_HIP(hipStreamSynchronize(stream));
    return MLVFS_AMD_OK;
}

int StripesWork::hist_dev(const void *d_frame, const void *d_rand, long long n_rand, int *d_hist, int *d_num,
                          hipStream_t stream)
{
    MLV_HIP(hipMemsetAsync(base + o_nre, 0, 4, stream));
    return launch_stripes_hist(d_frame, g.w, row0, row1, g.black, g.white, base + o_counts, (long long *)(base + o_boff),
                               d_rand, n_rand, d_hist, d_num, base + o_re, RECHECK_CAP, (int *)(base + o_nre), stream);
}

// bins the samples the device refused to bin (too close to a bin edge for its log2)
int StripesWork::recheck_into(int32_t *hist_host_or_null, int *d_hist, hipStream_t stream)
{
    int n_re = 0;
    MLV_HIP(hipMemcpyAsync(&n_re, base + o_nre, 4, hipMemcpyDeviceToHost, stream));
    MLV_HIP(hipStreamSynchronize(stream));
    if (n_re > RECHECK_CAP) { set_error("stripes: %d samples need host re-binning (cap %d)", n_re, RECHECK_CAP); return MLVFS_AMD_ERR_ARG; }
    if (n_re == 0) return MLVFS_AMD_OK;
    std::vector<RecheckHost> re(n_re);
    MLV_HIP(hipMemcpy(re.data(), base + o_re, sizeof(RecheckHost) * n_re, hipMemcpyDeviceToHost));
    std::vector<int> idx(n_re);
    for (int i = 0; i < n_re; i++) {
        const double af = re[i].a + re[i].r1 / 1024.0 - 0.5, bf = re[i].b + re[i].r2 / 1024.0 - 0.5;
        const double ev = log2(af / bf);
        int bin = (int)(65536 / 2 + ev * 65536 / 2);
        bin = bin < 0 ? 0 : (bin > 65535 ? 65535 : bin);
        idx[i] = re[i].hist * 65536 + bin;
    }
    if (hist_host_or_null) {
        for (int i : idx) hist_host_or_null[i]++;
    } else {
        // device histogram (multi-GPU path): add one by one, n_re is tiny
        for (int i : idx) {
            int v = 0;
            MLV_HIP(hipMemcpy(&v, d_hist + i, 4, hipMemcpyDeviceToHost));
            v++;
            MLV_HIP(hipMemcpy(d_hist + i, &v, 4, hipMemcpyHostToDevice));
        }
    }
    return MLVFS_AMD_OK;
}

int StripesWork::hist_to_host(const void *d_frame, const void *d_rand, long long n_rand, int32_t *hist, int32_t num[8],
                              hipStream_t stream)
{
    int *d_hist = (int *)(base + o_hist), *d_num = (int *)(base + o_num);
    MLV_HIP(hipMemsetAsync(d_hist, 0, (size_t)8 * 65536 * 4, stream));
    MLV_HIP(hipMemsetAsync(d_num, 0, 32, stream));
    int rc = hist_dev(d_frame, d_rand, n_rand, d_hist, d_num, stream);
    if (rc) return rc;
    MLV_HIP(hipMemcpyAsync(hist, d_hist, (size_t)8 * 65536 * 4, hipMemcpyDeviceToHost, stream));
    MLV_HIP(hipMemcpyAsync(num, d_num, 32, hipMemcpyDeviceToHost, stream));
    MLV_HIP(hipStreamSynchronize(stream));
    return recheck_into(hist, nullptr, stream);
}

}  // namespace mlv

// ==================================================================== C ABI
using namespace mlv;

extern "C" {

mlvfs_amd_clip_t *mlvfs_amd_clip_create(const mlvfs_amd_geom_t *geom)
{
    if (!geom || geom->width <= 0 || geom->height <= 0) { set_error("clip_create: bad geometry"); return nullptr; }
    ThreadCtx *c = thread_ctx();
    if (!c) return nullptr;
    Clip *clip = new Clip;
    clip->g = Geom{ geom->width, geom->height, geom->bpp, geom->black, geom->white };
    clip->pan_x = geom->pan_x;
    clip->pan_y = geom->pan_y;
    clip->device = c->dev->id;
    return reinterpret_cast<mlvfs_amd_clip_t *>(clip);
}

void mlvfs_amd_clip_destroy(mlvfs_amd_clip_t *clip) { delete reinterpret_cast<Clip *>(clip); }

int mlvfs_amd_clip_set_stripes(mlvfs_amd_clip_t *clip_, int needed, const int32_t coeffs[8])
{
    Clip *clip = reinterpret_cast<Clip *>(clip_);
    std::lock_guard<std::mutex> lk(clip->mu);
    clip->needed = needed;
    memcpy(clip->coef, coeffs, sizeof clip->coef);
    return MLVFS_AMD_OK;
}

int mlvfs_amd_clip_get_stripes(const mlvfs_amd_clip_t *clip_, int *needed, int32_t coeffs[8])
{
    const Clip *clip = reinterpret_cast<const Clip *>(clip_);
    if (needed) *needed = clip->needed;
    if (coeffs) memcpy(coeffs, clip->coef, sizeof clip->coef);
    return MLVFS_AMD_OK;
}

int mlvfs_amd_clip_set_pixel_map(mlvfs_amd_clip_t *clip, const int32_t *xy, size_t count, int kind, int dual_iso)
{
    return reinterpret_cast<Clip *>(clip)->set_pixel_map(xy, count, kind, dual_iso);
}

size_t mlvfs_amd_clip_get_pixel_map(const mlvfs_amd_clip_t *clip_, int32_t *xy, size_t cap)
{
    const Clip *clip = reinterpret_cast<const Clip *>(clip_);
    const size_t n = clip->xy.size() / 2;
    if (xy) memcpy(xy, clip->xy.data(), std::min(n, cap) * 8);
    return n;
}

static Geom to_geom(const mlvfs_amd_geom_t *g) { return Geom{ g->width, g->height, g->bpp, g->black, g->white }; }

int mlvfs_amd_unpack_dev(const mlvfs_amd_geom_t *geom, const void *d_packed, size_t packed_stride, void *d_out,
                         size_t out_stride, int nframes, void *stream)
{
    ThreadCtx *c = thread_ctx();
    if (!c) return MLVFS_AMD_ERR_HIP;
    return launch_unpack(d_packed, packed_stride, d_out, out_stride, 0, (uint32_t)geom->width * geom->height, geom->bpp,
                         nframes, pick_stream(stream, c));
}

int mlvfs_amd_chroma_smooth_dev(const mlvfs_amd_geom_t *geom, const void *d_in, void *d_out, size_t stride, int method,
                                int nframes, void *stream)
{
    ThreadCtx *c = thread_ctx();
    if (!c) return MLVFS_AMD_ERR_HIP;
    if (method != 2 && method != 3 && method != 5) { set_error("Unsupported chroma smooth method"); return MLVFS_AMD_ERR_ARG; }
    if (geom->black > 16384) { set_error("Black level too large for processing"); return MLVFS_AMD_ERR_ARG; }
    return launch_frame(c->dev, to_geom(geom), false, d_in, stride, d_out, stride, nframes, method, nullptr, false,
                        nullptr, pick_stream(stream, c));
}

int mlvfs_amd_detect_bad_pixels_dev(mlvfs_amd_clip_t *clip, const void *d_frame, int aggressive, void *stream)
{
    ThreadCtx *c = thread_ctx();
    if (!c) return MLVFS_AMD_ERR_HIP;
    return reinterpret_cast<Clip *>(clip)->detect_bad_pixels(d_frame, aggressive, 0, pick_stream(stream, c));
}

int mlvfs_amd_fix_pixels_dev(mlvfs_amd_clip_t *clip, void *d_frames, size_t stride, int nframes, void *stream)
{
    ThreadCtx *c = thread_ctx();
    if (!c) return MLVFS_AMD_ERR_HIP;
    return reinterpret_cast<Clip *>(clip)->fix_pixels(d_frames, stride, nframes, pick_stream(stream, c));
}

// The shard entry points share one scratch area per calling thread.
static thread_local Clip *t_shard_scratch = nullptr;
static thread_local StripesWork t_work;

int mlvfs_amd_stripes_count_dev(const mlvfs_amd_geom_t *geom, const void *d_frame, int row0, int row1, int64_t *accepted,
                                void *stream)
{
    ThreadCtx *c = thread_ctx();
    if (!c) return MLVFS_AMD_ERR_HIP;
    if (row0 < 0 || row1 > geom->height || row0 > row1) { set_error("stripes shard rows out of range"); return MLVFS_AMD_ERR_ARG; }
    if (!t_shard_scratch) t_shard_scratch = new Clip;
    int rc = t_work.init(t_shard_scratch, to_geom(geom), row0, row1);
    if (rc) return rc;
    long long acc = 0;
    rc = t_work.count(d_frame, &acc, pick_stream(stream, c));
    *accepted = acc;
    return rc;
}

int mlvfs_amd_stripes_hist_dev(const mlvfs_amd_geom_t *geom, const void *d_frame, int row0, int row1, const void *d_rand,
                               int64_t n_rand, void *d_hist, void *d_num, void *stream)
{
    ThreadCtx *c = thread_ctx();
    if (!c) return MLVFS_AMD_ERR_HIP;
    if (!t_shard_scratch || t_work.row0 != row0 || t_work.row1 != row1 || t_work.g.w != geom->width) {
        set_error("stripes_hist_dev must follow stripes_count_dev for the same shard");
        return MLVFS_AMD_ERR_ARG;
    }
    hipStream_t s = pick_stream(stream, c);
    int rc = t_work.hist_dev(d_frame, d_rand, n_rand, (int *)d_hist, (int *)d_num, s);
    if (rc) return rc;
    return t_work.recheck_into(nullptr, (int *)d_hist, s);
}

int mlvfs_amd_stripes_solve(const int32_t *hist, const int32_t num[8], int frame_size, int32_t coeffs[8])
{
    return stripes_solve(hist, num, frame_size, coeffs);
}

int mlvfs_amd_stripes_compute_dev(mlvfs_amd_clip_t *clip, const void *d_frame, int frame_size, int rand_mode, void *stream)
{
    ThreadCtx *c = thread_ctx();
    if (!c) return MLVFS_AMD_ERR_HIP;
    return reinterpret_cast<Clip *>(clip)->stripes_compute(d_frame, frame_size, rand_mode, pick_stream(stream, c));
}

int mlvfs_amd_stripes_apply_dev(const mlvfs_amd_clip_t *clip_, void *d_frames, size_t stride, int nframes, void *stream)
{
    ThreadCtx *c = thread_ctx();
    if (!c) return MLVFS_AMD_ERR_HIP;
    const Clip *clip = reinterpret_cast<const Clip *>(clip_);
    if (!clip->needed) return MLVFS_AMD_OK;                                     // stripes.c:252
    return launch_stripes_apply(d_frames, stride, (size_t)clip->g.w * clip->g.h, clip->g.w, clip->g.black, clip->g.white,
                                clip->coef, nframes, pick_stream(stream, c));
}

int mlvfs_amd_timer_begin(int max_launches)
{
    if (!thread_ctx()) return MLVFS_AMD_ERR_HIP;
    KernelTimer &tm = kernel_timer();
    while ((int)tm.ev.size() < 2 * max_launches) {
        hipEvent_t e;
        MLV_HIP(hipEventCreate(&e));
        tm.ev.push_back(e);
    }
    tm.used = 0;
    tm.on = true;
    return MLVFS_AMD_OK;
}

int mlvfs_amd_timer_end(float *ms, int cap)
{
    KernelTimer &tm = kernel_timer();
    tm.on = false;
    int n = 0;
    for (int i = 0; i + 1 < tm.used && n < cap; i += 2, n++) {
        MLV_HIP(hipEventSynchronize(tm.ev[i + 1]));
        MLV_HIP(hipEventElapsedTime(&ms[n], tm.ev[i], tm.ev[i + 1]));
    }
    tm.used = 0;
    return n;
}

void mlvfs_amd_rand_stream(uint16_t *out, size_t n, uint64_t skip, unsigned seed) { glibc_rand_stream(out, n, skip, seed); }

int mlvfs_amd_process_frames_dev(mlvfs_amd_clip_t *clip_, const void *d_packed, size_t packed_stride, void *d_out,
                                 size_t out_stride, int nframes, int cs_method, int fix_pixels, int apply_stripes,
                                 void *stream)
{
    ThreadCtx *c = thread_ctx();
    if (!c) return MLVFS_AMD_ERR_HIP;
    Clip *clip = reinterpret_cast<Clip *>(clip_);
    hipStream_t s = pick_stream(stream, c);
    if (cs_method != 0 && clip->g.black > 16384) cs_method = 0;                  // get_raw2ev() == NULL: stage skipped
    const bool patch = fix_pixels && clip->n_entries > 0;
    const bool stripes = apply_stripes && clip->needed && (clip->g.w % 8 == 0);
    if (!patch && !stripes && cs_method == 0)
        return launch_unpack(d_packed, packed_stride, d_out, out_stride, 0, (uint32_t)clip->g.w * clip->g.h, clip->g.bpp,
                             nframes, s);
    if (clip->g.bpp != 14) { set_error("fused pipeline needs 14-bit payloads (got %d)", clip->g.bpp); return MLVFS_AMD_ERR_ARG; }
    if (patch) {
        int rc = clip->ensure_patches(nframes);
        if (rc) return rc;
        rc = launch_pixfix(true, d_packed, packed_stride, clip->g.w, clip->g.black, clip->d_entries, clip->d_level_off,
                           clip->n_levels, clip->n_entries, clip->d_patches, nullptr, 0, nframes, c->dev->luts, s);
        if (rc) return rc;
    }
    const PatchView pv{ clip->d_patches, clip->n_entries, clip->d_tile_off, clip->d_tile_ent };
    return launch_frame(c->dev, clip->g, true, d_packed, packed_stride, d_out, out_stride, nframes, cs_method,
                        patch ? &pv : nullptr, stripes, clip->coef, s);
}

// ---- frames that live in HOST memory: chunked, triple-buffered H2D -> fused kernel -> D2H ------------------------
namespace {
struct HostPipe {                       // per host thread and device
    static constexpr int NS = 3;
    hipStream_t s[NS] = { nullptr, nullptr, nullptr };
    void *d_in[NS] = { nullptr, nullptr, nullptr }, *d_out[NS] = { nullptr, nullptr, nullptr }, *d_patch[NS] = { nullptr, nullptr, nullptr };
    size_t cap_in = 0, cap_out = 0, cap_patch = 0;
    int ensure(size_t in_bytes, size_t out_bytes, size_t patch_bytes)
    {
        for (int k = 0; k < NS; k++)
            if (!s[k]) MLV_HIP(hipStreamCreateWithFlags(&s[k], hipStreamNonBlocking));
        auto grow = [&](void *(&buf)[NS], size_t &cap, size_t need) -> int {
            if (need <= cap) return MLVFS_AMD_OK;
            for (int k = 0; k < NS; k++) {
                if (buf[k]) (void)hipFree(buf[k]);
                buf[k] = nullptr;
                MLV_HIP(hipMalloc(&buf[k], need));
            }
            cap = need;
            return MLVFS_AMD_OK;
        };
        int rc = grow(d_in, cap_in, in_bytes);
        if (!rc) rc = grow(d_out, cap_out, out_bytes);
        if (!rc && patch_bytes) rc = grow(d_patch, cap_patch, patch_bytes);
        return rc;
    }
};
thread_local std::map<int, HostPipe> t_pipe;
}  // namespace

int mlvfs_amd_process_frames_host(mlvfs_amd_clip_t *clip_, const void *h_packed, size_t packed_stride, void *h_out,
                                  size_t out_stride, int nframes, int cs_method, int fix_pixels, int apply_stripes,
                                  int chunk_frames)
{
    ThreadCtx *c = thread_ctx();
    if (!c) return MLVFS_AMD_ERR_HIP;
    Clip *clip = reinterpret_cast<Clip *>(clip_);
    if (nframes <= 0) return MLVFS_AMD_OK;
    if (chunk_frames <= 0) chunk_frames = 8;
    if (chunk_frames > nframes) chunk_frames = nframes;
    if (clip->g.bpp != 14) { set_error("host pipeline needs 14-bit payloads (got %d)", clip->g.bpp); return MLVFS_AMD_ERR_ARG; }
    if (cs_method != 0 && clip->g.black > 16384) cs_method = 0;
    const bool patch = fix_pixels && clip->n_entries > 0;
    const bool stripes = apply_stripes && clip->needed && (clip->g.w % 8 == 0);
    const size_t dstride_in = (packed_stride + 15) / 16 * 16, dstride_out = ((size_t)clip->g.w * clip->g.h * 2 + 15) / 16 * 16;
    HostPipe &hp = t_pipe[c->dev->id];
    const size_t patch_bytes = patch ? (size_t)chunk_frames * clip->n_entries * sizeof(int2) : 0;
    int rc = hp.ensure(dstride_in * chunk_frames + 16, dstride_out * chunk_frames, patch_bytes);
    if (rc) return rc;
    const size_t row_bytes = (size_t)clip->g.w * clip->g.h * 2;
    for (int f0 = 0, k = 0; f0 < nframes; f0 += chunk_frames, k++) {
        const int n = std::min(chunk_frames, nframes - f0), slot = k % HostPipe::NS;
        hipStream_t s = hp.s[slot];
        const uint8_t *src = (const uint8_t *)h_packed + (size_t)f0 * packed_stride;
        uint8_t *dst = (uint8_t *)h_out + (size_t)f0 * out_stride;
        // same-stream order makes the slot's buffers safe to reuse: this H2D is queued behind the slot's previous D2H
        if (packed_stride == dstride_in) MLV_HIP(hipMemcpyAsync(hp.d_in[slot], src, packed_stride * n, hipMemcpyHostToDevice, s));
        else MLV_HIP(hipMemcpy2DAsync(hp.d_in[slot], dstride_in, src, packed_stride, packed_stride, n, hipMemcpyHostToDevice, s));
        if (!patch && !stripes && cs_method == 0) {
            rc = launch_unpack(hp.d_in[slot], dstride_in, hp.d_out[slot], dstride_out, 0, (uint32_t)clip->g.w * clip->g.h, 14, n, s);
        } else {
            if (patch) {
                rc = launch_pixfix(true, hp.d_in[slot], dstride_in, clip->g.w, clip->g.black, clip->d_entries, clip->d_level_off,
                                   clip->n_levels, clip->n_entries, hp.d_patch[slot], nullptr, 0, n, c->dev->luts, s);
                if (rc) return rc;
            }
            const PatchView pv{ hp.d_patch[slot], clip->n_entries, clip->d_tile_off, clip->d_tile_ent };
            rc = launch_frame(c->dev, clip->g, true, hp.d_in[slot], dstride_in, hp.d_out[slot], dstride_out, n, cs_method,
                              patch ? &pv : nullptr, stripes, clip->coef, s);
        }
        if (rc) return rc;
        if (out_stride == dstride_out) MLV_HIP(hipMemcpyAsync(dst, hp.d_out[slot], out_stride * n, hipMemcpyDeviceToHost, s));
        else MLV_HIP(hipMemcpy2DAsync(dst, out_stride, hp.d_out[slot], dstride_out, row_bytes, n, hipMemcpyDeviceToHost, s));
    }
    for (int k = 0; k < HostPipe::NS; k++) MLV_HIP(hipStreamSynchronize(hp.s[k]));
    return MLVFS_AMD_OK;
}

void *mlvfs_amd_host_alloc(size_t bytes)
{
    if (!thread_ctx()) return nullptr;
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { set_error("hipHostMalloc(%zu) failed", bytes); return nullptr; }
    return p;
}

void mlvfs_amd_host_free(void *p) { if (p) (void)hipHostFree(p); }

}  // extern "C"
