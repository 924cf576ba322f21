// hdr.cpp -- host side of the dual-ISO preview (mlvfs/hdr.c:40-227): scalar
// analysis between the two kernels of k_hdr.hip, the drop-in hdr_convert_data
// and its device-resident form.
#include "clip.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace mlv {

int launch_hdr_row_hist(const void *d_frame, int w, int h, int white, unsigned *d_hist, hipStream_t stream);
int launch_hdr_preview(const void *d_frame, void *d_out, int w, int h, int black, int white, int dark_row_start, int shadow, double a,
                       double b, size_t shift_count, hipStream_t stream);

struct HdrFit {
    int dark_row_start;
    double a, b;
    uint16_t shadow;
};

// hist32: four histograms of (white+1) 32-bit counts as the GPU produced them.
// Returns 0 when no interlaced exposure pattern is found (hdr.c:98-102).
static int hdr_analyse(const unsigned *hist32, int w_in, int h_in, int black_in, int white_in, HdrFit *fit)
{
    const uint16_t width = (uint16_t)w_in, height = (uint16_t)h_in;
    const uint16_t black = (uint16_t)black_in, white = (uint16_t)white_in;
    const int bins = white + 1;
    // fold to the reference's uint16 counters (they wrap) and rebuild `count`
    std::vector<uint16_t> hist(4 * (size_t)bins);
    for (size_t i = 0; i < hist.size(); i++) hist[i] = (uint16_t)hist32[i];
    uint32_t count[4] = { 0, 0, 0, 0 };
    for (uint16_t y = 4; y < height - 4; y += 5) count[y % 4] += (uint32_t)(width - (y + 1) % 2) / 4;   // histogram.c:58

    int med[4];
    for (int k = 0; k < 4; k++) {                                      // hist_median, histogram.c:64-75
        const uint32_t middle = count[k] / 2;
        uint32_t acc = 0;
        int m = 0;
        for (int i = 0; i < bins; i++) {
            acc += hist[(size_t)k * bins + i];
            if (acc > middle) { m = i; break; }
        }
        med[k] = m - black;
    }
    static const int8_t layouts[4][6] = {                              // hdr.c:66-97: bright pair, dark pair, lo, hi
        { 2, 3, 0, 1, 0, 2 }, { 0, 3, 1, 2, 1, 0 }, { 0, 1, 2, 3, 2, 0 }, { 1, 2, 0, 3, 0, 2 },
    };
    int start = -1;
    for (int k = 0; k < 4 && start < 0; k++) {
        const int8_t *L = layouts[k];
        if (med[L[0]] > med[L[2]] * 2 && med[L[0]] > med[L[3]] * 2 && med[L[1]] > med[L[2]] * 2 && med[L[1]] > med[L[3]] * 2)
            start = k;
    }
    if (start < 0) return 0;
    const uint16_t *lo = &hist[(size_t)layouts[start][4] * bins], *hi = &hist[(size_t)layouts[start][5] * bins];
    auto at = [&](const uint16_t *hg, int i) -> int { return (i >= 0 && i < bins) ? hg[i] : 0; };

    // CDF matching, hdr.c:106-141 (reads past `white` happen only after the sums are
    // complete, where no point can be recorded; they are treated as 0)
    const int min_pix = 100;
    const int cap = width * height / min_pix + 1;
    std::vector<int> px(cap), py(cap);
    std::vector<double> pw(cap);
    int n = 0, acc_lo = 0, acc_hi = 0, raw_lo = 0, prev_acc_hi = 0;
    const int total = (int)count[0];
    for (int raw_hi = 0; raw_hi < total; raw_hi++) {
        acc_hi += at(hi, raw_hi);
        while (acc_lo < acc_hi && raw_lo <= 65536) { acc_lo += at(lo, raw_lo); raw_lo++; }
        if (raw_lo >= white) break;
        if (acc_hi - prev_acc_hi > min_pix) {
            if (acc_hi > total * 1 / 100 && acc_hi < total * 99.99 / 100) {
                const int xb = raw_hi - black;
                px[n] = xb; py[n] = raw_lo - black;
                pw[n] = (double)(xb + 100 > 0 ? xb + 100 : 0);
                n++;
                prev_acc_hi = acc_hi;
            }
        }
    }
    double mx = 0, my = 0, mxy = 0, mx2 = 0, wsum = 0;                 // hdr.c:150-166, same summation order
    for (int i = 0; i < n; i++) {
        mx += px[i] * pw[i];
        my += py[i] * pw[i];
        mxy += (double)px[i] * py[i] * pw[i];
        mx2 += (double)px[i] * px[i] * pw[i];
        wsum += pw[i];
    }
    mx /= wsum; my /= wsum; mxy /= wsum; mx2 /= wsum;
    fit->a = (mxy - mx * my) / (mx2 - mx * mx);
    fit->b = my - fit->a * mx;
    fit->dark_row_start = start;
    fit->shadow = (uint16_t)(int32_t)(black + 1 / (fit->a * fit->a) + fit->b);
    return 1;
}

// d_frame: one 16-bit frame in HBM; d_out: where the converted frame goes (another buffer of the same size); d_hist: scratch of
// 4*(white+1) unsigned.  pre_transform (optional) runs between detection and matching (focus pixels, hdr.c:104).
int hdr_preview_device(const Geom &g, const void *d_frame, void *d_out, size_t max_size, unsigned *d_hist, hipStream_t stream,
                       int (*pre_transform)(void *), void *pre_arg)
{
    const int white16 = (int)(uint16_t)g.white, black16 = (int)(uint16_t)g.black;
    int rc = launch_hdr_row_hist(d_frame, g.w, g.h, white16, d_hist, stream);
    if (rc) return rc;
    std::vector<unsigned> hist(4 * (size_t)(white16 + 1));
    MLV_HIP(hipMemcpyAsync(hist.data(), d_hist, hist.size() * 4, hipMemcpyDeviceToHost, stream));
    MLV_HIP(hipStreamSynchronize(stream));
    HdrFit fit;
    if (!hdr_analyse(hist.data(), g.w, g.h, g.black, g.white, &fit)) {
        fprintf(stderr, "Could not detect dual ISO interlaced lines\n");
        return 0;
    }
    if (pre_transform) {
        rc = pre_transform(pre_arg);
        if (rc) return rc;
    }
    rc = launch_hdr_preview(d_frame, d_out, g.w, g.h, black16, white16, fit.dark_row_start, fit.shadow, fit.a, fit.b,
                            max_size / 2, stream);
    if (rc) return rc;
    return 1;
}

}  // namespace mlv

using namespace mlv;

extern "C" {

void fix_focus_pixels(struct frame_headers *fh, uint16_t *image_data, int dual_iso);

int hdr_convert_data(struct frame_headers *fh, uint16_t *image_data, off_t offset, size_t max_size)
{
    LibcRandGuard rand_guard;                      // HIP code may run: keep the caller's rand() stream out of its reach
    (void)offset;                                                       // unused by the reference as well
    const int w = fh->rawi_hdr.xRes, h = fh->rawi_hdr.yRes;
    const Geom g{ w, h, fh->rawi_hdr.raw_info.bits_per_pixel, fh->rawi_hdr.raw_info.black_level,
                  fh->rawi_hdr.raw_info.white_level };
    ThreadCtx *c = thread_ctx();
    if (!c) return 0;
    // a stage of the drop-in sequence, right behind the unpack (dropin.cpp): it reads the device copy the unpack left (or an upload)
    // and writes the other frame buffer; inside a frame bracket the result stays on the device
    const size_t bytes = (size_t)w * h * 2;
    const size_t hist_bytes = 4 * (size_t)((uint16_t)g.white + 1) * sizeof(unsigned);
    void *d_frame = nullptr, *d_out = nullptr;
    int which = 0;
    bool was_dirty = false;
    if (inplace_stage_begin(c, STAGE_DUALISO, image_data, bytes, &d_frame, &which, &was_dirty, &d_out)) return 0;
    if (c->ensure(0, hist_bytes)) { inplace_stage_end(c, STAGE_DUALISO, image_data, bytes, which, was_dirty, false, false); return 0; }
    struct Pre { struct frame_headers *fh; ThreadCtx *c; void *d_frame; } pre{ fh, c, d_frame };
    // hdr.c:104: focus pixels are repaired (dual-ISO rule) before the exposure matching, once the frame is known to be dual ISO
    auto focus = [](void *p) -> int { Pre *q = (Pre *)p; return focus_pixels_device(q->fh, q->c, q->d_frame, 1, nullptr); };
    const int r = hdr_preview_device(g, d_frame, d_out, max_size, (unsigned *)c->d_b, c->stream, focus, &pre);
    if (r != 1) {                                                       // not dual ISO: the frame is what it was
        inplace_stage_end(c, STAGE_DUALISO, image_data, bytes, which, was_dirty, r == 0, false);
        return 0;
    }
    inplace_stage_end(c, STAGE_DUALISO, image_data, bytes, which ^ 1, was_dirty, true, true);
    fh->rawi_hdr.raw_info.black_level *= 4;                             // hdr.c:223-224
    fh->rawi_hdr.raw_info.white_level *= 4;
    return 1;
}

int mlvfs_amd_hdr_preview_dev(const mlvfs_amd_geom_t *geom, void *d_frame, size_t max_size, void *stream)
{
    ThreadCtx *c = thread_ctx();
    if (!c) return MLVFS_AMD_ERR_HIP;
    const Geom g{ geom->width, geom->height, geom->bpp, geom->black, geom->white };
    const size_t hist_bytes = 4 * (size_t)((uint16_t)g.white + 1) * sizeof(unsigned), bytes = (size_t)g.w * g.h * 2;
    const size_t out_at = (hist_bytes + 255) & ~(size_t)255;
    if (c->ensure(0, out_at + bytes)) return MLVFS_AMD_ERR_HIP;
    hipStream_t s = pick_stream(stream, c);
    const int r = hdr_preview_device(g, d_frame, (uint8_t *)c->d_b + out_at, max_size, (unsigned *)c->d_b, s, nullptr, nullptr);
    if (r == 1) {
        MLV_HIP(hipMemcpyAsync(d_frame, (uint8_t *)c->d_b + out_at, bytes, hipMemcpyDeviceToDevice, s));    // in place for the caller
        // d_b is the thread's scratch, shared by every stream the thread uses: the copy out of it is complete before the next call --
        // possibly on another stream -- may write it again (the fit above has synchronised this stream once already)
        MLV_HIP(hipStreamSynchronize(s));
    }
    return r;
}

// deflicker of main.c:895-906 on a device frame: exposure_bias as the reference stores it in raw_info (numerator = (int)
// (correction * 10000), denominator 10000).  `size_bytes` is what main.c:943 passes: the frame's size in BYTES.
int mlvfs_amd_deflicker_dev(const mlvfs_amd_geom_t *geom, const void *d_frame, size_t size_bytes, int target, int32_t exposure_bias[2],
                            void *stream)
{
    ThreadCtx *c = thread_ctx();
    if (!c) return MLVFS_AMD_ERR_HIP;
    if (geom->bpp < 1 || geom->bpp > 15 || size_bytes < 2) { set_error("deflicker: unsupported bit depth / size"); return MLVFS_AMD_ERR_ARG; }
    const uint32_t white = (1u << geom->bpp) + 1;                     // (uint16_t)((1 << bpp) + 1)
    const uint32_t size = (uint32_t)((size_bytes - 1) / 2);            // elements handed to hist_add behind data + 1
    const uint32_t samples = (size + 1) / 2;                           // i = 0, 2, 4, ... < size
    const size_t hist_bytes = sizeof(unsigned) * ((size_t)white + 1);
    if (c->ensure(0, hist_bytes)) return MLVFS_AMD_ERR_HIP;
    hipStream_t s = pick_stream(stream, c);
    int rc = launch_deflicker_hist(d_frame, samples, white, (unsigned *)c->d_b, s);
    if (rc) return rc;
    std::vector<unsigned> h(white + 1);
    MLV_HIP(hipMemcpyAsync(h.data(), c->d_b, hist_bytes, hipMemcpyDeviceToHost, s));
    MLV_HIP(hipStreamSynchronize(s));
    // hist_median with the reference's 16-bit counters and its count = size / 2 (histogram.c:57,63-76)
    const uint32_t middle = (size / 2) / 2;
    uint32_t cur = 0;
    uint16_t median = 0;
    for (uint32_t i = 0; i <= white; i++) {
        cur += (uint16_t)h[i];
        if (cur > middle) { median = (uint16_t)i; break; }
    }
    const uint16_t black = (uint16_t)geom->black;
    const double correction = log2((double)(target - black) / (median - black));
    exposure_bias[0] = (int32_t)(correction * 10000);
    exposure_bias[1] = 10000;
    return MLVFS_AMD_OK;
}

}  // extern "C"
