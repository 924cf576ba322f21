// k_hdr.hip -- fast dual-ISO preview (replaces hdr_convert_data, mlvfs/hdr.c:40-227).
//
//   k_hdr_row_hist   : the four row-phase green histograms (hdr.c:52-60): rows
//                      y = 4, 9, 14, ..., every 4th sample starting at the row's
//                      first green.  32-bit atomics; the host folds them to the
//                      reference's 16-bit counters (histogram.h:30).
//   host (hdr.cpp)   : medians, bright/dark row detection, CDF matching and the
//                      weighted least-squares fit -- a few thousand scalar double
//                      operations whose summation order must be the reference's.
//   k_hdr_preview    : the per-row exposure matching (hdr.c:178-215) and the final
//                      << 2 (hdr.c:217-222).  The reference rewrites rows top-down in
//                      place; a pixel reads row y+2 as it was and -- only where a bright
//                      pixel is clipped or a dark one lies in deep shadow -- the REWRITTEN
//                      row y-2 of its own column.  One thread per pixel, out of place: it
//                      walks up its column for as long as that dependency holds (one step
//                      at most on real material, never more than the column), then
//                      rewrites down to its own row.  (Rounds 1-2: a lane per column
//                      walking it top-down, 0.52 ms of dependent loads at 3584x1320.)
//                      Doubles, no FMA contraction -> bit-identical to the x86 path.
#include "clip.h"

namespace mlv {

// one workgroup per sampled row; with `lds_bins` the row is counted in LDS first and its non-empty bins flushed (neighbouring
// samples of a smooth row hit the same few counters: 0.09 ms of same-address global atomics at 3584x1320)
__global__ __launch_bounds__(256) void k_hdr_row_hist(const uint16_t *__restrict__ img, int w, int h, int white,
                                                      unsigned *__restrict__ hist /* [4][white+1] */, int lds_bins)
{
    extern __shared__ unsigned cnt[];
    // blockIdx.x enumerates the sampled rows y = 4 + 5*k
    const int y = 4 + 5 * blockIdx.x;
    if (y >= h - 4) return;
    const int first = (y + 1) % 2;
    const int size = w - first;
    unsigned *hg = hist + (size_t)(y % 4) * (white + 1);
    if (lds_bins) {
        for (int i = threadIdx.x; i < lds_bins; i += blockDim.x) cnt[i] = 0;
        __syncthreads();
    }
    for (int i = 4 * threadIdx.x; i < size; i += 4 * blockDim.x) {
        const int v = img[(size_t)y * w + first + i];
        atomicAdd(lds_bins ? &cnt[v < white ? v : white] : &hg[v < white ? v : white], 1u);
    }
    if (lds_bins) {
        __syncthreads();
        for (int i = threadIdx.x; i < lds_bins; i += blockDim.x) { const unsigned c = cnt[i]; if (c) atomicAdd(&hg[i], c); }
    }
}

__device__ __forceinline__ uint16_t d2u16(double v) { return (uint16_t)(int)v; }

__global__ __launch_bounds__(256) void k_hdr_preview(const uint16_t *__restrict__ in, uint16_t *__restrict__ out, int w, int h, int black,
                                                     int white, int dark_row_start, int shadow, double a, double b, size_t shift_count)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    auto scaled = [&](int p) {
        const double v = (p - black) * a + black + b;                   // hdr.c:198
        return (double)white < v ? (double)white : v;
    };
    auto px = [&](int yy) { return (int)in[(size_t)yy * w + x]; };
    auto bright = [&](int yy) { return ((yy - dark_row_start + 4) % 4) >= 2; };
    // where a pixel takes the rewritten value two rows up
    auto needs_above = [&](int yy, int p) { return yy > 2 && (bright(yy) ? p >= white : p < shadow); };
    int y0 = y, p0 = px(y);
    while (needs_above(y0, p0)) { y0 -= 2; p0 = px(y0); }
    int above = 0, res = 0;
    for (int yy = y0;; yy += 2) {
        const int p = yy == y0 ? p0 : px(yy);
        const int below = (yy + 2 < h) ? px(yy + 2) : 0;                // original row yy+2
        int o = p;
        if (bright(yy)) {
            if (p >= white) o = (yy > 2) ? ((yy < h - 2) ? (above + below) / 2 : above) : below;
            else o = d2u16(scaled(p));
        } else if (p < shadow) {                                        // dark row in deep shadow
            double v;
            if (yy > 2) v = (yy < h - 2) ? (above + scaled(below)) / 2 : (double)above;
            else v = scaled(below);
            o = d2u16(v);
        }
        o &= 0xFFFF;
        if (yy == y) { res = o; break; }
        above = o;
    }
    // the final 14 -> 16 bit shift applies to the first max_size/2 pixels (hdr.c:218-222)
    const size_t i = (size_t)y * w + x;
    out[i] = (uint16_t)(i < shift_count ? (res << 2) : res);
}

int launch_hdr_row_hist(const void *d_frame, int w, int h, int white, unsigned *d_hist, hipStream_t stream)
{
    MLV_HIP(hipMemsetAsync(d_hist, 0, sizeof(unsigned) * 4 * (size_t)(white + 1), stream));
    const int rows = (h - 4 - 4 + 4) / 5 + 1;       // generous; the kernel re-checks y < h-4
    if (rows <= 0) return MLVFS_AMD_OK;
    const int lds_bins = (size_t)(white + 1) * sizeof(unsigned) <= 64 * 1024 ? white + 1 : 0;        // 14-bit levels fit
    hipLaunchKernelGGL(k_hdr_row_hist, dim3(rows), dim3(256), (size_t)lds_bins * sizeof(unsigned), stream, (const uint16_t *)d_frame, w, h, white,
                       d_hist, lds_bins);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

// d_out: a second frame buffer (the rewrite reads rows as they were)
int launch_hdr_preview(const void *d_frame, void *d_out, int w, int h, int black, int white, int dark_row_start, int shadow, double a,
                       double b, size_t shift_count, hipStream_t stream)
{
    hipLaunchKernelGGL(k_hdr_preview, dim3((w + 255) / 256, h), dim3(256), 0, stream, (const uint16_t *)d_frame, (uint16_t *)d_out, w, h, black,
                       white, dark_row_start, shadow, a, b, shift_count);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

}  // namespace mlv

// ---------------------------------------------------------------- deflicker
// main.c:895-906: histogram of every second pixel of the flat frame starting at pixel 1 (hist_add(data + 1, (bytes - 1) / 2,
// skip 1), values clipped to 2^bpp + 1), its median, BaselineExposure = log2((target - black) / (median - black)).
// The kernel counts with 32-bit atomics; the host folds to the reference's 16-bit counters (histogram.h:30) and does the rest.
namespace mlv {
// samples img[first + step * s], s < samples
__global__ __launch_bounds__(256) void k_deflicker_hist(const uint16_t *__restrict__ img, uint32_t first, uint32_t step, uint32_t samples,
                                                        uint32_t white, unsigned *__restrict__ hist)
{
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < samples; s += gridDim.x * blockDim.x) {
        const uint32_t v = img[first + (size_t)step * s];
        atomicAdd(&hist[v < white ? v : white], 1u);
    }
}

int launch_deflicker_hist(const void *d_frame, uint32_t samples, uint32_t white, unsigned *d_hist, hipStream_t s)
{
    return launch_hist_add(d_frame, 1, 2, samples, white, d_hist, s);
}

// hist_add (histogram.c:52-59) on device memory: counts of img[first], img[first + step], ... (`samples` of them), values clipped to `white`
int launch_hist_add(const void *d_frame, uint32_t first, uint32_t step, uint32_t samples, uint32_t white, unsigned *d_hist, hipStream_t s)
{
    MLV_HIP(hipMemsetAsync(d_hist, 0, sizeof(unsigned) * ((size_t)white + 1), s));
    if (samples) hipLaunchKernelGGL(k_deflicker_hist, dim3(1024), dim3(256), 0, s, (const uint16_t *)d_frame, first, step, samples, white, d_hist);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}
}  // namespace mlv
