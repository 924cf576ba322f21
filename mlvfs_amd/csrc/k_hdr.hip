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
//                      place and every pixel only depends on rows y-2 (already
//                      rewritten) and y+2 (not yet rewritten) of ITS OWN COLUMN, so
//                      one lane owns one column and walks it top-down: coalesced
//                      row-wise across the wave, sequential in y.
//                      Doubles, no FMA contraction -> bit-identical to the x86 path.
#include "clip.h"

namespace mlv {

__global__ __launch_bounds__(256) void k_hdr_row_hist(const uint16_t *__restrict__ img, int w, int h, int white,
                                                      unsigned *__restrict__ hist /* [4][white+1] */)
{
    // blockIdx.y enumerates the sampled rows y = 4 + 5*k
    const int y = 4 + 5 * blockIdx.y;
    if (y >= h - 4) return;
    const int first = (y + 1) % 2;
    const int size = w - first;
    unsigned *hg = hist + (size_t)(y % 4) * (white + 1);
    for (int i = 4 * (blockIdx.x * blockDim.x + threadIdx.x); i < size; i += 4 * gridDim.x * blockDim.x) {
        const int v = img[(size_t)y * w + first + i];
        atomicAdd(&hg[v < white ? v : white], 1u);
    }
}

__device__ __forceinline__ uint16_t d2u16(double v) { return (uint16_t)(int)v; }

__global__ __launch_bounds__(64) void k_hdr_preview(uint16_t *__restrict__ img, int w, int h, int black, int white,
                                                    int dark_row_start, int shadow, double a, double b, size_t shift_count)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= w) return;
    auto scaled = [&](int p) {
        const double v = (p - black) * a + black + b;                   // hdr.c:198
        return (double)white < v ? (double)white : v;
    };
    // pm2 = rewritten value two rows up; nxt holds original rows y, y+1, y+2
    int done_m2 = 0, done_m1 = 0;
    for (int y = 0; y < h; y++) {
        const size_t i = (size_t)y * w + x;
        const int p = img[i];
        const int below = (y + 2 < h) ? img[i + 2 * (size_t)w] : 0;    // original row y+2 (not yet rewritten)
        const int above = done_m2;                                      // rewritten row y-2
        int out = p;
        if (((y - dark_row_start + 4) % 4) >= 2) {                      // bright row
            if (p >= white) out = (y > 2) ? ((y < h - 2) ? (above + below) / 2 : above) : below;
            else out = d2u16(scaled(p));
        } else if (p < shadow) {                                        // dark row in deep shadow
            double v;
            if (y > 2) v = (y < h - 2) ? (above + scaled(below)) / 2 : (double)above;
            else v = scaled(below);
            out = d2u16(v);
        }
        out &= 0xFFFF;
        done_m2 = done_m1;
        done_m1 = out;
        // the final 14 -> 16 bit shift applies to the first max_size/2 pixels (hdr.c:218-222)
        img[i] = (uint16_t)(i < shift_count ? (out << 2) : out);
    }
}

int launch_hdr_row_hist(const void *d_frame, int w, int h, int white, unsigned *d_hist, hipStream_t stream)
{
    MLV_HIP(hipMemsetAsync(d_hist, 0, sizeof(unsigned) * 4 * (size_t)(white + 1), stream));
    const int rows = (h - 4 - 4 + 4) / 5 + 1;       // generous; the kernel re-checks y < h-4
    if (rows <= 0) return MLVFS_AMD_OK;
    dim3 grid((w / 4 + 255) / 256 + 1, rows);
    hipLaunchKernelGGL(k_hdr_row_hist, grid, dim3(256), 0, stream, (const uint16_t *)d_frame, w, h, white, d_hist);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

int launch_hdr_preview(void *d_frame, int w, int h, int black, int white, int dark_row_start, int shadow, double a,
                       double b, size_t shift_count, hipStream_t stream)
{
    hipLaunchKernelGGL(k_hdr_preview, dim3((w + 63) / 64), dim3(64), 0, stream, (uint16_t *)d_frame, w, h, black, white,
                       dark_row_start, shadow, a, b, shift_count);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

}  // namespace mlv

// ---------------------------------------------------------------- deflicker
// main.c:895-906: histogram of every second pixel of the flat frame starting at pixel 1 (hist_add(data + 1, (bytes - 1) / 2,
// skip 1), values clipped to 2^bpp + 1), its median, BaselineExposure = log2((target - black) / (median - black)).
// The kernel counts with 32-bit atomics; the host folds to the reference's 16-bit counters (histogram.h:30) and does the rest.
namespace mlv {
__global__ __launch_bounds__(256) void k_deflicker_hist(const uint16_t *__restrict__ img, uint32_t samples, uint32_t white,
                                                        unsigned *__restrict__ hist)
{
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < samples; s += gridDim.x * blockDim.x) {
        const uint32_t v = img[1 + 2 * (size_t)s];
        atomicAdd(&hist[v < white ? v : white], 1u);
    }
}

int launch_deflicker_hist(const void *d_frame, uint32_t samples, uint32_t white, unsigned *d_hist, hipStream_t s)
{
    MLV_HIP(hipMemsetAsync(d_hist, 0, sizeof(unsigned) * ((size_t)white + 1), s));
    if (samples) hipLaunchKernelGGL(k_deflicker_hist, dim3(1024), dim3(256), 0, s, (const uint16_t *)d_frame, samples, white, d_hist);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}
}  // namespace mlv
