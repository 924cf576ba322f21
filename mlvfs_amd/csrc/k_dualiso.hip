// k_dualiso.hip -- kernels of the full dual-ISO conversion (cr2hdr 20-bit,
// mlvfs/hdr.c:230-1957) with the mean23 interpolation (hdr.c:1231-1304).
// The scalar decisions between the kernels (pattern, white levels, robust exposure
// fit) run on the host from small device-built histograms (dualiso.cpp).
//
//   k_di_analyse    one pass over the 16-bit frame: hdr_check sum (hdr.c:407-439), value histograms per
//                   (row phase, column parity) counted in LDS -- the host derives the four Bayer-phase
//                   histograms (identify_rggb_or_gbrg, :441-495) and the row-phase green histograms
//                   (:497-636) for both row parities (RGGB and GBRG hypothesis) from them -- and the
//                   every-3rd-pixel white histograms (:250-300), also for both parities
//   k_di_subsample  3x3-subsampled native/interpolated pairs + their histograms
//                   (match_exposures, :650-722)
//   k_di_score      RANSAC-like score of every candidate slope (:752-772)
//   k_di_match      14 -> 20 bit and per-pixel exposure correction in double (:781-803, :825-837)
//   k_di_squeeze, k_di_amaze_clamp, k_di_gray, k_di_edge_dir   the AMaZE-based interpolator around k_amaze.hip
//                   (:954-1173); k_di_interp<true> then interpolates along the chosen edge direction (:1181-1208)
//   k_di_interp     mean23 + borders + full-res pick + half-res mix + overexposure flag +
//                   alias-map error, fused per pixel (:1231-1380, :1588-1612, :1404-1418, :1620-1626)
//   k_di_alias_rank 6th largest of 37 neighbours (:1423-1443)
//   k_di_alias_blur integer gaussian (:1446-1466)
//   k_di_blend      2x2 max of the alias map, overexposure blur, final blend in EV space,
//                   20 -> 16 bit (:1469-1483, :1631-1651, :1663-1772)
// Doubles are evaluated operation by operation (no FMA contraction); the only
// transcendental evaluated on the device is the cos() of the per-frame mixing curve.
#include "clip.h"
#include "dualiso.h"
#define MLV_NET_FN __device__ __forceinline__
#include "median_nets.h"

namespace mlv {

#define DI_EVR 32768

__device__ __forceinline__ int di_bright(const DiParams &p, int y) { return (p.is_bright_bits >> (y & 3)) & 1; }

// ------------------------------------------------------------------ analysis
// One workgroup = the rows of one phase (y % 4) of a 16-row band.  The 14-bit values of its pixels are counted per
// column parity in LDS (two 16-bit counters per word: a band holds at most 4 * w / 2 < 65536 pixels per class), then
// flushed with one global atomic per non-empty bin into the 8 class histograms [y % 4][x & 1][16384], from which the
// host derives the Bayer-phase and the two green-by-row-phase histograms (dualiso.cpp).  The every-3rd-pixel "white"
// histograms (1/9 of the pixels) and the hdr_check sum use global atomics directly.
constexpr int DI_BAND = 16;
__global__ __launch_bounds__(256) void k_di_analyse(const uint16_t *__restrict__ img, int w, int H, int black, int white,
                                                    const double *__restrict__ evf /* [16384] log2(i)*32768 */,
                                                    unsigned *__restrict__ hist /* device layout, dualiso.h */, double *__restrict__ check /* sum, count */)
{
    __shared__ unsigned cnt[16384];                          // slot = (x & 1) * 16384 + value; two slots per word
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    const int q = blockIdx.y, y_base = blockIdx.x * DI_BAND + q;
    unsigned *h_w0 = hist + DI_D_WHITE0, *h_w1 = hist + DI_D_WHITE1;
    double sum = 0, n = 0;
    for (int k = threadIdx.x; k < (DI_BAND / 4) * w; k += blockDim.x) {
        const int y = y_base + 4 * (k / w), x = k % w;
        if (y >= H) break;
        const size_t i = (size_t)y * w + x;
        const int p = img[i];
        if (y >= 2 && y < H - 2 && x >= 2 && x < w - 2) {        // hdr_check
            const int p2 = img[i + 2 * (size_t)w];
            if ((p > black + 32 || p2 > black + 32) && p < white && p2 < white) {
                const int a = p - black, b = p2 - black;
                const double ea = (a >= 0 && a < 16384) ? evf[a] : 0.0, eb = (b >= 0 && b < 16384) ? evf[b] : 0.0;
                const double d = eb - ea;
                sum += d > 0 ? d : -d;
                n += 1;
            }
        }
        const int slot = (x & 1) * 16384 + (p & 16383);
        atomicAdd(&cnt[slot >> 1], 1u << (16 * (slot & 1)));
        const int vw = p < 32767 ? p : 32767;
        if (y % 3 == 0 && x % 3 == 0) atomicAdd(&h_w0[(y & 3) * 32768 + vw], 1u);
        const int y1 = y - 1;                                    // row index in the frame that starts one row lower
        if (y1 >= 1 && y1 % 3 == 1 && x % 3 == 0) atomicAdd(&h_w1[(y1 & 3) * 32768 + vw], 1u);
    }
    __syncthreads();
    unsigned *cls = hist + DI_D_CLASS + (size_t)q * 2 * 16384;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) {
        const unsigned v = cnt[i];
        if (v & 0xFFFFu) atomicAdd(&cls[2 * i], v & 0xFFFFu);
        if (v >> 16) atomicAdd(&cls[2 * i + 1], v >> 16);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o); n += __shfl_xor(n, o); }
    if ((threadIdx.x & 63) == 0 && n > 0) { atomicAdd(&check[0], sum); atomicAdd(&check[1], n); }
}

// ------------------------------------------------------------------ exposure matching
__device__ __forceinline__ int di_p16(const uint16_t *img, size_t i) { return (int)(((unsigned)img[i] << 2) & 0xFFFFu); }   // 14->20->16 bit

__global__ __launch_bounds__(256) void k_di_subsample(const uint16_t *__restrict__ img, DiParams p, int nsx, int nsy,
                                                      int *__restrict__ dark_s, int *__restrict__ bright_s,
                                                      unsigned *__restrict__ hist_b, unsigned *__restrict__ hist_d)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nsx * nsy) return;
    const int sx = idx % nsx, sy = idx / nsx;
    const int x = 3 * sx, y = p.ay1 + 2 + 3 * sy, w = p.w;
    const int black = p.black20 / 16, white = p.match_white20 / 16;
    const int clip0 = white - black, clip = (int)(clip0 * 0.95);
    const int pa = di_p16(img, x + (size_t)(y - 2) * w) - black, pb = di_p16(img, x + (size_t)(y + 2) * w) - black;
    int pn = di_p16(img, x + (size_t)y * w) - black;
    int pi = (pa + pb + 1) / 2;
    if (pa >= clip || pb >= clip) pi = clip0;
    if (pi >= clip) pn = clip0;
    const int br = di_bright(p, y);
    const int d = br ? pi : pn, b = br ? pn : pi;
    dark_s[idx] = d;
    bright_s[idx] = b;
    if (b < clip) {
        atomicAdd(&hist_b[min(max(b + DI_HIST_OFF, 0), DI_HIST_N - 1)], 1u);
        atomicAdd(&hist_d[min(max(d + DI_HIST_OFF, 0), DI_HIST_N - 1)], 1u);
    }
}

// highlight pairs of match_exposures (hdr.c:735-746): the samples with b_lo < bright < b_hi in raster order.  One workgroup
// per sample row counts them; the host turns the counts into how many each row contributes and where (the reference's
// cap only leaves the inner loop, dualiso.cpp); the second kernel writes them in order.
__device__ __forceinline__ bool di_hi_ok(int b, int b_lo, int b_hi) { return !(b >= b_hi || b <= b_lo); }

__global__ __launch_bounds__(256) void k_di_hi_count(const int *__restrict__ bs, int nsx, int b_lo, int b_hi, int *__restrict__ counts)
{
    __shared__ int red[4];
    const int *row = bs + (size_t)blockIdx.x * nsx;
    int n = 0;
    for (int x = threadIdx.x; x < nsx; x += blockDim.x) n += di_hi_ok(row[x], b_lo, b_hi);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void k_di_hi_compact(const int *__restrict__ ds, const int *__restrict__ bs, int nsx, int b_lo, int b_hi,
                                                       const int *__restrict__ take, const int *__restrict__ offset,
                                                       int *__restrict__ hd, int *__restrict__ hb)
{
    __shared__ int part[256];
    const int lim = take[blockIdx.x];
    if (lim <= 0) return;
    const size_t base = (size_t)blockIdx.x * nsx;
    const int per = (nsx + 255) / 256, x0 = threadIdx.x * per, x1 = min(x0 + per, nsx);     // consecutive samples per thread: keeps the order
    int n = 0;
    for (int x = x0; x < x1; x++) n += di_hi_ok(bs[base + x], b_lo, b_hi);
    part[threadIdx.x] = n;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {                          // inclusive scan
        const int v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int pos = part[threadIdx.x] - n;
    const int out0 = offset[blockIdx.x];
    for (int x = x0; x < x1 && pos < lim; x++) {
        const int b = bs[base + x];
        if (!di_hi_ok(b, b_lo, b_hi)) continue;
        hd[out0 + pos] = ds[base + x];
        hb[out0 + pos] = b;
        pos++;
    }
}

__global__ __launch_bounds__(256) void k_di_score(const int *__restrict__ hd, const int *__restrict__ hb, int hi_n,
                                                  const double *__restrict__ cand /* [2*ncand]: a, b */, int *__restrict__ score)
{
    __shared__ int red[4];
    const double ta = cand[2 * blockIdx.x], tb = cand[2 * blockIdx.x + 1];
    int s = 0;
    for (int i = threadIdx.x; i < hi_n; i += blockDim.x) {
        const int e = (int)(hd[i] - (hb[i] * ta + tb));
        s += (e > 0 ? e : -e) < 50;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) score[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void k_di_match(const uint16_t *__restrict__ img, uint32_t *__restrict__ raw, DiParams p)
{
    const size_t n = (size_t)p.w * p.h;
    const double a = p.a, b20 = p.b20;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        int v = (int)(((uint32_t)img[i] << 6) & 0xFFFFFu);
        if (v != 0) {
            const int y = (int)(i / p.w);
            double r;
            if (di_bright(p, y)) r = (v - p.black20) * a + p.black20 + b20 * a;
            else r = v - b20 + b20 * a;
            v = (int)r;
            v = v < 0 ? 0 : (v > 0xFFFFF ? 0xFFFFF : v);
        }
        raw[i] = (uint32_t)v;
    }
}

// ------------------------------------------------------------------ interpolation + mix (per pixel)
__device__ __forceinline__ int di_mean2(int a, int b, int white) { return (a >= white || b >= white) ? white : (a + b) / 2; }
__device__ __forceinline__ int di_mean3(int a, int b, int c, int white)
{
    const int m = (a + b + c) / 3;
    return (a >= white || b >= white || c >= white) ? max(m, white) : m;
}

// alias-map error term of one pixel, hdr.c:1404-1418
__device__ __forceinline__ int di_alias_err(const DiParams &p, const DiLuts &L, int b, int f, int hr)
{
    if (L.fullres_curve[b] > 0.8) return 0;
    int e_lin = f - hr;
    e_lin = e_lin > 0 ? e_lin : -e_lin;
    e_lin = max(e_lin - p.dark_noise * 3 / 2, 0);
    int e_log = L.mix_raw2ev[f] - L.mix_raw2ev[hr];
    e_log = e_log > 0 ? e_log : -e_log;
    return min(min(e_lin / 2, e_log / 16), 65530);
}

// everything that follows the interpolation for one pixel: full-res pick (hdr.c:1355-1380), half-res mix
// (hdr.c:1588-1612), overexposure flag (hdr.c:1620-1626) and, when no chroma smoothing sits in between, the alias error
__device__ __forceinline__ void di_mix_pixel(const DiParams &p, const DiLuts &L, size_t i, int br, int b, int d,
                                             uint32_t *__restrict__ fullres, uint32_t *__restrict__ halfres,
                                             uint16_t *__restrict__ over, uint16_t *__restrict__ amap)
{
    int f = 0;
    if (p.use_fullres) f = br ? (b < p.white_darkened ? b : max(b, d)) : d;
    fullres[i] = (uint32_t)f;
    const double ev = L.log2sig[b & 0xFFFFF] + p.corr_ev;
    double t = ev - (p.max_ev - p.overlap);
    t = t < p.overlap ? t : p.overlap;
    t = t > 0 ? t : 0;
    double k = (-cos(t * 3.14159265358979323846 / p.overlap) + 1) / 2;
    k = k < 0 ? 0 : (k > 1 ? 1 : k);
    const int mixed = (int)(L.mix_raw2ev[b] * (1 - k) + L.mix_raw2ev[d] * k);
    const int hr = L.mix_ev2raw[mixed];
    halfres[i] = (uint32_t)hr;
    over[i] = (b >= p.white_darkened || d >= p.white20) ? 100 : 0;
    if (amap) amap[i] = (uint16_t)di_alias_err(p, L, b, f, hr);
}

// edge directions of the AMaZE-based interpolation, hdr.c:916-938: {ack, a, b, bck} x {x, y}; y is multiplied by s
__constant__ signed char k_edge_dirs[11][8] = {
    { -4, 2, -2, 1, 4, -2, 6, -3 }, { -3, 2, -1, 1, 3, -2, 4, -3 }, { -2, 2, -1, 1, 2, -2, 3, -3 }, { -1, 2, -1, 1, 1, -2, 2, -3 },
    { -1, 2, 0, 1, 1, -2, 1, -3 },  { 0, 2, 0, 1, 0, -2, 0, -3 },   { 1, 2, 0, 1, -1, -2, -1, -3 }, { 1, 2, 1, 1, -1, -2, -2, -3 },
    { 2, 2, 1, 1, -2, -2, -3, -3 }, { 3, 2, 1, 1, -3, -2, -4, -3 }, { 4, 2, 2, 1, -4, -2, -6, -3 } };

struct DiAmazeIn {              // inputs of the edge-directed interpolation (null planes = mean23)
    const float *red, *green, *blue;
    const uint8_t *dir;
    const int *sq_row;
};

template <bool AMAZE>
__global__ __launch_bounds__(256) void k_di_interp(const uint32_t *__restrict__ raw, DiParams p, DiLuts L, DiAmazeIn A,
                                                   uint32_t *__restrict__ dark, uint32_t *__restrict__ bright,
                                                   uint32_t *__restrict__ fullres, uint32_t *__restrict__ halfres,
                                                   uint16_t *__restrict__ over, uint16_t *__restrict__ amap)
{
    const int w = p.w, h = p.h;
    const size_t n = (size_t)w * h;
    const int *ir2e = L.interp_raw2ev, *ie2r = L.interp_ev2raw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % w), y = (int)(i / w);
        const int br = di_bright(p, y);
        auto R = [&](int xx, int yy) { return (int)raw[xx + (size_t)yy * w]; };
        int native, interp;
        // precedence of the reference's loops: column borders (y >= 2) over row borders over the interior
        if (y >= 2 && x < 2) { interp = R(x, y - 2); native = R(x, y); }
        else if (y >= 2 && x >= w - 3) { interp = R(x - 2, y - 2); native = R(x - 2, y); }
        else if (y < 3) { interp = R(x, y + 2); native = R(x, y); }
        else if (y >= h - 4) { interp = R(x, y - 2); native = R(x, y); }
        else if (AMAZE) {                                         // hdr.c:940-952, 1181-1208
            const int s = (di_bright(p, y) == di_bright(p, y + 1)) ? -1 : 1;
            const float *plane = (y & 1) == 0 ? ((x & 1) == 0 ? A.red : A.green) : ((x & 1) == 0 ? A.green : A.blue);
            const int d = A.dir[i];
            const int dd[3] = { d, min(d + 1, 10), max(d - 1, 0) };
            int pi[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const signed char *e = k_edge_dirs[dd[k]];
                int pa = (int)plane[(size_t)A.sq_row[y + e[3] * s] * w + x + e[2]];
                int pb = (int)plane[(size_t)A.sq_row[y + e[5] * s] * w + x + e[4]];
                pa = pa < 0 ? 0 : (pa > 0xFFFFF ? 0xFFFFF : pa);
                pb = pb < 0 ? 0 : (pb > 0xFFFFF ? 0xFFFFF : pb);
                pi[k] = (ir2e[pa] * 2 + ir2e[pb]) / 3;
            }
            interp = ie2r[(2 * pi[0] + pi[1] + pi[2]) / 4];
            native = R(x, y);
        } else {
            const int wl = !br ? p.white_darkened : p.white20;
            const int wev = ir2e[wl];
            const int s = (di_bright(p, y) == di_bright(p, y + 1)) ? -1 : 1;
            const int xe = x & ~1;                                // the pair (xe, xe+1) is produced together
            int ev;
            if ((y & 1) == 0) {
                if (x == xe) ev = di_mean2(ir2e[R(xe, y - 2)], ir2e[R(xe, y + 2)], wev);
                else ev = di_mean3(ir2e[R(xe + 2, y + s)], ir2e[R(xe, y + s)], ir2e[R(xe + 1, y - 2 * s)], wev);
            } else {
                if (x == xe) ev = di_mean3(ir2e[R(xe + 1, y + s)], ir2e[R(xe - 1, y + s)], ir2e[R(xe, y - 2 * s)], wev);
                else ev = di_mean2(ir2e[R(xe + 1, y - 2)], ir2e[R(xe + 1, y + 2)], wev);
            }
            interp = ie2r[ev];
            native = R(x, y);
        }
        const int b = br ? native : interp, d = br ? interp : native;
        bright[i] = (uint32_t)b;
        dark[i] = (uint32_t)d;
        di_mix_pixel(p, L, i, br, b, d, fullres, halfres, over, amap);
    }
}

// ------------------------------------------------------------------ AMaZE-based interpolation, hdr.c:954-1229
// squeeze: rows of one exposure become adjacent, greens halved around black (hdr.c:977-1026)
__global__ __launch_bounds__(256) void k_di_squeeze(const uint32_t *__restrict__ raw, DiParams p, const int *__restrict__ sq_dst,
                                                    float *__restrict__ cfa)
{
    const size_t n = (size_t)p.w * p.h;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % p.w), y = (int)(i / p.w);
        const int yh = sq_dst[y];
        if (yh < 0) continue;
        int v = (int)raw[i];
        if ((x & 1) != (y & 1)) v = (v - p.black20) / 2 + p.black20;
        cfa[(size_t)yh * p.w + x] = (float)v;
    }
}

// undo the green scaling, clamp (hdr.c:1041-1050), in place on the squeezed planes
__global__ __launch_bounds__(256) void k_di_amaze_clamp(float *__restrict__ red, float *__restrict__ green, float *__restrict__ blue,
                                                        size_t n, int black)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float fb = (float)black, hi = 1048575.0f;
        const float g = (green[i] - fb) * 2.0f + fb, r = red[i], b = blue[i];
        green[i] = g < hi ? (g > 0.0f ? g : 0.0f) : hi;
        red[i] = r < hi ? (r > 0.0f ? r : 0.0f) : hi;
        blue[i] = b < hi ? (b > 0.0f ? b : 0.0f) : hi;
    }
}

// de-squeezed gray image in EV (hdr.c:1055-1059 + the raw2ev lookups of :1157-1168)
__global__ __launch_bounds__(256) void k_di_gray(const float *__restrict__ red, const float *__restrict__ green,
                                                 const float *__restrict__ blue, DiParams p, const int *__restrict__ sq_row,
                                                 const int *__restrict__ r2e, int *__restrict__ gray_ev)
{
    const size_t n = (size_t)p.w * p.h;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % p.w), y = (int)(i / p.w);
        const size_t o = (size_t)sq_row[y] * p.w + x;
        const unsigned gray = (unsigned)(green[o] / 2 + red[o] / 4 + blue[o] / 4);
        gray_ev[i] = r2e[gray];
    }
}

// best of 11 edge directions where the interpolation has to be good (hdr.c:1096-1173)
__global__ __launch_bounds__(256) void k_di_edge_dir(const uint32_t *__restrict__ raw, const int *__restrict__ gray_ev, DiParams p,
                                                     const double *__restrict__ fullres_curve, uint8_t *__restrict__ dir,
                                                     unsigned *__restrict__ stats)
{
    __shared__ unsigned s_stats[4];
    if (threadIdx.x < 4) s_stats[threadIdx.x] = 0;
    __syncthreads();
    const int w = p.w, h = p.h;
    const size_t n = (size_t)w * h;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % w), y = (int)(i / w);
        int best = 5;
        if (x >= 5 && x < w - 5 && y >= 5 && y < h - 5) {
            const int v = (int)raw[i];
            bool search;
            if (!di_bright(p, y)) { search = !(fullres_curve[v] > 0.8); atomicAdd(&s_stats[search ? 2 : 3], 1u); }
            else { search = !(v < p.white_darkened); atomicAdd(&s_stats[search ? 0 : 1], 1u); }
            if (search) {
                const int s = (di_bright(p, y) == di_bright(p, y + 1)) ? -1 : 1;
                int e_best = 0x7FFFFFFF;
                for (int d = 0; d < 11; d++) {
                    const signed char *e = k_edge_dirs[d];
                    const int *r1 = gray_ev + (size_t)(y + e[1] * s) * w + x + e[0], *r2 = gray_ev + (size_t)(y + e[3] * s) * w + x + e[2];
                    const int *r3 = gray_ev + (size_t)(y + e[5] * s) * w + x + e[4], *r4 = gray_ev + (size_t)(y + e[7] * s) * w + x + e[6];
                    int err = 0;
#pragma unroll
                    for (int j = -5; j <= 5; j++) {
                        const int p1 = r1[j], p2 = r2[j], p3 = r3[j], p4 = r4[j];
                        err += abs(p1 - p2) + abs(p2 - p3) + abs(p3 - p4);
                    }
                    err += abs(d - 5) * DI_EVR / 8;
                    if (err < e_best) { e_best = err; best = d; }
                }
            }
        }
        dir[i] = (uint8_t)best;
    }
    __syncthreads();
    if (threadIdx.x < 4 && s_stats[threadIdx.x]) atomicAdd(&stats[threadIdx.x], s_stats[threadIdx.x]);
}

// alias error from the chroma-smoothed planes (hdr.c:1620: build_alias_map gets fullres_smooth / halfres_smooth)
__global__ __launch_bounds__(256) void k_di_alias_err(const uint32_t *__restrict__ bright, const uint32_t *__restrict__ fullres_s,
                                                      const uint32_t *__restrict__ halfres_s, DiParams p, DiLuts L,
                                                      uint16_t *__restrict__ amap)
{
    const size_t n = (size_t)p.w * p.h;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        amap[i] = (uint16_t)di_alias_err(p, L, (int)bright[i], (int)fullres_s[i], (int)halfres_s[i]);
}

// ------------------------------------------------------------------ chroma smoothing of a 20-bit plane
// hdr.c:1488-1522 = chroma_smooth.c:22-71 for uint32_t pixels, black 0 and the 20-bit tables of mix_images.
// Pass 1 turns every Bayer cell into (green EV, R - green, B - green); pass 2 takes the medians over the cell
// neighbourhood and rewrites R and B of the cells the reference visits.  The planes stay L2-resident in between.
__global__ __launch_bounds__(256) void k_di_cs_cells(const uint32_t *__restrict__ plane, int w, int cw, int ch,
                                                     const int *__restrict__ r2e, int *__restrict__ cells /* [3][ch][cw] */)
{
    const int cx = blockIdx.x * blockDim.x + threadIdx.x, cy = blockIdx.y;
    if (cx >= cw) return;
    const uint32_t *c = plane + 2 * cx + (size_t)(2 * cy) * w;
    const int ge = (r2e[c[1]] + r2e[c[w]]) / 2;
    const size_t o = cx + (size_t)cy * cw, pl = (size_t)cw * ch;
    cells[o] = ge;
    cells[o + pl] = r2e[c[0]] - ge;
    cells[o + 2 * pl] = r2e[c[w + 1]] - ge;
}

template <int METHOD>
__global__ __launch_bounds__(256) void k_di_cs_apply(const int *__restrict__ cells, int w, int h, int cw, int ch,
                                                     const int *__restrict__ e2r, uint32_t *__restrict__ out)
{
    const int cx = blockIdx.x * blockDim.x + threadIdx.x, cy = blockIdx.y;
    const int x = 2 * cx, y = 2 * cy;
    if (x < 4 || x >= w - 4 || y < 4 || y >= h - 5) return;
    const size_t pl = (size_t)cw * ch;
    const int *G = cells + cx + (size_t)cy * cw, *Rr = G + pl, *Bb = G + 2 * pl;
    const int ge = G[0];
    if (ge < 2 * DI_EVR) return;
    constexpr int NV = METHOD == 5 ? 25 : (METHOD == 3 ? 9 : 5), REACH = METHOD == 5 ? 2 : 1;
    int vr[NV], vb[NV], k = 0;
#pragma unroll
    for (int i = -REACH; i <= REACH; i++)
#pragma unroll
        for (int j = -REACH; j <= REACH; j++) {
            if (METHOD == 2 && i != 0 && j != 0) continue;
            vr[k] = Rr[i + j * cw];
            vb[k] = Bb[i + j * cw];
            k++;
        }
    int dr[1], db[1];
    if constexpr (METHOD == 5) { mlv_median25(vr, dr); mlv_median25(vb, db); }
    else if constexpr (METHOD == 3) { mlv_median9(vr, dr); mlv_median9(vb, db); }
    else { mlv_median5(vr, dr); mlv_median5(vb, db); }
    if (ge + dr[0] <= DI_EVR || ge + db[0] <= DI_EVR) return;
    auto clampev = [](int v) { return v < 0 ? 0 : (v > 14 * DI_EVR - 1 ? 14 * DI_EVR - 1 : v); };
    out[x + (size_t)y * w] = (uint32_t)e2r[clampev(ge + dr[0])];
    out[x + 1 + (size_t)(y + 1) * w] = (uint32_t)e2r[clampev(ge + db[0])];
}

// 6th largest of the 37 neighbours (kth_smallest(negated, 37, 5)), hdr.c:1423-1443
__global__ __launch_bounds__(256) void k_di_alias_rank(const uint16_t *__restrict__ amap, const uint32_t *__restrict__ bright,
                                                       const double *__restrict__ fullres_curve, int w, int h,
                                                       uint16_t *__restrict__ aux)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const size_t i = x + (size_t)y * w;
    int out = amap[i];
    if (x >= 6 && x < w - 6 && y >= 6 && y < h - 6 && !(fullres_curve[bright[i]] > 0.8)) {
        int t0 = -1, t1 = -1, t2 = -1, t3 = -1, t4 = -1, t5 = -1;      // six largest, descending
        auto push = [&](int v) {
            if (v > t5) {
                t5 = v;
                int s;
                if (t5 > t4) { s = t4; t4 = t5; t5 = s; }
                if (t4 > t3) { s = t3; t3 = t4; t4 = s; }
                if (t3 > t2) { s = t2; t2 = t3; t3 = s; }
                if (t2 > t1) { s = t1; t1 = t2; t2 = s; }
                if (t1 > t0) { s = t0; t0 = t1; t1 = s; }
            }
        };
#pragma unroll
        for (int dy = -6; dy <= 6; dy += 2) {
            const int reach = (dy == -6 || dy == 6) ? 2 : ((dy == -4 || dy == 4) ? 4 : 6);
#pragma unroll
            for (int dx = -6; dx <= 6; dx += 2)
                if (dx >= -reach && dx <= reach) push((int)amap[(x + dx) + (size_t)(y + dy) * w]);
        }
        out = t5;
    }
    aux[i] = (uint16_t)out;
}

// integer gaussian, hdr.c:1446-1466 (terms exactly as written there, duplicates included)
__global__ __launch_bounds__(256) void k_di_alias_blur(const uint16_t *__restrict__ aux, const uint16_t *__restrict__ amap_in,
                                                       const uint32_t *__restrict__ bright, const double *__restrict__ fullres_curve,
                                                       int w, int h, uint16_t *__restrict__ out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const size_t i = x + (size_t)y * w;
    int v = amap_in[i];
    if (x >= 6 && x < w - 6 && y >= 6 && y < h - 6 && !(fullres_curve[bright[i]] > 0.8)) {
        auto A = [&](int dx, int dy) { return (int)aux[(x + dx) + (size_t)(y + dy) * w]; };
        const int plus2 = A(0, -2) + A(-2, 0) + A(2, 0) + A(0, 2);
        const int diag2 = A(-2, -2) + A(2, -2) + A(-2, 2) + A(2, 2);
        const int twice = A(-2, -2) + A(2, -2) + A(-2, -2) + A(2, -2) + A(-2, 2) + A(2, 2) + A(-2, 2) + A(2, 2);
        const int plus6 = A(0, -6) + A(-6, 0) + A(6, 0) + A(0, 6);
        const int ring = A(-2, -6) + A(2, -6) + A(-6, -2) + A(6, -2) + A(-6, 2) + A(6, 2) + A(-2, 6) + A(2, 6);
        v = A(0, 0) + plus2 * 820 / 1024 + diag2 * 657 / 1024 + plus2 * 421 / 1024 + twice * 337 / 1024 + diag2 * 173 / 1024 +
            plus6 * 139 / 1024 + ring * 111 / 1024 + ring * 57 / 1024;
    }
    out[i] = (uint16_t)v;
}

// final blend + 20 -> 16 bit; amap = gaussian output (pre 2x2 max), over = raw 100/0 flags
__global__ __launch_bounds__(256) void k_di_blend(const uint32_t *__restrict__ dark, const uint32_t *__restrict__ bright,
                                                  const uint32_t *__restrict__ fullres, const uint32_t *__restrict__ fullres_s,
                                                  const uint32_t *__restrict__ halfres_s, const uint16_t *__restrict__ over,
                                                  const uint16_t *__restrict__ amap, DiParams p, DiLuts L,
                                                  uint16_t *__restrict__ img_out)
{
    const int w = p.w, h = p.h;
    const size_t n = (size_t)w * h;
    const int *r2e = L.blend_raw2ev, *e2r = L.blend_ev2raw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % w), y = (int)(i / w);
        const int b = (int)bright[i], d = (int)dark[i];
        // overexposure blur, hdr.c:1631-1651
        int ov = over[i];
        if (x >= 3 && x < w - 3 && y >= 3 && y < h - 3) {
            auto O = [&](int dx, int dy) { return (int)over[(x + dx) + (size_t)(y + dy) * w]; };
            ov = (uint16_t)(O(0, 0) + (O(0, -1) + O(-1, 0) + O(1, 0) + O(0, 1)) * 820 / 1024 +
                            (O(-1, -1) + O(1, -1) + O(-1, 1) + O(1, 1)) * 657 / 1024);
        }
        // alias map made "grayscale": max over the 2x2 cell, capped (hdr.c:1469-1483)
        double c = 0;
        if (amap) {
            int co = amap[i];
            const int cx = x & ~1, cy = y & ~1;
            if (cx >= 2 && cx < w - 2 && cy >= 2 && cy < h - 2) {
                const size_t o = cx + (size_t)cy * w;
                co = min(max(max((int)amap[o], (int)amap[o + 1]), max((int)amap[o + w], (int)amap[o + w + 1])), 15000);
            }
            c = co / 15000.0;
            c = c < 0 ? 0 : (c > 1 ? 1 : c);
        }
        const int hrev = r2e[halfres_s[i]], frev = r2e[fullres[i]], frsev = r2e[fullres_s[i]];
        double f = L.fullres_curve[b & 0xFFFFF];
        double ovf = ov / 200.0;
        ovf = ovf < 0 ? 0 : (ovf > 1 ? 1 : ovf);
        c = c > ovf ? c : ovf;
        const double noisy = ovf > 1 - f ? ovf : 1 - f;
        f = f > c ? f : c;
        const double fev = noisy * frsev + (1 - noisy) * frev;
        const int sig = (d + b) / 2;
        const double lim = (double)(sig - p.black20) / (4 * p.dark_noise);
        const double fm = f < lim ? f : lim;
        f = fm > 0 ? fm : 0;
        int out = (int)(hrev * (1 - f) + fev * f);
        out = out < -10 * DI_EVR ? -10 * DI_EVR : (out > 14 * DI_EVR - 1 ? 14 * DI_EVR - 1 : out);
        const int v20 = e2r[out];
        int v = (int)(v20 / 16.0 + 0.0f + 0.5);                                      // hdr.c:243, dither term is 0
        v = v < 0 ? 0 : (v > 0xFFFF ? 0xFFFF : v);
        img_out[i] = (uint16_t)v;
    }
}

// ------------------------------------------------------------------ launchers
static inline dim3 flat_grid(size_t n) { size_t b = (n + 255) / 256; if (b > 8192) b = 8192; return dim3((unsigned)b); }

int di_launch_analyse(const void *d_img, int w, int H, int black, int white, const double *d_evf, unsigned *d_hist,
                      double *d_check, hipStream_t s)
{
    const ptrdiff_t gap = (const uint8_t *)d_check - (const uint8_t *)d_hist;
    if (gap >= (ptrdiff_t)(sizeof(unsigned) * DI_D_WORDS) && gap < (ptrdiff_t)(sizeof(unsigned) * DI_D_WORDS) + 4096)
        MLV_HIP(hipMemsetAsync(d_hist, 0, (size_t)gap + 2 * sizeof(double), s));                   // the check sums lie right behind: one call
    else {
        MLV_HIP(hipMemsetAsync(d_hist, 0, sizeof(unsigned) * DI_D_WORDS, s));
        MLV_HIP(hipMemsetAsync(d_check, 0, 2 * sizeof(double), s));
    }
    hipLaunchKernelGGL(k_di_analyse, dim3((H + DI_BAND - 1) / DI_BAND, 4), dim3(256), 0, s, (const uint16_t *)d_img, w, H, black, white,
                       d_evf, d_hist, d_check);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

int di_launch_subsample(const void *d_img, const DiParams &p, int nsx, int nsy, int *d_dark_s, int *d_bright_s,
                        unsigned *d_hist_b, unsigned *d_hist_d, hipStream_t s)
{
    if (d_hist_d == d_hist_b + DI_HIST_N) MLV_HIP(hipMemsetAsync(d_hist_b, 0, 2 * sizeof(unsigned) * DI_HIST_N, s));      // back to back: one call
    else {
        MLV_HIP(hipMemsetAsync(d_hist_b, 0, sizeof(unsigned) * DI_HIST_N, s));
        MLV_HIP(hipMemsetAsync(d_hist_d, 0, sizeof(unsigned) * DI_HIST_N, s));
    }
    if (nsx * nsy > 0)
        hipLaunchKernelGGL(k_di_subsample, dim3((nsx * nsy + 255) / 256), dim3(256), 0, s, (const uint16_t *)d_img, p, nsx, nsy,
                           d_dark_s, d_bright_s, d_hist_b, d_hist_d);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

int di_launch_hi_count(const int *d_bs, int nsx, int nsy, int b_lo, int b_hi, int *d_counts, hipStream_t s)
{
    if (nsy > 0) hipLaunchKernelGGL(k_di_hi_count, dim3(nsy), dim3(256), 0, s, d_bs, nsx, b_lo, b_hi, d_counts);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

int di_launch_hi_compact(const int *d_ds, const int *d_bs, int nsx, int nsy, int b_lo, int b_hi, const int *d_take, const int *d_offset,
                         int *d_hd, int *d_hb, hipStream_t s)
{
    if (nsy > 0) hipLaunchKernelGGL(k_di_hi_compact, dim3(nsy), dim3(256), 0, s, d_ds, d_bs, nsx, b_lo, b_hi, d_take, d_offset, d_hd, d_hb);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

int di_launch_score(const int *d_hd, const int *d_hb, int hi_n, const double *d_cand, int ncand, int *d_score, hipStream_t s)
{
    hipLaunchKernelGGL(k_di_score, dim3(ncand), dim3(256), 0, s, d_hd, d_hb, hi_n, d_cand, d_score);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

static int di_chroma_smooth(const uint32_t *plane, uint32_t *plane_s, const DiParams &p, const DiLuts &L, int *cells, hipStream_t s)
{
    const int w = p.w, h = p.h, cw = w / 2, ch = h / 2;
    MLV_HIP(hipMemcpyAsync(plane_s, plane, (size_t)w * h * 4, hipMemcpyDeviceToDevice, s));
    if (cw <= 0 || ch <= 0) return MLVFS_AMD_OK;
    dim3 g((cw + 255) / 256, ch);
    hipLaunchKernelGGL(k_di_cs_cells, g, dim3(256), 0, s, plane, w, cw, ch, L.mix_raw2ev, cells);
    switch (p.chroma_smooth) {
    case 2: hipLaunchKernelGGL(k_di_cs_apply<2>, g, dim3(256), 0, s, cells, w, h, cw, ch, L.mix_ev2raw, plane_s); break;
    case 3: hipLaunchKernelGGL(k_di_cs_apply<3>, g, dim3(256), 0, s, cells, w, h, cw, ch, L.mix_ev2raw, plane_s); break;
    default: hipLaunchKernelGGL(k_di_cs_apply<5>, g, dim3(256), 0, s, cells, w, h, cw, ch, L.mix_ev2raw, plane_s); break;
    }
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

// raw (exposure-matched 20-bit) -> dark/bright by mean23; returns with fullres/halfres/over (and the alias error) filled in
int di_launch_match(const void *d_img, const DiParams &p, const DiPlanes &P, hipStream_t s)
{
    hipLaunchKernelGGL(k_di_match, flat_grid((size_t)p.w * p.h), dim3(256), 0, s, (const uint16_t *)d_img, P.raw, p);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

// squeeze -> AMaZE -> clamp -> gray -> edge directions; the planes feed k_di_interp<true>
int di_launch_amaze_interp(const DiParams &p, const DiLuts &L, const DiPlanes &P, hipStream_t s)
{
    const size_t n = (size_t)p.w * p.h;
    MLV_HIP(hipMemsetAsync(P.cfa, 0, n * sizeof(float), s));          // rows no exposure lands on stay zero (hdr.c:971)
    MLV_HIP(hipMemsetAsync(P.stats, 0, 4 * sizeof(unsigned), s));
    hipLaunchKernelGGL(k_di_squeeze, flat_grid(n), dim3(256), 0, s, P.raw, p, P.sq_dst, P.cfa);
    int rc = amaze_launch(P.cfa, p.w, p.h, P.red, P.green, P.blue, P.amaze_scratch, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_di_amaze_clamp, flat_grid(n), dim3(256), 0, s, P.red, P.green, P.blue, n, p.black20);
    hipLaunchKernelGGL(k_di_gray, flat_grid(n), dim3(256), 0, s, P.red, P.green, P.blue, p, P.sq_row, L.interp_raw2ev, P.gray_ev);
    hipLaunchKernelGGL(k_di_edge_dir, flat_grid(n), dim3(256), 0, s, P.raw, P.gray_ev, p, L.fullres_curve, P.dir, P.stats);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

// amaze: the planes of di_launch_amaze_interp are ready and the edge-directed interpolation replaces mean23
int di_launch_convert(const DiParams &p, const DiLuts &L, const DiPlanes &P, bool amaze, void *d_out, hipStream_t s)
{
    const size_t n = (size_t)p.w * p.h;
    uint16_t *amap_fused = (p.use_alias_map && !p.chroma_smooth) ? P.amap : nullptr;
    const DiAmazeIn A{ P.red, P.green, P.blue, P.dir, P.sq_row };
    if (amaze)
        hipLaunchKernelGGL(k_di_interp<true>, flat_grid(n), dim3(256), 0, s, P.raw, p, L, A, P.dark, P.bright, P.fullres, P.halfres,
                           P.over, amap_fused);
    else
        hipLaunchKernelGGL(k_di_interp<false>, flat_grid(n), dim3(256), 0, s, P.raw, p, L, A, P.dark, P.bright, P.fullres, P.halfres,
                           P.over, amap_fused);
    MLV_HIP(hipGetLastError());
    const uint32_t *fullres_s = P.fullres, *halfres_s = P.halfres;
    if (p.chroma_smooth) {                                             // hdr.c:1612-1619
        int rc = di_chroma_smooth(P.halfres, P.halfres_s, p, L, P.cells, s);
        if (rc) return rc;
        halfres_s = P.halfres_s;
        if (p.use_fullres) {                                           // otherwise fullres_smooth aliases the all-zero fullres (hdr.c:1822)
            rc = di_chroma_smooth(P.fullres, P.fullres_s, p, L, P.cells, s);
            if (rc) return rc;
            fullres_s = P.fullres_s;
        }
        if (p.use_alias_map)
            hipLaunchKernelGGL(k_di_alias_err, flat_grid(n), dim3(256), 0, s, P.bright, fullres_s, halfres_s, p, L, P.amap);
    }
    const uint16_t *amap_final = nullptr;
    if (p.use_alias_map) {
        dim3 g((p.w + 255) / 256, p.h);
        hipLaunchKernelGGL(k_di_alias_rank, g, dim3(256), 0, s, P.amap, P.bright, L.fullres_curve, p.w, p.h, P.aux);
        hipLaunchKernelGGL(k_di_alias_blur, g, dim3(256), 0, s, P.aux, P.amap, P.bright, L.fullres_curve, p.w, p.h, P.amap2);
        amap_final = P.amap2;
    }
    hipLaunchKernelGGL(k_di_blend, flat_grid(n), dim3(256), 0, s, P.dark, P.bright, P.fullres, fullres_s, halfres_s, P.over, amap_final,
                       p, L, (uint16_t *)d_out);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

}  // namespace mlv
