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
//   k_di_match (with the squeeze), k_di_amaze_ev, k_di_edge_dir   the AMaZE-based interpolator around k_amaze.hip
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

// Which 256-pixel piece of a pass a workgroup of the grid-stride kernels takes.  The dispatcher deals workgroups b, b + 1, ... to the
// chip's eight XCDs in turn, each with an L2 of its own: with piece = b the fourteen pieces of a 3584-pixel row, and the rows two
// above and below that every one of these kernels also reads, lie in eight different L2s and every line is fetched several times.
// Workgroup b takes piece (b % 8) * (G / 8) + b / 8 instead (G = the grid's size, a multiple of 8: flat_grid): an XCD's workgroups
// cover one band of consecutive rows per pass, its neighbours' rows are in its own L2.  OFF by default (flat_grid: measured, slower);
// MLVFS_AMD_DI_XCD=1 in the launcher's environment switches it on (a grid that is no multiple of 8 keeps the plain order).
__device__ __forceinline__ unsigned di_xcd_block()
{
    const unsigned b = blockIdx.x, G = gridDim.x;
    return (G & 7u) ? b : (b & 7u) * (G >> 3) + (b >> 3);
}

// The frame of a batch this workgroup works on -- blockIdx.y, or blockIdx.z for the kernels whose grid is two-dimensional -- and
// its parameters (dualiso.h: DiBatch); false: the batch leaves this frame alone.
template <int DIM>
__device__ __forceinline__ bool di_frame(const DiBatch &b, int &f, DiParams &p)
{
    f = b.f0 + (DIM == 1 ? (int)blockIdx.y : (int)blockIdx.z);
    p = b.pp ? b.pp[f] : b.p0;
    return p.h > 0;
}
// the 16-bit frame f of a batch as the conversion sees it: one row further down for GBRG (hdr.c:1790-1795)
__device__ __forceinline__ const uint16_t *di_img(const uint16_t *base, const DiBatch &b, int f, const DiParams &p)
{
    return (const uint16_t *)((const uint8_t *)base + (size_t)f * b.img_stride) + (size_t)p.ay1 * p.w;
}

// ------------------------------------------------------------------ analysis
// One workgroup = the rows of one phase (y % 4) of a 16-row band.  The 14-bit values of its pixels are counted per
// column parity in LDS (two 16-bit counters per word: a band holds at most 4 * w / 2 < 65536 pixels per class), then
// flushed with one global atomic per non-empty bin into the 8 class histograms [y % 4][x & 1][16384], from which the
// host derives the Bayer-phase and the two green-by-row-phase histograms (dualiso.cpp).  The every-3rd-pixel "white"
// histograms (1/9 of the pixels) and the hdr_check sum use global atomics directly.
// band: rows per workgroup (a multiple of 4).  The taller, the fewer global atomics flush the same bins (a 16-row band of 3584 pixels
// touches ~5 000 of its 32 768 counters for 14 336 pixels: 13 M flush atomics per batch of 8 were most of the kernel); the 16-bit
// counters hold band / 4 * w / 2 pixels per class.
// Timing experiments on a batch of 8 (round 4, rocprofv3 --stats, 419 us): without the white histograms' global atomics 323, without
// hdr_check 329, without the class counters 417; 512 / 1024 threads per workgroup 409 / 453.
__global__ __launch_bounds__(256) void k_di_analyse(const uint16_t *__restrict__ img, int w, int H, int black, int white,
                                                    const double *__restrict__ evf /* [16384] log2(i)*32768 */,
                                                    unsigned *__restrict__ hist /* device layout, dualiso.h */, double *__restrict__ check /* sum, count */,
                                                    size_t img_stride /* bytes */, size_t hist_stride /* words */, size_t check_stride /* doubles */,
                                                    int band)
{
    __shared__ unsigned cnt[16384];                          // slot = (x & 1) * 16384 + value; two slots per word
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    img = (const uint16_t *)((const uint8_t *)img + (size_t)blockIdx.z * img_stride);        // frame of the batch
    hist += (size_t)blockIdx.z * hist_stride;
    check += (size_t)blockIdx.z * check_stride;
    const int q = blockIdx.y, y_base = blockIdx.x * band + q;
    unsigned *h_w0 = hist + DI_D_WHITE0, *h_w1 = hist + DI_D_WHITE1;
    double sum = 0, n = 0;
    for (int k = threadIdx.x; k < (band / 4) * w; k += blockDim.x) {          // (a run of pixels per thread instead -- fewer collisions in the
        const int y = y_base + 4 * (k / w), x = k % w;                             // LDS counters -- was slower: 0.59 -> 0.71 ms per batch of 8)
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

// samples of frame f: rows y = ay1 + 2 + 3 sy while y < h - 2 (hdr.c:664)
__device__ __forceinline__ int di_nsy(const DiParams &p) { const int y0 = p.ay1 + 2; return (p.h - 2 > y0) ? (p.h - 2 - y0 + 2) / 3 : 0; }

// Layout of the sample arrays of a batch: frame f at f * ns_stride (dark_s, bright_s), its two histograms back to back at
// f * 2 * DI_HIST_N (hist_b | hist_d).
// The histograms: a sample's bin is value + DI_HIST_OFF, and value + black is a 16-bit number (a pixel << 2, the mean of two, or the
// white level), so ONE histogram fits into LDS as 65 536 16-bit counters.  grid = (chunks, frames, 2): a workgroup counts hist_b
// (z = 0) or hist_d (z = 1) of one chunk of a frame's samples (at most 65 535, so no counter wraps) and flushes the bins it touched
// with one global atomic each -- a global atomic per sample and histogram (8.4 M per batch of 8, most of them on the few thousand
// bins a scene occupies) was 340 of the kernel's 357 us.  The z = 0 workgroups write the sample arrays.
__global__ __launch_bounds__(1024) void k_di_subsample(const uint16_t *__restrict__ img_base, DiBatch bt, int nsx, size_t ns_stride,
                                                       int per_chunk, int *__restrict__ dark_s, int *__restrict__ bright_s,
                                                       unsigned *__restrict__ hist_bd)
{
    __shared__ unsigned cnt[32768];                          // bin (value + black) >> 1, two 16-bit counters per word
    int f; DiParams p;
    if (!di_frame<1>(bt, f, p)) return;
    const int n = nsx * di_nsy(p), first = blockIdx.x * per_chunk, last = min(first + per_chunk, n), which = blockIdx.z;
    if (first >= n) return;
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    const uint16_t *img = di_img(img_base, bt, f, p);
    dark_s += (size_t)f * ns_stride; bright_s += (size_t)f * ns_stride;
    const int w = p.w, black = p.black20 / 16, white = p.match_white20 / 16;
    const int clip0 = white - black, clip = (int)(clip0 * 0.95);
    for (int idx = first + (int)threadIdx.x; idx < last; idx += blockDim.x) {
        const int sx = idx % nsx, sy = idx / nsx;
        const int x = 3 * sx, y = p.ay1 + 2 + 3 * sy;
        const int pa = di_p16(img, x + (size_t)(y - 2) * w) - black, pb = di_p16(img, x + (size_t)(y + 2) * w) - black;
        int pn = di_p16(img, x + (size_t)y * w) - black;
        int pi = (pa + pb + 1) / 2;
        if (pa >= clip || pb >= clip) pi = clip0;
        if (pi >= clip) pn = clip0;
        const int br = di_bright(p, y);
        const int d = br ? pi : pn, b = br ? pn : pi;
        if (which == 0) { dark_s[idx] = d; bright_s[idx] = b; }
        if (b < clip) {
            const unsigned slot = (unsigned)((which ? d : b) + black) & 0xFFFFu;          // (in range by construction; masked for the LDS bound)
            atomicAdd(&cnt[slot >> 1], 1u << (16 * (slot & 1)));
        }
    }
    __syncthreads();
    // bin of slot s: s - black + DI_HIST_OFF, clamped as the per-sample form clamped it (never at the ends for 16-bit values)
    unsigned *hist = hist_bd + (size_t)f * 2 * DI_HIST_N + (size_t)which * DI_HIST_N;
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) {
        const unsigned v = cnt[i];
        if (v & 0xFFFFu) atomicAdd(&hist[min(max(2 * i - black + DI_HIST_OFF, 0), DI_HIST_N - 1)], v & 0xFFFFu);
        if (v >> 16) atomicAdd(&hist[min(max(2 * i + 1 - black + DI_HIST_OFF, 0), DI_HIST_N - 1)], v >> 16);
    }
}

// highlight pairs of match_exposures (hdr.c:735-746): the samples with b_lo < bright < b_hi in raster order.  One workgroup
// per sample row counts them; the counts are turned into how many each row contributes and where (the reference's cap only leaves
// the inner loop: dualiso.cpp for one frame, k_di_decide_rows for a batch); the second kernel writes them in order.
// grid = (nsy_max, frames); dd != null: the frame's limits come from its decisions on the device
__device__ __forceinline__ bool di_hi_ok(int b, int b_lo, int b_hi) { return !(b >= b_hi || b <= b_lo); }

__global__ __launch_bounds__(256) void k_di_hi_count(const int *__restrict__ bs, int nsx, size_t ns_stride, int b_lo, int b_hi,
                                                     const DiDecide *__restrict__ dd, DiBatch bt, int *__restrict__ counts, int rows_stride)
{
    __shared__ int red[4];
    int f; DiParams p;
    const bool live = di_frame<1>(bt, f, p);
    if (dd) { b_lo = dd[f].b_lo; b_hi = dd[f].b_hi; }
    const int *row = bs + (size_t)f * ns_stride + (size_t)blockIdx.x * nsx;
    int n = 0;
    if (live && (int)blockIdx.x < di_nsy(p))
        for (int x = threadIdx.x; x < nsx; x += blockDim.x) n += di_hi_ok(row[x], b_lo, b_hi);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) counts[(size_t)f * rows_stride + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// rows: [counts | take | offset] of nsy_max ints each per frame (rows_stride = 3 nsy_max); hi: hd | hb of hi_stride ints each per frame
__global__ __launch_bounds__(256) void k_di_hi_compact(const int *__restrict__ ds, const int *__restrict__ bs, int nsx, size_t ns_stride,
                                                       int b_lo, int b_hi, const DiDecide *__restrict__ dd, DiBatch bt,
                                                       const int *__restrict__ rows, int nsy_max, int *__restrict__ hd_base, size_t hi_stride)
{
    __shared__ int part[256];
    int f; DiParams p;
    if (!di_frame<1>(bt, f, p) || (int)blockIdx.x >= di_nsy(p)) return;
    if (dd) { b_lo = dd[f].b_lo; b_hi = dd[f].b_hi; }
    const int *take = rows + (size_t)f * 3 * nsy_max + nsy_max, *offset = take + nsy_max;
    int *hd = hd_base + (size_t)f * 2 * hi_stride, *hb = hd + hi_stride;
    const int lim = take[blockIdx.x];
    if (lim <= 0) return;
    const size_t base = (size_t)f * ns_stride + (size_t)blockIdx.x * nsx;
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

// score of every candidate slope test_a = ta[k] (hdr.c:752-772); its offset test_b = dmed - bmed * test_a in double, uncontracted,
// like the reference's.  grid = (ncand, frames)
__global__ __launch_bounds__(256) void k_di_score(const int *__restrict__ hd_base, size_t hi_stride, int hi_n, const double *__restrict__ ta_tab,
                                                  int dmed, int bmed, const DiDecide *__restrict__ dd, int *__restrict__ score, int score_stride)
{
    __shared__ int red[4];
    const int f = blockIdx.y;
    if (dd) { hi_n = dd[f].hi_n; dmed = dd[f].dmed; bmed = dd[f].bmed; }
    const int *hd = hd_base + (size_t)f * 2 * hi_stride, *hb = hd + hi_stride;
    const double ta = ta_tab[blockIdx.x], tb = dmed - bmed * ta;
    int s = 0;
    for (int i = threadIdx.x; i < hi_n; i += blockDim.x) {
        const int e = (int)(hd[i] - (hb[i] * ta + tb));
        s += (e > 0 ? e : -e) < 50;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) score[(size_t)f * score_stride + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// With `cfa` (the AMaZE-based interpolator) the squeeze of hdr.c:977-1026 rides along: rows of one exposure become adjacent, greens
// halved around black, and a squeezed row that no exposure lands on is zeroed by the thread whose image row has its number (sq: per
// frame sq_dst | sq_row | source row of a squeezed row) -- one pass over the frame instead of a 151 MB memset and two.
// With `ev` (mean23) the interpolator's raw2ev of every matched pixel rides along instead: each is asked for by two or three neighbours.
__global__ __launch_bounds__(256) void k_di_match(const uint16_t *__restrict__ img_base, uint32_t *__restrict__ raw, DiBatch bt,
                                                  const int *__restrict__ sq, size_t sq_stride, int hs, float *__restrict__ cfa,
                                                  const int *__restrict__ r2e, int *__restrict__ ev)
{
    int f; DiParams p;
    if (!di_frame<1>(bt, f, p)) return;
    const uint16_t *img = di_img(img_base, bt, f, p);
    raw += (size_t)f * bt.S;
    if (cfa) { cfa += (size_t)f * bt.S; sq += (size_t)f * sq_stride; }
    if (ev) ev += (size_t)f * bt.S;
    const size_t n = (size_t)p.w * p.h;
    const double a = p.a, b20 = p.b20;
    for (size_t i = (size_t)di_xcd_block() * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
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
        if (ev) ev[i] = r2e[v];
        if (cfa) {
            const int x = (int)(i % p.w), y = (int)(i / p.w);
            const int yh = sq[y];
            if (yh >= 0) cfa[(size_t)yh * p.w + x] = (float)(((x & 1) != (y & 1)) ? (v - p.black20) / 2 + p.black20 : v);
            if (sq[2 * hs + y] < 0) cfa[i] = 0.0f;
        }
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

// everything that follows the interpolation for one pixel: full-res pick (hdr.c:1355-1380), half-res mix (hdr.c:1588-1612),
// overexposure flag (hdr.c:1620-1626) and, when no chroma smoothing sits in between, the alias error --
// with the re-packed table (DiLuts::mix_pair; until round 5 also a 16-byte entry per bright value): 4 gathers instead of 8 -- what is looked
// up at the bright value, the full-res pick f is b or d (its table value is already here), and ev2raw comes with its raw2ev
// ev_out: the two planes go out as EV values (what k_di_blend looks up of them anyway, when no chroma smoothing sits in between)
// the weight of the half-res mix at t (hdr.c:1572-1575)
__device__ __forceinline__ double di_mix_weight(const DiParams &p, double t)
{
    double k = (-cos(t * 3.14159265358979323846 / p.overlap) + 1) / 2;
    return k < 0 ? 0 : (k > 1 ? 1 : k);
}
// Round 5: the weight depends on the bright value alone and is constant outside a band of it (DiParams::mix_lo / mix_hi): k_lo / k_hi are
// the weights at t = 0 and t = overlap, evaluated once per thread with the same expression; only pixels inside the band read log2 of
// their signal (8 MB of doubles) and take the cosine.  What was looked up at the bright value in one 16-byte entry (DiLuts::by_bright,
// 16 MB) is then mix_raw2ev -- the table the dark value goes through anyway -- and a comparison with fullres_thr.
__device__ __forceinline__ void di_mix_pixel_packed(const DiParams &p, const DiLuts &L, size_t i, int br, int b, int d,
                                                    uint32_t *__restrict__ fullres, uint32_t *__restrict__ halfres,
                                                    uint16_t *__restrict__ over, uint16_t *__restrict__ amap, bool ev_out, double k_lo, double k_hi)
{
    const int bm = b & 0xFFFFF;
    const bool in_band = bm >= p.mix_lo && bm < p.mix_hi;
    double log2sig = 0;
    if (in_band) log2sig = L.log2sig[bm];                                          // (issued with the other look-ups, not behind them)
    struct { int mix_raw2ev; int fullres_hi; } tb = { L.mix_raw2ev[bm], bm >= L.fullres_thr };
    const int ev_b = tb.mix_raw2ev, ev_d = L.mix_raw2ev[d];
    int f = 0, ev_f = 0;
    if (p.use_fullres) {
        const bool take_b = br && (b < p.white_darkened || b >= d);              // f = br ? (b < white_darkened ? b : max(b, d)) : d
        f = take_b ? b : d;
        ev_f = take_b ? ev_b : ev_d;
    } else if (ev_out || amap) ev_f = L.mix_raw2ev[0];
    fullres[i] = ev_out ? (uint32_t)ev_f : (uint32_t)f;
    double k = bm < p.mix_lo ? k_lo : k_hi;
    if (in_band) {
        const double ev = log2sig + p.corr_ev;
        double t = ev - (p.max_ev - p.overlap);
        t = t < p.overlap ? t : p.overlap;
        t = t > 0 ? t : 0;
        k = di_mix_weight(p, t);
    }
    const int mixed = (int)(ev_b * (1 - k) + ev_d * k);
    const int2 hp = L.mix_pair[mixed];
    const int hr = hp.x;
    halfres[i] = ev_out ? (uint32_t)hp.y : (uint32_t)hr;
    over[i] = (b >= p.white_darkened || d >= p.white20) ? 100 : 0;
    if (amap) {                                                                   // di_alias_err
        int err = 0;
        if (!tb.fullres_hi) {
            int e_lin = f - hr;
            e_lin = e_lin > 0 ? e_lin : -e_lin;
            e_lin = max(e_lin - p.dark_noise * 3 / 2, 0);
            int e_log = ev_f - hp.y;
            e_log = e_log > 0 ? e_log : -e_log;
            err = min(min(e_lin / 2, e_log / 16), 65530);
        }
        amap[i] = (uint16_t)err;
    }
}

// edge directions of the AMaZE-based interpolation, hdr.c:916-938: {ack, a, b, bck} x {x, y}; y is multiplied by s
__constant__ signed char k_edge_dirs[11][8] = {
    { -4, 2, -2, 1, 4, -2, 6, -3 }, { -3, 2, -1, 1, 3, -2, 4, -3 }, { -2, 2, -1, 1, 2, -2, 3, -3 }, { -1, 2, -1, 1, 1, -2, 2, -3 },
    { -1, 2, 0, 1, 1, -2, 1, -3 },  { 0, 2, 0, 1, 0, -2, 0, -3 },   { 1, 2, 0, 1, -1, -2, -1, -3 }, { 1, 2, 1, 1, -1, -2, -2, -3 },
    { 2, 2, 1, 1, -2, -2, -3, -3 }, { 3, 2, 1, 1, -3, -2, -4, -3 }, { 4, 2, 2, 1, -4, -2, -6, -3 } };

struct DiAmazeIn {              // inputs of the edge-directed interpolation (null planes = mean23)
    const int *red, *green, *blue;      // interp_raw2ev of the clamped demosaic (k_di_amaze_clamp)
    const uint8_t *dir;
    const int *sq_row;
    size_t sq_stride;           // ints between the frames' squeezed-row maps
};

// one pixel of the interpolation and everything behind it (the body of k_di_interp).  DIR >= 0: the edge direction comes with the
// call (k_di_edge_interp), else from A.dir
template <bool AMAZE>
__device__ __forceinline__ void di_interp_pixel(const DiParams &p, const DiLuts &L, const DiAmazeIn &A, const uint32_t *__restrict__ raw,
                                                size_t i, int x, int y, int dir_given,
                                                uint32_t *__restrict__ dark, uint32_t *__restrict__ bright,
                                                uint32_t *__restrict__ fullres, uint32_t *__restrict__ halfres,
                                                uint16_t *__restrict__ over, uint16_t *__restrict__ amap, bool ev_out, double k_lo, double k_hi)
{
    const int w = p.w, h = p.h;
    const int *ir2e = L.interp_raw2ev, *ie2r = L.interp_ev2raw;
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
        const int *plane = (y & 1) == 0 ? ((x & 1) == 0 ? A.red : A.green) : ((x & 1) == 0 ? A.green : A.blue);
        const int d = dir_given >= 0 ? dir_given : (int)A.dir[i];
        const int dd[3] = { d, min(d + 1, 10), max(d - 1, 0) };
        // of a direction's table row only a.x and b.x vary (a.y = 1, b.y = -2 for all eleven, hdr.c:916-938): two rows of the
        // plane for the three directions, the column offsets from two packed constants instead of twelve byte loads per pixel
        const int *row_a = plane + (size_t)A.sq_row[y + s] * w + x, *row_b = plane + (size_t)A.sq_row[y - 2 * s] * w + x;
        constexpr unsigned long long AX = 0x43332221110ull, BX = 0x01233455678ull;       // a.x + 2, b.x + 4 of directions 0..10, 4 bits each
        int pi[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int ax = (int)((AX >> (4 * dd[k])) & 15) - 2, bx = (int)((BX >> (4 * dd[k])) & 15) - 4;
            const int ea = row_a[ax];                                                    // raw2ev of the clamped plane value
            const int eb = row_b[bx];
            pi[k] = (ea * 2 + eb) / 3;
        }
        interp = ie2r[(2 * pi[0] + pi[1] + pi[2]) / 4];
        native = R(x, y);
    } else {
        const int wl = !br ? p.white_darkened : p.white20;
        const int wev = ir2e[wl];
        auto E = [&](int xx, int yy) { return A.red[xx + (size_t)yy * w]; };        // (mean23: A.red is raw2ev of the matched frame)
        const int s = (di_bright(p, y) == di_bright(p, y + 1)) ? -1 : 1;
        const int xe = x & ~1;                                // the pair (xe, xe+1) is produced together
        int ev;
        if ((y & 1) == 0) {
            if (x == xe) ev = di_mean2(E(xe, y - 2), E(xe, y + 2), wev);
            else ev = di_mean3(E(xe + 2, y + s), E(xe, y + s), E(xe + 1, y - 2 * s), wev);
        } else {
            if (x == xe) ev = di_mean3(E(xe + 1, y + s), E(xe - 1, y + s), E(xe, y - 2 * s), wev);
            else ev = di_mean2(E(xe + 1, y - 2), E(xe + 1, y + 2), wev);
        }
        interp = ie2r[ev];
        native = R(x, y);
    }
    const int b = br ? native : interp, d = br ? interp : native;
    bright[i] = (uint32_t)b;
    dark[i] = (uint32_t)d;
    di_mix_pixel_packed(p, L, i, br, b, d, fullres, halfres, over, amap, ev_out, k_lo, k_hi);
}

template <bool AMAZE>
__global__ __launch_bounds__(256) void k_di_interp(const uint32_t *__restrict__ raw, DiBatch bt, DiLuts L, DiAmazeIn A,
                                                   uint32_t *__restrict__ dark, uint32_t *__restrict__ bright,
                                                   uint32_t *__restrict__ fullres, uint32_t *__restrict__ halfres,
                                                   uint16_t *__restrict__ over, uint16_t *__restrict__ amap, bool ev_out)
{
    int f; DiParams p;
    if (!di_frame<1>(bt, f, p)) return;
    {
        const size_t o = (size_t)f * bt.S;
        raw += o; dark += o; bright += o; fullres += o; halfres += o; over += o;
        if (amap) amap += o;
        if (AMAZE) { A.red += o; A.green += o; A.blue += o; A.dir += o; A.sq_row += (size_t)f * A.sq_stride; }
        else A.red += o;
    }
    const int w = p.w, h = p.h;
    const size_t n = (size_t)w * h;
    const double k_lo = di_mix_weight(p, 0.0), k_hi = di_mix_weight(p, p.overlap);
    for (size_t i = (size_t)di_xcd_block() * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        di_interp_pixel<AMAZE>(p, L, A, raw, i, (int)(i % w), (int)(i / w), -1, dark, bright, fullres, halfres, over, amap, ev_out, k_lo, k_hi);
}

// ------------------------------------------------------------------ AMaZE-based interpolation, hdr.c:954-1229
// undo the green scaling, clamp (hdr.c:1041-1050); what leaves this kernel are the table values everything downstream looks up:
// raw2ev of the three clamped planes (the edge-directed interpolation, hdr.c:1181-1208: up to six pixels ask for each) and raw2ev of
// the gray image (hdr.c:1055-1059, 1157-1168) -- still squeezed: k_di_edge_dir de-squeezes through the row map when it stages its rows
__global__ __launch_bounds__(256) void k_di_amaze_ev(const float *__restrict__ red, const float *__restrict__ green, const float *__restrict__ blue,
                                                     DiBatch bt, const int *__restrict__ r2e, int *__restrict__ ev_red,
                                                     int *__restrict__ ev_green, int *__restrict__ ev_blue, int *__restrict__ gray_sq)
{
    int f; DiParams p;
    if (!di_frame<1>(bt, f, p)) return;
    red += (size_t)f * bt.S; green += (size_t)f * bt.S; blue += (size_t)f * bt.S;
    ev_red += (size_t)f * bt.S; ev_green += (size_t)f * bt.S; ev_blue += (size_t)f * bt.S; gray_sq += (size_t)f * bt.S;
    const size_t n = (size_t)p.w * p.h;
    const int black = p.black20;
    // (four pixels per thread -- float4 in, int4 out, sixteen look-ups in flight -- made this kernel 10 % shorter and the batch of 8
    // 4.5 % LONGER, three rounds round-robin: it runs beside AMaZE's kernels on the other stream and took more of the chip from them;
    // profiles/r04/ab_di_bench.log)
    for (size_t i = (size_t)di_xcd_block() * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float fb = (float)black, hi = 1048575.0f;
        const float g = (green[i] - fb) * 2.0f + fb, r = red[i], b = blue[i];
        const float gc = g < hi ? (g > 0.0f ? g : 0.0f) : hi, rc = r < hi ? (r > 0.0f ? r : 0.0f) : hi, bc = b < hi ? (b > 0.0f ? b : 0.0f) : hi;
        ev_green[i] = r2e[(int)gc]; ev_red[i] = r2e[(int)rc]; ev_blue[i] = r2e[(int)bc];
        gray_sq[i] = r2e[(unsigned)(gc / 2 + rc / 4 + bc / 4)];
    }
}

// best of 11 edge directions where the interpolation has to be good (hdr.c:1096-1173)
// A workgroup owns 256 pixels of one row.  All eleven directions compare the same four rows (y + 2s, y + s, y - 2s, y - 3s) at
// column shifts within +-11: the rows' 278 values are staged in LDS once (only when a pixel of the segment searches at all),
// biased to unsigned, each searching lane takes its 4 x 23 window into registers and the 363 |a - b| + c of the search are one
// v_sad_u32 each -- 92 LDS reads and 363 instructions where the per-direction loops made 484 loads and ~1 100 instructions.
// Columns left of 0 / right of w - 1 are the flat neighbours of the reference's indexing (the previous / next row).
constexpr signed char EDGE_DIRS[11][8] = {
    { -4, 2, -2, 1, 4, -2, 6, -3 }, { -3, 2, -1, 1, 3, -2, 4, -3 }, { -2, 2, -1, 1, 2, -2, 3, -3 }, { -1, 2, -1, 1, 1, -2, 2, -3 },
    { -1, 2, 0, 1, 1, -2, 1, -3 },  { 0, 2, 0, 1, 0, -2, 0, -3 },   { 1, 2, 0, 1, -1, -2, -1, -3 }, { 1, 2, 1, 1, -1, -2, -2, -3 },
    { 2, 2, 1, 1, -2, -2, -3, -3 }, { 3, 2, 1, 1, -3, -2, -4, -3 }, { 4, 2, 2, 1, -4, -2, -6, -3 } };
__device__ __forceinline__ unsigned di_sad(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__global__ __launch_bounds__(256) void k_di_edge_dir(const uint32_t *__restrict__ raw, const int *__restrict__ gray_sq, DiBatch bt,
                                                     const int *__restrict__ sq_row, size_t sq_stride, int fullres_thr,
                                                     uint8_t *__restrict__ dir, unsigned *__restrict__ stats)
{
    constexpr int REACH = 11, SPAN = 256 + 2 * REACH, BIAS = 10 * DI_EVR;        // raw2ev >= -10 EV
    __shared__ unsigned s_rows[4][SPAN];
    int f; DiParams p;
    if (!di_frame<2>(bt, f, p)) return;
    raw += (size_t)f * bt.S; gray_sq += (size_t)f * bt.S; dir += (size_t)f * bt.S; sq_row += (size_t)f * sq_stride;
    stats += ((size_t)f * DI_STAT_SLOTS + ((blockIdx.x + blockIdx.y) & (DI_STAT_SLOTS - 1))) * 4;
    const int w = p.w, h = p.h, x0 = blockIdx.x * 256, x = x0 + (int)threadIdx.x;
    unsigned n_search = 0, n_plain = 0;                                          // this lane's pixels, bright rows in the low half, dark rows << 16
    for (int y = blockIdx.y; y < h; y += gridDim.y) {                            // (a band of rows per workgroup: 4 atomics per workgroup, not per row)
    const size_t i = (size_t)y * w + x;
    const int br = di_bright(p, y);
    bool search = false;
    if (x >= 5 && x < w - 5 && y >= 5 && y < h - 5) {
        const int v = (int)raw[i];
        search = br ? !(v < p.white_darkened) : v < fullres_thr;                  // !(fullres_curve[v] > 0.8)
        if (search) n_search += br ? 1u : 0x10000u; else n_plain += br ? 1u : 0x10000u;
    }
    int best = 5;
    if (__syncthreads_or(search)) {
        const int s = (br == di_bright(p, y + 1)) ? -1 : 1;
        for (int k = threadIdx.x; k < 4 * SPAN; k += 256) {
            const int rr = k / SPAN, cc = k - rr * SPAN;
            const int row = y + (rr == 0 ? 2 : rr == 1 ? 1 : rr == 2 ? -2 : -3) * s;
            int gy = row, gx = x0 + cc - REACH;                                  // the reference indexes the gray image flat: columns off
            if (gx < 0) { gy--; gx += w; }                                       // the row's ends are the neighbouring rows'
            while (gx >= w && gy < h - 1) { gy++; gx -= w; }
            gx = gx < w ? gx : w - 1;                                            // (beyond the image's last pixel: nobody reads it)
            s_rows[rr][cc] = (unsigned)(gray_sq[(size_t)sq_row[gy] * w + gx] + BIAS);    // de-squeezed here (hdr.c:1055-1059)
        }
        __syncthreads();
        if (search) {
            unsigned win[4][2 * REACH + 1];
#pragma unroll
            for (int rr = 0; rr < 4; rr++)
#pragma unroll
                for (int c = 0; c < 2 * REACH + 1; c++) win[rr][c] = s_rows[rr][threadIdx.x + c];
            unsigned e_best = 0xFFFFFFFFu;
#pragma unroll
            for (int d = 0; d < 11; d++) {
                unsigned err = (unsigned)((d > 5 ? d - 5 : 5 - d) * DI_EVR / 8);
#pragma unroll
                for (int j = -5; j <= 5; j++) {
                    const unsigned p1 = win[0][REACH + EDGE_DIRS[d][0] + j], p2 = win[1][REACH + EDGE_DIRS[d][2] + j];
                    const unsigned p3 = win[2][REACH + EDGE_DIRS[d][4] + j], p4 = win[3][REACH + EDGE_DIRS[d][6] + j];
                    err = di_sad(p1, p2, err); err = di_sad(p2, p3, err); err = di_sad(p3, p4, err);
                }
                if (err < e_best) { e_best = err; best = d; }
            }
        }
    }
    if (x < w) dir[i] = (uint8_t)best;
    __syncthreads();                                                             // (s_rows is reused by the next row)
    }
    {   // the reference's four counters: semi-overexposed / not (bright rows), deep shadow / not (dark rows)
        unsigned a = n_search & 0xFFFFu, b = n_plain & 0xFFFFu, c = n_search >> 16, d = n_plain >> 16;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); c += __shfl_xor(c, o); d += __shfl_xor(d, o); }
        if ((threadIdx.x & 63) == 0) {
            if (a) atomicAdd(&stats[0], a);
            if (b) atomicAdd(&stats[1], b);
            if (c) atomicAdd(&stats[2], c);
            if (d) atomicAdd(&stats[3], d);
        }
    }
}

// k_di_edge_dir and k_di_interp<true> in one (round 5): the direction a pixel's search finds goes straight into its interpolation --
// no direction plane, one pass over the matched frame instead of two, and the search's arithmetic runs beside the interpolation's
// table look-ups in the same workgroup
__global__ __launch_bounds__(256, 4) void k_di_edge_interp(const uint32_t *__restrict__ raw, const int *__restrict__ gray_sq, DiBatch bt,
                                                        const int *__restrict__ sq_row, size_t sq_stride, int fullres_thr,
                                                        unsigned *__restrict__ stats, DiLuts L, DiAmazeIn A,
                                                        uint32_t *__restrict__ dark, uint32_t *__restrict__ bright,
                                                        uint32_t *__restrict__ fullres, uint32_t *__restrict__ halfres,
                                                        uint16_t *__restrict__ over, uint16_t *__restrict__ amap, bool ev_out)
{
    constexpr int REACH = 11, SPAN = 256 + 2 * REACH, BIAS = 10 * DI_EVR;        // raw2ev >= -10 EV
    __shared__ unsigned s_rows[4][SPAN];
    int f; DiParams p;
    if (!di_frame<2>(bt, f, p)) return;
    raw += (size_t)f * bt.S; gray_sq += (size_t)f * bt.S; sq_row += (size_t)f * sq_stride;
    {
        const size_t o = (size_t)f * bt.S;
        dark += o; bright += o; fullres += o; halfres += o; over += o;
        if (amap) amap += o;
        A.red += o; A.green += o; A.blue += o; A.sq_row = sq_row;
    }
    const double k_lo = di_mix_weight(p, 0.0), k_hi = di_mix_weight(p, p.overlap);
    stats += ((size_t)f * DI_STAT_SLOTS + ((blockIdx.x + blockIdx.y) & (DI_STAT_SLOTS - 1))) * 4;
    const int w = p.w, h = p.h, x0 = blockIdx.x * 256, x = x0 + (int)threadIdx.x;
    unsigned n_search = 0, n_plain = 0;                                          // this lane's pixels, bright rows in the low half, dark rows << 16
    // (consecutive rows per workgroup, unlike k_di_edge_dir's stride of gridDim.y: the rows of the planes and of the gray image that
    // one row reads are the next row's too -- with the stride the kernel fetched 582 MB per frame, 136 more than the two kernels it replaces)
    const int rows_per = (h + (int)gridDim.y - 1) / (int)gridDim.y;
    for (int y = (int)blockIdx.y * rows_per, y_end = min(y + rows_per, h); y < y_end; y++) {                            // (a band of rows per workgroup: 4 atomics per workgroup, not per row)
    const size_t i = (size_t)y * w + x;
    const int br = di_bright(p, y);
    bool search = false;
    if (x >= 5 && x < w - 5 && y >= 5 && y < h - 5) {
        const int v = (int)raw[i];
        search = br ? !(v < p.white_darkened) : v < fullres_thr;                  // !(fullres_curve[v] > 0.8)
        if (search) n_search += br ? 1u : 0x10000u; else n_plain += br ? 1u : 0x10000u;
    }
    int best = 5;
    if (__syncthreads_or(search)) {
        const int s = (br == di_bright(p, y + 1)) ? -1 : 1;
        for (int k = threadIdx.x; k < 4 * SPAN; k += 256) {
            const int rr = k / SPAN, cc = k - rr * SPAN;
            const int row = y + (rr == 0 ? 2 : rr == 1 ? 1 : rr == 2 ? -2 : -3) * s;
            int gy = row, gx = x0 + cc - REACH;                                  // the reference indexes the gray image flat: columns off
            if (gx < 0) { gy--; gx += w; }                                       // the row's ends are the neighbouring rows'
            while (gx >= w && gy < h - 1) { gy++; gx -= w; }
            gx = gx < w ? gx : w - 1;                                            // (beyond the image's last pixel: nobody reads it)
            s_rows[rr][cc] = (unsigned)(gray_sq[(size_t)sq_row[gy] * w + gx] + BIAS);    // de-squeezed here (hdr.c:1055-1059)
        }
        __syncthreads();
        if (search) {
            unsigned win[4][2 * REACH + 1];
#pragma unroll
            for (int rr = 0; rr < 4; rr++)
#pragma unroll
                for (int c = 0; c < 2 * REACH + 1; c++) win[rr][c] = s_rows[rr][threadIdx.x + c];
            unsigned e_best = 0xFFFFFFFFu;
#pragma unroll
            for (int d = 0; d < 11; d++) {
                unsigned err = (unsigned)((d > 5 ? d - 5 : 5 - d) * DI_EVR / 8);
#pragma unroll
                for (int j = -5; j <= 5; j++) {
                    const unsigned p1 = win[0][REACH + EDGE_DIRS[d][0] + j], p2 = win[1][REACH + EDGE_DIRS[d][2] + j];
                    const unsigned p3 = win[2][REACH + EDGE_DIRS[d][4] + j], p4 = win[3][REACH + EDGE_DIRS[d][6] + j];
                    err = di_sad(p1, p2, err); err = di_sad(p2, p3, err); err = di_sad(p3, p4, err);
                }
                if (err < e_best) { e_best = err; best = d; }
            }
        }
    }
    if (x < w) di_interp_pixel<true>(p, L, A, raw, i, x, y, best, dark, bright, fullres, halfres, over, amap, ev_out, k_lo, k_hi);
    __syncthreads();                                                             // (s_rows is reused by the next row)
    }
    {   // the reference's four counters: semi-overexposed / not (bright rows), deep shadow / not (dark rows)
        unsigned a = n_search & 0xFFFFu, b = n_plain & 0xFFFFu, c = n_search >> 16, d = n_plain >> 16;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); c += __shfl_xor(c, o); d += __shfl_xor(d, o); }
        if ((threadIdx.x & 63) == 0) {
            if (a) atomicAdd(&stats[0], a);
            if (b) atomicAdd(&stats[1], b);
            if (c) atomicAdd(&stats[2], c);
            if (d) atomicAdd(&stats[3], d);
        }
    }
}

// alias error from the chroma-smoothed planes (hdr.c:1620: build_alias_map gets fullres_smooth / halfres_smooth)
__global__ __launch_bounds__(256) void k_di_alias_err(const uint32_t *__restrict__ bright, const uint32_t *__restrict__ fullres_s,
                                                      const uint32_t *__restrict__ halfres_s, DiBatch bt, DiLuts L,
                                                      uint16_t *__restrict__ amap)
{
    int f; DiParams p;
    if (!di_frame<1>(bt, f, p)) return;
    bright += (size_t)f * bt.S; fullres_s += (size_t)f * bt.S; halfres_s += (size_t)f * bt.S; amap += (size_t)f * bt.S;
    const size_t n = (size_t)p.w * p.h;
    for (size_t i = (size_t)di_xcd_block() * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        amap[i] = (uint16_t)di_alias_err(p, L, (int)bright[i], (int)fullres_s[i], (int)halfres_s[i]);
}

// ------------------------------------------------------------------ chroma smoothing of a 20-bit plane
// hdr.c:1488-1522 = chroma_smooth.c:22-71 for uint32_t pixels, black 0 and the 20-bit tables of mix_images.
// Pass 1 turns every Bayer cell into (green EV, R - green, B - green); pass 2 takes the medians over the cell
// neighbourhood and rewrites R and B of the cells the reference visits.  The planes stay L2-resident in between.
__global__ __launch_bounds__(256) void k_di_cs_cells(const uint32_t *__restrict__ plane, DiBatch bt, size_t cells_stride,
                                                     const int *__restrict__ r2e, int *__restrict__ cells /* [3][ch][cw] */)
{
    int f; DiParams p;
    if (!di_frame<2>(bt, f, p)) return;
    const int w = p.w, cw = p.w / 2, ch = p.h / 2;
    plane += (size_t)f * bt.S; cells += (size_t)f * cells_stride;
    const int cx = blockIdx.x * blockDim.x + threadIdx.x, cy = blockIdx.y;
    if (cx >= cw || cy >= ch) return;
    const uint32_t *c = plane + 2 * cx + (size_t)(2 * cy) * w;
    const int ge = (r2e[c[1]] + r2e[c[w]]) / 2;
    const size_t o = cx + (size_t)cy * cw, pl = (size_t)cw * ch;
    cells[o] = ge;
    cells[o + pl] = r2e[c[0]] - ge;
    cells[o + 2 * pl] = r2e[c[w + 1]] - ge;
}

template <int METHOD>
__global__ __launch_bounds__(256) void k_di_cs_apply(const int *__restrict__ cells, DiBatch bt, size_t cells_stride,
                                                     const int *__restrict__ e2r, uint32_t *__restrict__ out)
{
    int f; DiParams p;
    if (!di_frame<2>(bt, f, p)) return;
    const int w = p.w, h = p.h, cw = p.w / 2, ch = p.h / 2;
    out += (size_t)f * bt.S; cells += (size_t)f * cells_stride;
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
                                                       int fullres_thr, DiBatch bt, uint16_t *__restrict__ aux)
{
    int f; DiParams p;
    if (!di_frame<2>(bt, f, p)) return;
    const int w = p.w, h = p.h;
    amap += (size_t)f * bt.S; bright += (size_t)f * bt.S; aux += (size_t)f * bt.S;
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    const size_t i = x + (size_t)y * w;
    int out = amap[i];
    if (x >= 6 && x < w - 6 && y >= 6 && y < h - 6 && (int)bright[i] < fullres_thr) {                          // !(fullres_curve[bright] > 0.8)
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
                                                       const uint32_t *__restrict__ bright, int fullres_thr,
                                                       DiBatch bt, uint16_t *__restrict__ out)
{
    int f; DiParams p;
    if (!di_frame<2>(bt, f, p)) return;
    const int w = p.w, h = p.h;
    aux += (size_t)f * bt.S; amap_in += (size_t)f * bt.S; bright += (size_t)f * bt.S; out += (size_t)f * bt.S;
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    const size_t i = x + (size_t)y * w;
    int v = amap_in[i];
    if (x >= 6 && x < w - 6 && y >= 6 && y < h - 6 && (int)bright[i] < fullres_thr) {
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
                                                  const uint16_t *__restrict__ amap, DiBatch bt, DiLuts L,
                                                  uint16_t *__restrict__ img_base, bool ev_planes)
{
    int f; DiParams p;
    if (!di_frame<1>(bt, f, p)) return;
    {
        const size_t o = (size_t)f * bt.S;
        dark += o; bright += o; fullres += o; fullres_s += o; halfres_s += o; over += o;
        if (amap) amap += o;
    }
    uint16_t *img_out = const_cast<uint16_t *>(di_img(img_base, bt, f, p));
    const int w = p.w, h = p.h;
    const size_t n = (size_t)w * h;
    const int *r2e = L.blend_raw2ev, *e2r = L.blend_ev2raw;
    for (size_t i = (size_t)di_xcd_block() * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
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
        int hrev, frev, frsev;
        if (ev_planes) { hrev = (int)halfres_s[i]; frev = frsev = (int)fullres[i]; }                       // k_di_interp<true> wrote EV values
        else { hrev = r2e[halfres_s[i]]; frev = r2e[fullres[i]]; frsev = fullres_s == fullres ? frev : r2e[fullres_s[i]]; }
        const int bm = b & 0xFFFFF;                                                  // (the curve is constant outside a band of b: DiLuts::fr_lo / fr_hi)
        double f = bm < L.fr_lo ? L.fr_lo_val : (bm >= L.fr_hi ? L.fr_hi_val : L.fullres_curve[bm]);
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

// ------------------------------------------------------------------ decisions of a batch, on the device
// The scalar decisions that dualiso.cpp makes on the host between the kernels of ONE conversion -- pattern, bright / dark fields,
// white levels (hdr.c:441-636, 250-300), the order statistics, the highlight rows' shares and the winning slope of match_exposures
// (hdr.c:638-772) -- for every frame of a batch at once, one workgroup per frame, so that a batch needs ONE round trip to the
// host (for the libm scalars log2 / pow and the reference's progress lines) instead of five per frame.  All of it is integer work
// on histograms; each step restates the host code it replaces (named in its comment), and tests/test_gpu_dualiso.py requires
// the two to agree value by value.
namespace {

constexpr int DNT = 1024;                                  // threads of a decision workgroup (one per frame: latency is what it costs)

__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, unsigned long long *sh)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned long long r = 0;
    for (int k = 0; k < DNT / 64; k++) r += sh[k];
    __syncthreads();
    return r;
}
__device__ __forceinline__ long long block_min_i64(long long v, long long *sh)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const long long u = __shfl_xor(v, o); v = u < v ? u : v; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    long long r = sh[0];
    for (int k = 1; k < (int)blockDim.x / 64; k++) r = sh[k] < r ? sh[k] : r;
    __syncthreads();
    return r;
}
// exclusive prefix of one value per thread (DNT threads), total in *total: per wave by shuffles, the waves' sums in LDS
__device__ __forceinline__ unsigned long long block_excl_scan(unsigned long long v, unsigned long long *sh, unsigned long long *total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned long long u = __shfl_up(inc, o); if (lane >= o) inc += u; }
    __syncthreads();
    if (lane == 63) sh[wv] = inc;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
    for (int k = 0; k < DNT / 64; k++) { const unsigned long long t = sh[k]; if (k < wv) base += t; tot += t; }
    if (total) *total = tot;
    __syncthreads();
    return base + inc - v;
}
// in place: hist[0..n) -> C[r] = sum of hist[v], v < r, for r = 0..n (n + 1 words, the array has room); DNT threads, n % DNT == 0
template <int n> __device__ void block_prefix_in_place(unsigned *hist, unsigned long long *sh)
{
    constexpr int per = n / DNT;
    const int i0 = threadIdx.x * per;
    unsigned v[per];
    unsigned long long s = 0;
#pragma unroll
    for (int i = 0; i < per; i++) { v[i] = hist[i0 + i]; s += v[i]; }
    unsigned long long tot;
    unsigned long long run = block_excl_scan(s, sh, &tot);
#pragma unroll
    for (int i = 0; i < per; i++) { hist[i0 + i] = (unsigned)run; run += v[i]; }
    if (threadIdx.x == DNT - 1) hist[n] = (unsigned)run;
    __syncthreads();
}
// smallest r in [0, n] with C[r] >= ref, n if none (C non-decreasing, n + 1 entries)
__device__ __forceinline__ int quantile_of(const unsigned *C, int n, long long ref)
{
    if (ref <= 0) return 0;
    if ((long long)C[n] < ref) return n;
    int lo = 0, hi = n;                                   // C[lo] < ref <= C[hi]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((long long)C[mid] >= ref) hi = mid; else lo = mid; }
    return hi;
}
// kth_from_hist of rounds 1-2's host code -- the first index i (ascending) with sum of hist[0..i] > k, BINS - 1 if none -- in two steps,
// so that several k of one histogram share the sweep: kth_scan sums a chunk of BINS / DNT bins per thread and scans the sums
// (DNT threads, BINS % DNT == 0, at most 128 bins per thread); kth_resolve finds up to four k: the thread whose chunk holds a
// crossing names it, and wave t finds k[t] inside that chunk with one scan over its lanes (the walk of one thread through its chunk,
// a dependent load per bin, was most of the two decision kernels' 300 us).
struct KthScan { unsigned long long s, run, tot; };
struct KthScratch { unsigned long long run[4]; int chunk[4], res[4]; };
template <int BINS> __device__ __forceinline__ KthScan kth_scan(const unsigned *hist, unsigned long long *sh)
{
    constexpr int per = BINS / DNT;
    const int i0 = threadIdx.x * per;
    KthScan r;
    r.s = 0;
#pragma unroll 16
    for (int i = 0; i < per; i++) r.s += hist[i0 + i];
    r.run = block_excl_scan(r.s, sh, &r.tot);
    return r;
}
template <int BINS> __device__ __forceinline__ void kth_resolve(const unsigned *hist, const KthScan &sc, const long long *k, int nk, int *out,
                                                                KthScratch &ks)
{
    constexpr int per = BINS / DNT, m = (per + 63) / 64;      // bins per lane of the resolving wave
    static_assert(per <= 128, "a chunk is resolved by one wave, two bins per lane");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < 4) { ks.chunk[tid] = -1; ks.res[tid] = BINS - 1; }
    __syncthreads();
    for (int t = 0; t < nk; t++)
        if ((long long)sc.run <= k[t] && (long long)(sc.run + sc.s) > k[t]) { ks.chunk[t] = tid * per; ks.run[t] = sc.run; }
    __syncthreads();
    if (wv < nk && ks.chunk[wv] >= 0) {
        const int c0 = ks.chunk[wv];
        const long long kk = k[wv];
        unsigned v[m];
        unsigned long long ls = 0;
#pragma unroll
        for (int j = 0; j < m; j++) { const int b = lane * m + j; v[j] = b < per ? hist[c0 + b] : 0u; ls += v[j]; }
        unsigned long long inc = ls;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned long long u = __shfl_up(inc, o); if (lane >= o) inc += u; }
        unsigned long long before = ks.run[wv] + inc - ls;
        if ((long long)before <= kk && (long long)(before + ls) > kk) {
#pragma unroll
            for (int j = 0; j < m; j++) { before += v[j]; if ((long long)before > kk) { ks.res[wv] = c0 + lane * m + j; break; } }
        }
    }
    __syncthreads();
    for (int t = 0; t < nk; t++) out[t] = ks.res[t];
    __syncthreads();
}

}  // namespace

// derived (per frame, words): hb [4][16384] | g [4][16385] (greens by row phase of the frame's pattern, then their prefix sums) |
// wh [2][32768] (white histograms by exposure, overwritten samples taken out)
constexpr size_t DI_DERIVED_WORDS = 4 * 16384 + 4 * 16385 + 3 + 2 * 32768;

#ifdef DI_DIAG
__device__ unsigned long long g_dp_stamps[16];
#define DP_STAMP(k) do { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x == 0) g_dp_stamps[k] = wall_clock64(); } while (0)
#else
#define DP_STAMP(k) do {} while (0)
#endif
// analyse()'s host half, the pattern, the bright / dark fields and the white levels of dualiso.cpp (is_rggb_from_hist,
// bright_dark_from_hist, whites_from_hist) for frame blockIdx.x; writes dd[f] and the geometry part of pp[f]
__global__ __launch_bounds__(DNT) void k_di_decide_pattern(const uint16_t *__restrict__ frames, size_t img_stride, int w, int H, int black14,
                                                           DiDecideBuffers D)
{
    __shared__ unsigned long long sh[DNT / 64];
    __shared__ long long shl[DNT / 64];
    __shared__ int s_rggb, s_bright[4], s_rows[512][2], s_nrows, s_raw[4], s_off[4];
    __shared__ KthScratch ks;
    const int f = blockIdx.x, tid = threadIdx.x;
    const uint16_t *frame = (const uint16_t *)((const uint8_t *)frames + (size_t)f * img_stride);
    const unsigned *dev = D.hist + (size_t)f * D.hist_stride;
    unsigned *hb = D.derived + (size_t)f * D.derived_stride, *g = hb + 4 * 16384, *wh = g + 4 * 16385 + 3;
    DiDecide &dd = D.dd[f];
    if (tid == 0) {
        dd.check_sum = D.check[(size_t)f * D.check_stride];
        dd.check_n = D.check[(size_t)f * D.check_stride + 1];
        dd.n = 0; dd.bmed = dd.b_lo = dd.b_hi = dd.dmed = 0; dd.hi_n = 0; dd.best = -1; dd.best_score = 0;
        dd.check_ok = D.check_passed || (dd.check_n > 0 && dd.check_sum / dd.check_n > 0.5);      // hdr.c:432-438
    }
    DP_STAMP(0);
    // ---- the four Bayer-phase histograms over rows [0, H / 4 * 4) (hdr.c:453): all rows' classes minus the rows below that range
#pragma unroll 16
    for (int i = tid; i < 4 * 16384; i += DNT) {
        const int k = i >> 14, v = i & 16383, qb = k >> 1, px = k & 1;
        hb[i] = dev[DI_D_CLASS + (size_t)(qb * 2 + px) * 16384 + v] + dev[DI_D_CLASS + (size_t)((qb + 2) * 2 + px) * 16384 + v];
    }
    __syncthreads();
    const int R0 = H / 4 * 4, R1 = (H - 1) / 4 * 4;
    for (int y = R0; y < H; y++)
        for (int x = tid; x < w; x += DNT) atomicSub(&hb[(size_t)((y & 1) * 2 + (x & 1)) * 16384 + (frame[(size_t)y * w + x] & 16383)], 1u);
    __syncthreads();
    DP_STAMP(1);
    {   // is_rggb_from_hist: sum over v of |acc1 - acc2| against |acc0 - acc3| (integers far below 2^53: the doubles of the host are exact)
        const int per = 16384 / DNT, i0 = tid * per;
        unsigned long long run[4];
        for (int k = 0; k < 4; k++) {
            unsigned long long s = 0;
#pragma unroll
            for (int i = 0; i < per; i++) s += hb[(size_t)k * 16384 + i0 + i];
            run[k] = block_excl_scan(s, sh, nullptr);
        }
        unsigned long long d_rggb = 0, d_gbrg = 0;
#pragma unroll
        for (int i = 0; i < per; i++) {
#pragma unroll
            for (int k = 0; k < 4; k++) run[k] += hb[(size_t)k * 16384 + i0 + i];
            d_rggb += run[1] > run[2] ? run[1] - run[2] : run[2] - run[1];
            d_gbrg += run[0] > run[3] ? run[0] - run[3] : run[3] - run[0];
        }
        d_rggb = block_sum_u64(d_rggb, sh);
        d_gbrg = block_sum_u64(d_gbrg, sh);
        if (tid == 0) s_rggb = d_rggb < d_gbrg;
        __syncthreads();
    }
    DP_STAMP(2);
    const int rggb = s_rggb, ay1 = rggb ? 0 : 1, h = rggb ? H : H - 1;
    // ---- greens by row phase: the frame as it is (RGGB: x & 1 != y & 1, rows [0, R0)) or one row lower (GBRG: rows 4 <= y - 1 < R1)
#pragma unroll 16
    for (int i = tid; i < 4 * 16384; i += DNT) {
        const int ph = i >> 14, v = i & 16383;
        const int q = rggb ? ph : ((ph + 1) & 3);                 // GBRG: class q feeds phase (q + 3) & 3
        const int px = rggb ? 1 - (q & 1) : (q & 1);
        g[(size_t)ph * 16385 + v] = dev[DI_D_CLASS + (size_t)(q * 2 + px) * 16384 + v];
    }
    __syncthreads();
    DP_STAMP(3);
    // rows outside the range (dualiso.cpp: take_out): RGGB [R0, H); GBRG y - 1 < 4 or y - 1 >= R1, i.e. [0, 5) and [R1 + 1, H)
    for (int pass = 0; pass < 2; pass++) {
        const int ya = rggb ? (pass ? H : R0) : (pass ? R1 + 1 : 0), yb = rggb ? H : (pass ? H : min(5, H));
        for (int y = ya; y < yb; y++) {
            const int ph = rggb ? (y & 3) : ((y - 1) & 3);
            for (int x = tid; x < w; x += DNT)
                if (rggb ? ((x & 1) != (y & 1)) : ((x & 1) == (y & 1))) atomicSub(&g[(size_t)ph * 16385 + (frame[(size_t)y * w + x] & 16383)], 1u);
        }
    }
    __syncthreads();
    DP_STAMP(4);
    // ---- bright_dark_from_hist in closed form.  The host walks ref upwards and moves every class to raw_i(ref) = the smallest r whose
    // count of values below r reaches ref; it records `off` while ref < ref_off and all classes are below the threshold, and stops at
    // the first ref that takes a class to 10000 or at ref_max.  The state only depends on ref, so the stops are evaluated directly.
    for (int k = 0; k < 4; k++) block_prefix_in_place<16384>(g + (size_t)k * 16385, sh);
    DP_STAMP(5);
    if (tid < 64) {
        // lanes 0..3: raw of row phase k = lane, lanes 4..7: off of phase k = lane - 4 (eight binary searches side by side; the other
        // lanes repeat them); the minima over the four phases by shuffles within the groups of four lanes
        const int k4 = tid & 3;
        const unsigned *Ck = g + (size_t)k4 * 16385;
        const int white = 10000;
        const long long total = g[16384];                  // hdr.c:553-555: the count of the FIRST row phase stands for all four
        const int ref_max = (int)(total * 0.998), ref_off = (int)(total * 0.05);
        long long ref_b = (long long)Ck[white - 1] + 1;      // first ref with raw_k >= white
        { long long u = __shfl_xor(ref_b, 1); ref_b = u < ref_b ? u : ref_b; u = __shfl_xor(ref_b, 2); ref_b = u < ref_b ? u : ref_b; }
        const long long ref_end = ref_b < ref_max ? ref_b : (long long)ref_max - 1;
        const int thr = black14 + (white - black14) / 4;
        long long r_thr = thr > 0 ? (long long)Ck[min(thr - 1, 16384)] : 0;                    // largest ref with every raw_k < thr
        { long long u = __shfl_xor(r_thr, 1); r_thr = u < r_thr ? u : r_thr; u = __shfl_xor(r_thr, 2); r_thr = u < r_thr ? u : r_thr; }
        const long long ref_o = min(min((long long)ref_off - 1, r_thr), ref_end);
        const bool is_off = (tid >> 2) & 1;
        int val = 0;
        if (ref_max > 0 && (!is_off || (thr > 0 && ref_o >= 0))) val = quantile_of(Ck, 16384, is_off ? ref_o : ref_end);
        if (tid < 4) s_raw[k4] = val;
        else if (tid < 8) s_off[k4] = val;
    }
    __syncthreads();
    if (tid == 0) {
        const int raw[4] = { s_raw[0], s_raw[1], s_raw[2], s_raw[3] }, off[4] = { s_off[0], s_off[1], s_off[2], s_off[3] };
        int d[4], sv[4];
        for (int k = 0; k < 4; k++) { d[k] = raw[k] - off[k]; sv[k] = d[k]; }
        for (int a = 0; a < 4; a++) for (int b = a + 1; b < 4; b++) if (sv[b] < sv[a]) { const int t = sv[a]; sv[a] = sv[b]; sv[b] = t; }
        const double med = (sv[1] + sv[2]) / 2;
        for (int k = 0; k < 4; k++) { s_bright[k] = d[k] > med; dd.is_bright[k] = s_bright[k]; dd.bd_raw[k] = d[k]; }
        dd.rggb = rggb;
    }
    __syncthreads();
    DP_STAMP(6);
    // ---- whites_from_hist: the every-3rd-pixel histograms by exposure; the samples the reference's list cap overwrote (class
    // indices max_pix - 1 .. total - 2, only ever in the last rows) are taken out; 11th / 51st largest
    const int isb[4] = { s_bright[0], s_bright[1], s_bright[2], s_bright[3] };
    const uint16_t *img = frame + (size_t)ay1 * w;
    const unsigned *hw = dev + (rggb ? DI_D_WHITE0 : DI_D_WHITE1);
#pragma unroll 8
    for (int v = tid; v < 32768; v += DNT) {                                     // all four phases loaded, then dealt to the two exposures
        unsigned s0 = 0, s1 = 0;
#pragma unroll
        for (int ph = 0; ph < 4; ph++) { const unsigned n = hw[(size_t)ph * 32768 + v]; if (isb[ph]) s1 += n; else s0 += n; }
        wh[v] = s0;
        wh[32768 + v] = s1;
    }
    DP_STAMP(7);
    const int spr = (w + 2) / 3, tail_rows = min(h, 32);
    const long long max_pix = (long long)w * h / 2 / 9;
    if (tid == 0) {
        // sample rows y = ay1 + 3 j, j < J; the class of a row depends on j % 4, so what the reference's running indices hold at row j
        // is a count of earlier rows per residue; only the last rows can reach the cap
        const int J = h > ay1 ? (h - ay1 + 2) / 3 : 0;
        auto before = [&](int j, int c) {                              // samples of class c in the rows before row j
            long long nrows = 0;
            for (int r = 0; r < 4; r++) if (isb[(ay1 + 3 * r) % 4] == c) nrows += (j - r + 3) / 4;
            return nrows * spr;
        };
        const long long total[2] = { before(J, 0), before(J, 1) };
        shl[0] = total[0]; shl[1] = total[1];
        int n = 0;
        const int y_first = h - tail_rows;
        for (int j = y_first > ay1 ? (y_first - ay1 + 2) / 3 : 0; j < J; j++) {
            const int y = ay1 + 3 * j, c = isb[y % 4];
            const long long idx = before(j, c);
            if (total[c] > max_pix && idx + spr > max_pix - 1 && n < 512) { s_rows[n][0] = y; s_rows[n][1] = (int)idx; n++; }
        }
        s_nrows = n;
    }
    __syncthreads();
    DP_STAMP(8);
    unsigned long long removed[2] = { 0, 0 };
    for (int r = 0; r < s_nrows; r++) {
        const int y = s_rows[r][0], c = isb[y % 4];
        const long long i0 = s_rows[r][1], tot = shl[c];
        for (int sx = tid; sx < spr; sx += DNT) {
            const long long k = i0 + sx;
            if (k >= max_pix - 1 && k <= tot - 2) {
                const int v0 = img[(size_t)y * w + 3 * sx];
                atomicSub(&wh[(size_t)c * 32768 + (v0 < 32767 ? v0 : 32767)], 1u);
                removed[c]++;
            }
        }
    }
    removed[0] = block_sum_u64(removed[0], sh);
    removed[1] = block_sum_u64(removed[1], sh);
    __syncthreads();
    DP_STAMP(9);
    int wlev[2];
    for (int c = 0; c < 2; c++) {
        const long long kept = shl[c] - (long long)removed[c];
        long long k = c == 0 ? 10 : 50;
        int val = 0;
        if (kept > 0) {
            if (k > kept - 1) k = kept - 1;
            // (k + 1)-th largest = first index from the top whose running count exceeds k = from the bottom: exceeds kept - 1 - k
            const long long kk = kept - 1 - k;
            const KthScan sc = kth_scan<32768>(wh + (size_t)c * 32768, sh);
            kth_resolve<32768>(wh + (size_t)c * 32768, sc, &kk, 1, &val, ks);
        }
        wlev[c] = val;
    }
    DP_STAMP(10);
    if (tid == 0) {
        const int w0 = wlev[0] - 100, w1 = wlev[1] - 1500;
        dd.white_dark = w0 < 10000 ? 10000 : (w0 > 16383 ? 16383 : w0);
        dd.white_bright = w1 < 5000 ? 5000 : (w1 > 16383 ? 16383 : w1);
        // the geometry and levels the sampling kernels need (dualiso.cpp fills the rest after the round trip)
        DiParams &p = D.pp[f];
        const int nb = isb[0] + isb[1] + isb[2] + isb[3];
        const bool ok = nb == 2 && isb[0] != isb[2] && isb[1] != isb[3] && dd.check_ok;
        p.w = w; p.h = ok ? h : 0; p.ay1 = ay1;
        p.is_bright_bits = isb[0] | (isb[1] << 1) | (isb[2] << 2) | (isb[3] << 3);
        p.black20 = black14 * 64; p.white20 = dd.white_dark * 64;
        p.match_white20 = min(dd.white_dark, dd.white_bright) * 64;
    }
}

// medians and percentiles of match_exposures from its two histograms (hdr.c:700-733; dualiso.cpp: kth_from_hist)
__global__ __launch_bounds__(DNT) void k_di_decide_quantiles(DiDecideBuffers D)
{
    __shared__ unsigned long long sh[DNT / 64];
    __shared__ KthScratch ks;
    const int f = blockIdx.x;
    const unsigned *hb = D.hist_bd + (size_t)f * 2 * DI_HIST_N, *hd = hb + DI_HIST_N;
    DiDecide &dd = D.dd[f];
    if (D.pp[f].h <= 0) return;
    const KthScan sb = kth_scan<DI_HIST_N>(hb, sh);             // one sweep over each histogram: n, then three k of hist_b and one of hist_d
    const long long n = (long long)sb.tot;
    const long long mk = (n & 1) ? n / 2 : n / 2 - 1;
    int bmed = 0, b_lo = 0, b_hi = 0, dmed = 0;
    if (n > 0) {
        const long long kb[3] = { mk, n * 98 / 100, (long long)(n * 99.9 / 100) };
        int rb[3];
        kth_resolve<DI_HIST_N>(hb, sb, kb, 3, rb, ks);
        const KthScan sd = kth_scan<DI_HIST_N>(hd, sh);
        int rd;
        kth_resolve<DI_HIST_N>(hd, sd, &mk, 1, &rd, ks);
        bmed = rb[0] - DI_HIST_OFF; b_lo = rb[1] - DI_HIST_OFF; b_hi = rb[2] - DI_HIST_OFF; dmed = rd - DI_HIST_OFF;
    }
    if (threadIdx.x == 0) { dd.n = n; dd.bmed = bmed; dd.b_lo = b_lo; dd.b_hi = b_hi; dd.dmed = dmed; }
}

// what every sample row contributes to the list of highlight pairs and where (hdr.c:735-746: the cap leaves only the inner loop).
// Sequentially: take = min(count, max(hi_nmax - hi_n, 1)); hi_n += take.  In closed form over the exclusive prefix P of the counts:
// a row takes all of its samples while P + count <= hi_nmax, the row that crosses takes what is left, and every row after the list
// is full still takes one sample if it has any.
__global__ __launch_bounds__(256) void k_di_decide_rows(DiDecideBuffers D)
{
    __shared__ int s_tot[4];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const DiParams p = D.pp[f];
    if (p.h <= 0) return;
    int *counts = D.rows + (size_t)f * 3 * D.nsy_max, *take = counts + D.nsy_max, *offset = take + D.nsy_max;
    const int nsy = di_nsy(p), nmax = (p.w + 2) * (p.h + 2) / 9, hi_nmax = nmax / 50;
    const bool any = D.dd[f].n > 0;
    // two scans in one sweep: P = samples before the row; Q = rows before it that have samples AND start at or behind the cap
    int base_p = 0, base_q = 0;
    for (int r0 = 0; r0 < nsy; r0 += 256) {
        const int sy = r0 + tid;
        const int c = (any && sy < nsy) ? counts[sy] : 0;
        int incp = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incp, o); if (lane >= o) incp += u; }
        if (lane == 63) s_tot[wv] = incp;
        __syncthreads();
        int wp = 0, tot = 0;
        for (int k = 0; k < 4; k++) { if (k < wv) wp += s_tot[k]; tot += s_tot[k]; }
        const int P = base_p + wp + incp - c;                          // exclusive prefix of this row
        __syncthreads();
        // rows at or behind the cap: P >= hi_nmax (hi_n has reached the cap exactly when the prefix has)
        const int late = (P >= hi_nmax && c > 0) ? 1 : 0;
        int incq = late;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incq, o); if (lane >= o) incq += u; }
        if (lane == 63) s_tot[wv] = incq;
        __syncthreads();
        int wq = 0, totq = 0;
        for (int k = 0; k < 4; k++) { if (k < wv) wq += s_tot[k]; totq += s_tot[k]; }
        const int Q = base_q + wq + incq - late;
        __syncthreads();
        if (sy < nsy) {
            int t, o;
            if (P + c <= hi_nmax) { t = c; o = P; }                      // fits entirely
            else if (P < hi_nmax) { t = hi_nmax - P; o = P; }              // the row that fills the list
            else { t = min(c, 1); o = hi_nmax + Q; }                       // the list is full: one sample per row that has any
            take[sy] = t;
            offset[sy] = o;
        }
        base_p += tot;
        base_q += totq;
    }
    if (tid == 0) D.dd[f].hi_n = any ? (base_p <= hi_nmax ? base_p : hi_nmax + base_q) : 0;
}

// the winning slope: the FIRST candidate with the highest score, if that score is above zero (hdr.c:764-771)
__global__ __launch_bounds__(256) void k_di_decide_fit(DiDecideBuffers D)
{
    __shared__ long long shl[4];
    const int f = blockIdx.x;
    if (D.pp[f].h <= 0) return;
    const int *score = D.score + (size_t)f * D.score_stride;
    long long best = 0;                                   // (score << 32) | (0x7FFFFFFF - index): maximum = highest score, lowest index
    if (D.dd[f].hi_n > 0)
        for (int k = threadIdx.x; k < D.ncand; k += 256) {
            const long long key = ((long long)score[k] << 32) | (long long)(0x7FFFFFFF - k);
            best = key > best ? key : best;
        }
    const long long top = -block_min_i64(-best, shl);
    if (threadIdx.x == 0) {
        const int sc = (int)(top >> 32);
        D.dd[f].best_score = sc;
        D.dd[f].best = sc > 0 ? 0x7FFFFFFF - (int)(top & 0x7FFFFFFF) : -1;
    }
}

// ------------------------------------------------------------------ launchers
// grid-stride kernels: x over the pixels of one frame, y = frame of the batch
static inline dim3 flat_grid(size_t n, int nframes = 1)
{
    size_t b = (n + 255) / 256;
    const size_t cap = nframes > 4 ? 2048 : 8192;       // a batch fills the chip with its frames
    if (b > cap) b = cap;
    // (di_xcd_block: measured and left OFF -- profiles/r05/di_xcd/: 6 % less traffic per conversion, k_di_blend's fetches -31 %, but
    // k_di_interp 216 -> 270 us, k_di_amaze_ev 93 -> 113 us per frame: a band of rows per XCD concentrates each pass on few memory
    // channels, and the traffic of these kernels is gathers in the 4 MB tables, not re-read rows.  MLVFS_AMD_DI_XCD=1 switches it on.)
    static const bool xcd = [] { const char *e = getenv("MLVFS_AMD_DI_XCD"); return e && e[0] == '1'; }();
    if (xcd) b = (b + 7) / 8 * 8;                       // (workgroups beyond the last piece find nothing to do)
    else if ((b & 7) == 0) b += 1;                      // (an odd grid keeps the plain order)
    return dim3((unsigned)b, (unsigned)nframes);
}

int di_launch_analyse(const void *d_img, int w, int H, int black, int white, const double *d_evf, unsigned *d_hist,
                      double *d_check, hipStream_t s, int nframes, size_t img_stride, size_t hist_stride, size_t check_stride)
{
    if (nframes > 1) {
        MLV_HIP(hipMemsetAsync(d_hist, 0, sizeof(unsigned) * hist_stride * nframes, s));
        MLV_HIP(hipMemsetAsync(d_check, 0, sizeof(double) * check_stride * nframes, s));
    } else {
        const ptrdiff_t gap = (const uint8_t *)d_check - (const uint8_t *)d_hist;
        if (gap >= (ptrdiff_t)(sizeof(unsigned) * DI_D_WORDS) && gap < (ptrdiff_t)(sizeof(unsigned) * DI_D_WORDS) + 4096)
            MLV_HIP(hipMemsetAsync(d_hist, 0, (size_t)gap + 2 * sizeof(double), s));                   // the check sums lie right behind: one call
        else {
            MLV_HIP(hipMemsetAsync(d_hist, 0, sizeof(unsigned) * DI_D_WORDS, s));
            MLV_HIP(hipMemsetAsync(d_check, 0, 2 * sizeof(double), s));
        }
    }
    static const int band_max = [] { const char *e = getenv("MLVFS_AMD_ANALYSE_BAND"); const int v = e ? atoi(e) : 128; return v >= 4 ? v / 4 * 4 : 128; }();
    int band = H * 4 * nframes / 256 / 4 * 4;                                   // at least a workgroup per CU
    band = band < 16 ? 16 : (band > band_max ? band_max : band);
    while (band > 4 && (band / 4) * (w / 2 + 1) >= 65536) band -= 4;            // the 16-bit counters of a class
    hipLaunchKernelGGL(k_di_analyse, dim3((H + band - 1) / band, 4, nframes), dim3(256), 0, s, (const uint16_t *)d_img, w, H, black, white,
                       d_evf, d_hist, d_check, img_stride, hist_stride, check_stride, band);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

// hist_bd: per frame hist_b | hist_d back to back
int di_launch_subsample(const void *d_img, const DiBatch &b, int nsx, int nsy_max, size_t ns_stride, int *d_dark_s, int *d_bright_s,
                        unsigned *d_hist_bd, hipStream_t s)
{
    MLV_HIP(hipMemsetAsync(d_hist_bd, 0, 2 * sizeof(unsigned) * DI_HIST_N * b.nframes, s));
    const int n = nsx * nsy_max;
    if (n > 0) {
        // chunks per frame: enough workgroups for the chip from a small batch, few flushes from a large one; a chunk's samples fit a
        // 16-bit counter
        int chunks = std::min(64, std::max(16, 256 / (2 * std::max(b.nframes, 1))));
        chunks = std::max(chunks, (n + 65534) / 65535);
        const int per_chunk = (n + chunks - 1) / chunks;
        hipLaunchKernelGGL(k_di_subsample, dim3(chunks, b.nframes, 2), dim3(1024), 0, s, (const uint16_t *)d_img, b, nsx, ns_stride, per_chunk,
                           d_dark_s, d_bright_s, d_hist_bd);
    }
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

int di_launch_hi_count(const int *d_bs, int nsx, int nsy_max, size_t ns_stride, int b_lo, int b_hi, const DiDecide *dd, const DiBatch &b,
                       int *d_rows, hipStream_t s)
{
    if (nsy_max > 0)
        hipLaunchKernelGGL(k_di_hi_count, dim3(nsy_max, b.nframes), dim3(256), 0, s, d_bs, nsx, ns_stride, b_lo, b_hi, dd, b, d_rows, 3 * nsy_max);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

int di_launch_hi_compact(const int *d_ds, const int *d_bs, int nsx, int nsy_max, size_t ns_stride, int b_lo, int b_hi, const DiDecide *dd,
                         const DiBatch &b, const int *d_rows, int *d_hi, size_t hi_stride, hipStream_t s)
{
    if (nsy_max > 0)
        hipLaunchKernelGGL(k_di_hi_compact, dim3(nsy_max, b.nframes), dim3(256), 0, s, d_ds, d_bs, nsx, ns_stride, b_lo, b_hi, dd, b, d_rows,
                           nsy_max, d_hi, hi_stride);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

int di_launch_score(const int *d_hi, size_t hi_stride, int hi_n, const double *d_ta, int ncand, int dmed, int bmed, const DiDecide *dd,
                    int nframes, int *d_score, int score_stride, hipStream_t s)
{
    hipLaunchKernelGGL(k_di_score, dim3(ncand, nframes), dim3(256), 0, s, d_hi, hi_stride, hi_n, d_ta, dmed, bmed, dd, d_score, score_stride);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

static int di_chroma_smooth(const uint32_t *plane, uint32_t *plane_s, const DiBatch &b, int h_launch, const DiLuts &L, const DiPlanes &P,
                            hipStream_t s)
{
    const DiParams &p = b.p0;                            // width and options are the batch's; the rows are each frame's
    const int w = p.w, cw = w / 2, ch = h_launch / 2;
    MLV_HIP(hipMemcpyAsync(plane_s + (size_t)b.f0 * b.S, plane + (size_t)b.f0 * b.S, (b.nframes > 1 || b.f0 ? b.S * b.nframes : (size_t)w * h_launch) * 4,
                           hipMemcpyDeviceToDevice, s));
    if (cw <= 0 || ch <= 0) return MLVFS_AMD_OK;
    dim3 g((cw + 255) / 256, ch, b.nframes);
    hipLaunchKernelGGL(k_di_cs_cells, g, dim3(256), 0, s, plane, b, P.cells_stride, L.mix_raw2ev, P.cells);
    switch (p.chroma_smooth) {
    case 2: hipLaunchKernelGGL(k_di_cs_apply<2>, g, dim3(256), 0, s, P.cells, b, P.cells_stride, L.mix_ev2raw, plane_s); break;
    case 3: hipLaunchKernelGGL(k_di_cs_apply<3>, g, dim3(256), 0, s, P.cells, b, P.cells_stride, L.mix_ev2raw, plane_s); break;
    default: hipLaunchKernelGGL(k_di_cs_apply<5>, g, dim3(256), 0, s, P.cells, b, P.cells_stride, L.mix_ev2raw, plane_s); break;
    }
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

// 14 -> 20 bit with the exposures matched (hdr.c:781-803); d_img: frame 0 of the batch, NOT offset for GBRG (the kernels do that)
int di_launch_match(const void *d_img, const DiBatch &b, int h_launch, const DiLuts &L, const DiPlanes &P, hipStream_t s)
{
    hipLaunchKernelGGL(k_di_match, flat_grid((size_t)b.p0.w * h_launch, b.nframes), dim3(256), 0, s, (const uint16_t *)d_img, P.raw, b,
                       (const int *)nullptr, (size_t)0, 0, (float *)nullptr, L.interp_raw2ev, P.ev_red);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

// squeeze -> AMaZE -> clamp -> gray -> edge directions; the planes feed k_di_interp<true>.  P.sq_dst: per frame sq_dst | sq_row of
// h_launch ints each
// rows per workgroup of the edge search.  MLVFS_AMD_DI_EDGE_FUSED=1: direction search and interpolation in one kernel (k_di_edge_interp;
// identical results).  OFF: 4 % more conversions per second in batches of 8 (985 -> 1 026 on one box), but 582-636 MB of HBM-side
// traffic per frame where the two kernels need 446 (profiles/r05/di_experiments.log) -- the wrong direction for VERDICT r4 #3 until
// the table look-ups of the interpolation are localised.
static int di_edge_rows() { static const int v = [] { const char *e = getenv("MLVFS_AMD_EDGE_ROWS"); const int r = e ? atoi(e) : 4; return r > 0 ? r : 4; }(); return v; }
static bool di_edge_fused() { const char *e = getenv("MLVFS_AMD_DI_EDGE_FUSED"); return e && atoi(e) != 0; }
int di_launch_amaze_interp(const void *d_img, const DiBatch &b, int h_launch, const DiLuts &L, const DiPlanes &P, hipStream_t s, hipEvent_t after_amaze,
                           hipStream_t tail)
{
    const int w = b.p0.w, nf = b.nframes;
    const size_t n = (size_t)w * h_launch, sq_stride = 3 * (size_t)h_launch, fo = (size_t)b.f0 * b.S;
    MLV_HIP(hipMemsetAsync(P.stats + (size_t)b.f0 * 4 * DI_STAT_SLOTS, 0, 4 * sizeof(unsigned) * nf * DI_STAT_SLOTS, s));
    hipLaunchKernelGGL(k_di_match, flat_grid(n, nf), dim3(256), 0, s, (const uint16_t *)d_img, P.raw, b, P.sq_dst, sq_stride, h_launch, P.cfa,
                       (const int *)nullptr, (int *)nullptr);
    // AMaZE's output stage makes the look-ups of the planes itself (amaze_math.h: ev_of_planes): the three int planes and the gray
    // plane are what it writes, the float planes and k_di_amaze_ev's pass over them (134 MB and 93 us per 3584x1320 frame) are gone.
    // MLVFS_AMD_DI_EV_FUSED=0: the separate pass (A/B; identical results)
    static const bool ev_fused = [] { const char *e = getenv("MLVFS_AMD_DI_EV_FUSED"); return !e || atoi(e) != 0; }();
    const bool fused = ev_fused && w % 4 == 0;
    float *const o_red = fused ? (float *)P.ev_red : P.red, *const o_green = fused ? (float *)P.ev_green : P.green, *const o_blue = fused ? (float *)P.ev_blue : P.blue;
    const int *const r2e = fused ? L.interp_raw2ev : nullptr;
    int *const o_gray = fused ? P.gray_ev : nullptr;
    // a frame's AMaZE geometry follows its own row count (one less for GBRG): the launch plan is made per distinct height
    int rc = MLVFS_AMD_OK;
    if (b.pp) {
        static_assert(sizeof(DiParams) % sizeof(int) == 0, "h of frame f sits f * sizeof(DiParams) / 4 ints behind h of frame 0");
        for (int k = 0; k < b.nheights && !rc; k++)
            rc = amaze_launch(P.cfa + fo, w, b.heights[k], o_red + fo, o_green + fo, o_blue + fo, P.amaze_scratch + (size_t)b.f0 * P.amaze_scratch_stride, s,
                              nf, b.S, P.amaze_scratch_stride, &b.pp[b.f0].h, (int)(sizeof(DiParams) / sizeof(int)), nullptr, r2e, b.p0.black20,
                              o_gray ? o_gray + fo : nullptr);
    } else rc = amaze_launch(P.cfa, w, b.p0.h, o_red, o_green, o_blue, P.amaze_scratch, s, 1, 0, 0, nullptr, 0, nullptr, r2e, b.p0.black20, o_gray);
    if (rc) return rc;
    if (after_amaze) MLV_HIP(hipEventRecord(after_amaze, s));
    if (tail && tail != s) {                                           // what follows AMaZE goes on with the planes on another stream
        if (!after_amaze) return MLVFS_AMD_ERR_ARG;
        MLV_HIP(hipStreamWaitEvent(tail, after_amaze, 0));
        s = tail;
    }
    if (!fused)
        hipLaunchKernelGGL(k_di_amaze_ev, flat_grid(n, nf), dim3(256), 0, s, P.red, P.green, P.blue, b, L.interp_raw2ev, P.ev_red, P.ev_green, P.ev_blue,
                           P.gray_ev);
    if (!di_edge_fused())                                             // (fused: the search runs inside the interpolation's kernel, di_launch_convert)
        hipLaunchKernelGGL(k_di_edge_dir, dim3((w + 255) / 256, (h_launch + di_edge_rows() - 1) / di_edge_rows(), nf), dim3(256), 0, s, P.raw, P.gray_ev, b, P.sq_row, sq_stride,
                           L.fullres_thr, P.dir, P.stats);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

// amaze: the planes of di_launch_amaze_interp are ready and the edge-directed interpolation replaces mean23
int di_launch_convert(const DiBatch &b, int h_launch, const DiLuts &L, const DiPlanes &P, bool amaze, void *d_out, hipStream_t s)
{
    const DiParams &p = b.p0;
    const int nf = b.nframes;
    const size_t n = (size_t)p.w * h_launch;
    uint16_t *amap_fused = (p.use_alias_map && !p.chroma_smooth) ? P.amap : nullptr;
    const DiAmazeIn A{ P.ev_red, P.ev_green, P.ev_blue, P.dir, P.sq_row, 3 * (size_t)h_launch };
    const bool ev_planes = !p.chroma_smooth && L.blend_is_mix;                // halfres / fullres travel as EV (the blend's lookups, done early)
    if (amaze && di_edge_fused())
        hipLaunchKernelGGL(k_di_edge_interp, dim3((p.w + 255) / 256, (h_launch + di_edge_rows() - 1) / di_edge_rows(), nf), dim3(256), 0, s, P.raw, P.gray_ev, b, P.sq_row,
                           3 * (size_t)h_launch, L.fullres_thr, P.stats, L, A, P.dark, P.bright, P.fullres, P.halfres, P.over, amap_fused, ev_planes);
    else if (amaze)
        hipLaunchKernelGGL(k_di_interp<true>, flat_grid(n, nf), dim3(256), 0, s, P.raw, b, L, A, P.dark, P.bright, P.fullres, P.halfres,
                           P.over, amap_fused, ev_planes);
    else
        hipLaunchKernelGGL(k_di_interp<false>, flat_grid(n, nf), dim3(256), 0, s, P.raw, b, L, A, P.dark, P.bright, P.fullres, P.halfres,
                           P.over, amap_fused, ev_planes);
    MLV_HIP(hipGetLastError());
    const uint32_t *fullres_s = P.fullres, *halfres_s = P.halfres;
    if (p.chroma_smooth) {                                             // hdr.c:1612-1619
        int rc = di_chroma_smooth(P.halfres, P.halfres_s, b, h_launch, L, P, s);
        if (rc) return rc;
        halfres_s = P.halfres_s;
        if (p.use_fullres) {                                           // otherwise fullres_smooth aliases the all-zero fullres (hdr.c:1822)
            rc = di_chroma_smooth(P.fullres, P.fullres_s, b, h_launch, L, P, s);
            if (rc) return rc;
            fullres_s = P.fullres_s;
        }
        if (p.use_alias_map)
            hipLaunchKernelGGL(k_di_alias_err, flat_grid(n, nf), dim3(256), 0, s, P.bright, fullres_s, halfres_s, b, L, P.amap);
    }
    const uint16_t *amap_final = nullptr;
    if (p.use_alias_map) {
        dim3 g((p.w + 255) / 256, h_launch, nf);
        hipLaunchKernelGGL(k_di_alias_rank, g, dim3(256), 0, s, P.amap, P.bright, L.fullres_thr, b, P.aux);
        hipLaunchKernelGGL(k_di_alias_blur, g, dim3(256), 0, s, P.aux, P.amap, P.bright, L.fullres_thr, b, P.amap2);
        amap_final = P.amap2;
    }
    hipLaunchKernelGGL(k_di_blend, flat_grid(n, nf), dim3(256), 0, s, P.dark, P.bright, P.fullres, fullres_s, halfres_s, P.over, amap_final,
                       b, L, (uint16_t *)d_out, ev_planes);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

size_t di_derived_words() { return DI_DERIVED_WORDS; }

int di_launch_decide_pattern(const void *d_frames, const DiBatch &b, int H, int black14, const DiDecideBuffers &D, hipStream_t s)
{
    hipLaunchKernelGGL(k_di_decide_pattern, dim3(b.nframes), dim3(DNT), 0, s, (const uint16_t *)d_frames, b.img_stride, b.p0.w, H, black14, D);
#ifdef DI_DIAG
    {
        static int shown = 0;
        if (shown++ == 3) {
            unsigned long long st[16];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_dp_stamps), sizeof st);
            fprintf(stderr, "DI_DIAG k_di_decide_pattern, 10 ns ticks between stamps:");
            for (int k = 0; k < 10; k++) fprintf(stderr, " %lld", (long long)(st[k + 1] - st[k]));
            fprintf(stderr, "\n");
        }
    }
#endif
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}
int di_launch_decide_quantiles(const DiBatch &b, const DiDecideBuffers &D, hipStream_t s)
{
    hipLaunchKernelGGL(k_di_decide_quantiles, dim3(b.nframes), dim3(DNT), 0, s, D);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}
int di_launch_decide_rows(const DiBatch &b, const DiDecideBuffers &D, hipStream_t s)
{
    hipLaunchKernelGGL(k_di_decide_rows, dim3(b.nframes), dim3(256), 0, s, D);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}
int di_launch_decide_fit(const DiBatch &b, const DiDecideBuffers &D, hipStream_t s)
{
    hipLaunchKernelGGL(k_di_decide_fit, dim3(b.nframes), dim3(256), 0, s, D);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

}  // namespace mlv
