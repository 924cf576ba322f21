// k_pnoise.hip -- row/column pattern-noise correction (replaces fix_pattern_noise,
// mlvfs/patternnoise.c:49-380, debug_flags = 0).
//
// One direction = "column pass" (patternnoise.c:312-355); the row pass is the same
// pass on the transposed frame (patternnoise.c:371-378).  All values are int16 and
// wrap like the reference's int16 stores; every median is the LOWER median
// (wirth.h:129-131), obtained here by bisection on the value (the k-th smallest of n
// values is the smallest v with #{x <= v} >= k+1), which needs no sorting, no
// dynamic register arrays, and is exact for integers.
//
//   k_pn_transpose       frame <-> transposed frame
//   k_pn_smooth          per half-res pixel: run of horizontally adjacent pixels whose
//                        average green stays within thr of the centre (<= 25 left,
//                        <= 24 right), lower medians of g1, g2, r-avg_g, b-avg_g over
//                        the run (patternnoise.c:88-180); one workgroup per plane row,
//                        the row's five int16 arrays staged in LDS
//   k_pn_noise_t         the noise samples (orig - smoothed, or "masked": |flat-array gradient| > 500 or >= white) of
//                        every plane, written transposed so that a column's samples lie side by side
//   k_pn_column_offsets  per plane column: lower median of its unmasked samples, offset = -median when >= 10 samples
//                        (patternnoise.c:185-254)
//                        (the same kernel, "plain", gives the lower median of a plane's column offsets, patternnoise.c:268)
//   k_pn_apply           add offsets (clamp +-32767), remove their median (clamp 0..32760)
#include "clip.h"

namespace mlv {

__device__ __forceinline__ int plane_px(const int16_t *raw, int w, int c, int x, int y)
{
    return raw[(2 * x + (c & 1)) + (size_t)(2 * y + (c >> 1)) * w];
}

__global__ __launch_bounds__(256) void k_pn_transpose(const int16_t *__restrict__ in, int16_t *__restrict__ out, int w, int h)
{
    __shared__ int16_t tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int x = bx + threadIdx.x, y = by + r;
        if (x < w && y < h) tile[r][threadIdx.x] = in[x + (size_t)y * w];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int x = bx + r, y = by + threadIdx.x;          // out[y + x*h] = in[x + y*w]
        if (x < w && y < h) out[y + (size_t)x * h] = tile[threadIdx.x][r];
    }
}

// The lower median of a run of at most 50 values around a pixel, four times per half-res pixel, is what the correction costs.
// A thread takes the 54 values around its pixel (27 aligned dwords, [ws, ws + 54) with ws = x - 26 rounded down to even) from the
// staged row into registers, replaces what lies outside the run by a value no bisection step ever counts (32767: a step asks for
// #{v <= mid} with mid below the run's maximum), and bisects on the value between the run's extremes with packed 16-bit arithmetic:
// sat(mid - v) is negative exactly where v > mid, its sign bits are summed two per instruction.  (The first form bisected over LDS,
// a 16-bit read, a compare and an add per value and step inside lane-divergent loops: 1.62 ms per pass at 3584x1320.)
typedef short pn_s2 __attribute__((ext_vector_type(2)));
constexpr int PN_WD = 27;                                    // dwords of a pixel's window
constexpr int PN_PAD = 32;                                   // int16 elements either side of a staged row (never read as values: masked)

__device__ __forceinline__ pn_s2 pn_as_s2(unsigned u) { return __builtin_bit_cast(pn_s2, u); }

// arr: a staged row (element 0 at arr[PN_PAD]); m: per dword of the window, which halves belong to the run; need = rank of the lower median
__device__ __forceinline__ int pn_lower_median(const int16_t *arr, int ws, const unsigned (&m)[PN_WD], int need)
{
    const unsigned *p = (const unsigned *)(arr + PN_PAD + ws);          // 4-byte aligned: PN_PAD and ws are even
    unsigned v[PN_WD];
    pn_s2 mn = { 32767, 32767 }, mx = { -32768, -32768 };
#pragma unroll
    for (int d = 0; d < PN_WD; d++) {
        const unsigned u = p[d];
        v[d] = (u & m[d]) | (~m[d] & 0x7FFF7FFFu);
        mn = __builtin_elementwise_min(mn, pn_as_s2(v[d]));
        mx = __builtin_elementwise_max(mx, pn_as_s2((u & m[d]) | (~m[d] & 0x80008000u)));
    }
    int a = min((int)mn.x, (int)mn.y), b = max((int)mx.x, (int)mx.y);
    while (a < b) {
        const int mid = (a + b) >> 1;                        // floor
        const pn_s2 mp = { (short)mid, (short)mid };
        pn_s2 acc = { 0, 0 };
#pragma unroll
        for (int d = 0; d < PN_WD; d++) acc += __builtin_elementwise_sub_sat(mp, pn_as_s2(v[d])) >> 15;      // -1 where v > mid
        const int le = 2 * PN_WD + acc.x + acc.y;
        if (le >= need) b = mid; else a = mid + 1;
    }
    return a;
}

__global__ __launch_bounds__(256) void k_pn_smooth(const int16_t *__restrict__ raw, int w, int hw, int reach, int thr,
                                                   int16_t *__restrict__ smooth /* [4][hh][hw] */, int hh)
{
    extern __shared__ __attribute__((aligned(16))) int16_t row[];     // avg, g1, g2, drg, dbg: 5 x (hw + 2 PN_PAD), hw rounded up to even
    const int hs = ((hw + 1) & ~1) + 2 * PN_PAD;
    int16_t *avg = row, *g1 = row + hs, *g2 = row + 2 * hs, *drg = row + 3 * hs, *dbg = row + 4 * hs;
    const int y = blockIdx.x;
    for (int x = threadIdx.x; x < hw; x += blockDim.x) {
        const int r = plane_px(raw, w, 0, x, y), a = plane_px(raw, w, 1, x, y), b2 = plane_px(raw, w, 2, x, y),
                  b = plane_px(raw, w, 3, x, y);
        const int16_t av = (int16_t)((a + b2) / 2);            // patternnoise.c:59-65 (C division)
        avg[PN_PAD + x] = av; g1[PN_PAD + x] = (int16_t)a; g2[PN_PAD + x] = (int16_t)b2;
        drg[PN_PAD + x] = (int16_t)(r - av);
        dbg[PN_PAD + x] = (int16_t)(b - av);
    }
    __syncthreads();
    const size_t plane = (size_t)hw * hh;
    for (int x = threadIdx.x; x < hw; x += blockDim.x) {
        const int16_t *av = avg + PN_PAD;
        const int centre = av[x];
        const int hi_lim = min(x + reach, hw), lo_lim = max(x - reach, 0);
        int xr = x + 1, xl = x - 1;
        while (xr < hi_lim && abs(av[xr] - centre) <= thr) xr++;
        while (xl >= lo_lim && abs(av[xl] - centre) <= thr) xl--;
        const int lo = xl + 1, hi = xr - 1, n = hi - lo + 1, need = (n - 1) / 2 + 1;
        const int ws = (x - 26) & ~1;                          // lo >= x - 25 > ws, hi <= x + 24 < ws + 54
        const unsigned long long run = ((1ull << n) - 1) << (lo - ws);       // bit j: element ws + j belongs to the run
        unsigned m[PN_WD];
#pragma unroll
        for (int d = 0; d < PN_WD; d++)
            m[d] = (((run >> (2 * d)) & 1) ? 0x0000FFFFu : 0u) | (((run >> (2 * d + 1)) & 1) ? 0xFFFF0000u : 0u);
        const int mg1 = pn_lower_median(g1, ws, m, need), mg2 = pn_lower_median(g2, ws, m, need);
        const int mg = (mg1 + mg2) / 2;
        const size_t at = (size_t)y * hw + x;
        smooth[at] = (int16_t)(pn_lower_median(drg, ws, m, need) + mg);          // r
        smooth[plane + at] = (int16_t)mg1;                                        // g1
        smooth[2 * plane + at] = (int16_t)mg2;                                    // g2
        smooth[3 * plane + at] = (int16_t)(pn_lower_median(dbg, ws, m, need) + mg);   // b
    }
}

// noise sample of plane c at flat index i, or "masked"
__device__ __forceinline__ bool noise_at(const int16_t *raw, int w, int hw, size_t n, int c, const int16_t *smooth_c,
                                         size_t i, int white, int &noise)
{
    const int x = (int)(i % hw), y = (int)(i / hw);
    const int o = plane_px(raw, w, c, x, y);
    int grad = 0;
    if (i >= 2 && i + 2 < n) {
        const size_t im = i - 2, ip = i + 2;
        grad = (int16_t)(plane_px(raw, w, c, (int)(im % hw), (int)(im / hw)) - plane_px(raw, w, c, (int)(ip % hw), (int)(ip / hw)));
    }
    noise = (int16_t)(o - smooth_c[i]);
    return !((abs(grad) > 500) || (o >= white));
}

// The noise samples of a plane, transposed: T[c][x][y] = orig - smoothed, or MASKED (|flat-array gradient| > 500 or >= white).  Read
// row-wise (coalesced), written column-wise through a 32 x 32 tile in LDS, so that the next kernel finds a column's samples side by
// side.  (Until the end of round 3 a workgroup per column and plane walked the column in the frame itself: 83 us per pass.)
constexpr int PN_MASKED = 1 << 20;                           // above every int16
__global__ __launch_bounds__(256) void k_pn_noise_t(const int16_t *__restrict__ raw, int w, int hw, int hh, int white,
                                                    const int16_t *__restrict__ smooth, int *__restrict__ T /* [4][hw][hh] */)
{
    __shared__ int tile[32][33];
    const int c = blockIdx.z, bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const size_t n = (size_t)hw * hh;
    const int16_t *sm = smooth + (size_t)c * n;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int x = bx + threadIdx.x, y = by + r;
        if (x < hw && y < hh) {
            int nz;
            const bool ok = noise_at(raw, w, hw, n, c, sm, (size_t)x + (size_t)y * hw, white, nz);
            tile[r][threadIdx.x] = ok ? nz : PN_MASKED;
        }
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int x = bx + r, y = by + threadIdx.x;
        if (x < hw && y < hh) T[((size_t)c * hw + x) * hh + y] = tile[threadIdx.x][r];
    }
}

// one workgroup per column, one wave per plane: lower median of the column's unmasked samples by bisection between their extremes
// -- a step is a ballot per 64 rows --, offset = -median when >= 10 samples (patternnoise.c:185-254)
// R: samples per lane held in registers (hh <= 64 R); R = 0: any height, the column is re-read in every step
// plain: the lower median itself, whatever the number of samples (the median of a plane's column offsets, patternnoise.c:268:
// T = the offsets [4][1][hw], one "column" per plane)
template <int R>
__global__ __launch_bounds__(256) void k_pn_column_offsets(const int *__restrict__ T, int hw, int hh, int *__restrict__ offs /* [4][hw] */, bool plain)
{
    const int x = blockIdx.x, c = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int *col = T + ((size_t)c * hw + x) * hh;
    int v[R > 0 ? R : 1];
    int cnt = 0, lo = 0x7FFFFFFF, hi = -0x7FFFFFFF - 1;
    if (R > 0) {
#pragma unroll
        for (int i = 0; i < R; i++) {
            const int y = lane + 64 * i;
            v[i] = y < hh ? col[y] : PN_MASKED;
            if (v[i] != PN_MASKED) { cnt++; lo = min(lo, v[i]); hi = max(hi, v[i]); }
        }
    } else {
        for (int y = lane; y < hh; y += 64) {
            const int s = col[y];
            if (s != PN_MASKED) { cnt++; lo = min(lo, s); hi = max(hi, s); }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cnt += __shfl_xor(cnt, o); lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
    const int k = cnt;
    if (plain ? k < 1 : k < 10) { if (lane == 0) offs[c * hw + x] = 0; return; }      // patternnoise.c:250
    int a = lo, b = hi;
    const int need = (k - 1) / 2 + 1;
    while (a < b) {
        const int mid = (a + b) >> 1;
        int le = 0;
        if (R > 0) {
#pragma unroll
            for (int i = 0; i < R; i++) le += __popcll(__ballot(v[i] <= mid));
        } else {
            for (int y0 = 0; y0 < hh; y0 += 64) {
                const int y = y0 + lane;
                le += __popcll(__ballot(y < hh && col[y] <= mid));
            }
        }
        if (le >= need) b = mid; else a = mid + 1;
    }
    if (lane == 0) offs[c * hw + x] = plain ? a : -a;
}

__global__ __launch_bounds__(256) void k_pn_apply(int16_t *__restrict__ raw, int w, int h, int hw, const int *__restrict__ offs,
                                                  const int *__restrict__ mc)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= (w & ~1) || y >= (h & ~1)) return;
    const int c = (x & 1) + 2 * (y & 1);
    int v = (int)raw[x + (size_t)y * w] + offs[c * hw + (x >> 1)];
    v = v < -32767 ? -32767 : (v > 32767 ? 32767 : v);                       // patternnoise.c:260
    v = (int16_t)v - mc[c];
    raw[x + (size_t)y * w] = (int16_t)(v < 0 ? 0 : (v > 32760 ? 32760 : v));  // patternnoise.c:273
}

// The reference's debug views (patternnoise.c:215-240, debug_flags & FIXPN_DBG_DENOISED / _NOISE / _MASK; MLVFS itself passes 0):
// what one direction's pass sees instead of its correction -- the smoothed planes, the noise samples (+100, masked ones 0), or the
// mask (x 1000) -- written into the Bayer frame `view`; the gradient reads neighbours, so not in place.
__global__ __launch_bounds__(256) void k_pn_debug_view(const int16_t *__restrict__ raw, int w, int hw, int hh, int white,
                                                       const int16_t *__restrict__ smooth, int flags, int16_t *__restrict__ view)
{
    const size_t n = (size_t)hw * hh, i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (i >= n) return;
    const int16_t *sm = smooth + (size_t)c * n;
    int nz;
    const bool ok = noise_at(raw, w, hw, n, c, sm, i, white, nz);
    const int x = (int)(i % hw), y = (int)(i / hw);
    int16_t v;
    if (flags & 2) v = sm[i];
    else if (flags & 4) v = ok ? (int16_t)(nz + 100) : (int16_t)0;
    else v = ok ? 0 : 1000;
    view[(size_t)(2 * y + (c >> 1)) * w + 2 * x + (c & 1)] = v;
}

// one direction on a device frame (w x h int16).  scratch: smooth 4*hw*hh int16, offs 4*hw int, mc 4 int
static int column_pass(int16_t *d_raw, int w, int h, int white, int16_t *d_smooth, int *d_offs, int *d_mc, int *d_noise_t, hipStream_t stream, int flags)
{
    const int hw = w / 2, hh = h / 2;
    if (hw <= 0 || hh <= 0) return MLVFS_AMD_OK;
    const size_t shmem = (size_t)5 * (((hw + 1) & ~1) + 2 * PN_PAD) * sizeof(int16_t);
    if (shmem > 150 * 1024) { set_error("fix_pattern_noise: rows of %d pixels do not fit in LDS", w); return MLVFS_AMD_ERR_ARG; }
    MLV_HIP(hipFuncSetAttribute((const void *)k_pn_smooth, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(k_pn_smooth, dim3(hh), dim3(256), shmem, stream, d_raw, w, hw, 50 / 2, 500, d_smooth, hh);
    if (flags & (2 | 4 | 8)) {
        int16_t *d_view = (int16_t *)d_noise_t;                       // the samples' buffer is free in this mode
        hipLaunchKernelGGL(k_pn_debug_view, dim3((unsigned)(((size_t)hw * hh + 255) / 256), 4), dim3(256), 0, stream, d_raw, w, hw, hh, white, d_smooth, flags, d_view);
        MLV_HIP(hipGetLastError());
        MLV_HIP(hipMemcpyAsync(d_raw, d_view, (size_t)w * h * 2, hipMemcpyDeviceToDevice, stream));
        return MLVFS_AMD_OK;
    }
    hipLaunchKernelGGL(k_pn_noise_t, dim3((hw + 31) / 32, (hh + 31) / 32, 4), dim3(32, 8), 0, stream, d_raw, w, hw, hh, white, d_smooth, d_noise_t);
    auto offsets = hh <= 64 * 12 ? k_pn_column_offsets<12> : (hh <= 64 * 32 ? k_pn_column_offsets<32> : k_pn_column_offsets<0>);
    hipLaunchKernelGGL(offsets, dim3(hw), dim3(256), 0, stream, d_noise_t, hw, hh, d_offs, false);
    auto median = hw <= 64 * 12 ? k_pn_column_offsets<12> : (hw <= 64 * 32 ? k_pn_column_offsets<32> : k_pn_column_offsets<0>);
    hipLaunchKernelGGL(median, dim3(1), dim3(256), 0, stream, d_offs, 1, hw, d_mc, true);      // lower median of each plane's offsets
    hipLaunchKernelGGL(k_pn_apply, dim3((w + 255) / 256, h), dim3(256), 0, stream, d_raw, w, h, hw, d_offs, d_mc);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

size_t pattern_noise_scratch_bytes(int w, int h)
{
    const size_t n = (size_t)w * h;
    return n * 2 /* transposed frame */ + n * 2 /* 4 smoothed half-res planes */ + (size_t)4 * (w > h ? w : h) * 4 + 64 + 256 +
           n * 4 /* transposed noise samples */;
}

// flags: the reference's debug_flags (patternnoise.h:19-24); with any of them set only one direction runs (patternnoise.c:363-379)
int launch_pattern_noise(void *d_raw, int w, int h, int white, void *d_scratch, hipStream_t stream, int flags)
{
    int16_t *raw = (int16_t *)d_raw;
    const size_t n = (size_t)w * h;
    int16_t *d_t = (int16_t *)d_scratch;
    int16_t *d_smooth = d_t + n;
    int *d_offs = (int *)(d_smooth + n);
    int *d_mc = d_offs + (size_t)4 * (w > h ? w : h) / 2 + 4;
    int *d_noise_t = (int *)(((uintptr_t)((uint8_t *)d_scratch + n * 4 + (size_t)4 * (w > h ? w : h) * 4 + 64) + 255) & ~(uintptr_t)255);
    int rc = MLVFS_AMD_OK;
    if (!flags || !(flags & 1)) rc = column_pass(raw, w, h, white, d_smooth, d_offs, d_mc, d_noise_t, stream, flags);
    if (rc || (flags && !(flags & 1))) return rc;
    hipLaunchKernelGGL(k_pn_transpose, dim3((w + 31) / 32, (h + 31) / 32), dim3(32, 8), 0, stream, raw, d_t, w, h);
    rc = column_pass(d_t, h, w, white, d_smooth, d_offs, d_mc, d_noise_t, stream, flags);
    if (rc) return rc;
    hipLaunchKernelGGL(k_pn_transpose, dim3((h + 31) / 32, (w + 31) / 32), dim3(32, 8), 0, stream, d_t, raw, h, w);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

}  // namespace mlv
