// k_stripes.hip -- vertical-stripe correction: histogram pass for the per-clip
// coefficients and the stand-alone apply pass.
//
// Replaces stripes_compute_correction's raster loop (mlvfs/stripes.c:143-205,
// add_pixel :108-140) and stripes_apply_correction (:250-266).
//
// Bit-exactness of the histogram:
//  * every ACCEPTED add_pixel call consumes two rand() values in raster order.
//    Pass 1 counts the accepted calls per 8-pixel group, an exclusive scan turns
//    the counts into each group's offset into the rand() stream, pass 2 indexes
//    a pre-generated stream of rand()%1024 values (host: libc rand() itself or
//    the glibc-compatible generator in stripes_host.cpp).
//  * the bin is (int)(32768 + log2(af/bf) * 32768) in double.  The division and
//    the final add are IEEE-exact on the GPU; the device log2 may differ from
//    glibc's in the last ulp, so a sample whose bin position lies within 1e-6 of
//    a bin edge is not binned on the device: it is appended to a small "recheck"
//    list and the host bins it with its own libm (mlvfs_amd_stripes_* in clip.cpp).
#include "clip.h"
#include <algorithm>

namespace mlv {

// the 24 add_pixel calls of a group (stripes.c:175-203): histogram, reference px, corrected px
// pixel slots 0..9 = pa..ph, pa2, pb2
// (constexpr: the loops over it are unrolled, so the pixel slots are register names, not indices into a spilled array)
constexpr unsigned char k_call[24][3] = {
    {2,0,2},{2,0,2},{2,0,2},{2,8,2},  {3,1,3},{3,1,3},{3,1,3},{3,9,3},
    {4,0,4},{4,0,4},{4,8,4},{4,8,4},  {5,1,5},{5,1,5},{5,9,5},{5,9,5},
    {6,0,6},{6,8,6},{6,8,6},{6,8,6},  {7,1,7},{7,9,7},{7,9,7},{7,9,7},
};

struct Recheck { int hist, a, b, r1, r2; };

__device__ __forceinline__ bool accepted(int a, int b, double too_bright)
{
    const int lo = min(a, b), hi = max(a, b);
    return !(lo < 32) && !((double)hi > too_bright);            // stripes.c:113-117
}

__device__ __forceinline__ void load_group(const uint16_t *row, int x, int black, int (&px)[10])
{
#pragma unroll
    for (int k = 0; k < 10; k++) px[k] = (int)row[x + k] - black;
}

// pass 1: accepted calls per group; per-block totals
__global__ __launch_bounds__(256) void k_stripes_count(const uint16_t *__restrict__ img, int w, int row0, int groups_per_row,
                                                       int n_groups, int black, double too_bright,
                                                       unsigned char *__restrict__ counts, int *__restrict__ block_sum)
{
    __shared__ int wsum[4];
    const int g = blockIdx.x * 256 + threadIdx.x;
    int c = 0;
    if (g < n_groups) {
        const int y = row0 + g / groups_per_row, x = (g % groups_per_row) * 8;
        int px[10];
        load_group(img + (size_t)y * w, x, black, px);
#pragma unroll
        for (int i = 0; i < 24; i++) c += accepted(px[k_call[i][1]], px[k_call[i][2]], too_bright) ? 1 : 0;
        counts[g] = (unsigned char)c;
    }
    int s = c;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) block_sum[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of the block totals (single workgroup, n up to a few thousand)
__global__ __launch_bounds__(1024) void k_scan_blocks(const int *__restrict__ block_sum, int n,
                                                      long long *__restrict__ block_off, long long *__restrict__ total)
{
    __shared__ long long carry;
    __shared__ long long wtot[16];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const long long v = i < n ? block_sum[i] : 0;
        long long inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const long long up = __shfl_up(inc, o);
            if ((threadIdx.x & 63) >= o) inc += up;
        }
        if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = inc;
        __syncthreads();
        long long before = carry;
        for (int k = 0; k < (int)(threadIdx.x >> 6); k++) before += wtot[k];
        if (i < n) block_off[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

// The bins that matter sit within a few hundred of 32768 (ratios of neighbouring pixels): millions of atomic adds on a few
// hundred addresses.  The workgroups therefore add into HIST_COPIES private copies (block index modulo), which
// k_stripes_hist_fold then adds into the caller's histogram.
constexpr int HIST_COPIES = 16;

__global__ __launch_bounds__(256) void k_stripes_hist_fold(const int *__restrict__ copies, int *__restrict__ hist)
{
    const int i = blockIdx.x * 256 + threadIdx.x;              // 8 * 65536 bins
    int s = 0;
#pragma unroll
    for (int c = 0; c < HIST_COPIES; c++) s += copies[(size_t)c * (8 * 65536) + i];
    if (s) hist[i] += s;
}

__global__ void k_hist_bump(int *__restrict__ hist, const int *__restrict__ idx, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&hist[idx[i]], 1);
}

int launch_hist_bump(int *d_hist, const int *d_idx, int n, hipStream_t stream)
{
    if (n <= 0) return MLVFS_AMD_OK;
    hipLaunchKernelGGL(k_hist_bump, dim3((n + 255) / 256), dim3(256), 0, stream, d_hist, d_idx, n);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

size_t stripes_hist_copies_bytes() { return (size_t)HIST_COPIES * 8 * 65536 * sizeof(int); }

// pass 2: histogram.  The bins that matter sit within a few thousand of 32768 (ratios of neighbouring pixels): 4.7 M atomic adds of
// a 3584x1320 frame on a few hundred addresses took 650 us even spread over 16 copies in memory.  Each workgroup -- 1024 threads,
// one per CU: four blocks of 256 groups per round, the blocks of pass 1 -- now counts the window [32768 - LWIN/2, 32768 + LWIN/2) of
// all eight histograms in 16-bit counters in LDS (a workgroup sees at most n_groups * 24 / gridDim.x < 65 536 calls: launcher) and
// adds what it touched to its copy at the end; bins outside the window go to memory directly as before.
constexpr int LWIN = 8192, LWIN_LO = 32768 - LWIN / 2;
__global__ __launch_bounds__(1024) void k_stripes_hist(const uint16_t *__restrict__ img, int w, int row0, int groups_per_row,
                                                       int n_groups, int nblk, int black, double too_bright,
                                                       const unsigned char *__restrict__ counts,
                                                       const long long *__restrict__ block_off,
                                                       const uint16_t *__restrict__ rnd, long long n_rand,
                                                       int *__restrict__ hist, int *__restrict__ num,
                                                       Recheck *__restrict__ recheck, int recheck_cap, int *__restrict__ n_recheck)
{
    __shared__ unsigned cnt[8 * LWIN / 2];                        // two 16-bit counters per word
    __shared__ int wtot[4][4];
    __shared__ int lnum[8];
    for (int i = threadIdx.x; i < 8 * LWIN / 2; i += 1024) cnt[i] = 0;
    if (threadIdx.x < 8) lnum[threadIdx.x] = 0;
    __syncthreads();
    int *const mine = hist + (size_t)(blockIdx.x % HIST_COPIES) * (8 * 65536);
    const int sub = threadIdx.x >> 8, lt = threadIdx.x & 255;
    for (int b0 = blockIdx.x * 4; b0 < nblk; b0 += gridDim.x * 4) {
        const int blk = b0 + sub;
        const int g = blk * 256 + lt;
        const int c = (blk < nblk && g < n_groups) ? counts[g] : 0;
        int inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(inc, o);
            if ((lt & 63) >= o) inc += up;
        }
        if ((lt & 63) == 63) wtot[sub][lt >> 6] = inc;
        __syncthreads();
        long long call = (blk < nblk ? block_off[blk] : 0) + (inc - c);      // accepted calls before this group
        for (int k = 0; k < (int)(lt >> 6); k++) call += wtot[sub][k];
        if (c > 0) {
            const int y = row0 + g / groups_per_row, x = (g % groups_per_row) * 8;
            int px[10];
            load_group(img + (size_t)y * w, x, black, px);
#pragma unroll
            for (int i = 0; i < 24; i++) {
                const int j = k_call[i][0], a = px[k_call[i][1]], b = px[k_call[i][2]];
                if (!accepted(a, b, too_bright)) continue;
                const long long ri = 2 * call;
                call++;
                if (ri + 1 >= n_rand) continue;                         // host sized the stream; never taken
                const int r1 = rnd[ri], r2 = rnd[ri + 1];
                const double af = a + r1 / 1024.0 - 0.5;               // stripes.c:129-130
                const double bf = b + r2 / 1024.0 - 0.5;
                const double ev = log2(af / bf);
                const double pos = 65536 / 2 + ev * 65536 / 2;          // F2H, stripes.c:105
                const double nearest = rint(pos);
                if (fabs(pos - nearest) < 1e-6) {
                    const int slot = atomicAdd(n_recheck, 1);
                    if (slot < recheck_cap) recheck[slot] = Recheck{ j, a, b, r1, r2 };
                } else {
                    int bin = (int)pos;
                    bin = bin < 0 ? 0 : (bin > 65535 ? 65535 : bin);
                    const unsigned rel = (unsigned)(bin - LWIN_LO);
                    if (rel < (unsigned)LWIN) atomicAdd(&cnt[j * (LWIN / 2) + (rel >> 1)], 1u << (16 * (rel & 1)));
                    else atomicAdd(&mine[j * 65536 + bin], 1);
                }
                atomicAdd(&lnum[j], 1);
            }
        }
        __syncthreads();                                              // wtot is rewritten in the next round
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 8 * LWIN / 2; i += 1024) {
        const unsigned v = cnt[i];
        if (!v) continue;
        const int j = i / (LWIN / 2), rel = 2 * (i % (LWIN / 2));
        if (v & 0xFFFFu) atomicAdd(&mine[j * 65536 + LWIN_LO + rel], (int)(v & 0xFFFFu));
        if (v >> 16) atomicAdd(&mine[j * 65536 + LWIN_LO + rel + 1], (int)(v >> 16));
    }
    if (threadIdx.x < 8 && lnum[threadIdx.x]) atomicAdd(&num[threadIdx.x], lnum[threadIdx.x]);
}

// stand-alone apply (the fused kernel has its own epilogue): 8 px per lane
__global__ __launch_bounds__(256) void k_stripes_apply(uint8_t *__restrict__ frames, size_t stride, size_t n_vec,
                                                       int black16, int white16, int c0, int c1, int c2, int c3, int c4,
                                                       int c5, int c6, int c7)
{
    const int coef[8] = { c0, c1, c2, c3, c4, c5, c6, c7 };
    uint4 *img = (uint4 *)(frames + (size_t)blockIdx.y * stride);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = img[i];
        uint32_t d[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t lo = d[k] & 0xFFFFu, hi = d[k] >> 16;
            {
                const int cf = coef[2 * k];
                if (cf != 0 && (int)lo > black16 + 64) {
                    const long long nm = (long long)((int)lo - black16) * cf + ((long long)black16 << 16);
                    lo = (((long long)white16 << 16) < nm) ? (uint32_t)white16 : ((uint32_t)(int)(nm / 65536) & 0xFFFFu);
                }
            }
            {
                const int cf = coef[2 * k + 1];
                if (cf != 0 && (int)hi > black16 + 64) {
                    const long long nm = (long long)((int)hi - black16) * cf + ((long long)black16 << 16);
                    hi = (((long long)white16 << 16) < nm) ? (uint32_t)white16 : ((uint32_t)(int)(nm / 65536) & 0xFFFFu);
                }
            }
            d[k] = lo | (hi << 16);
        }
        img[i] = make_uint4(d[0], d[1], d[2], d[3]);
    }
}

// ------------------------------------------------------------------ launchers
int stripes_groups_per_row(int w) { return w > 10 ? (w - 10 + 7) / 8 : 0; }

int launch_stripes_count(const void *d_frame, int w, int row0, int row1, int black, int white, unsigned char *d_counts,
                         int *d_block_sum, long long *d_block_off, long long *d_total, hipStream_t stream)
{
    const int gpr = stripes_groups_per_row(w), n_groups = gpr * (row1 - row0);
    if (n_groups <= 0) { MLV_HIP(hipMemsetAsync(d_total, 0, sizeof(long long), stream)); return MLVFS_AMD_OK; }
    const int nblk = (n_groups + 255) / 256;
    hipLaunchKernelGGL(k_stripes_count, dim3(nblk), dim3(256), 0, stream, (const uint16_t *)d_frame, w, row0, gpr, n_groups,
                       black, white / 1.5, d_counts, d_block_sum);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, stream, d_block_sum, nblk, d_block_off, d_total);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

int launch_stripes_hist(const void *d_frame, int w, int row0, int row1, int black, int white, const unsigned char *d_counts,
                        const long long *d_block_off, const void *d_rand, long long n_rand, int *d_hist, int *d_num,
                        void *d_recheck, int recheck_cap, int *d_n_recheck, void *d_copies, hipStream_t stream)
{
    const int gpr = stripes_groups_per_row(w), n_groups = gpr * (row1 - row0);
    if (n_groups <= 0) return MLVFS_AMD_OK;
    const int nblk = (n_groups + 255) / 256;
    MLV_HIP(hipMemsetAsync(d_copies, 0, stripes_hist_copies_bytes(), stream));
    // one workgroup per CU (its counters fill the LDS), but never so few that one could see 65 536 calls (16-bit counters)
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount; }
    const int quads = (nblk + 3) / 4;                             // (a histogram gets at most 4 calls per group: k_call)
    int wgs = std::min(quads, cus);
    while ((long long)((quads + wgs - 1) / wgs) * 4 * 256 * 4 >= 65536 && wgs < quads) wgs = std::min(quads, wgs * 2);
    hipLaunchKernelGGL(k_stripes_hist, dim3(wgs), dim3(1024), 0, stream, (const uint16_t *)d_frame, w, row0, gpr, n_groups, nblk,
                       black, white / 1.5, d_counts, d_block_off, (const uint16_t *)d_rand, n_rand, (int *)d_copies, d_num,
                       (Recheck *)d_recheck, recheck_cap, d_n_recheck);
    hipLaunchKernelGGL(k_stripes_hist_fold, dim3(8 * 65536 / 256), dim3(256), 0, stream, (const int *)d_copies, d_hist);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

int launch_stripes_apply(void *d_frames, size_t stride, size_t npix, int w, int black, int white, const int32_t *coef,
                         int nframes, hipStream_t stream)
{
    if (w % 8 != 0) return MLVFS_AMD_OK;                 // stripes.c:253
    if (npix % 8 != 0 || ((uintptr_t)d_frames % 16) != 0 || (nframes > 1 && stride % 16 != 0)) {
        set_error("stripes apply needs 16-byte aligned frames of a multiple of 8 pixels");
        return MLVFS_AMD_ERR_ARG;
    }
    const size_t n_vec = npix / 8;
    dim3 grid((unsigned)((n_vec + 255) / 256), nframes);
    if (grid.x > 4096) grid.x = 4096;
    hipLaunchKernelGGL(k_stripes_apply, grid, dim3(256), 0, stream, (uint8_t *)d_frames, stride, n_vec,
                       (int)(uint16_t)black, (int)(uint16_t)white, coef[0], coef[1], coef[2], coef[3], coef[4], coef[5],
                       coef[6], coef[7]);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

}  // namespace mlv

// ---------------------------------------------------------------- glibc rand() % 1024 stream on the device
// x[i] = x[i-31] + x[i-3] (mod 2^32), output (x >> 1) % 1024 (see clip.cpp).  The recurrence is linear, so the state at the
// start of chunk c is (A^RAND_CHUNK)^c times the state at the start of the stream: k_rand_states applies the host-built
// powers (A^RAND_CHUNK)^(2^j) for the set bits of c, k_rand_fill then lets one thread per chunk run its 992 = 31 * 32 values
// with the 31-word ring in registers (31 unrolled steps per round), eight values per 16-byte store.
namespace mlv {

// One workgroup per RAND_GROUP consecutive chunks: the state at the group's first chunk by the set bits of its index (up to npow
// products with the host-built powers), the states of the chunks behind it by one product each with the first power (A^RAND_CHUNK) --
// a workgroup per chunk repeated the long walk for every chunk (9 500 chunks of a 3584x1320 frame's dither: 150 us; now 10).
constexpr int RAND_GROUP = 32;
__global__ __launch_bounds__(32) void k_rand_states(const uint32_t *__restrict__ start, const uint32_t *__restrict__ pow2 /* [j][31][31] */,
                                                    int npow, uint32_t nchunks, uint32_t *__restrict__ states)
{
    const uint32_t c0 = blockIdx.x * RAND_GROUP;
    if (c0 >= nchunks) return;
    __shared__ uint32_t v[2][32];
    const int i = threadIdx.x;
    if (i < 31) v[0][i] = start[i];
    __syncthreads();
    int cur = 0;
    auto times = [&](int j) {                                    // v <- pow2[j] * v
        if (i < 31) {
            const uint32_t *row = pow2 + ((size_t)j * 31 + i) * 31;
            uint32_t acc = 0;
            for (int k = 0; k < 31; k++) acc += row[k] * v[cur][k];
            v[cur ^ 1][i] = acc;
        }
        __syncthreads();
        cur ^= 1;
    };
    for (int j = 0; j < npow; j++)
        if ((c0 >> j) & 1u) times(j);                            // uniform
    for (uint32_t c = c0; c < min(c0 + RAND_GROUP, nchunks); c++) {
        if (i < 31) states[(size_t)c * 31 + i] = v[cur][i];
        if (c + 1 < min(c0 + RAND_GROUP, nchunks)) times(0);
    }
}

__global__ __launch_bounds__(256) void k_rand_fill(const uint32_t *__restrict__ states, uint32_t nchunks, uint16_t *__restrict__ out, size_t n)
{
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= nchunks) return;
    uint32_t x[31];
#pragma unroll
    for (int k = 0; k < 31; k++) x[k] = states[(size_t)c * 31 + k];
    const size_t base = (size_t)c * RAND_CHUNK;
    // the last eight values travel in a 128-bit shift register (31 steps per round do not line up with groups of eight)
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    int held = 0, done = 0;
    for (int round = 0; round < RAND_CHUNK / 31; round++) {
#pragma unroll
        for (int k = 0; k < 31; k++) {
            const uint32_t v = x[k] + x[(k + 28) % 31];     // oldest + the value three steps back
            x[k] = v;
            const uint32_t o = (v >> 1) & 1023u;
            w0 = (w0 >> 16) | (w1 << 16); w1 = (w1 >> 16) | (w2 << 16); w2 = (w2 >> 16) | (w3 << 16); w3 = (w3 >> 16) | (o << 16);
            if (++held == 8) {                               // the same step in every thread
                const size_t at = base + (size_t)done;
                if (at + 8 <= n) *(uint4 *)(out + at) = make_uint4(w0, w1, w2, w3);
                else {
                    const uint32_t ws[4] = { w0, w1, w2, w3 };
                    for (int q = 0; q < 8; q++)
                        if (at + q < n) out[at + q] = (uint16_t)(q & 1 ? ws[q >> 1] >> 16 : ws[q >> 1] & 0xFFFFu);
                }
                held = 0;
                done += 8;
            }
        }
    }
}

int launch_rand_stream(const uint32_t *d_start, const uint32_t *d_pow2, int npow, uint32_t nchunks, uint32_t *d_states, uint16_t *d_out,
                       size_t n, hipStream_t stream)
{
    if (!nchunks) return MLVFS_AMD_OK;
    hipLaunchKernelGGL(k_rand_states, dim3((nchunks + RAND_GROUP - 1) / RAND_GROUP), dim3(32), 0, stream, d_start, d_pow2, npow, nchunks, d_states);
    hipLaunchKernelGGL(k_rand_fill, dim3((nchunks + 255) / 256), dim3(256), 0, stream, (const uint32_t *)d_states, nchunks, d_out, n);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}


// the first launch of any kernel of this file loads the file's code object (HIP loads them lazily): the device context asks for a
// kernel's attributes when it is created, so that a clip's first frame does not pay for it (runtime.cpp: get_device)
void preload_k_stripes() { hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, (const void *)k_stripes_count); (void)hipGetLastError(); }

}  // namespace mlv
