// k_pixfix.hip -- bad-pixel detection and ORDERED pixel repair.
//
// Replaces mlvfs/cs.c:87-168 (interpolate_horizontal/vertical/pixel),
// cs.c:255-306 (detection loop of fix_bad_pixels) and the application loops of
// fix_bad_pixels (cs.c:314-330) and fix_focus_pixels (cs.c:463-500).
//
// Ordering.  The reference repairs the list sequentially in place, so an entry
// may read a pixel that an EARLIER entry already rewrote (taps at +-1,+-2,+-3
// along x and y).  The host (clip.cpp) turns the list into a dependency graph
// once per clip: per entry and tap the index of the latest earlier entry at that
// position (or -1), and a level = 1 + max(level of its dependencies).  The GPU
// then processes level after level in parallel; a tap with a dependency reads the
// dependency's repaired value, every other tap reads the ORIGINAL frame (nothing
// is written back until all levels are done), which is exactly the sequential
// semantics.  Level 0 (no dependencies: nearly every entry of a real map) runs
// as a flat grid over entries x frames, the few entries of the levels above in
// one workgroup per frame.  The result is a per-frame patch list {position, value}; it is
// either scattered into the 16-bit frame (in-place entry points) or consumed by
// the fused kernel's tile loader (k_frame.hip), which never materialises the
// intermediate frame.
#include "clip.h"

namespace mlv {


// tap order: x-3, x-2, x-1, x+1, x+2, x+3, y-3, y-2, y-1, y+1, y+2, y+3
__device__ __forceinline__ int tap_offset(int t, int w)
{
    const int d = (t % 6) < 3 ? (t % 6) - 3 : (t % 6) - 2;
    return t < 6 ? d : d * w;
}

// bpp: bits per pixel of a packed stream (14; 12 / 10 for Magic Lantern's reduced bit depths) -- a run-time number here: these
// kernels are bound by their gathers, and the fused kernel's launch plan (k_frame.hip) decides which streams come in packed at all
template <bool PACKED>
__device__ __forceinline__ int fetch_px(const uint8_t *frame, int pos, int bpp)
{
    if (PACKED) {
        // the two 16-bit words that hold the pixel as ONE 32-bit load (2-byte aligned: global memory takes that), halves swapped
        // back into stream order: a lane-wide gather costs the texture addresser the same for 2 or 4 bytes, and a dense map
        // (a focus-pixel map: 150 000 entries x 12 taps) is bound by exactly those gathers
        const uint16_t *s = (const uint16_t *)frame;
        const size_t bit = (size_t)pos * (size_t)bpp;
        uint32_t v;
        __builtin_memcpy(&v, s + (bit >> 4), 4);
        const uint32_t two = (v << 16) | (v >> 16);
        return (int)((two >> (32 - bpp - (int)(bit & 15))) & ((1u << bpp) - 1u));
    }
    return ((const uint16_t *)frame)[pos];
}

__device__ __forceinline__ int sdiv_safe(int a, int b) { return (a == (int)0x80000000 && b == -1) ? a : a / b; }

// one repaired value; tap(t) returns the pixel value the sequential algorithm would see
template <typename Tap>
__device__ __forceinline__ int repair_value(int kind, int black, const uint16_t *t16, const uint16_t *u16, Tap tap)
{
    auto ev = [&](int t) { return ev_of_pixel(tap(t), black, t16); };
    if (kind == 4) return tap(4);
    if (kind == 5) return tap(1);
    if (kind == 2 || kind == 3) {                       // cs.c:87-129
        const int b = kind == 2 ? 0 : 6;                // tap base: x taps 0..5, y taps 6..11
        const int dp = wabs(wsub(ev(b + 5), ev(b + 3)));
        const int dm = wabs(wsub(ev(b + 2), ev(b + 0)));
        const int sum = wadd(dp, dm);
        if (sum == 0) return tap(b + 4);
        const int cp = sdiv_safe((int)((unsigned)wsub(sum, dp) << 8), sum);
        const int cm = sdiv_safe((int)((unsigned)wsub(sum, dm) << 8), sum);
        const int e = wadd(wmul(ev(b + 4), cp) >> 8, wmul(ev(b + 1), cm) >> 8);
        return pixel_of_ev(e, black, u16);
    }
    // cross, cs.c:131-168
    const int vp = wabs(wsub(ev(11), ev(9))), vm = wabs(wsub(ev(8), ev(6)));
    const int hp = wabs(wsub(ev(5), ev(3))), hm = wabs(wsub(ev(2), ev(0)));
    const int sum = wadd(wadd(hp, hm), wadd(vp, vm));
    if (sum == 0) return tap(4);
    const int den = wmul(3, sum);
    const int cvp = sdiv_safe((int)((unsigned)wsub(sum, vp) << 8), den);
    const int cvm = sdiv_safe((int)((unsigned)wsub(sum, vm) << 8), den);
    const int chp = sdiv_safe((int)((unsigned)wsub(sum, hp) << 8), den);
    const int chm = sdiv_safe((int)((unsigned)wsub(sum, hm) << 8), den);
    const int e = wadd(wadd(wmul(ev(10), cvp) >> 8, wmul(ev(7), cvm) >> 8),
                       wadd(wmul(ev(4), chp) >> 8, wmul(ev(1), chm) >> 8));
    return pixel_of_ev(e, black, u16);
}

// Level 0 -- entries none of whose taps was rewritten by an earlier entry; on a real map (isolated hot pixels, the regular grid
// of a focus-pixel map) that is all of them, tens of thousands per frame for some cameras -- has no order to keep: one lane per
// entry and frame over the whole grid.
// The raw2ev table (16 KiB) is staged in LDS -- twelve look-ups per cross-shaped repair that would otherwise be twelve more
// gathers through the texture path --; a workgroup takes FLAT_PER_WG entries, reads them ONCE (8 bytes each: the compact list
// that clip.cpp puts behind the entries) and walks the frames of its share of the batch with them in registers.
constexpr int FLAT_PER_WG = 1024;
// frames one workgroup walks with its entries / records in registers: as many as leave the launch with some 4000 workgroups (a
// sparse map on a short batch keeps one frame per workgroup: the chip's parallelism is worth more than the re-reads)
static inline int frames_per_wg(int wgs_x, int nframes) { const long long all = (long long)wgs_x * nframes; const int fr = (int)(all / 4096); return fr < 1 ? 1 : (fr > 16 ? 16 : fr); }
template <bool PACKED>
__global__ __launch_bounds__(256) void k_pixfix_flat(const uint8_t *__restrict__ frames, size_t stride, int w, int black,
                                                     const PixEntry *__restrict__ entries, int n_level0, int n_entries, int nframes, int fpw,
                                                     int2 *__restrict__ patches, const uint16_t *__restrict__ t16,
                                                     const uint16_t *__restrict__ u16, int bpp)
{
    __shared__ uint16_t s_t16[MLV_T16_N];
    {
        const uint4 *src = (const uint4 *)t16;
        uint4 *dst = (uint4 *)s_t16;
        for (int i = threadIdx.x; i < MLV_T16_N * 2 / 16; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const int2 *flat = (const int2 *)(entries + n_entries);
    constexpr int PER = FLAT_PER_WG / 256;
    int2 ent[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int m = blockIdx.x * FLAT_PER_WG + k * 256 + threadIdx.x;
        ent[k] = m < n_level0 ? flat[m] : make_int2(-1, 0);
    }
    const int f_end = min((int)(blockIdx.y + 1) * fpw, nframes);
    for (int f = blockIdx.y * fpw; f < f_end; f++) {
        const uint8_t *frame = frames + (size_t)f * stride;
#pragma unroll 1
        for (int k = 0; k < PER; k++) {                            // (not unrolled: four inlined repairs halve the occupancy)
            const int m = blockIdx.x * FLAT_PER_WG + k * 256 + threadIdx.x;
            if (m >= n_level0) continue;
            const int pos = ent[k].x, kind = ent[k].y & 0xFF, emit = ent[k].y >> 8;
            int val = 0;
            if (kind != 0) {
                auto tap = [&](int t) { return fetch_px<PACKED>(frame, pos + tap_offset(t, w), bpp); };
                val = repair_value(kind, black, (const uint16_t *)s_t16, u16, tap) & 0xFFFF;
            }
            patches[(size_t)f * n_entries + m] = make_int2((kind != 0 && emit) ? pos : -1, val);
        }
    }
}

// The levels above: one workgroup per frame, levels separated by workgroup barriers (few entries: pairs of bad pixels within
// three pixels of each other)
template <bool PACKED>
__global__ __launch_bounds__(256) void k_pixfix(const uint8_t *__restrict__ frames, size_t stride, int w, int black,
                                                const PixEntry *__restrict__ entries, const int *__restrict__ level_off,
                                                int n_levels, int n_entries, int2 *__restrict__ patches,
                                                const uint16_t *__restrict__ t16, const uint16_t *__restrict__ u16, int bpp)
{
    const uint8_t *frame = frames + (size_t)blockIdx.x * stride;
    int2 *out = patches + (size_t)blockIdx.x * n_entries;
    for (int lv = 1; lv < n_levels; lv++) {
        const int beg = level_off[lv], end = level_off[lv + 1];
        for (int m = beg + threadIdx.x; m < end; m += blockDim.x) {
            const PixEntry e = entries[m];
            int val = 0;
            if (e.kind != 0) {
                auto tap = [&](int t) {
                    const int d = e.dep[t];
                    return d >= 0 ? out[d].y : fetch_px<PACKED>(frame, e.pos + tap_offset(t, w), bpp);
                };
                val = repair_value(e.kind, black, t16, u16, tap) & 0xFFFF;
            }
            out[m] = make_int2((e.kind != 0 && e.emit) ? e.pos : -1, val);
        }
        __threadfence_block();
        __syncthreads();
    }
}

// in-place entry points: the finished list into the 16-bit frames (after every level: a tap never sees a value of this call
// unless the list order says so)
__global__ __launch_bounds__(256) void k_pixfix_scatter(const int2 *__restrict__ patches, int n_entries, uint16_t *scatter_base,
                                                        size_t scatter_stride)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_entries) return;
    const int2 p = patches[(size_t)blockIdx.y * n_entries + m];
    if (p.x >= 0) ((uint16_t *)((uint8_t *)scatter_base + (size_t)blockIdx.y * scatter_stride))[p.x] = (uint16_t)p.y;
}

int launch_pixfix(int packed, const void *frames, size_t stride, int w, int black, const void *entries,
                  const int *level_off, int n_levels, int n_level0, int n_entries, void *patches, void *scatter,
                  size_t scatter_stride, int nframes, const DeviceLuts &luts, hipStream_t stream)
{
    if (n_entries <= 0 || nframes <= 0) return MLVFS_AMD_OK;
    const int bpp = packed == 1 ? 14 : packed;           // 1: what callers that pass `true` mean
    const int flat_x = (n_level0 + FLAT_PER_WG - 1) / FLAT_PER_WG, fpw = frames_per_wg(flat_x, nframes);
    const dim3 flat(flat_x, (nframes + fpw - 1) / fpw), all((n_entries + 255) / 256, nframes);
    if (packed) {
        if (n_level0 > 0)
            hipLaunchKernelGGL(k_pixfix_flat<true>, flat, dim3(256), 0, stream, (const uint8_t *)frames, stride, w, black,
                               (const PixEntry *)entries, n_level0, n_entries, nframes, fpw, (int2 *)patches, luts.t16, luts.u16, bpp);
        if (n_levels > 1)
            hipLaunchKernelGGL(k_pixfix<true>, dim3(nframes), dim3(256), 0, stream, (const uint8_t *)frames, stride, w, black,
                               (const PixEntry *)entries, level_off, n_levels, n_entries, (int2 *)patches, luts.t16, luts.u16, bpp);
    } else {
        if (n_level0 > 0)
            hipLaunchKernelGGL(k_pixfix_flat<false>, flat, dim3(256), 0, stream, (const uint8_t *)frames, stride, w, black,
                               (const PixEntry *)entries, n_level0, n_entries, nframes, fpw, (int2 *)patches, luts.t16, luts.u16, bpp);
        if (n_levels > 1)
            hipLaunchKernelGGL(k_pixfix<false>, dim3(nframes), dim3(256), 0, stream, (const uint8_t *)frames, stride, w, black,
                               (const PixEntry *)entries, level_off, n_levels, n_entries, (int2 *)patches, luts.t16, luts.u16, bpp);
    }
    if (scatter)
        hipLaunchKernelGGL(k_pixfix_scatter, all, dim3(256), 0, stream, (const int2 *)patches, n_entries, (uint16_t *)scatter,
                           scatter_stride);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

// The fused kernel (k_frame.hip) takes the repair cell by cell: the four pixels of every Bayer cell that holds a repaired pixel
// -- the repaired ones from the patch list, the others from the frame --, in the order of its per-tile lists.  One lane per
// listed cell and frame; the kernel then reads one 16-byte record per cell and never looks at the patch list.
// (a lane reads its record once and walks the frames of its share of the batch: the records are 20 bytes each, 190 000 of them for
// the densest focus-pixel map)
template <bool PACKED>
__global__ __launch_bounds__(256) void k_pixfix_cells(const uint8_t *__restrict__ frames, size_t stride, int w, int h,
                                                      const CellRec *__restrict__ recs, int n_rec, int nframes, int fpw,
                                                      const int2 *__restrict__ patches, int n_entries, int4 *__restrict__ cells, int bpp)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rec) return;
    const CellRec rec = recs[r];
    const int cx = rec.cell & 0xFFFF, cy = rec.cell >> 16;
    int pos[4];
#pragma unroll
    for (int q = 0; q < 4; q++) pos[q] = min(2 * cy + (q >> 1), h - 1) * w + min(2 * cx + (q & 1), w - 1);
    const int f_end = min((int)(blockIdx.y + 1) * fpw, nframes);
    for (int f = blockIdx.y * fpw; f < f_end; f++) {
        const uint8_t *frame = frames + (size_t)f * stride;
        int v[4];
#pragma unroll
        for (int q = 0; q < 4; q++)
            v[q] = rec.e[q] >= 0 ? (patches[(size_t)f * n_entries + rec.e[q]].y & 0xFFFF) : fetch_px<PACKED>(frame, pos[q], bpp);
        cells[(size_t)f * n_rec + r] = make_int4(rec.cell, v[0] | (v[1] << 16), v[2] | (v[3] << 16), 0);
    }
}

int launch_pixfix_cells(int packed, const void *frames, size_t stride, int w, int h, const CellRec *recs, int n_rec,
                        const void *patches, int n_entries, void *cells, int nframes, hipStream_t stream)
{
    if (n_rec <= 0 || nframes <= 0) return MLVFS_AMD_OK;
    const int bpp = packed == 1 ? 14 : packed;
    const int fpw = frames_per_wg((n_rec + 255) / 256, nframes);
    const dim3 grid((n_rec + 255) / 256, (nframes + fpw - 1) / fpw);
    if (packed)
        hipLaunchKernelGGL(k_pixfix_cells<true>, grid, dim3(256), 0, stream, (const uint8_t *)frames, stride, w, h, recs, n_rec, nframes, fpw,
                           (const int2 *)patches, n_entries, (int4 *)cells, bpp);
    else
        hipLaunchKernelGGL(k_pixfix_cells<false>, grid, dim3(256), 0, stream, (const uint8_t *)frames, stride, w, h, recs, n_rec, nframes, fpw,
                           (const int2 *)patches, n_entries, (int4 *)cells, bpp);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

// A small map (a clip's hot and cold pixels: a hundred entries) for the fused kernel in ONE launch: every level and the cell
// records, one workgroup per frame.  The three launches above cost a batch of 100 frames 2 % of its time in launch gaps.
template <bool PACKED>
__global__ __launch_bounds__(256) void k_pixfix_small(const uint8_t *__restrict__ frames, size_t stride, int w, int h, int black,
                                                      const PixEntry *__restrict__ entries, const int *__restrict__ level_off,
                                                      int n_levels, int n_entries, int2 *__restrict__ patches,
                                                      const CellRec *__restrict__ recs, int n_rec, int4 *__restrict__ cells,
                                                      const uint16_t *__restrict__ t16, const uint16_t *__restrict__ u16, int bpp)
{
    const uint8_t *frame = frames + (size_t)blockIdx.x * stride;
    int2 *out = patches + (size_t)blockIdx.x * n_entries;
    for (int lv = 0; lv < n_levels; lv++) {
        const int beg = level_off[lv], end = level_off[lv + 1];
        for (int m = beg + threadIdx.x; m < end; m += blockDim.x) {
            const PixEntry e = entries[m];
            int val = 0;
            if (e.kind != 0) {
                auto tap = [&](int t) {
                    const int d = e.dep[t];
                    return d >= 0 ? out[d].y : fetch_px<PACKED>(frame, e.pos + tap_offset(t, w), bpp);
                };
                val = repair_value(e.kind, black, t16, u16, tap) & 0xFFFF;
            }
            out[m] = make_int2((e.kind != 0 && e.emit) ? e.pos : -1, val);
        }
        __threadfence_block();
        __syncthreads();
    }
    for (int r = threadIdx.x; r < n_rec; r += blockDim.x) {
        const CellRec rec = recs[r];
        const int cx = rec.cell & 0xFFFF, cy = rec.cell >> 16;
        int v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int x = min(2 * cx + (q & 1), w - 1), y = min(2 * cy + (q >> 1), h - 1);
            v[q] = rec.e[q] >= 0 ? (out[rec.e[q]].y & 0xFFFF) : fetch_px<PACKED>(frame, y * w + x, bpp);
        }
        cells[(size_t)blockIdx.x * n_rec + r] = make_int4(rec.cell, v[0] | (v[1] << 16), v[2] | (v[3] << 16), 0);
    }
}

// patch list + cell records for the fused kernel: one launch for a small map, the flat grid + levels + cells for a large one
int launch_pixfix_for_frame_kernel(int packed, const void *frames, size_t stride, int w, int h, int black, const void *entries,
                                   const int *level_off, int n_levels, int n_level0, int n_entries, void *patches,
                                   const CellRec *recs, int n_rec, void *cells, int nframes, const DeviceLuts &luts, hipStream_t stream)
{
    if (n_entries <= 0 || nframes <= 0) return MLVFS_AMD_OK;
    const int bpp = packed == 1 ? 14 : packed;
    if (n_entries > 1024 || n_rec > 2048) {
        int rc = launch_pixfix(packed, frames, stride, w, black, entries, level_off, n_levels, n_level0, n_entries, patches, nullptr, 0,
                               nframes, luts, stream);
        if (rc) return rc;
        return launch_pixfix_cells(packed, frames, stride, w, h, recs, n_rec, patches, n_entries, cells, nframes, stream);
    }
    if (packed)
        hipLaunchKernelGGL(k_pixfix_small<true>, dim3(nframes), dim3(256), 0, stream, (const uint8_t *)frames, stride, w, h, black,
                           (const PixEntry *)entries, level_off, n_levels, n_entries, (int2 *)patches, recs, n_rec, (int4 *)cells,
                           luts.t16, luts.u16, bpp);
    else
        hipLaunchKernelGGL(k_pixfix_small<false>, dim3(nframes), dim3(256), 0, stream, (const uint8_t *)frames, stride, w, h, black,
                           (const PixEntry *)entries, level_off, n_levels, n_entries, (int2 *)patches, recs, n_rec, (int4 *)cells,
                           luts.t16, luts.u16, bpp);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

// ------------------------------------------------------------------ detection
// pass 1: one lane per pixel; per 64-pixel span a ballot word, per row a count
__global__ __launch_bounds__(256) void k_badpix_flags(const uint16_t *__restrict__ img, int w, int h, int black,
                                                      int aggressive, unsigned long long *__restrict__ mask,
                                                      int words_per_row, int *__restrict__ row_count,
                                                      const uint16_t *__restrict__ t16)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    bool bad = false;
    if (x >= 6 && x < w - 6 && y >= 6 && y < h - 6) {
        const int p = img[x + (size_t)y * w];
        int t0 = -1, t1 = -1, t2 = -1;                 // three largest same-colour neighbours
#pragma unroll
        for (int dy = -2; dy <= 2; dy += 2)
#pragma unroll
            for (int dx = -2; dx <= 2; dx += 2) {
                if (dx == 0 && dy == 0) continue;
                const int q = img[(x + dx) + (size_t)(y + dy) * w];
                if (q >= t0) { t2 = t1; t1 = t0; t0 = q; }
                else if (q >= t1) { t2 = t1; t1 = q; }
                else if (q > t2) { t2 = q; }
            }
        const int dark_lo = black - 96, dark_hi = black + 96;       // cs.c:255-257
        if (p < dark_lo) bad = true;
        else {
            const int ep = ev_of_pixel(p, black, t16);
            const int d1 = wsub(ep, ev_of_pixel(t1, black, t16));
            if (d1 > 2 * MLV_EV_RES && p > dark_hi) bad = true;
            else if (aggressive) {
                const int d2 = wsub(ep, ev_of_pixel(t2, black, t16));
                if ((d1 > MLV_EV_RES || d2 > MLV_EV_RES) && p > dark_hi) bad = true;
            }
        }
    }
    const unsigned long long b = __ballot(bad);
    if ((threadIdx.x & 63) == 0 && x < words_per_row * 64) {
        mask[(size_t)y * words_per_row + x / 64] = b;
        if (b) atomicAdd(&row_count[y], __popcll(b));
    }
}

// pass 2: one workgroup per row; raster order = rows in order, bits in order
__global__ __launch_bounds__(64) void k_badpix_compact(const unsigned long long *__restrict__ mask, int words_per_row,
                                                       const int *__restrict__ row_count, int crop_x, int crop_y,
                                                       int2 *__restrict__ list, int cap)
{
    const int y = blockIdx.x;
    if (row_count[y] == 0) return;
    int part = 0;
    for (int r = threadIdx.x; r < y; r += 64) part += row_count[r];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if (threadIdx.x != 0) return;
    int at = part;
    for (int wd = 0; wd < words_per_row; wd++) {
        unsigned long long b = mask[(size_t)y * words_per_row + wd];
        while (b) {
            const int bit = __ffsll((long long)b) - 1;
            b &= b - 1;
            if (at < cap) list[at] = make_int2(wd * 64 + bit + crop_x, y + crop_y);
            at++;
        }
    }
}

int launch_badpix_detect(const void *d_frame, int w, int h, int black, int aggressive, int crop_x, int crop_y,
                         void *d_mask, int words_per_row, int *d_row_count, void *d_list, int cap,
                         const DeviceLuts &luts, hipStream_t stream)
{
    MLV_HIP(hipMemsetAsync(d_row_count, 0, sizeof(int) * h, stream));
    dim3 grid(words_per_row * 64 / 256 + ((words_per_row * 64) % 256 ? 1 : 0), h);
    hipLaunchKernelGGL(k_badpix_flags, grid, dim3(256), 0, stream, (const uint16_t *)d_frame, w, h, black, aggressive,
                       (unsigned long long *)d_mask, words_per_row, d_row_count, luts.t16);
    hipLaunchKernelGGL(k_badpix_compact, dim3(h), dim3(64), 0, stream, (const unsigned long long *)d_mask, words_per_row,
                       d_row_count, crop_x, crop_y, (int2 *)d_list, cap);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}


// the first launch of any kernel of this file loads the file's code object (HIP loads them lazily): the device context asks for a
// kernel's attributes when it is created, so that a clip's first frame does not pay for it (runtime.cpp: get_device)
void preload_k_pixfix() { hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, (const void *)k_pixfix_cells<true>); (void)hipGetLastError(); }

}  // namespace mlv
