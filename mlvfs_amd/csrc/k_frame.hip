// k_frame.hip -- the fused per-frame kernel:
//     [14-bit unpack] -> [pixel-map patches] -> [chroma smooth 2x2/3x3/5x5] -> [stripes apply]
// in ONE pass over HBM (packed in, 16-bit out = 3.75 B/px), in the stage order of
// process_frame (mlvfs/main.c:942-997).
//
// Replaces, per stage:
//   unpack          mlvfs/dng.c:813-843
//   patches         values produced by k_pixfix (ordered repair, mlvfs/cs.c:314-330)
//   chroma smooth   mlvfs/chroma_smooth.c:22-71 via mlvfs/cs.c:49-84
//   stripes apply   mlvfs/stripes.c:250-266
//
// Work decomposition (gfx950: 256 CUs, 8 XCDs, wave64, 160 KiB LDS/CU):
//   * tile = 64 x 32 Bayer cells (128 x 64 px) per 256-thread workgroup, halo of
//     2 cells; raw tile (72 x 160 u16) staged in LDS
//   * per cell (2x2 px) the EV triple {ge, dr = ev(R)-ge, db = ev(B)-ge} is
//     computed ONCE into LDS planes; the 5x5 (3x3, plus-5) medians then run on
//     the planes, one thread producing a strip of 8 horizontally adjacent cells
//     with shared column sorts / pair merges / quad selections (median_nets.h)
//   * T16 (raw2ev, 16 KiB) lives in LDS, U16 (ev2raw, 64 KiB) is gathered from L2
//   * workgroups are persistent and walk the tile list; the block -> tile map
//     keeps each XCD on a contiguous band of tiles so halo re-reads hit its L2
// No MFMA: this is a stencil / gather / selection path.
#include "clip.h"

#define MLV_NET_FN __device__ __forceinline__
#define mlv_mn(a, b) min((a), (b))
#define mlv_mx(a, b) max((a), (b))
#include "median_nets.h"

namespace mlv {

constexpr int TCW = 64;                 // tile width  in cells
constexpr int TCH = 32;                 // tile height in cells
constexpr int HC = 2;                   // halo in cells
constexpr int PW = TCW + 2 * HC;        // plane width  (68)
constexpr int PH = TCH + 2 * HC;        // plane height (36)
constexpr int RAW_W = 2 * TCW + 32;     // raw tile row: 16 px margin each side (160)
constexpr int RAW_H = 2 * PH;           // 72
constexpr int RAW_X0 = 16;              // raw column of the tile's first pixel
constexpr int STRIP = 8;                // cells per thread in the median phase

struct FrameArgs {
    const uint8_t *src;      // packed stream or u16 frames
    size_t src_stride;       // bytes between frames
    uint8_t *dst;
    size_t dst_stride;
    int w, h, black, white;
    int nframes;
    int tiles_x, tiles_y;
    const uint16_t *t16, *u16;
    // patches: per frame `n_patch` entries {pos, value}; pos = y*w + x (or -1)
    const int2 *patches;
    int n_patch;
    // stripes
    int coef[8];
};

struct __align__(16) Smem {
    uint16_t raw[RAW_H][RAW_W];
    int dr[PH][PW];
    int db[PH][PW];
    int ge[TCH][TCW];
    uint16_t t16[MLV_T16_N];
};

__device__ __forceinline__ int med3i(int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); }

// sort 5 with 12 three-input-friendly ops: sort3 + sort2, split off the extremes, sort3
__device__ __forceinline__ void sort5(int (&v)[5])
{
    const int lo = min(min(v[0], v[1]), v[2]), hi = max(max(v[0], v[1]), v[2]), mid = med3i(v[0], v[1], v[2]);
    const int d = min(v[3], v[4]), e = max(v[3], v[4]);
    const int p = max(lo, d), q = min(hi, e);
    v[0] = min(lo, d);
    v[4] = max(hi, e);
    v[1] = min(min(p, mid), q);
    v[2] = med3i(p, mid, q);
    v[3] = max(max(p, mid), q);
}

// ---------------------------------------------------------------- phase 1: tile load
template <bool PACKED>
__device__ __forceinline__ void load_tile(Smem &sm, const FrameArgs &a, const uint8_t *frame, int tx0, int ty0)
{
    const int w = a.w, h = a.h;
    const bool vec = (w % 16) == 0;
    // work item = (raw row, group of 16 px)
    for (int it = threadIdx.x; it < RAW_H * (RAW_W / 16); it += blockDim.x) {
        const int rr = it / (RAW_W / 16), g = it % (RAW_W / 16);
        int y = ty0 - 2 * HC + rr;
        int x = tx0 - RAW_X0 + 16 * g;
        y = y < 0 ? 0 : (y >= h ? h - 1 : y);
        uint32_t px[16];
        if (vec) {
            x = x < 0 ? 0 : (x > w - 16 ? w - 16 : x);
            if (PACKED) {
                const uint32_t *s = (const uint32_t *)(frame + ((size_t)y * w + x) / 16 * 28);
                uint32_t sw[7];
#pragma unroll
                for (int i = 0; i < 7; i++) { uint32_t d = s[i]; sw[i] = (d << 16) | (d >> 16); }
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int bit = 14 * k, wi = bit >> 5, sh = bit & 31;
                    if (sh + 14 <= 32) px[k] = (sw[wi] >> (32 - 14 - sh)) & 0x3FFFu;
                    else px[k] = (uint32_t)((((uint64_t)sw[wi] << 32) | sw[wi + 1]) >> (64 - 14 - sh)) & 0x3FFFu;
                }
            } else {
                const uint4 *s = (const uint4 *)(frame + ((size_t)y * w + x) * 2);
                const uint4 v0 = s[0], v1 = s[1];
                const uint32_t d[8] = { v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w };
#pragma unroll
                for (int k = 0; k < 8; k++) { px[2 * k] = d[k] & 0xFFFFu; px[2 * k + 1] = d[k] >> 16; }
            }
        } else {
            // any width: per-pixel fetch with clamped coordinates
#pragma unroll 1
            for (int k = 0; k < 16; k++) {
                int xx = x + k;
                xx = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
                const size_t i = (size_t)y * w + xx;
                if (PACKED) {
                    const uint16_t *s = (const uint16_t *)frame;
                    const size_t bit = i * 14;
                    const uint32_t two = ((uint32_t)s[bit >> 4] << 16) | s[(bit >> 4) + 1];
                    px[k] = (two >> (32 - 14 - (bit & 15))) & 0x3FFFu;
                } else {
                    px[k] = ((const uint16_t *)frame)[i];
                }
            }
        }
        uint4 lo, hi;
        lo.x = px[0] | (px[1] << 16);   lo.y = px[2] | (px[3] << 16);
        lo.z = px[4] | (px[5] << 16);   lo.w = px[6] | (px[7] << 16);
        hi.x = px[8] | (px[9] << 16);   hi.y = px[10] | (px[11] << 16);
        hi.z = px[12] | (px[13] << 16); hi.w = px[14] | (px[15] << 16);
        uint4 *d = (uint4 *)&sm.raw[rr][16 * g];
        d[0] = lo;
        d[1] = hi;
    }
}

// ---------------------------------------------------------------- phase 2: EV planes
__device__ __forceinline__ void build_planes(Smem &sm, int black)
{
    for (int c = threadIdx.x; c < PH * PW; c += blockDim.x) {
        const int j = c / PW, i = c % PW;
        const int rx = RAW_X0 - 2 * HC + 2 * i;
        const uint32_t top = *(const uint32_t *)&sm.raw[2 * j][rx];        // R | G1<<16
        const uint32_t bot = *(const uint32_t *)&sm.raw[2 * j + 1][rx];    // G2 | B<<16
        const int er = ev_of_pixel((int)(top & 0xFFFFu), black, sm.t16);
        const int eg1 = ev_of_pixel((int)(top >> 16), black, sm.t16);
        const int eg2 = ev_of_pixel((int)(bot & 0xFFFFu), black, sm.t16);
        const int eb = ev_of_pixel((int)(bot >> 16), black, sm.t16);
        const int ge = half_trunc(wadd(eg1, eg2));                          // chroma_smooth.c:32,54
        sm.dr[j][i] = wsub(er, ge);
        sm.db[j][i] = wsub(eb, ge);
        const int ji = j - HC, ii = i - HC;
        if (ji >= 0 && ji < TCH && ii >= 0 && ii < TCW) sm.ge[ji][ii] = ge;
    }
}

// ---------------------------------------------------------------- phase 3: medians
// 5x5: strip of 8 outputs from 12 sorted columns
__device__ __forceinline__ void strip_median25(const int (*plane)[PW], int row_top, int col_left, int (&med)[STRIP])
{
    int col[12][5];
#pragma unroll
    for (int r = 0; r < 5; r++) {
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int4 v = *(const int4 *)&plane[row_top + r][col_left + 4 * q];
            col[4 * q + 0][r] = v.x; col[4 * q + 1][r] = v.y; col[4 * q + 2][r] = v.z; col[4 * q + 3][r] = v.w;
        }
    }
#pragma unroll
    for (int c = 0; c < 12; c++) sort5(col[c]);
    int pr[6][10];
#pragma unroll
    for (int p = 0; p < 6; p++) mlv_merge55(col[2 * p], col[2 * p + 1], pr[p]);
    int qd[5][6];
#pragma unroll
    for (int q = 0; q < 5; q++) mlv_quad_mid6(pr[q], pr[q + 1], qd[q]);
#pragma unroll
    for (int c = 0; c < STRIP; c++) {
        const int x = c + 2;
        int o[1];
        if (x % 2 == 0) mlv_final6of11(qd[(x - 2) / 2], col[x + 2], o);
        else            mlv_final6of11(qd[(x - 1) / 2], col[x - 2], o);
        med[c] = o[0];
    }
}

// 3x3: sorted columns of 3, classic max-of-mins / med-of-meds / min-of-maxes
__device__ __forceinline__ void strip_median9(const int (*plane)[PW], int row_top, int col_left, int (&med)[STRIP])
{
    int lo[10], mi[10], hi[10];
#pragma unroll
    for (int c = 0; c < 10; c++) {
        const int a = plane[row_top][col_left + c], b = plane[row_top + 1][col_left + c], d = plane[row_top + 2][col_left + c];
        lo[c] = min(min(a, b), d);
        hi[c] = max(max(a, b), d);
        mi[c] = med3i(a, b, d);
    }
#pragma unroll
    for (int c = 0; c < STRIP; c++)
        med[c] = med3i(max(max(lo[c], lo[c + 1]), lo[c + 2]), med3i(mi[c], mi[c + 1], mi[c + 2]),
                       min(min(hi[c], hi[c + 1]), hi[c + 2]));
}

// plus-shaped 5 (chroma_smooth.c:44-47 with CHROMA_SMOOTH_2X2)
__device__ __forceinline__ void strip_median5(const int (*plane)[PW], int row_top, int col_left, int (&med)[STRIP])
{
#pragma unroll
    for (int c = 0; c < STRIP; c++) {
        const int v[5] = { plane[row_top][col_left + c + 1], plane[row_top + 1][col_left + c],
                           plane[row_top + 1][col_left + c + 1], plane[row_top + 1][col_left + c + 2],
                           plane[row_top + 2][col_left + c + 1] };
        int o[1];
        mlv_median5(v, o);
        med[c] = o[0];
    }
}

// stripes.c:250-266: p' = (uint16)min(white, (p-black)*coef/65536 + black), exact in integers
__device__ __forceinline__ uint32_t stripe_px(uint32_t p, int coef, int black16, int white16)
{
    if (coef == 0 || (int)p <= black16 + 64) return p;
    const long long num = (long long)((int)p - black16) * coef + ((long long)black16 << 16);   // value * 65536
    if (((long long)white16 << 16) < num) return (uint32_t)white16;
    return (uint32_t)(int)(num / 65536) & 0xFFFFu;
}

template <int METHOD, bool PACKED, bool PATCH, bool STRIPES>
__global__ __launch_bounds__(256) void k_frame(const FrameArgs a)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    Smem &sm = *reinterpret_cast<Smem *>(smem_raw);

    if (METHOD != 0) {
        const uint4 *src = (const uint4 *)a.t16;
        uint4 *dstl = (uint4 *)sm.t16;
        for (int i = threadIdx.x; i < MLV_T16_N * 2 / 16; i += blockDim.x) dstl[i] = src[i];
    }

    // XCD-aware persistent tile walk: blocks b, b+8, b+16, ... share an XCD; give
    // every XCD a contiguous band of the tile list
    const int tiles_per_frame = a.tiles_x * a.tiles_y;
    const long long total = (long long)tiles_per_frame * a.nframes;
    const int nx = 8;
    const long long band = (total + nx - 1) / nx;
    const int xcd = blockIdx.x % nx, slot = blockIdx.x / nx, slots = (gridDim.x + nx - 1) / nx;
    const long long band_end = min(total, (long long)(xcd + 1) * band);
    const int black16 = (int)(uint16_t)a.black, white16 = (int)(uint16_t)a.white;

    for (long long t = (long long)xcd * band + slot; t < band_end; t += slots) {
        const int f = (int)(t / tiles_per_frame), tr = (int)(t % tiles_per_frame);
        const int tx0 = (tr % a.tiles_x) * 2 * TCW, ty0 = (tr / a.tiles_x) * 2 * TCH;
        const uint8_t *frame = a.src + (size_t)f * a.src_stride;
        uint16_t *out = (uint16_t *)(a.dst + (size_t)f * a.dst_stride);

        __syncthreads();                       // previous tile's readers are done (also covers the T16 copy)
        load_tile<PACKED>(sm, a, frame, tx0, ty0);
        if (PATCH) {
            __syncthreads();
            const int2 *pl = a.patches + (size_t)f * a.n_patch;
            for (int i = threadIdx.x; i < a.n_patch; i += blockDim.x) {
                const int2 e = pl[i];
                if (e.x < 0) continue;
                const int py = e.x / a.w - (ty0 - 2 * HC), px = e.x % a.w - (tx0 - RAW_X0);
                if (py >= 0 && py < RAW_H && px >= 0 && px < RAW_W) sm.raw[py][px] = (uint16_t)e.y;
            }
        }
        __syncthreads();
        if (METHOD != 0) {
            build_planes(sm, a.black);
            __syncthreads();
        }

        // one thread = 8 cells = 16 px on two rows
        const int k = threadIdx.x % (TCW / STRIP), j = threadIdx.x / (TCW / STRIP);
        const int y = ty0 + 2 * j, x = tx0 + 2 * STRIP * k;
        uint32_t top[STRIP], bot[STRIP];        // (R | G1<<16), (G2 | B<<16)
        {
            const uint4 *r0 = (const uint4 *)&sm.raw[2 * (j + HC)][RAW_X0 + 2 * STRIP * k];
            const uint4 *r1 = (const uint4 *)&sm.raw[2 * (j + HC) + 1][RAW_X0 + 2 * STRIP * k];
            const uint4 a0 = r0[0], a1 = r0[1], b0 = r1[0], b1 = r1[1];
            top[0] = a0.x; top[1] = a0.y; top[2] = a0.z; top[3] = a0.w; top[4] = a1.x; top[5] = a1.y; top[6] = a1.z; top[7] = a1.w;
            bot[0] = b0.x; bot[1] = b0.y; bot[2] = b0.z; bot[3] = b0.w; bot[4] = b1.x; bot[5] = b1.y; bot[6] = b1.z; bot[7] = b1.w;
        }
        if (METHOD != 0 && y >= 4 && y < a.h - 5) {
            int mr[STRIP], mb[STRIP];
            if (METHOD == 5) {
                strip_median25(sm.dr, j, STRIP * k, mr);
                strip_median25(sm.db, j, STRIP * k, mb);
            } else if (METHOD == 3) {
                strip_median9(sm.dr, j + 1, STRIP * k + 1, mr);
                strip_median9(sm.db, j + 1, STRIP * k + 1, mb);
            } else {
                strip_median5(sm.dr, j + 1, STRIP * k + 1, mr);
                strip_median5(sm.db, j + 1, STRIP * k + 1, mb);
            }
#pragma unroll
            for (int c = 0; c < STRIP; c++) {
                const int xc = x + 2 * c;
                const int ge = sm.ge[j][STRIP * k + c];
                const int er = wadd(ge, mr[c]), eb = wadd(ge, mb[c]);
                // chroma_smooth.c:28, 35, 64-65
                if (xc >= 4 && xc < a.w - 4 && ge >= 2 * MLV_EV_RES && er > MLV_EV_RES && eb > MLV_EV_RES) {
                    top[c] = (top[c] & 0xFFFF0000u) | pixel_of_ev(er, a.black, a.u16);
                    bot[c] = (bot[c] & 0x0000FFFFu) | ((uint32_t)pixel_of_ev(eb, a.black, a.u16) << 16);
                }
            }
        }
        if (STRIPES) {
#pragma unroll
            for (int c = 0; c < STRIP; c++) {
                const int p0 = (2 * c) & 7, p1 = (2 * c + 1) & 7;
                top[c] = stripe_px(top[c] & 0xFFFFu, a.coef[p0], black16, white16) |
                         (stripe_px(top[c] >> 16, a.coef[p1], black16, white16) << 16);
                bot[c] = stripe_px(bot[c] & 0xFFFFu, a.coef[p0], black16, white16) |
                         (stripe_px(bot[c] >> 16, a.coef[p1], black16, white16) << 16);
            }
        }
        if (y < a.h) {
            if ((a.w % 16) == 0) {
                if (x < a.w) {
                    uint4 *o0 = (uint4 *)(out + (size_t)y * a.w + x);
                    o0[0] = make_uint4(top[0], top[1], top[2], top[3]);
                    o0[1] = make_uint4(top[4], top[5], top[6], top[7]);
                    if (y + 1 < a.h) {
                        uint4 *o1 = (uint4 *)(out + (size_t)(y + 1) * a.w + x);
                        o1[0] = make_uint4(bot[0], bot[1], bot[2], bot[3]);
                        o1[1] = make_uint4(bot[4], bot[5], bot[6], bot[7]);
                    }
                }
            } else {
#pragma unroll 1
                for (int c = 0; c < STRIP; c++) {
                    const int xc = x + 2 * c;
                    if (xc < a.w) out[(size_t)y * a.w + xc] = (uint16_t)top[c];
                    if (xc + 1 < a.w) out[(size_t)y * a.w + xc + 1] = (uint16_t)(top[c] >> 16);
                    if (y + 1 < a.h) {
                        if (xc < a.w) out[(size_t)(y + 1) * a.w + xc] = (uint16_t)bot[c];
                        if (xc + 1 < a.w) out[(size_t)(y + 1) * a.w + xc + 1] = (uint16_t)(bot[c] >> 16);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------- host launcher
template <int METHOD, bool PACKED>
static int launch_frame_t(const FrameArgs &a, bool patch, bool stripes, int num_cu, hipStream_t stream)
{
    const long long total = (long long)a.tiles_x * a.tiles_y * a.nframes;
    int grid = num_cu > 0 ? num_cu * 2 : 512;
    grid = (grid + 7) / 8 * 8;
    if (grid > total) grid = (int)((total + 7) / 8 * 8);
    const size_t shmem = sizeof(Smem);
#define MLV_LAUNCH(P, S)                                                                                  \
    do {                                                                                                  \
        auto kern = k_frame<METHOD, PACKED, P, S>;                                                        \
        MLV_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
        KernelTimer &tm = kernel_timer();                                                                 \
        const bool timed = tm.on && tm.used + 2 <= (int)tm.ev.size();                                     \
        if (timed) MLV_HIP(hipEventRecord(tm.ev[tm.used], stream));                                       \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), shmem, stream, a);                                \
        if (timed) { MLV_HIP(hipEventRecord(tm.ev[tm.used + 1], stream)); tm.used += 2; }                 \
    } while (0)
    if (patch && stripes) MLV_LAUNCH(true, true);
    else if (patch) MLV_LAUNCH(true, false);
    else if (stripes) MLV_LAUNCH(false, true);
    else MLV_LAUNCH(false, false);
#undef MLV_LAUNCH
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

int launch_frame(const Device *dev, const Geom &g, bool packed, const void *src, size_t src_stride, void *dst,
                 size_t dst_stride, int nframes, int method, const int2 *patches, int n_patch, bool stripes,
                 const int32_t *coef, hipStream_t stream)
{
    if (nframes <= 0) return MLVFS_AMD_OK;
    if (g.w < 2 || g.h < 2 || (g.w & 1)) { set_error("frame geometry %dx%d unsupported", g.w, g.h); return MLVFS_AMD_ERR_ARG; }
    if (packed && g.bpp != 14) { set_error("fused path needs 14-bit input"); return MLVFS_AMD_ERR_ARG; }
    FrameArgs a{};
    a.src = (const uint8_t *)src; a.src_stride = src_stride;
    a.dst = (uint8_t *)dst; a.dst_stride = dst_stride;
    a.w = g.w; a.h = g.h; a.black = g.black; a.white = g.white;
    a.nframes = nframes;
    a.tiles_x = (g.w + 2 * TCW - 1) / (2 * TCW);
    a.tiles_y = (g.h + 2 * TCH - 1) / (2 * TCH);
    a.t16 = dev->luts.t16; a.u16 = dev->luts.u16;
    a.patches = patches; a.n_patch = patches ? n_patch : 0;
    for (int i = 0; i < 8; i++) a.coef[i] = (stripes && coef) ? coef[i] : 0;
    const bool patch = a.n_patch > 0;
#define MLV_DISPATCH(M)                                                                          \
    return packed ? launch_frame_t<M, true>(a, patch, stripes, dev->num_cu, stream)              \
                  : launch_frame_t<M, false>(a, patch, stripes, dev->num_cu, stream)
    switch (method) {
        case 0: MLV_DISPATCH(0);
        case 2: MLV_DISPATCH(2);
        case 3: MLV_DISPATCH(3);
        case 5: MLV_DISPATCH(5);
        default: set_error("Unsupported chroma smooth method %d", method); return MLVFS_AMD_ERR_ARG;
    }
#undef MLV_DISPATCH
}

}  // namespace mlv
