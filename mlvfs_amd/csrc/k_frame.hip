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
//   * tile = 64 x 16 Bayer cells (128 x 32 px) per 256-thread workgroup, halo 2 cells;
//     39 KiB of LDS and <= 128 VGPRs so that FOUR workgroups (16 waves) share a CU and
//     cover each other's barriers, LDS and HBM latencies
//   * LOADER: one thread = one "item" = 8 cells (2 rows x 16 px, i.e. 2 x 28 B of
//     packed stream as 7 dwords each).  It unpacks in registers and writes, per
//     cell, the EV triple {ge, dr = ev(R)-ge, db = ev(B)-ge} to LDS planes (the
//     planes are what the medians run on; every pixel's EV is computed once).
//     The 2-cell halo columns left and right are separate small "edge" items.
//     The packed dwords of the NEXT tile are prefetched into registers before
//     the median phase, so HBM latency hides behind the selection networks.
//   * MEDIANS: one thread = a strip of 4 horizontally adjacent cells with shared
//     column sorts / pair merges / quad selections (median_nets.h), 10 ds_read_b128
//     per plane, conflict-free lane -> (row, strip) map.
//   * raw2ev lives in LDS as the 8192-entry mantissa-normalised 16-bit table (common.h),
//     ev2raw's 64 KiB table is gathered from L2 (all gathers of a strip in flight at once)
//   * pixel-map patches: per-tile lists of repaired cells (built once per clip on the host; values per frame: k_pixfix_cells)
//     recompute just the cells they touch
//   * persistent workgroups; the four residents of a CU draw tiles by ticket from that CU's contiguous
//     range of the tile list (the ranges of an XCD's CUs adjacent, so halo re-reads hit its own L2)
// No MFMA: this is a stencil / gather / selection path.
#include "clip.h"
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>
#include <algorithm>
#include <cstdio>

// two 16-bit lanes per register: v_pk_min_i16 / v_pk_max_i16 issue at the rate of v_min_i32 (tools/valu_rate.hip)
typedef short mlv_pk16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int mlv_min_(int a, int b) { return min(a, b); }
__device__ __forceinline__ int mlv_max_(int a, int b) { return max(a, b); }
__device__ __forceinline__ mlv_pk16 mlv_min_(mlv_pk16 a, mlv_pk16 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ mlv_pk16 mlv_max_(mlv_pk16 a, mlv_pk16 b) { return __builtin_elementwise_max(a, b); }
#define MLV_NET_FN __device__ __forceinline__
#define mlv_mn(a, b) mlv_min_((a), (b))
#define mlv_mx(a, b) mlv_max_((a), (b))
#include "median_nets.h"

namespace mlv {

constexpr int TCW = FRAME_TCW;          // tile width  in cells (64)
constexpr int HC = FRAME_HC;            // halo in cells (2)
constexpr int PW = TCW + 2 * HC;        // plane width  (68)
constexpr int STRIP = 4;                // cells per thread in the median phase
constexpr int GROUPS = TCW / 8;         // full items per plane row (8)
#ifndef KF_DARK_ITEMS_MIN
#define KF_DARK_ITEMS_MIN 16
#endif
constexpr int DARK_ITEMS_MIN = KF_DARK_ITEMS_MIN;      // of the loader items of a tile
#ifndef KF_FB_ROBUST
#define KF_FB_ROBUST 12
#endif
constexpr int FB_ROBUST = KF_FB_ROBUST;           // 5x5: more uncertain strips than this: the next tiles' lanes share their references row by row
#ifndef KF_FB_WAIT_MIN
#define KF_FB_WAIT_MIN 1
#endif
#ifndef KF_FB_WAIT_MAX
#define KF_FB_WAIT_MAX 15
#endif
constexpr int FB_WAIT_MIN = KF_FB_WAIT_MIN, FB_WAIT_MAX = KF_FB_WAIT_MAX;     // 5x5: tiles that skip the packed networks after a busy one
#ifndef KF_FB_DIRECT
#define KF_FB_DIRECT 60
#endif
constexpr int FB_DIRECT = KF_FB_DIRECT;          // 5x5: more uncertain strips than this (of 240): the next tiles go to the 32-bit networks directly
constexpr int PMAP_WORDS = 64;          // tiles per frame covered by the LDS patch bitmap: 2048 (3584x1320 has 1176 / 1232)
// Tile height in cells: 16 rows of 16 strips fill the 256 threads; 5x5 tiles have 15 rows, the 16 lanes that this frees
// compute the right-hand halo group of every row for the neighbour-sharing medians (strip_chain_*, below).
constexpr int tile_rows_of(int method) { return method == 5 ? FRAME_TCH5 : FRAME_TCH; }

struct FrameArgs {
    const uint8_t *src;      // packed stream or u16 frames
    size_t src_stride;       // bytes between frames
    uint8_t *dst;
    size_t dst_stride;
    int w, h, black, white;
    int nframes;
    int tiles_x, tiles_y;
    const uint16_t *t16;
    const uint16_t *e2r;     // (uint16)(ev2raw[ev] + black), ev in [0, 14 * 32768): the output pixel by EV, one buffer look-up
    // pixel map: per frame `n_rec` cell records {cell, R | G1 << 16, G2 | B << 16, -} (k_pixfix_cells), listed tile by tile (CSR)
    const int4 *cells;
    int n_rec;
    const int *tile_off;
    // stripes
    int coef[8];
    int coef_fast;           // all |coef - 65536| < 32768: 32-bit epilogue
    int coef_pk;             // additionally 14-bit input, white > black + 64, black <= 16384: packed 16-bit epilogue
    int patch, stripes;      // wave-uniform stage switches
    int *tickets;            // [groups] next tile of each group's range + [1] workgroups done (the last one zeroes them all)
    int groups;              // workgroups b, b + groups, b + 2 groups, ... form a group (one CU's residents) and share a tile range
#ifdef KF_DIAG_TIMES
    unsigned long long *times;
#endif
};

// Table look-ups as buffer loads with idxen: the address unit scales the index by the descriptor's stride, no VALU address arithmetic
// (tools/gather_probe.hip checks the semantics on gfx950).  The LLVM intrinsics are bound by name: hipcc has no builtin for
// the struct forms, and unlike inline asm the compiler counts these loads in its s_waitcnt bookkeeping.
typedef int mlv_i32x4 __attribute__((ext_vector_type(4)));
__device__ unsigned short mlv_sbl_u16(mlv_i32x4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.i16");
__device__ __forceinline__ mlv_i32x4 table_rsrc(const void *p, unsigned stride, unsigned entries)
{
    const unsigned long long a = (unsigned long long)p;
    mlv_i32x4 r;
    r.x = (int)(unsigned)a;
    r.y = (int)(((unsigned)(a >> 32) & 0xFFFFu) | (stride << 16));
    r.z = (int)entries;
    r.w = 0x00020000;
    return r;
}
constexpr int E2R_ENTRIES = 14 * MLV_EV_RES;

// SPREAD: the T16 table with entry i at i + (i >> 7).  A pixel below 2^e above black uses only every 2^(13-e)-th entry, so
// the look-ups of dark footage crowd into a few LDS banks (below 128 DN: one); the spread form puts those entries into
// different banks for two more operations per pixel.  Chosen per clip from its first frame (launch_frame's `spread`).
constexpr int XCHG_WORDS = 28;          // per tile row: what the halo group lane hands to the last strip of the row (7 x 16 bytes)
template <bool SPREAD_, int TCH_>
struct __align__(16) SmemT {
    static constexpr bool SPREAD = SPREAD_;
    static constexpr int TCH = TCH_;                    // tile height in cells
    static constexpr int PH = TCH_ + 2 * HC;            // plane height
    static constexpr int N_FULL = PH * GROUPS;          // full loader items per tile (threads 0 .. N_FULL-1)
    static constexpr int N_ITEMS = N_FULL + PH;         // + one edge item per plane row
    static constexpr bool CHAIN = TCH_ == FRAME_TCH5;   // the 5x5 geometry
    uint16_t raw[2 * TCH][2 * TCW];     // interior pixels (post patch), 8 KiB
    int dr[PH][PW];                     // 5.3 KiB
    int db[PH][PW];
    int ge[TCH][TCW];                   // 4 KiB
    uint16_t t16[MLV_T16_N + (SPREAD_ ? 64 : 0)];   // mantissa-normalised raw2ev (common.h), 16 KiB
    uint32_t has_patch[PMAP_WORDS];     // one bit per tile of a frame: some pixel-map entry touches it
    uint32_t xchg[CHAIN ? TCH_ : 1][XCHG_WORDS];     // 5x5: sorted columns / pair list / rank window of each row's halo group
    uint8_t fb_queue[CHAIN ? 256 : 4];  // 5x5: strips whose packed medians are not certain (row * 16 + strip), settled densely
    int fb_count;
    int next_ticket;
    int dark_items[2];                  // loader items of the current / next tile that hold pixels at or below black (5x5 only)
};
static_assert(SmemT<false, FRAME_TCH>::N_ITEMS <= 256 && FRAME_TCH * (TCW / STRIP) == 256, "one item and one strip per thread");
static_assert(SmemT<false, FRAME_TCH5>::N_ITEMS <= 256 && FRAME_TCH5 * (TCW / STRIP) + FRAME_TCH5 <= 256, "strips + one halo-group lane per row");

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the
// vector-memory counter, which would stall every wave on its own global stores (and
// on the prefetch loads of the next tile) at each of the three barriers per tile.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

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

// raw2ev through the LDS mantissa table (main.c:163-167 semantics, see common.h):
//   ev(lin) = T16[(lin << (13 - e)) & 8191] + (e << 15),  e = floor(log2 lin)
// Pixels at or below black (ev = INT_MIN / 0) or beyond the table are the rare case: a
// wave-wide vote picks the branch-free fast path unless some lane needs the fix-up.
// Exponent and 13-bit mantissa fraction come out of the float conversion (exact for l < 2^24): one v_cvt + one v_bfe
// instead of count-leading-zeros, variable shift and mask.
__device__ __forceinline__ int ev_index(int l) { return (int)((__float_as_uint((float)(unsigned)l) >> 10) & 8191u); }
__device__ __forceinline__ int ev_value(int l, int tv)
{
    return tv + (int)((__float_as_uint((float)(unsigned)l) >> 8) & 0xFFFF8000u) - (127 << 15);
}

// v_bfe_u32 as written: the optimiser otherwise re-expands a bit-field extract whose result is shifted or scaled into
// shift + and (two quarter-rate instructions instead of one)
template <int OFF, int WIDTH>
__device__ __forceinline__ uint32_t bfe_asm(uint32_t v)
{
    uint32_t r;
    asm("v_bfe_u32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "n"(OFF), "n"(WIDTH));
    return r;
}

// EV triples of two adjacent Bayer cells (8 pixels): r/g1 on the top row, g2/b below
// byte offset of the table entry of a pixel whose float is fb
template <bool SPREAD>
__device__ __forceinline__ uint32_t t16_offset(uint32_t fb)
{
    if (SPREAD) return (bfe_asm<17, 6>(fb) << 1) + bfe_asm<9, 14>(fb);       // v_bfe + v_lshl_add
    return bfe_asm<9, 14>(fb);
}

template <bool SPREAD>
__device__ __forceinline__ void cell_pair_ev(const uint32_t *p0, const uint32_t *p1, int black, const uint16_t *t, bool slow,
                                             int (&ge)[2], int (&dr)[2], int (&db)[2])
{
    const int px[8] = { (int)p0[0], (int)p0[1], (int)p1[0], (int)p1[1], (int)p0[2], (int)p0[3], (int)p1[2], (int)p1[3] };
    int lin[8], tv[8], ev[8];
#pragma unroll
    for (int i = 0; i < 8; i++) lin[i] = px[i] - black;
    if (!slow) {                                        // wave-uniform: two separate paths, so the common one carries no selects or copies
        // Every lin is in [1, 16383]: its float has at most 13 mantissa bits below the leading one, so bits 0..9 are zero and
        // bits 9..22 ARE the byte offset of the table entry (one v_bfe).  The EVs carry the exponent bias (127 << 15): it
        // cancels in dr and db, the sum of two biased EVs is positive (a plain shift halves it) and ge drops it at the end.
        uint32_t fb[8];
#pragma unroll
        for (int i = 0; i < 8; i++) fb[i] = __float_as_uint((float)(unsigned)lin[i]);
#pragma unroll
        for (int i = 0; i < 8; i++) tv[i] = *(const uint16_t *)((const char *)t + t16_offset<SPREAD>(fb[i]));
        // opaque use: keeps the eight LDS reads unconditional and back to back (the compiler
        // otherwise sinks each read next to its use and waits for it there)
        uint32_t ex[8], eb[8];
#pragma unroll
        for (int i = 0; i < 8; i++) ex[i] = bfe_asm<23, 8>(fb[i]);                                    // while the reads are in flight
        asm volatile("" :: "v"(ex[0]), "v"(ex[1]), "v"(ex[2]), "v"(ex[3]), "v"(ex[4]), "v"(ex[5]), "v"(ex[6]), "v"(ex[7]));
        asm volatile("" :: "v"(tv[0]), "v"(tv[1]), "v"(tv[2]), "v"(tv[3]), "v"(tv[4]), "v"(tv[5]), "v"(tv[6]), "v"(tv[7]));
#pragma unroll
        for (int i = 0; i < 8; i++) eb[i] = (ex[i] << 15) + (uint32_t)tv[i];                          // v_lshl_add_u32
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const uint32_t gb = (eb[4 * c + 1] + eb[4 * c + 2]) >> 1;           // chroma_smooth.c:32,54 (both EVs >= 0: trunc == floor)
            ge[c] = (int)(gb - (127u << 15));
            dr[c] = (int)(eb[4 * c + 0] - gb);
            db[c] = (int)(eb[4 * c + 3] - gb);
        }
        return;
    }
    {
        int l[8];
#pragma unroll
        for (int i = 0; i < 8; i++) l[i] = min(max(lin[i], 1), 16383);
#pragma unroll
        for (int i = 0; i < 8; i++) tv[i] = t[SPREAD ? ev_index(l[i]) + (ev_index(l[i]) >> 7) : ev_index(l[i])];
        asm volatile("" :: "v"(tv[0]), "v"(tv[1]), "v"(tv[2]), "v"(tv[3]), "v"(tv[4]), "v"(tv[5]), "v"(tv[6]), "v"(tv[7]));
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int e = ev_value(l[i], tv[i]);
            ev[i] = ((unsigned)(lin[i] - 1) < 16383u) ? e : (lin[i] == 0 ? (int)0x80000000 : 0);
        }
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
        ge[c] = half_trunc(wadd(ev[4 * c + 1], ev[4 * c + 2]));             // chroma_smooth.c:32,54
        dr[c] = wsub(ev[4 * c + 0], ge[c]);
        db[c] = wsub(ev[4 * c + 3], ge[c]);
    }
}

// the common path of cell_pair_ev for NC adjacent cells at once: 4 NC table reads in flight, one wait
template <int NC, bool SPREAD>
__device__ __forceinline__ void cell_multi_ev_fast(const uint32_t *p0, const uint32_t *p1, int black, const uint16_t *t,
                                                   int (&ge)[NC], int (&dr)[NC], int (&db)[NC])
{
    uint32_t fb[4 * NC], tv[4 * NC], ex[4 * NC], eb[4 * NC];
    // float(px) - float(black) == float(px - black), exactly (integers below 2^24): v_sub_f32 issues at twice the rate of
    // v_sub_u32 on gfx950 (tools/valu_rate2.hip: 0.45 against 0.29 per clock and SIMD)
    const float fblack = (float)black;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const uint32_t px[4] = { p0[2 * c], p0[2 * c + 1], p1[2 * c], p1[2 * c + 1] };
#pragma unroll
        for (int i = 0; i < 4; i++) fb[4 * c + i] = __float_as_uint((float)px[i] - fblack);
    }
#pragma unroll
    for (int i = 0; i < 4 * NC; i++) tv[i] = *(const uint16_t *)((const char *)t + t16_offset<SPREAD>(fb[i]));
#pragma unroll
    for (int i = 0; i < 4 * NC; i++) ex[i] = bfe_asm<23, 8>(fb[i]);
#pragma unroll
    for (int i = 0; i < 4 * NC; i += 8) {                // opaque uses: the reads stay unconditional and back to back
        asm volatile("" :: "v"(ex[i]), "v"(ex[i + 1]), "v"(ex[i + 2]), "v"(ex[i + 3]), "v"(ex[i + 4]), "v"(ex[i + 5]), "v"(ex[i + 6]), "v"(ex[i + 7]));
    }
#pragma unroll
    for (int i = 0; i < 4 * NC; i += 8) {
        asm volatile("" :: "v"(tv[i]), "v"(tv[i + 1]), "v"(tv[i + 2]), "v"(tv[i + 3]), "v"(tv[i + 4]), "v"(tv[i + 5]), "v"(tv[i + 6]), "v"(tv[i + 7]));
    }
#pragma unroll
    for (int i = 0; i < 4 * NC; i++) eb[i] = (ex[i] << 15) + tv[i];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const uint32_t gb = (eb[4 * c + 1] + eb[4 * c + 2]) >> 1;
        ge[c] = (int)(gb - (127u << 15));
        dr[c] = (int)(eb[4 * c + 0] - gb);
        db[c] = (int)(eb[4 * c + 3] - gb);
    }
}

// single cell (pixel-map path); slow (wave-uniform): with the out-of-table fix-ups
template <bool SPREAD>
__device__ __forceinline__ void cell_ev(int r, int g1, int g2, int b, int black, const uint16_t *t, bool slow, int &ge, int &dr, int &db)
{
    const uint32_t p0[4] = { (uint32_t)r, (uint32_t)g1, (uint32_t)r, (uint32_t)g1 }, p1[4] = { (uint32_t)g2, (uint32_t)b, (uint32_t)g2, (uint32_t)b };
    int g[2], a[2], c[2];
    cell_pair_ev<SPREAD>(p0, p1, black, t, slow, g, a, c);
    ge = g[0]; dr = a[0]; db = c[0];
}

// ---------------------------------------------------------------- loader
// Thread t < 160 owns the full item (plane row t/8, group t%8); threads 160..179 own the
// edge item (two halo cells left + right) of plane row t-160.  An item is NW dwords per
// row in plain register arrays.  An edge item loads the two dwords that hold the right
// halo's first 4 px into positions 0,1 and the two dwords that hold the left halo's last
// 4 px into the last two positions, so the SAME unpack yields right halo = px 0..3 and
// left halo = px 12..15: one uniform code path, no per-lane variant of the register layout.
template <bool PACKED> struct Words { static constexpr int N = PACKED ? 7 : 8; };

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// byte offset of the 16-px group starting at pixel (x, y) inside a frame
// (32-bit: a frame has fewer than 2^28 pixels, checked by the launcher; the 64-bit form cost ten 64-bit multiply-adds per prefetch)
template <bool PACKED>
__device__ __forceinline__ uint32_t group_offset(int w, int x, int y)
{
    const uint32_t px = (uint32_t)y * (uint32_t)w + (uint32_t)x;
    return PACKED ? (px >> 4) * 28u : px * 2u;
}

template <bool PACKED, int N_FULL>
__device__ __forceinline__ void issue_item(uint32_t (&r0)[Words<PACKED>::N], uint32_t (&r1)[Words<PACKED>::N],
                                           const uint8_t *frame, int w, int h, int tx0, int ty0, int tid)
{
    constexpr int N = Words<PACKED>::N;
    const bool edge = tid >= N_FULL;
    const int pr = edge ? tid - N_FULL : tid >> 3;
    const int y = ty0 - 2 * HC + 2 * pr;
    const int x_main = edge ? clampi(tx0 - 16, 0, w - 16) : clampi(tx0 + 16 * (tid & 7), 0, w - 16);
    const int x_right = clampi(tx0 + 2 * TCW, 0, w - 16);
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const int yy = clampi(y + rr, 0, h - 1);
        const uint32_t *sm_ = (const uint32_t *)(frame + group_offset<PACKED>(w, x_main, yy));
        const uint32_t *sr_ = (const uint32_t *)(frame + group_offset<PACKED>(w, x_right, yy));
        uint32_t (&r)[N] = rr ? r1 : r0;
#pragma unroll
        for (int i = 0; i < N; i++) r[i] = ((edge && i < 2) ? sr_ : sm_)[i];
    }
}

__device__ __forceinline__ uint32_t sw16(uint32_t d) { return (d << 16) | (d >> 16); }

// pixel K (0..15) of a 16-pixel group from its seven MSB-first stream words
template <int K>
__device__ __forceinline__ uint32_t px14(const uint32_t (&s)[7])
{
    constexpr int bit = 14 * K, wi = bit >> 5, sh = bit & 31;
    if constexpr (sh + 14 <= 32) return (s[wi] >> (32 - 14 - sh)) & 0x3FFFu;
    else return (uint32_t)((((uint64_t)s[wi] << 32) | s[wi + 1]) >> (64 - 14 - sh)) & 0x3FFFu;
}

template <bool PACKED>
__device__ __forceinline__ void unpack16(const uint32_t (&d)[Words<PACKED>::N], uint32_t (&px)[16])
{
    if constexpr (PACKED) {
        uint32_t s[7];
#pragma unroll
        for (int i = 0; i < 7; i++) s[i] = sw16(d[i]);
        px[0] = px14<0>(s);   px[1] = px14<1>(s);   px[2] = px14<2>(s);   px[3] = px14<3>(s);
        px[4] = px14<4>(s);   px[5] = px14<5>(s);   px[6] = px14<6>(s);   px[7] = px14<7>(s);
        px[8] = px14<8>(s);   px[9] = px14<9>(s);   px[10] = px14<10>(s); px[11] = px14<11>(s);
        px[12] = px14<12>(s); px[13] = px14<13>(s); px[14] = px14<14>(s); px[15] = px14<15>(s);
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) { px[2 * k] = d[k] & 0xFFFFu; px[2 * k + 1] = d[k] >> 16; }
    }
}

// ---- widths that are a multiple of 8 but not of 16 (1736: the 3x crop of most APS-C bodies; 1880: the 5D2).  Rows then start
// alternately at a 16-pixel group and in the middle of one: a 14-bit group begins 14 bytes into the stream words of the pair it
// shares (dword-aligned address minus 2), and the row's last group has eight pixels in the frame.  Everything else -- item =
// 16 pixels x 2 rows, strips of 8 pixels, vector stores -- is the w % 16 == 0 path.  (Until round 3 such widths took the any-geometry
// path, a load per pixel: cs5x5 9.6 us per 1736x976 frame where 1728 takes 3.5.)
struct Half8 {                     // where one row of an item comes from
    uint32_t at;                   // byte offset in the frame of the first dword to load
    bool mis;                      // packed stream: the group starts in the upper half of that dword
    bool shift8;                   // the window lies 8 pixels further left than the item (the last row's half group): take pixels 8..15
};
// x: a multiple of 16, or beyond the frame; yy: a row of the frame.  whole: all 16 pixels are loaded (of a right halo only the first
// 12 bytes, which lie inside the frame wherever the group starts)
template <bool PACKED>
__device__ __forceinline__ Half8 half8_row(int w, int h, int x, int yy, bool whole)
{
    Half8 r;
    x = x < 0 ? 0 : (x >= w ? w - 16 : x);               // (beyond the frame: any group of the row will do)
    r.shift8 = whole && (yy == h - 1) && (x == w - 8);   // its 16 pixels would run past the end of the frame
    if (r.shift8) x -= 8;
    const uint32_t px = (uint32_t)yy * (uint32_t)w + (uint32_t)x;      // a multiple of 8
    const uint32_t byte = PACKED ? (px >> 3) * 14u : px * 2u;
    r.mis = PACKED && (byte & 2u);
    r.at = PACKED ? (byte & ~3u) : byte;
    return r;
}
// t0 / t1: the 16 stream bits behind the seven dwords of each row (a misaligned group ends there)
template <bool PACKED, int N_FULL>
__device__ __forceinline__ void issue_item_half8(uint32_t (&r0)[Words<PACKED>::N], uint32_t (&r1)[Words<PACKED>::N], uint32_t &t0, uint32_t &t1,
                                                 const uint8_t *frame, int w, int h, int tx0, int ty0, int tid)
{
    constexpr int N = Words<PACKED>::N;
    const bool edge = tid >= N_FULL;
    const int pr = edge ? tid - N_FULL : tid >> 3;
    const int y = ty0 - 2 * HC + 2 * pr;
    const int x_main = edge ? tx0 - 16 : tx0 + 16 * (tid & 7), x_right = tx0 + 2 * TCW;
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const int yy = clampi(y + rr, 0, h - 1);
        const Half8 m = half8_row<PACKED>(w, h, x_main, yy, true), r = half8_row<PACKED>(w, h, x_right, yy, false);
        const uint32_t *sm_ = (const uint32_t *)(frame + m.at);
        const uint32_t *sr_ = (const uint32_t *)(frame + r.at);
        uint32_t (&d)[N] = rr ? r1 : r0;
        // edge items: the right halo's four pixels are the first 56 bits of its group -- dwords 0, 1 and, when the group is
        // misaligned, the lower half of dword 2 --, the left halo's the last 56 of its own (dwords 5, 6 and the tail)
#pragma unroll
        for (int i = 0; i < N; i++) d[i] = ((edge && i < (PACKED ? 3 : 2)) ? sr_ : sm_)[i];
        if (PACKED) (rr ? t1 : t0) = *(const uint16_t *)((const uint8_t *)sm_ + (m.mis ? 28 : 24));
    }
}
// the seven MSB-first stream words of a group from seven dwords + tail; nxt_sel / own_sel: v_perm selectors of a misaligned / aligned group
__device__ __forceinline__ uint32_t half8_word(uint32_t d, uint32_t nxt, bool mis)
{
    return __builtin_amdgcn_perm(nxt, d, mis ? 0x03020504u : 0x01000302u);      // (d & 0xFFFF0000) | (nxt & 0xFFFF)  :  halves of d swapped
}
template <bool PACKED>
__device__ __forceinline__ void unpack16_half8(const uint32_t (&d)[Words<PACKED>::N], uint32_t tail, bool edge, Half8 m, Half8 r, uint32_t (&px)[16])
{
    if constexpr (PACKED) {
        uint32_t s[7];
#pragma unroll
        for (int i = 0; i < 7; i++) s[i] = half8_word(d[i], i < 6 ? d[i + 1] : tail, (edge && i < 2) ? r.mis : m.mis);
        px[0] = px14<0>(s);   px[1] = px14<1>(s);   px[2] = px14<2>(s);   px[3] = px14<3>(s);
        px[4] = px14<4>(s);   px[5] = px14<5>(s);   px[6] = px14<6>(s);   px[7] = px14<7>(s);
        px[8] = px14<8>(s);   px[9] = px14<9>(s);   px[10] = px14<10>(s); px[11] = px14<11>(s);
        px[12] = px14<12>(s); px[13] = px14<13>(s); px[14] = px14<14>(s); px[15] = px14<15>(s);
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) { px[2 * k] = d[k] & 0xFFFFu; px[2 * k + 1] = d[k] >> 16; }
    }
    // the frame's last row: a window that was moved 8 pixels to the left (one item per frame; the left halo never is such a group)
    const bool sh = m.shift8 && !edge;
    if (__any(sh)) {
#pragma unroll
        for (int k = 0; k < 8; k++) px[k] = sh ? px[k + 8] : px[k];
    }
}

// any geometry: one pixel with clamped coordinates
template <bool PACKED>
__device__ __forceinline__ uint32_t fetch_clamped(const uint8_t *frame, int w, int h, int x, int y)
{
    x = clampi(x, 0, w - 1);
    y = clampi(y, 0, h - 1);
    const uint32_t i = (uint32_t)y * (uint32_t)w + (uint32_t)x;
    if (PACKED) {
        const uint16_t *s = (const uint16_t *)frame;
        const uint32_t bit = i * 14u;                    // < 2^28 pixels per frame (launcher): fits
        const uint32_t two = ((uint32_t)s[bit >> 4] << 16) | s[(bit >> 4) + 1];
        return (two >> (32 - 14 - (bit & 15))) & 0x3FFFu;
    }
    return ((const uint16_t *)frame)[i];
}

// slow path (w % 16 != 0): fill the two pixel rows of an item pixel by pixel
template <bool PACKED>
__device__ __forceinline__ void fetch_rows(const uint8_t *frame, int w, int h, int x, int y, bool edge, int tx0,
                                           uint32_t (&p0)[16], uint32_t (&p1)[16])
{
#pragma unroll 1
    for (int k = 0; k < 16; k++) {
        // edge items: px 0..3 = right halo, px 12..15 = left halo (same layout as the fast path)
        const int xx = edge ? (k < 4 ? tx0 + 2 * TCW + k : tx0 - 16 + k) : x + k;
        p0[k] = fetch_clamped<PACKED>(frame, w, h, xx, y);
        p1[k] = fetch_clamped<PACKED>(frame, w, h, xx, y + 1);
    }
}

// EV triples of NCELL (even) cells into the planes: plane row r, first plane column col0
template <int NCELL, class SM>
__device__ __forceinline__ void store_cells(SM &sm, int black, bool slow, int r, int col0, const uint32_t *p0, const uint32_t *p1)
{
    const bool row_in = r >= HC && r < HC + SM::TCH;
    if (NCELL == 8 && !slow) {                           // wave-uniform; against pairs: cs2x2 -2.5 %, cs5x5 -2 %; all eight at once: no better
#pragma unroll
        for (int c = 0; c < 8; c += 4) {
            int ge[4], dr[4], db[4];
            cell_multi_ev_fast<4, SM::SPREAD>(p0 + 2 * c, p1 + 2 * c, black, sm.t16, ge, dr, db);
            *(int4 *)&sm.dr[r][col0 + c] = make_int4(dr[0], dr[1], dr[2], dr[3]);
            *(int4 *)&sm.db[r][col0 + c] = make_int4(db[0], db[1], db[2], db[3]);
            if (row_in) *(int4 *)&sm.ge[r - HC][col0 + c - HC] = make_int4(ge[0], ge[1], ge[2], ge[3]);
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < NCELL; c += 2) {
        int ge[2], dr[2], db[2];
        cell_pair_ev<SM::SPREAD>(p0 + 2 * c, p1 + 2 * c, black, sm.t16, slow, ge, dr, db);
        *(int2 *)&sm.dr[r][col0 + c] = make_int2(dr[0], dr[1]);
        *(int2 *)&sm.db[r][col0 + c] = make_int2(db[0], db[1]);
        const int ci = col0 + c - HC;
        if (row_in && ci >= 0 && ci < TCW) *(int2 *)&sm.ge[r - HC][ci] = make_int2(ge[0], ge[1]);
    }
}

template <class SM>
__device__ __forceinline__ void store_raw(SM &sm, int row, int g, const uint32_t (&p)[16])
{
    uint4 *o = (uint4 *)&sm.raw[row][16 * g];
    o[0] = make_uint4(p[0] | (p[1] << 16), p[2] | (p[3] << 16), p[4] | (p[5] << 16), p[6] | (p[7] << 16));
    o[1] = make_uint4(p[8] | (p[9] << 16), p[10] | (p[11] << 16), p[12] | (p[13] << 16), p[14] | (p[15] << 16));
}

// pixels of one item -> planes (+ interior raw)
template <int METHOD, class SM>
__device__ __forceinline__ void emit_item(SM &sm, int black, bool slow, int pr, int g, bool edge, const uint32_t (&p0)[16],
                                          const uint32_t (&p1)[16])
{
    if (!edge) {
        if (METHOD != 0) store_cells<8, SM>(sm, black, slow, pr, HC + 8 * g, p0, p1);
        if (pr >= HC && pr < HC + SM::TCH) {
            store_raw(sm, 2 * (pr - HC), g, p0);
            store_raw(sm, 2 * (pr - HC) + 1, g, p1);
        }
    } else if (METHOD != 0) {
        store_cells<2, SM>(sm, black, slow, pr, 0, p0 + 12, p1 + 12);    // left halo  (px 12..15)
        store_cells<2, SM>(sm, black, slow, pr, HC + TCW, p0, p1);       // right halo (px 0..3)
    }
}

// ---------------------------------------------------------------- pixel map
// A tile's repaired cells arrive as records {cell, R | G1 << 16, G2 | B << 16} (k_pixfix_cells): the EV triple of a record is
// computed by a lane that has no loader item (the fourth wave, while the other three convert the tile), and goes into the
// planes -- with the four pixels into the interior raw tile -- once the loader's stores are behind a barrier.
struct PatchCell { int i, j, ge, dr, db; uint32_t top, bot; };      // i < 0: nothing to store

template <int METHOD, bool PACKED, class SM>
__device__ __forceinline__ PatchCell patch_cell(const SM &sm, const FrameArgs &a, int4 rec, int tx0, int ty0)
{
    PatchCell c;
    c.i = -1; c.j = 0; c.ge = c.dr = c.db = 0;
    c.top = (uint32_t)rec.y; c.bot = (uint32_t)rec.z;
    const bool have = rec.x >= 0;
    const int cx = rec.x & 0xFFFF, cy = rec.x >> 16;
    const int i = cx - (tx0 / 2 - HC), j = cy - (ty0 / 2 - HC);
    if (have && i >= 0 && i < PW && j >= 0 && j < SM::PH) { c.i = i; c.j = j; }
    if (METHOD != 0) {
        const int px[4] = { (int)(c.top & 0xFFFFu), (int)(c.top >> 16), (int)(c.bot & 0xFFFFu), (int)(c.bot >> 16) };
        bool odd = false;
        if (c.i >= 0) {
#pragma unroll
            for (int q = 0; q < 4; q++) odd = odd || (unsigned)(px[q] - a.black - 1) >= 16383u;
        }
        cell_ev<SM::SPREAD>(px[0], px[1], px[2], px[3], a.black, sm.t16, __any(odd), c.ge, c.dr, c.db);
    }
    return c;
}

template <int METHOD, class SM>
__device__ __forceinline__ void patch_store(SM &sm, const PatchCell &c)
{
    if (c.i < 0) return;
    if (METHOD != 0) {
        sm.dr[c.j][c.i] = c.dr;
        sm.db[c.j][c.i] = c.db;
    }
    const int ii = c.i - HC, jj = c.j - HC;
    if (ii >= 0 && ii < TCW && jj >= 0 && jj < SM::TCH) {
        if (METHOD != 0) sm.ge[jj][ii] = c.ge;
        *(uint32_t *)&sm.raw[2 * jj][2 * ii] = c.top;
        *(uint32_t *)&sm.raw[2 * jj + 1][2 * ii] = c.bot;
    }
}

// ---------------------------------------------------------------- medians
// 5x5: strip of 4 outputs from 8 sorted columns
__device__ __forceinline__ void strip_median25(const int (*plane)[PW], int row_top, int col_left, int (&med)[STRIP])
{
    int col[8][5];
#pragma unroll
    for (int r = 0; r < 5; r++) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int4 v = *(const int4 *)&plane[row_top + r][col_left + 4 * q];
            col[4 * q + 0][r] = v.x; col[4 * q + 1][r] = v.y; col[4 * q + 2][r] = v.z; col[4 * q + 3][r] = v.w;
        }
    }
#pragma unroll
    for (int c = 0; c < 8; c++) sort5(col[c]);
    int pr[4][10];
#pragma unroll
    for (int p = 0; p < 4; p++) mlv_merge55(col[2 * p], col[2 * p + 1], pr[p]);
    int qd[3][6];
#pragma unroll
    for (int q = 0; q < 3; q++) mlv_quad_mid6(pr[q], pr[q + 1], qd[q]);
#pragma unroll
    for (int c = 0; c < STRIP; c++) {
        const int x = c + 2;
        int o[1];
        if (x % 2 == 0) mlv_final6of11(qd[(x - 2) / 2], col[x + 2], o);
        else            mlv_final6of11(qd[(x - 1) / 2], col[x - 2], o);
        med[c] = o[0];
    }
}

// 5x5 on both colour-difference planes at once: (dr, db) of a cell, taken relative to a reference from the strip's own centre row
// (saturating subtract) and saturated to a pair of 16-bit lanes (v_cvt_pk_i16_i32); the same sorted-column / merge /
// rank-window networks then run on packed min/max.  Both saturations are monotone, so the packed median is the
// saturated, shifted true median: exact unless it sits ON a 16-bit bound, which the caller treats as "unknown"
// (returns true) and settles with the 32-bit networks.  The local reference keeps real footage (R and B one or two
// EV below G before white balance) inside the 16-bit window; only strips across a hard colour edge fall back.
__device__ __forceinline__ bool strip_median25_packed(const int (*pr_)[PW], const int (*pb_)[PW], int row_top, int col_left,
                                                      int (&mr)[STRIP], int (&mb)[STRIP])
{
    // reference = median of three cells of the centre row (columns 2, 4, 5): in noisy shadows a single cell is often more than
    // 1 EV away from the median of its neighbourhood (EVs of small integers), which sent the whole wave to the 32-bit networks
    const int4 cr0 = *(const int4 *)&pr_[row_top + 2][col_left], cr1 = *(const int4 *)&pr_[row_top + 2][col_left + 4];
    const int4 cb0 = *(const int4 *)&pb_[row_top + 2][col_left], cb1 = *(const int4 *)&pb_[row_top + 2][col_left + 4];
    const int ref_r = med3i(cr0.z, cr1.x, cr1.y), ref_b = med3i(cb0.z, cb1.x, cb1.y);
    auto pack = [&](int r, int b) {
        return __builtin_amdgcn_cvt_pk_i16(__builtin_elementwise_sub_sat(r, ref_r), __builtin_elementwise_sub_sat(b, ref_b));
    };
    mlv_pk16 col[8][5];
#pragma unroll
    for (int r = 0; r < 5; r++) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int4 a = *(const int4 *)&pr_[row_top + r][col_left + 4 * q];
            const int4 b = *(const int4 *)&pb_[row_top + r][col_left + 4 * q];
            col[4 * q + 0][r] = pack(a.x, b.x);
            col[4 * q + 1][r] = pack(a.y, b.y);
            col[4 * q + 2][r] = pack(a.z, b.z);
            col[4 * q + 3][r] = pack(a.w, b.w);
        }
    }
    mlv_pk16 srt[8][5];
#pragma unroll
    for (int c = 0; c < 8; c++) mlv_sort5(col[c], srt[c]);
    mlv_pk16 pr[4][10];
#pragma unroll
    for (int p = 0; p < 4; p++) mlv_merge55(srt[2 * p], srt[2 * p + 1], pr[p]);
    mlv_pk16 qd[3][6];
#pragma unroll
    for (int q = 0; q < 3; q++) mlv_quad_mid6(pr[q], pr[q + 1], qd[q]);
    bool unknown = false;
#pragma unroll
    for (int c = 0; c < STRIP; c++) {
        const int x = c + 2;
        mlv_pk16 o[1];
        if (x % 2 == 0) mlv_final6of11(qd[(x - 2) / 2], srt[x + 2], o);
        else            mlv_final6of11(qd[(x - 1) / 2], srt[x - 2], o);
        const int vr = (int)o[0].x, vb = (int)o[0].y;
        unknown |= (unsigned)(vr + 32767) >= 65534u || (unsigned)(vb + 32767) >= 65534u;   // -32768 or 32767
        mr[c] = vr + ref_r;
        mb[c] = vb + ref_b;
    }
    return unknown;
}

// ---------------------------------------------------------------- 5x5 with neighbour sharing
// A strip's window is 8 columns: its own group of four and the four of the strip to its right.  Two neighbouring strips
// would each sort, merge and rank the same four columns; instead every lane does that for ONE group (mlv::ChainGroup: four
// sorted columns, two pair lists, one rank window) and takes, from the lane to its right (v_mov_b32 wave_shl:1: the 16 strips of a
// tile row sit in 16 consecutive lanes), the two sorted columns, the pair list and the rank window it needs of that lane's group
// -- 26 values instead of 4 column sorts, 2 merges and a rank window, and half of the packing.  The group to the right of a
// row's last strip (plane columns 64..67, the halo) is computed by a lane that has no strip (the 5x5 tile has 15 rows: lanes
// 48..62 of the fourth wave) and handed over through LDS.
//
// Exactness.  Lanes pack relative to their OWN reference r (as before: saturating subtract, saturating 16-bit pack), the
// neighbour's values arrive relative to ITS reference r' and are rebased with a saturating add of D = sat16(r' - r).  For a
// neighbour cell x that saturated at the first stage the rebased value is not sat16(x - r), but it lies in the band of width |D|
// at the same end of the 16-bit range as sat16(x - r) does (x - r >= 32767 + D and sat(32767 + D) >= 32767 - |D|; mirrored at
// the low end).  So for every threshold c in [-32768 + |D|, 32766 - |D|] each window value is <= c exactly when its true
// relative value is: the 13th smallest of the 25 is exact whenever it comes out strictly inside (-32768 + |D|, 32767 - |D|).
// Anything else (that includes a saturated D) is "unknown" and settled by the 32-bit networks.  |D| = 0 gives the old rule.
struct ChainGroup {
    mlv_pk16 s[4][5];        // sorted columns
    mlv_pk16 p0[10], p1[10]; // columns 0+1 and 2+3 merged
    mlv_pk16 q[6];           // ranks 8..13 of the 20
    int ref_r, ref_b;
};

// Noisy shadows: the colour difference of a cell is, at a signal of a few DN, an EV or more away from the median of its
// neighbourhood, and so are many of the lanes' references -- from the window's median (first reason to be uncertain) and from
// each other (|D| eats the window: second reason).  A row's 16 lanes then agree on ONE reference, the median of five of
// theirs (lanes 1, 4, 8, 11, 14 of the row: v_mov_b32 row_share), provided at least ten of the sixteen lie within 1 EV of it;
// rows across a colour edge do not and keep their own.  Any reference gives exact medians (the criterion of chain_finish holds
// for whatever the lanes subtracted); this only decides how many strips are certain: underexposed footage 19 % -> 5 % uncertain.
// Costs 16 instructions per plane, so it runs only after a tile that had uncertain strips (k_frame: `robust`).
template <int N>
__device__ __forceinline__ int dpp_row_share(int v) { return __builtin_amdgcn_mov_dpp(v, 0x150 + N, 0xf, 0xf, true); }

__device__ __forceinline__ int robust_ref(int own)
{
    const int a = dpp_row_share<1>(own), b = dpp_row_share<4>(own), c = dpp_row_share<8>(own), d = dpp_row_share<11>(own),
              e = dpp_row_share<14>(own);
    const int shared = med3i(e, max(min(a, b), min(c, d)), min(max(a, b), max(c, d)));       // median of five
    const unsigned long long agree = __ballot((unsigned)(own - shared + 32767) < 65535u);
    unsigned long long use = 0;
#pragma unroll
    for (int r = 0; r < 4; r++)
        if (__builtin_popcount((unsigned)(agree >> (16 * r)) & 0xFFFFu) >= 10) use |= 0xFFFFull << (16 * r);
    int ref;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(ref) : "v"(own), "v"(shared), "s"(use));
    return ref;
}

__device__ __forceinline__ void chain_group(const int (*pr_)[PW], const int (*pb_)[PW], int row_top, int col_left, bool robust,
                                            ChainGroup &g)
{
    // reference = median of three cells of the group's centre row (a single cell is, in noisy shadows, often more than 1 EV
    // away from the median of its neighbourhood)
    const int4 cr = *(const int4 *)&pr_[row_top + 2][col_left], cb = *(const int4 *)&pb_[row_top + 2][col_left];
    g.ref_r = med3i(cr.y, cr.z, cr.w);
    g.ref_b = med3i(cb.y, cb.z, cb.w);
    if (robust) {                                          // wave-uniform
        g.ref_r = robust_ref(g.ref_r);
        g.ref_b = robust_ref(g.ref_b);
    }
    auto pack = [&](int r, int b) {
        return __builtin_amdgcn_cvt_pk_i16(__builtin_elementwise_sub_sat(r, g.ref_r), __builtin_elementwise_sub_sat(b, g.ref_b));
    };
    mlv_pk16 col[4][5];
#pragma unroll
    for (int r = 0; r < 5; r++) {
        const int4 a = r == 2 ? cr : *(const int4 *)&pr_[row_top + r][col_left];
        const int4 b = r == 2 ? cb : *(const int4 *)&pb_[row_top + r][col_left];
        col[0][r] = pack(a.x, b.x); col[1][r] = pack(a.y, b.y); col[2][r] = pack(a.z, b.z); col[3][r] = pack(a.w, b.w);
    }
#pragma unroll
    for (int c = 0; c < 4; c++) mlv_sort5(col[c], g.s[c]);
    mlv_merge55(g.s[0], g.s[1], g.p0);
    mlv_merge55(g.s[2], g.s[3], g.p1);
}
__device__ __forceinline__ void chain_group_window(ChainGroup &g) { mlv_quad_mid6(g.p0, g.p1, g.q); }

// what a strip needs of the group to its right
struct ChainNext {
    mlv_pk16 s0[5], s2[5], p0[10], q[6];
    int ref_r, ref_b;
};
// wave_shl:1 with bound_ctrl: lane i reads lane i + 1, the wave's last lane reads 0; no "old" operand, so no move to set one up
__device__ __forceinline__ int dpp_next_i(int v) { return __builtin_amdgcn_mov_dpp(v, 0x130, 0xf, 0xf, true); }
__device__ __forceinline__ mlv_pk16 dpp_next(mlv_pk16 v) { return __builtin_bit_cast(mlv_pk16, dpp_next_i(__builtin_bit_cast(int, v))); }

// in two parts, so that the LDS reads of a row's last strip (chain_collect_*) have arithmetic to hide behind: the sorted
// columns, the pair list and the references first, then -- after the lane's own rank window -- the neighbour's rank window
__device__ __forceinline__ void chain_fetch_lists(const ChainGroup &g, ChainNext &n)
{
#pragma unroll
    for (int i = 0; i < 5; i++) { n.s0[i] = dpp_next(g.s[0][i]); n.s2[i] = dpp_next(g.s[2][i]); }
#pragma unroll
    for (int i = 0; i < 10; i++) n.p0[i] = dpp_next(g.p0[i]);
    n.ref_r = dpp_next_i(g.ref_r);
    n.ref_b = dpp_next_i(g.ref_b);
}
__device__ __forceinline__ void chain_fetch_window(const ChainGroup &g, ChainNext &n)
{
#pragma unroll
    for (int i = 0; i < 6; i++) n.q[i] = dpp_next(g.q[i]);
}

// the halo group's lane -> LDS -> the row's last strip
__device__ __forceinline__ void chain_publish(const ChainGroup &g, uint32_t *x)
{
    uint4 *o = (uint4 *)x;
    auto u = [](mlv_pk16 v) { return __builtin_bit_cast(uint32_t, v); };
    o[0] = make_uint4((uint32_t)g.ref_r, (uint32_t)g.ref_b, u(g.s[0][0]), u(g.s[0][1]));
    o[1] = make_uint4(u(g.s[0][2]), u(g.s[0][3]), u(g.s[0][4]), u(g.s[2][0]));
    o[2] = make_uint4(u(g.s[2][1]), u(g.s[2][2]), u(g.s[2][3]), u(g.s[2][4]));
    o[3] = make_uint4(u(g.p0[0]), u(g.p0[1]), u(g.p0[2]), u(g.p0[3]));
    o[4] = make_uint4(u(g.p0[4]), u(g.p0[5]), u(g.p0[6]), u(g.p0[7]));
    o[5] = make_uint4(u(g.p0[8]), u(g.p0[9]), u(g.q[0]), u(g.q[1]));
    o[6] = make_uint4(u(g.q[2]), u(g.q[3]), u(g.q[4]), u(g.q[5]));
}
__device__ __forceinline__ void chain_collect_lists(const uint32_t *x, ChainNext &n)
{
    const uint4 *o = (const uint4 *)x;
    auto k = [](uint32_t v) { return __builtin_bit_cast(mlv_pk16, v); };
    const uint4 a0 = o[0], a1 = o[1], a2 = o[2], a3 = o[3], a4 = o[4];
    const uint2 a5 = *(const uint2 *)&o[5];
    n.ref_r = (int)a0.x; n.ref_b = (int)a0.y;
    n.s0[0] = k(a0.z); n.s0[1] = k(a0.w); n.s0[2] = k(a1.x); n.s0[3] = k(a1.y); n.s0[4] = k(a1.z);
    n.s2[0] = k(a1.w); n.s2[1] = k(a2.x); n.s2[2] = k(a2.y); n.s2[3] = k(a2.z); n.s2[4] = k(a2.w);
    n.p0[0] = k(a3.x); n.p0[1] = k(a3.y); n.p0[2] = k(a3.z); n.p0[3] = k(a3.w);
    n.p0[4] = k(a4.x); n.p0[5] = k(a4.y); n.p0[6] = k(a4.z); n.p0[7] = k(a4.w);
    n.p0[8] = k(a5.x); n.p0[9] = k(a5.y);
}
__device__ __forceinline__ void chain_collect_window(const uint32_t *x, ChainNext &n)
{
    auto k = [](uint32_t v) { return __builtin_bit_cast(mlv_pk16, v); };
    const uint2 a5 = *(const uint2 *)(x + 22);
    const uint4 a6 = *(const uint4 *)(x + 24);
    n.q[0] = k(a5.x); n.q[1] = k(a5.y); n.q[2] = k(a6.x); n.q[3] = k(a6.y); n.q[4] = k(a6.z); n.q[5] = k(a6.w);
}

// medians of the strip's four cells from its own group and the neighbour's; true = not certain (see above)
__device__ __forceinline__ bool chain_finish(const ChainGroup &g, ChainNext &n, int (&mr)[STRIP], int (&mb)[STRIP])
{
    const int dr_ = __builtin_elementwise_sub_sat(n.ref_r, g.ref_r), db_ = __builtin_elementwise_sub_sat(n.ref_b, g.ref_b);
    const mlv_pk16 D = __builtin_amdgcn_cvt_pk_i16(dr_, db_);
#pragma unroll
    for (int i = 0; i < 5; i++) { n.s0[i] = __builtin_elementwise_add_sat(n.s0[i], D); n.s2[i] = __builtin_elementwise_add_sat(n.s2[i], D); }
#pragma unroll
    for (int i = 0; i < 10; i++) n.p0[i] = __builtin_elementwise_add_sat(n.p0[i], D);
    mlv_pk16 q1[6];
    mlv_quad_mid6(g.p1, n.p0, q1);
    mlv_pk16 o[STRIP][1];
    mlv_final6of11(g.q, n.s0, o[0]);          // window columns 0..3 | 4
    mlv_final6of11(q1, g.s[1], o[1]);         // 2..5 | 1
    mlv_final6of11(q1, n.s2, o[2]);           // 2..5 | 6
#pragma unroll
    for (int i = 0; i < 6; i++) n.q[i] = __builtin_elementwise_add_sat(n.q[i], D);     // (the neighbour's rank window is the last to arrive)
    mlv_final6of11(n.q, g.s[3], o[3]);        // 4..7 | 3
    // certain when strictly inside (-32768 + |D|, 32767 - |D|): v - lo <= hi - lo as unsigned, lo = -32767 + |D|, hi = 32766 - |D|
    const int ar = (int)min((unsigned)wabs(dr_), 32767u), ab = (int)min((unsigned)wabs(db_), 32767u);
    const int lo_r = ar - 32767, span_r = 65533 - 2 * ar, lo_b = ab - 32767, span_b = 65533 - 2 * ab;
    bool unknown = ar >= 32767 || ab >= 32767;           // the references themselves are more than the 16-bit range apart
#pragma unroll
    for (int c = 0; c < STRIP; c++) {
        const int vr = (int)o[c][0].x, vb = (int)o[c][0].y;
        unknown |= (unsigned)(vr - lo_r) > (unsigned)span_r;
        unknown |= (unsigned)(vb - lo_b) > (unsigned)span_b;
        mr[c] = vr + g.ref_r;
        mb[c] = vb + g.ref_b;
    }
    return unknown;
}


// ---------------------------------------------------------------- 5x5 with neighbour sharing, 32-bit
// The same chain on one plane of plain int32 values: no reference, no rebasing, nothing uncertain.  For the tiles
// that skip the packed attempt (every strip would go through the stand-alone 32-bit networks at 348 operations per strip and
// plane; here a lane does 4 column sorts, 2 merges, 2 rank windows and 4 selections = 212, plus 26 moves): underexposed footage
// 11.8 -> 11.3 us per frame, colour patches 15.6 -> 14.6, the benchmark's frames unchanged (A/B in one run).
struct Chain32 {
    int s[4][5];
    int p0[10], p1[10];
    int q[6];
};
struct Next32 { int s0[5], s2[5], p0[10], q[6]; };

__device__ __forceinline__ void chain32_group(const int (*pl)[PW], int row_top, int col_left, Chain32 &g)
{
#pragma unroll
    for (int r = 0; r < 5; r++) {
        const int4 a = *(const int4 *)&pl[row_top + r][col_left];
        g.s[0][r] = a.x; g.s[1][r] = a.y; g.s[2][r] = a.z; g.s[3][r] = a.w;
    }
#pragma unroll
    for (int c = 0; c < 4; c++) sort5(g.s[c]);
    mlv_merge55(g.s[0], g.s[1], g.p0);
    mlv_merge55(g.s[2], g.s[3], g.p1);
    mlv_quad_mid6(g.p0, g.p1, g.q);
}
__device__ __forceinline__ void chain32_publish(const Chain32 &g, uint32_t *x)
{
    uint4 *o = (uint4 *)x;
    auto u = [](int v) { return (uint32_t)v; };
    o[0] = make_uint4(u(g.s[0][0]), u(g.s[0][1]), u(g.s[0][2]), u(g.s[0][3]));
    o[1] = make_uint4(u(g.s[0][4]), u(g.s[2][0]), u(g.s[2][1]), u(g.s[2][2]));
    o[2] = make_uint4(u(g.s[2][3]), u(g.s[2][4]), u(g.p0[0]), u(g.p0[1]));
    o[3] = make_uint4(u(g.p0[2]), u(g.p0[3]), u(g.p0[4]), u(g.p0[5]));
    o[4] = make_uint4(u(g.p0[6]), u(g.p0[7]), u(g.p0[8]), u(g.p0[9]));
    o[5] = make_uint4(u(g.q[0]), u(g.q[1]), u(g.q[2]), u(g.q[3]));
    *(uint2 *)&o[6] = make_uint2(u(g.q[4]), u(g.q[5]));
}
__device__ __forceinline__ void chain32_collect(const uint32_t *x, Next32 &n)
{
    const uint4 *o = (const uint4 *)x;
    const uint4 a0 = o[0], a1 = o[1], a2 = o[2], a3 = o[3], a4 = o[4], a5 = o[5];
    const uint2 a6 = *(const uint2 *)&o[6];
    n.s0[0] = (int)a0.x; n.s0[1] = (int)a0.y; n.s0[2] = (int)a0.z; n.s0[3] = (int)a0.w; n.s0[4] = (int)a1.x;
    n.s2[0] = (int)a1.y; n.s2[1] = (int)a1.z; n.s2[2] = (int)a1.w; n.s2[3] = (int)a2.x; n.s2[4] = (int)a2.y;
    n.p0[0] = (int)a2.z; n.p0[1] = (int)a2.w; n.p0[2] = (int)a3.x; n.p0[3] = (int)a3.y; n.p0[4] = (int)a3.z; n.p0[5] = (int)a3.w;
    n.p0[6] = (int)a4.x; n.p0[7] = (int)a4.y; n.p0[8] = (int)a4.z; n.p0[9] = (int)a4.w;
    n.q[0] = (int)a5.x; n.q[1] = (int)a5.y; n.q[2] = (int)a5.z; n.q[3] = (int)a5.w; n.q[4] = (int)a6.x; n.q[5] = (int)a6.y;
}
__device__ __forceinline__ void chain32_fetch(const Chain32 &g, Next32 &n)
{
#pragma unroll
    for (int i = 0; i < 5; i++) { n.s0[i] = dpp_next_i(g.s[0][i]); n.s2[i] = dpp_next_i(g.s[2][i]); }
#pragma unroll
    for (int i = 0; i < 10; i++) n.p0[i] = dpp_next_i(g.p0[i]);
#pragma unroll
    for (int i = 0; i < 6; i++) n.q[i] = dpp_next_i(g.q[i]);
}
__device__ __forceinline__ void chain32_finish(const Chain32 &g, const Next32 &n, int (&med)[STRIP])
{
    int q1[6], o[1];
    mlv_quad_mid6(g.p1, n.p0, q1);
    mlv_final6of11(g.q, n.s0, o);  med[0] = o[0];      // window columns 0..3 | 4
    mlv_final6of11(q1, g.s[1], o); med[1] = o[0];      // 2..5 | 1
    mlv_final6of11(q1, n.s2, o);   med[2] = o[0];      // 2..5 | 6
    mlv_final6of11(n.q, g.s[3], o); med[3] = o[0];     // 4..7 | 3
}

// 3x3: sorted columns of 3, classic max-of-mins / med-of-meds / min-of-maxes
__device__ __forceinline__ void strip_median9(const int (*plane)[PW], int row_top, int col_left, int (&med)[STRIP])
{
    int lo[STRIP + 2], mi[STRIP + 2], hi[STRIP + 2];
#pragma unroll
    for (int c = 0; c < STRIP + 2; c++) {
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
template <bool FAST>
__device__ __forceinline__ uint32_t stripe_px(uint32_t p, int coef, int black16, int white16)
{
    if (FAST) {
        // coef = 65536 + d with |d| < 2^15: ((p-black)*coef) >> 16 == (p-black) + (((p-black)*d) >> 16)
        const int a = (int)p - black16;
        const int v = (int)p + (__mul24(a, coef - 65536) >> 16);
        return (a > 64) ? (uint32_t)min(v, white16) : p;
    }
    if (coef == 0 || (int)p <= black16 + 64) return p;
    const long long num = (long long)((int)p - black16) * coef + ((long long)black16 << 16);   // value * 65536
    if (((long long)white16 << 16) < num) return (uint32_t)white16;
    return (uint32_t)(int)(num / 65536) & 0xFFFFu;
}

template <bool FAST>
__device__ __forceinline__ void stripe_strip(uint32_t (&top)[STRIP], uint32_t (&bot)[STRIP], const int (&coef)[8], int black16, int white16)
{
#pragma unroll
    for (int c = 0; c < STRIP; c++) {
        const int p0 = (2 * c) & 7, p1 = (2 * c + 1) & 7;
        top[c] = stripe_px<FAST>(top[c] & 0xFFFFu, coef[p0], black16, white16) |
                 (stripe_px<FAST>(top[c] >> 16, coef[p1], black16, white16) << 16);
        bot[c] = stripe_px<FAST>(bot[c] & 0xFFFFu, coef[p0], black16, white16) |
                 (stripe_px<FAST>(bot[c] >> 16, coef[p1], black16, white16) << 16);
    }
}

// The same on both pixels of a dword with 16-bit lanes (14-bit input, |coef - 65536| < 2^15, white > black + 64, checked
// by the launcher): a = p - black and the "a > 64" mask as packed ops, the two 24-bit products through SDWA operands, their
// upper halves gathered by one v_perm_b32.  Where a <= 64 the masked correction is 0 and min(p, white) = p.
__device__ __forceinline__ uint32_t stripe_pair(uint32_t x, int d0, int d1, uint32_t black_pk, uint32_t white_pk)
{
    typedef unsigned short upk16 __attribute__((ext_vector_type(2)));
    const mlv_pk16 a = __builtin_bit_cast(mlv_pk16, x) - __builtin_bit_cast(mlv_pk16, black_pk);
    const mlv_pk16 c64 = { 64, 64 }, s15 = { 15, 15 };
    const mlv_pk16 m = (c64 - a) >> s15;                                                     // -1 where a > 64
    const int p0 = __mul24((int)a.x, d0), p1 = __mul24((int)a.y, d1);                         // |.| < 2^29
    const uint32_t delta = __builtin_amdgcn_perm((uint32_t)p1, (uint32_t)p0, 0x07060302u);    // {p1 >> 16, p0 >> 16}
    const upk16 v = __builtin_bit_cast(upk16, x) + __builtin_bit_cast(upk16, delta & __builtin_bit_cast(uint32_t, m));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(v, __builtin_bit_cast(upk16, white_pk)));
}

__device__ __forceinline__ void stripe_strip_pk(uint32_t (&top)[STRIP], uint32_t (&bot)[STRIP], const int (&coef)[8], int black16, int white16)
{
    const uint32_t black_pk = (uint32_t)black16 * 0x10001u, white_pk = (uint32_t)white16 * 0x10001u;
#pragma unroll
    for (int c = 0; c < STRIP; c++) {
        const int d0 = coef[(2 * c) & 7] - 65536, d1 = coef[(2 * c + 1) & 7] - 65536;
        if ((d0 | d1) == 0) {
            // Unit gain on both columns -- always the case for column phases 0 and 1, which stripes.c:236-237 pins to 1.0 --
            // leaves min(p, white) (for p <= black + 64 < white that is p itself): one packed op instead of nine
            typedef unsigned short upk16 __attribute__((ext_vector_type(2)));
            top[c] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(upk16, top[c]), __builtin_bit_cast(upk16, white_pk)));
            bot[c] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(upk16, bot[c]), __builtin_bit_cast(upk16, white_pk)));
            continue;
        }
        top[c] = stripe_pair(top[c], d0, d1, black_pk, white_pk);
        bot[c] = stripe_pair(bot[c], d0, d1, black_pk, white_pk);
    }
}

template <int METHOD, bool PACKED, int VEC, bool SPREAD>
__global__ __launch_bounds__(256, 4) void k_frame(const FrameArgs a)
{
    using Smem = SmemT<SPREAD, tile_rows_of(METHOD)>;
    constexpr int TCH = Smem::TCH, N_FULL = Smem::N_FULL, N_ITEMS = Smem::N_ITEMS;
    constexpr bool CHAIN = Smem::CHAIN;
    __shared__ Smem sm;                                  // static: a compile-time LDS base (a dynamic one costs an add per access)

    if (METHOD != 0) {
        if (SPREAD) {
            for (int i = threadIdx.x; i < MLV_T16_N; i += blockDim.x) sm.t16[i + (i >> 7)] = a.t16[i];
        } else {
            const uint4 *src = (const uint4 *)a.t16;
            uint4 *dstl = (uint4 *)sm.t16;
            for (int i = threadIdx.x; i < MLV_T16_N * 2 / 16; i += blockDim.x) dstl[i] = src[i];
        }
    }
    // Which tiles of a frame have pixel-map entries: one bit per tile in LDS.  Reading the tile's list bounds from HBM in
    // every iteration made each wave wait for ALL its outstanding loads (the prefetch of the next tile included) before the
    // median phase; now only the few tiles that are touched fetch their bounds.
    const bool pmap_ok = a.tiles_x * a.tiles_y <= PMAP_WORDS * 32;
    if (a.patch && pmap_ok) {
        for (int i = threadIdx.x; i < PMAP_WORDS; i += blockDim.x) sm.has_patch[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < a.tiles_x * a.tiles_y; i += blockDim.x)
            if (a.tile_off[i + 1] != a.tile_off[i]) atomicOr(&sm.has_patch[i >> 5], 1u << (i & 31));
    }

    // Persistent tile walk.  Group g = blockIdx % groups: with 4 x CUs workgroups in the grid the dispatcher puts blocks
    // b, b + CUs, b + 2 CUs, b + 3 CUs on one CU (tools/hwid_probe.hip), so a group is one CU's four residents.  Each group
    // owns a contiguous range of the tile list -- the groups of an XCD (blocks b, b + 8, ... share one) next to each other,
    // so halo re-reads hit that XCD's L2 -- and its members draw tiles from it by ticket: the four workgroups of a CU do not
    // progress at the same pace (the oldest waves issue first), and with a fixed share each the slowest one ended up alone
    // on its CU (measured with s_memrealtime stamps: the first-dispatched quarter of the grid finished at 70 % of the kernel
    // time).  Whatever the placement, every tile is drawn exactly once.
    const int tiles_per_frame = a.tiles_x * a.tiles_y;
    const int total = tiles_per_frame * a.nframes;          // < 2^30 (checked by the launcher)
    const int nx = 8;
    const int grp = blockIdx.x % a.groups;
    const int gpx = (a.groups + nx - 1) / nx;                                    // groups per XCD
    const int grank = (a.groups % nx == 0) ? (grp % nx) * gpx + grp / nx : grp;  // position of the group's range in the tile list
    const int gq = total / a.groups, grem = total - gq * a.groups;
    const int band_start = grank * gq + min(grank, grem);
    const int band_end = band_start + gq + (grank < grem ? 1 : 0);
    const int black16 = (int)(uint16_t)a.black, white16 = (int)(uint16_t)a.white;
    constexpr bool vec = VEC != 0;                       // w % 16 == 0 (1) or w % 16 == 8 (2): dword/vector loads and stores
    constexpr bool half8 = VEC == 2;

    const mlv_i32x4 rs_e2r = table_rsrc(a.e2r, 2, E2R_ENTRIES);
    constexpr int NW = Words<PACKED>::N;
    uint32_t r0[NW], r1[NW];                             // prefetch registers of this thread's item
    uint32_t tl0 = 0, tl1 = 0;                           // half8: the stream's 16 bits behind them
    const int tid = threadIdx.x;
    const int item_row = tid >= N_FULL ? tid - N_FULL : tid >> 3, item_g = tid & 7;

    // median phase: lane -> (row, strip).  The 16 lanes that one ds_read_b128 pass serves
    // together ({0-3,12-15,20-27} / {4-11,16-19,28-31} of each half wave) share one row and
    // take its 16 strips, so their 16-byte accesses fall into 16 different bank groups.
    const int lane = tid & 63;
    const int la = (lane >> 4) & 1, lb = (lane >> 3) & 1, lc = (lane >> 2) & 1;
    // 5x5 (neighbour sharing): the 16 strips of a row in 16 consecutive lanes, four rows per wave; the fourth wave holds rows
    // 12..14 and, in lanes 48..62, the halo groups of rows 0..14 (lane 63 repeats row 14's).  Other methods: the conflict-free map.
    const bool is_strip = !CHAIN || tid < TCH * 16;
    const int k = CHAIN ? (lane & 15) : ((la << 3) | (lb << 2) | (lane & 3));
    const int j = CHAIN ? (is_strip ? (tid >> 4) : min(tid - TCH * 16, TCH - 1)) : (tid >> 6) * 4 + ((lane >> 5) << 1) + (la ^ lb ^ lc);

    if (threadIdx.x == 0) { sm.next_ticket = atomicAdd(&a.tickets[grp], 1); sm.dark_items[0] = 0; sm.dark_items[1] = 0; sm.fb_count = 0; }
    __syncthreads();
    int t = band_start + sm.next_ticket;
    auto tile_coords = [&](int tt, int &f, int &tr, int &tx0, int &ty0) {
        f = tt / tiles_per_frame;
        tr = tt - f * tiles_per_frame;
        const int trow = tr / a.tiles_x;
        tx0 = (tr - trow * a.tiles_x) * 2 * TCW;
        ty0 = trow * 2 * TCH;
    };
    // The prefetch is unconditional on purpose: threads without an item and the last
    // iteration re-load a valid item / tile.  A conditional load would need the old
    // register value on the other path, and the copies the compiler inserts for that
    // merge wait for the load right where it is issued.
    const int load_tid = min(tid, N_ITEMS - 1);
    auto issue_tile = [&](int tt) {
        int f, tr, tx0, ty0;
        tile_coords(tt, f, tr, tx0, ty0);
        if (half8) issue_item_half8<PACKED, N_FULL>(r0, r1, tl0, tl1, a.src + (size_t)f * a.src_stride, a.w, a.h, tx0, ty0, load_tid);
        else issue_item<PACKED, N_FULL>(r0, r1, a.src + (size_t)f * a.src_stride, a.w, a.h, tx0, ty0, load_tid);
    };
    if (vec) issue_tile(min(t, max(total - 1, 0)));
    __syncthreads();                           // T16 copy complete

#ifdef KF_DIAG_TIMES
    const uint64_t rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    int par = 0;                               // tile parity: which of the two dark_items counters this tile uses
    int fb_skip = 0, fb_wait = FB_WAIT_MIN;    // 5x5: tiles still to go straight to the 32-bit chain; how many after the next busy tile
    bool robust = false;                       // 5x5: rows of lanes agree on their references (robust_ref)
    int calm = 0;                              //      tiles in a row without an uncertain strip
    while (t < band_end) {
        int my_ticket = 0;
        if (threadIdx.x == 0) my_ticket = atomicAdd(&a.tickets[grp], 1);       // the tile after this one: back long before it is needed
        int f, tr, tx0, ty0;
        tile_coords(t, f, tr, tx0, ty0);
        const uint8_t *frame = a.src + (size_t)f * a.src_stride;
        uint16_t *out = (uint16_t *)(a.dst + (size_t)f * a.dst_stride);
        // ---- pixel-map entries of this tile (few tiles have any): list bounds now -- the wait that the uniform load implies is
        // for data the loader needs anyway --, the entries themselves in flight while the loader phase runs
        const bool tile_patched = a.patch && (!pmap_ok || ((sm.has_patch[tr >> 5] >> (tr & 31)) & 1u));      // wave-uniform
        int pbeg = 0, pend = 0;
        int4 my_rec = make_int4(-1, 0, 0, 0);
        static_assert(N_ITEMS <= 192, "the fourth wave has no loader item: it takes the pixel-map cells");
        const bool patch_wave = tid >= 192;                    // wave-uniform
        const int4 *cells = a.cells + (size_t)f * a.n_rec;
        if (tile_patched) {
            pbeg = a.tile_off[tr];
            pend = a.tile_off[tr + 1];
            if (patch_wave && pbeg + (tid - 192) < pend) my_rec = cells[pbeg + (tid - 192)];
        }
        // ---- loader: prefetched registers -> EV planes + interior raw pixels
        __builtin_amdgcn_s_setprio(0);
        // Lane predicates and wave-uniform switches are re-derived per tile from opaque copies: hoisted out of the loop they
        // became 64-bit SGPR masks, two scalar registers each, of which the kernel kept more than it has -- they were spilt
        // to VGPR lanes and came back through v_readlane, VECTOR instructions (about 40 per tile and wave).
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
        const bool has_item = tid_o < N_ITEMS;
        const bool edge = tid_o >= N_FULL;
        if (has_item) {
            uint32_t p0[16], p1[16];
            if (half8) {
                const int iy = ty0 - 2 * HC + 2 * item_row;
                const int x_main = edge ? tx0 - 16 : tx0 + 16 * item_g, x_right = tx0 + 2 * TCW;
                const int y0c = clampi(iy, 0, a.h - 1), y1c = clampi(iy + 1, 0, a.h - 1);
                unpack16_half8<PACKED>(r0, tl0, edge, half8_row<PACKED>(a.w, a.h, x_main, y0c, true), half8_row<PACKED>(a.w, a.h, x_right, y0c, false), p0);
                unpack16_half8<PACKED>(r1, tl1, edge, half8_row<PACKED>(a.w, a.h, x_main, y1c, true), half8_row<PACKED>(a.w, a.h, x_right, y1c, false), p1);
            } else if (vec) { unpack16<PACKED>(r0, p0); unpack16<PACKED>(r1, p1); }
            else fetch_rows<PACKED>(frame, a.w, a.h, tx0 + 16 * item_g, ty0 - 2 * HC + 2 * item_row, edge, tx0, p0, p1);
            // Pixels at or below black (ev = INT_MIN / 0) or beyond the table need the fix-ups of cell_pair_ev: decided once
            // per item from the extremes of its 32 pixels (three-input min/max), wave-uniformly.
            bool odd = false;
            if (METHOD != 0) {
                uint32_t lo = min(p0[0], p1[0]), hi = max(p0[0], p1[0]);
#pragma unroll
                for (int i = 1; i < 16; i++) {
                    lo = min(min(lo, p0[i]), p1[i]);
                    if (!PACKED) hi = max(max(hi, p0[i]), p1[i]);
                }
                odd = (int)lo <= a.black || (!PACKED && (int)hi - a.black > 16383) || (PACKED && a.black < 0);
            }
            const bool slow = __any(odd);
            if (METHOD == 5 && SPREAD && slow) {
                const unsigned long long who = __ballot(odd);
                if (lane == 0) atomicAdd(&sm.dark_items[par], __popcll(who));
            }
            emit_item<METHOD, Smem>(sm, a.black, slow, item_row, item_g, edge, p0, p1);
        }
        PatchCell my_cell;
        my_cell.i = -1;
        if (tile_patched && patch_wave) my_cell = patch_cell<METHOD, PACKED, Smem>(sm, a, my_rec, tx0, ty0);      // the first 64 cells of the tile
        if (threadIdx.x == 0) sm.next_ticket = my_ticket;
        lds_barrier();
        const int t_next = band_start + __builtin_amdgcn_readfirstlane(sm.next_ticket);
        if (tile_patched) {
            if (patch_wave) patch_store<METHOD, Smem>(sm, my_cell);
            for (int base = pbeg + 64; base < pend; base += 256) {          // a dense map (focus pixels): the rest, all lanes
                const int4 rec = base + tid < pend ? cells[base + tid] : make_int4(-1, 0, 0, 0);
                const PatchCell c = patch_cell<METHOD, PACKED, Smem>(sm, a, rec, tx0, ty0);
                patch_store<METHOD, Smem>(sm, c);
            }
            lds_barrier();
        }
        // Waves that are past the loader issue ahead of waves (of the CU's other workgroups) that are still in it: a tile that
        // is about to finish finishes sooner, its workgroup's barrier opens sooner, and the loader instructions of the others
        // fill the gaps.  Measured (tools/kbench.py): cs5x5 11.2-11.5 -> 10.2-10.4 us per frame, cs2x2 8.2 -> 7.8; which of the
        // levels 1..3 is used, and a third level for the output stage, make no difference.
        if (METHOD != 0) __builtin_amdgcn_s_setprio(1);       // (without chroma smoothing the kernel is load-bound and this costs 8 %)
        // ---- prefetch the next tile while the medians run
        if (vec) issue_tile(min(t_next, band_end - 1));

        // ---- medians + output: one thread = 4 cells = 8 px on two rows
        int stripe_mode = a.stripes ? ((PACKED && a.coef_pk) ? 1 : (a.coef_fast ? 2 : 3)) : 0;      // scalar, re-read per tile (see above)
        asm volatile("" : "+s"(stripe_mode));
        // the rest of a strip once its medians are known: R / B replacement, stripes, store
        auto finish_strip = [&](int jj, int kk, bool smooth, const int (&mr)[STRIP], const int (&mb)[STRIP], bool store) {
            const int y = ty0 + 2 * jj, x = tx0 + 2 * STRIP * kk;
            uint32_t top[STRIP], bot[STRIP];        // (R | G1<<16), (G2 | B<<16)
            auto read_raw = [&]() {
                const uint4 v0 = *(const uint4 *)&sm.raw[2 * jj][2 * STRIP * kk];
                const uint4 v1 = *(const uint4 *)&sm.raw[2 * jj + 1][2 * STRIP * kk];
                top[0] = v0.x; top[1] = v0.y; top[2] = v0.z; top[3] = v0.w;
                bot[0] = v1.x; bot[1] = v1.y; bot[2] = v1.z; bot[3] = v1.w;
            };
            if (METHOD == 0) read_raw();
            if (METHOD != 0) {
                // (the green EVs first: the look-ups' indices wait for them; the raw pixels are read once the look-ups are under way)
                const int4 g4 = *(const int4 *)&sm.ge[jj][STRIP * kk];
                const int gev[STRIP] = { g4.x, g4.y, g4.z, g4.w };
                // the output pixel by EV (896 KiB table in L2): all 8 look-ups issued before the first use
                int er[STRIP], eb[STRIP], ur[STRIP], ub[STRIP];
#pragma unroll
                for (int c = 0; c < STRIP; c++) {
                    er[c] = wadd(gev[c], mr[c]);
                    eb[c] = wadd(gev[c], mb[c]);
                    ur[c] = mlv_sbl_u16(rs_e2r, min(max(er[c], 0), MLV_EV_MAX), 0, 0, 0);
                    ub[c] = mlv_sbl_u16(rs_e2r, min(max(eb[c], 0), MLV_EV_MAX), 0, 0, 0);
                }
                read_raw();
                // chroma_smooth.c:27 leaves columns 0..3 and w-4.. alone: only the tiles at the frame's left and right margin test for that
                const bool x_margin = tx0 < 4 || tx0 + 2 * TCW > a.w - 4;                   // scalar
                // which cells take the smoothed values (chroma_smooth.c:28, 35, 64-65): decided while the look-ups are under way
                bool okc[STRIP];
#pragma unroll
                for (int c = 0; c < STRIP; c++) {
                    const int xc = x + 2 * c;
                    okc[c] = smooth && gev[c] >= 2 * MLV_EV_RES && er[c] > MLV_EV_RES && eb[c] > MLV_EV_RES;
                    if (x_margin) okc[c] = okc[c] && xc >= 4 && xc < a.w - 4;
                }
                // (the fence keeps the eight look-ups together and comes after the decisions in program order: the compiler
                // schedules them under the look-ups' latency; with the masks as operands of the fence it also copied them)
                asm volatile("" :: "v"(ur[0]), "v"(ur[1]), "v"(ur[2]), "v"(ur[3]), "v"(ub[0]), "v"(ub[1]), "v"(ub[2]), "v"(ub[3]));
#pragma unroll
                for (int c = 0; c < STRIP; c++) {
                    const bool ok = okc[c];
                    const uint32_t pr_ = (uint32_t)ur[c], pb_ = (uint32_t)ub[c];
                    top[c] = ok ? ((top[c] & 0xFFFF0000u) | pr_) : top[c];
                    bot[c] = ok ? ((bot[c] & 0x0000FFFFu) | (pb_ << 16)) : bot[c];
                }
            }
            if (stripe_mode != 0) {
                // a strip starts at an x that is a multiple of 8, so pixel n of the strip has column phase n
                int co[8];
#pragma unroll
                for (int i = 0; i < 8; i++) { co[i] = a.coef[i]; asm volatile("" : "+s"(co[i])); }
                if (stripe_mode == 1) stripe_strip_pk(top, bot, co, black16, white16);
                else if (stripe_mode == 2) stripe_strip<true>(top, bot, co, black16, white16);
                else stripe_strip<false>(top, bot, co, black16, white16);
            }
            if (store && y < a.h) {
                if (vec) {
                    if (x < a.w) {
                        const uint32_t o = (uint32_t)y * (uint32_t)a.w + (uint32_t)x;           // < 2^28 pixels per frame
                        *(uint4 *)(out + o) = make_uint4(top[0], top[1], top[2], top[3]);
                        if (y + 1 < a.h) *(uint4 *)(out + o + (uint32_t)a.w) = make_uint4(bot[0], bot[1], bot[2], bot[3]);
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
        };
        const int y = ty0 + 2 * j;
        const bool smooth_row = METHOD != 0 && y >= 4 && y < a.h - 5;                       // chroma_smooth.c:25
        int mr[STRIP] = { 0, 0, 0, 0 }, mb[STRIP] = { 0, 0, 0, 0 };
        bool skip_packed = false;
        if (CHAIN) {
            // Deep shadows (EVs of neighbouring small integers are more than the packed window apart) would fail the packed
            // attempt almost everywhere: tiles with many items at or below black skip it (dark clips only, i.e. the SPREAD
            // instantiation), and so do tiles that follow a tile a quarter of whose strips were uncertain (hard colour edges
            // everywhere: that is where the packed attempt plus the dense pass cost more than the 32-bit chain).
            const bool dark_tile = SPREAD && __builtin_amdgcn_readfirstlane(sm.dark_items[par]) >= DARK_ITEMS_MIN;
            skip_packed = dark_tile || fb_skip > 0;            // the same for every wave of the workgroup
            if (fb_skip > 0) fb_skip--;
            if (SPREAD && tid == 0) sm.dark_items[par ^ 1] = 0;
            bool unknown = true;
            const bool chain32 = skip_packed;                  // the same for every wave of the workgroup
            if (chain32) {
                // a tile that skips the packed attempt: both planes through the 32-bit chain, one after the other (the rows'
                // exchange records are reused)
#pragma unroll 1
                for (int pln = 0; pln < 2; pln++) {
                    Chain32 g;
                    chain32_group(pln ? sm.db : sm.dr, j, is_strip ? STRIP * k : TCW, g);
                    if (!is_strip) chain32_publish(g, sm.xchg[j]);
                    lds_barrier();
                    Next32 n;
                    chain32_fetch(g, n);
                    if (k == 15) chain32_collect(sm.xchg[j], n);
                    if (pln) chain32_finish(g, n, mb);
                    else {
                        chain32_finish(g, n, mr);
                        lds_barrier();                         // every row's record has been read: the second plane may overwrite it
                    }
                }
                unknown = false;
            } else
            if (!skip_packed) {
                ChainGroup g;
                chain_group(sm.dr, sm.db, j, is_strip ? STRIP * k : TCW, robust, g);  // a strip's own group, or a row's halo group
                const bool halo_wave = tid >= 192;                     // the wave that holds the halo groups' lanes (uniform)
                if (halo_wave) {
                    chain_group_window(g);
                    if (!is_strip) chain_publish(g, sm.xchg[j]);
                }
                lds_barrier();                                         // the halo groups are in LDS
                ChainNext n;
                chain_fetch_lists(g, n);
                if (k == 15) chain_collect_lists(sm.xchg[j], n);       // (lanes without a strip read a row's record too: harmless)
                if (!halo_wave) chain_group_window(g);                 // the lane's own rank window, while those reads are under way
                chain_fetch_window(g, n);
                if (k == 15) chain_collect_window(sm.xchg[j], n);
                unknown = chain_finish(g, n, mr, mb);
            }
            unknown = unknown && is_strip && smooth_row;
            if (unknown) sm.fb_queue[atomicAdd(&sm.fb_count, 1)] = (uint8_t)(j * 16 + k);      // settled below, densely
            if (is_strip && !unknown) finish_strip(j, k, smooth_row && (!skip_packed || chain32), mr, mb, true);
        } else {
            if (smooth_row) {
                if (METHOD == 3) {
                    strip_median9(sm.dr, j + 1, STRIP * k + 1, mr);
                    strip_median9(sm.db, j + 1, STRIP * k + 1, mb);
                } else if (METHOD == 2) {
                    strip_median5(sm.dr, j + 1, STRIP * k + 1, mr);
                    strip_median5(sm.db, j + 1, STRIP * k + 1, mb);
                }
            }
            finish_strip(j, k, smooth_row, mr, mb, true);
        }
        if (CHAIN) {
            // Strips whose packed medians were not certain: all of the tile's, gathered in LDS, go through the 32-bit networks
            // one per lane -- as many waves as ceil(count / 64) run them, instead of every wave that had one such strip.
            lds_barrier();
            const int nfb = __builtin_amdgcn_readfirstlane(sm.fb_count);
            if (!skip_packed) {
                // more than FB_DIRECT uncertain strips: the next fb_wait tiles go to the 32-bit chain directly, and the wait doubles
                // (up to FB_WAIT_MAX) each time the tile after it is no better -- a stray busy tile costs its few successors a
                // fifth more, a clip full of hard colour edges tries the packed networks once in sixteen tiles
                if (nfb > FB_DIRECT) { fb_skip = fb_wait; fb_wait = min(2 * fb_wait + 1, FB_WAIT_MAX); }
                else fb_wait = FB_WAIT_MIN;
                // shared references (robust_ref) from the tile after one with uncertain strips until four tiles in a row had none
                calm = nfb == 0 ? calm + 1 : 0;
                robust = nfb > FB_ROBUST || (robust && calm < 4);
            }
#ifdef KF_DIAG_TIMES
            if (tid == 0 && a.times) {                         // tiles, tiles that skipped the packed networks, strips settled densely
                atomicAdd(&a.times[4096], 1ull);
                if (skip_packed) atomicAdd(&a.times[4097], 1ull);
                atomicAdd(&a.times[4098], (unsigned long long)nfb);
            }
#endif
            if (nfb > 0) {
                if ((tid & ~63) < nfb) {                               // this wave has entries
                    const int e = sm.fb_queue[min(tid, nfb - 1)];
                    const int j2 = e >> 4, k2 = e & 15;
                    int z = 0;
                    asm volatile("" : "+s"(z));                        // (keeps these plane loads apart from the packed path's)
                    strip_median25(sm.dr, j2 + z, STRIP * k2, mr);
                    strip_median25(sm.db, j2 + z, STRIP * k2, mb);
                    finish_strip(j2, k2, true, mr, mb, tid < nfb);
                }
                lds_barrier();
                if (tid == 0) sm.fb_count = 0;
            }
        } else
        lds_barrier();                       // all strips done with the planes before the next tile's loader
        t = t_next;
        if (SPREAD) par ^= 1;
    }
    if (threadIdx.x == 0 && atomicAdd(&a.tickets[a.groups], 1) == (int)gridDim.x - 1) {
        for (int i = 0; i <= a.groups; i++) a.tickets[i] = 0;       // last workgroup out: ready for the next launch on this stream
    }
#ifdef KF_DIAG_TIMES
    if (threadIdx.x == 0 && a.times) { a.times[2 * blockIdx.x] = rt0; a.times[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

// ---------------------------------------------------------------- host launcher
// zeroed counters per stream (launches on one stream run one after the other and leave the counters zeroed)
constexpr int MAX_GROUPS = 1024;
namespace {
std::mutex g_ticket_mu;
std::map<std::pair<int, hipStream_t>, int *> g_tickets;
}
static int *ticket_counters(hipStream_t stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_ticket_mu);
    int *&p = g_tickets[{ dev, stream }];
    if (!p) {
        // zeroed ON THE LAUNCHING STREAM: the streams are non-blocking, a null-stream hipMemset is not ordered before their
        // kernels (it once landed in the middle of the first launch, the "done" count never completed and the next launch
        // on that stream started from stale tickets)
        if (hipMalloc(&p, (MAX_GROUPS + 1) * sizeof(int)) != hipSuccess ||
            hipMemsetAsync(p, 0, (MAX_GROUPS + 1) * sizeof(int), stream) != hipSuccess) {
            set_error("ticket counters: allocation failed");
            if (p) (void)hipFree(p);
            g_tickets.erase({ dev, stream });
            return nullptr;
        }
    }
    return p;
}
// The streams this library creates (per host thread, per host pipeline) give their counters back when they are destroyed: a
// recycled stream handle then starts from freshly zeroed counters instead of whatever an aborted launch left behind, and
// retired worker threads leak nothing.  Streams the caller owns keep their 4 KiB until the process ends.
void release_stream_state(int device, hipStream_t stream)
{
    std::lock_guard<std::mutex> lk(g_ticket_mu);
    auto it = g_tickets.find({ device, stream });
    if (it == g_tickets.end()) return;
    if (it->second) (void)hipFree(it->second);
    g_tickets.erase(it);
}
// ---------------------------------------------------------------- per-black output table in HBM
// E2R[ev] = (uint16)(ev2raw[ev] + black) for ev in [0, 14 * 32768): exactly what chroma_smooth.c:67-68 stores for a clamped EV.
// One 16-bit look-up by index replaces mask, address, quotient, two shifts, add and mask per output pixel (cs5x5 -2.5 %, A/B in
// profiles/r02/ab_table_gathers_kbench.log: "e1").  Built on the device from the exact 16-bit re-encoding U16 (common.h) the
// first time a black level is seen on a device; 896 KiB per black level, served from L2.
// (The same trick for the loader -- raw2ev by pixel value from a 256 KiB table, no conversion arithmetic at all, results
// identical -- makes the kernel wait for the texture addresser instead: cs5x5 +11 %, cs2x2 +27 %, "e2" in the same log.)
__global__ __launch_bounds__(256) void k_build_e2r(const uint16_t *u16, int black, uint16_t *e2r)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < E2R_ENTRIES) e2r[i] = (uint16_t)((((int)u16[i & 32767]) >> (13 - (i >> 15))) + black);
}
// The black level comes from file headers (and dual ISO multiplies it by 4): a long-running host that serves many clips would
// otherwise collect one table per level it has ever seen.  At most E2R_CACHE tables per device stay; the least recently used one
// goes -- not at once: a thread that fetched its pointer a moment ago may not have launched yet, so an evicted table is parked and
// freed at the NEXT eviction on that device, after the device has drained (the same rule as DeviceTables::retire in dualiso.cpp).  The table of a new level is built OUTSIDE the lock, so first launches of
// different clips do not queue behind each other's synchronisation.
namespace {
constexpr size_t E2R_CACHE = 8;
struct E2rEntry { uint16_t *table; unsigned long long used; };
std::mutex g_tables_mu;
std::map<std::pair<int, int>, E2rEntry> g_e2r;               // (device, black)
std::map<int, uint16_t *> g_e2r_parked;                       // per device: the table evicted last
unsigned long long g_e2r_clock = 0;
}
static int e2r_table(const Device *dev, int black, const uint16_t **out, hipStream_t stream)
{
    {
        std::lock_guard<std::mutex> lk(g_tables_mu);
        auto it = g_e2r.find({ dev->id, black });
        if (it != g_e2r.end()) { it->second.used = ++g_e2r_clock; *out = it->second.table; return MLVFS_AMD_OK; }
    }
    uint16_t *t = nullptr;
    MLV_HIP(hipMalloc(&t, sizeof(uint16_t) * E2R_ENTRIES));
    hipLaunchKernelGGL(k_build_e2r, dim3((E2R_ENTRIES + 255) / 256), dim3(256), 0, stream, dev->luts.u16, black, t);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {       // other streams use the table from now on
        (void)hipFree(t);
        set_error("building the output table for black level %d failed", black);
        return MLVFS_AMD_ERR_HIP;
    }
    uint16_t *victim = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_tables_mu);
        auto it = g_e2r.find({ dev->id, black });
        if (it != g_e2r.end()) {                              // another thread built the same level meanwhile: keep theirs
            it->second.used = ++g_e2r_clock;
            *out = it->second.table;
            victim = t;
        } else {
            size_t mine = 0;
            auto oldest = g_e2r.end();
            for (auto e = g_e2r.begin(); e != g_e2r.end(); ++e)
                if (e->first.first == dev->id) {
                    mine++;
                    if (oldest == g_e2r.end() || e->second.used < oldest->second.used) oldest = e;
                }
            if (mine >= E2R_CACHE) {                           // (rare: a ninth black level on this device)
                victim = g_e2r_parked[dev->id];                // evicted one eviction ago: long out of every thread's hands
                g_e2r_parked[dev->id] = oldest->second.table;
                g_e2r.erase(oldest);
            }
            g_e2r[{ dev->id, black }] = E2rEntry{ t, ++g_e2r_clock };
            *out = t;
        }
    }
    if (victim) {
        if (victim != t) (void)hipDeviceSynchronize();         // launches that still read it have finished
        (void)hipFree(victim);
    }
    return MLVFS_AMD_OK;
}

template <int METHOD, bool PACKED, int VEC, bool SPREAD>
static int launch_frame_t(const FrameArgs &a_in, int num_cu, hipStream_t stream)
{
    const long long total = (long long)a_in.tiles_x * a_in.tiles_y * a_in.nframes;
    int grid = num_cu > 0 ? num_cu * 4 : 1024;          // 4 workgroups per CU (39 KiB LDS, <= 128 VGPRs)
    grid = (grid + 7) / 8 * 8;
    if (grid > total) grid = (int)((total + 7) / 8 * 8);
    static_assert(sizeof(SmemT<SPREAD, tile_rows_of(METHOD)>) <= 40 * 1024, "four workgroups per CU need <= 40 KiB of LDS each");
    auto kern = k_frame<METHOD, PACKED, VEC, SPREAD>;
    FrameArgs a = a_in;
    a.tickets = ticket_counters(stream);
    if (!a.tickets) return MLVFS_AMD_ERR_HIP;
    a.groups = std::min(std::max(grid / 4, 1), MAX_GROUPS);
#ifdef KF_DIAG_TIMES
    static unsigned long long *d_times = nullptr;
    if (!d_times) hipMalloc(&d_times, (2048 * 2 + 8) * sizeof(unsigned long long));
    hipMemsetAsync(d_times + 4096, 0, 8 * sizeof(unsigned long long), stream);
    const_cast<FrameArgs &>(a).times = d_times;
#endif
    KernelTimer &tm = kernel_timer();
    const bool timed = tm.on && tm.used + 2 <= (int)tm.ev.size();
    if (timed) MLV_HIP(hipEventRecord(tm.ev[tm.used], stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, stream, a);
    if (timed) { MLV_HIP(hipEventRecord(tm.ev[tm.used + 1], stream)); tm.used += 2; }
    MLV_HIP(hipGetLastError());
#ifdef KF_DIAG_TIMES
    if (METHOD == 5 && a.nframes >= 50) {
        static int shown = 0;
        if (shown++ % 8 == 3) {
            hipStreamSynchronize(stream);
            std::vector<unsigned long long> h(2 * grid);
            hipMemcpy(h.data(), d_times, h.size() * 8, hipMemcpyDeviceToHost);
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int b = 0; b < grid; b++) { t0 = std::min(t0, h[2 * b]); t1 = std::max(t1, h[2 * b + 1]); }
            double end_q[4] = { 0, 0, 0, 0 }, start_q[4] = { 0, 0, 0, 0 };
            for (int b = 0; b < grid; b++) { end_q[b * 4 / grid] += (double)(h[2 * b + 1] - t0); start_q[b * 4 / grid] += (double)(h[2 * b] - t0); }
            unsigned long long fbc[3];
            hipMemcpy(fbc, d_times + 4096, sizeof(fbc), hipMemcpyDeviceToHost);
            fprintf(stderr, "KF_TIMES tiles %llu, of which %llu skipped the packed networks; strips settled densely %llu (%.1f %% of all)\n", fbc[0], fbc[1],
                    fbc[2], fbc[0] ? 100.0 * fbc[2] / (240.0 * fbc[0]) : 0.0);
            fprintf(stderr, "KF_TIMES grid %d: kernel %.1f us; mean start / end of the workgroups of each quarter of the grid (us):", grid, (t1 - t0) * 0.01);
            for (int q = 0; q < 4; q++) fprintf(stderr, "  %.1f / %.1f", start_q[q] / (grid / 4) * 0.01, end_q[q] / (grid / 4) * 0.01);
            fprintf(stderr, "\n");
            double xe[8] = { 0 }, xm[8] = { 0 };
            for (int b = 0; b < grid; b++) { const double e = (double)(h[2 * b + 1] - t0) * 0.01; xe[b % 8] += e / (grid / 8); xm[b % 8] = std::max(xm[b % 8], e); }
            fprintf(stderr, "KF_TIMES per XCD mean/max end:");
            for (int x = 0; x < 8; x++) fprintf(stderr, "  %.0f/%.0f", xe[x], xm[x]);
            fprintf(stderr, "\nKF_TIMES end of blocks 0..255 step 8 (XCD 0's groups):");
            for (int b = 0; b < 256 && b < grid; b += 8) fprintf(stderr, " %.0f", (double)(h[2 * b + 1] - t0) * 0.01);
            fprintf(stderr, "\n");
        }
    }
#endif
    return MLVFS_AMD_OK;
}

int launch_frame(const Device *dev, const Geom &g, bool packed, const void *src, size_t src_stride, void *dst,
                 size_t dst_stride, int nframes, int method, const PatchView *pv, bool stripes,
                 const int32_t *coef, hipStream_t stream, bool spread)
{
    if (nframes <= 0) return MLVFS_AMD_OK;
    const int geo = frame_geo_of(method);
    if ((long long)frame_tiles_x(g.w) * frame_tiles_y(g.h, geo) * nframes >= (1ll << 30)) {
        set_error("too many tiles in one launch (%d frames): split the batch", nframes);
        return MLVFS_AMD_ERR_ARG;
    }
    if (g.w < 2 || g.h < 2 || (g.w & 1) || (long long)g.w * g.h >= (1ll << 28)) {      // 32-bit bit / byte offsets inside a frame
        set_error("frame geometry %dx%d unsupported", g.w, g.h);
        return MLVFS_AMD_ERR_ARG;
    }
    if (packed && g.bpp != 14) { set_error("fused path needs 14-bit input"); return MLVFS_AMD_ERR_ARG; }
    FrameArgs a{};
    a.src = (const uint8_t *)src; a.src_stride = src_stride;
    a.dst = (uint8_t *)dst; a.dst_stride = dst_stride;
    a.w = g.w; a.h = g.h; a.black = g.black; a.white = g.white;
    a.nframes = nframes;
    a.tiles_x = frame_tiles_x(g.w);
    a.tiles_y = frame_tiles_y(g.h, geo);
    a.t16 = dev->luts.t16;
    if (method != 0) {
        int rc = e2r_table(dev, g.black, &a.e2r, stream);
        if (rc) return rc;
    }
    a.patch = pv && pv->n_rec > 0;
    if (a.patch) { a.cells = (const int4 *)pv->cells; a.n_rec = pv->n_rec; a.tile_off = pv->tile_off; }
    a.stripes = stripes ? 1 : 0;
    a.coef_fast = 1;
    for (int i = 0; i < 8; i++) {
        a.coef[i] = (stripes && coef) ? coef[i] : 0;
        if (a.coef[i] - 65536 <= -32768 || a.coef[i] - 65536 >= 32768) a.coef_fast = 0;
    }
    a.coef_pk = a.coef_fast && packed && (int)(uint16_t)g.white > (int)(uint16_t)g.black + 64 && g.black >= 0 && g.black <= 16384;
    // vector path: rows are whole 16-pixel groups and every row starts 16-byte aligned (1), or rows of whole 8-pixel half groups (2)
    static const bool no_half8 = [] { const char *e = getenv("MLVFS_AMD_KF_HALF8"); return e && atoi(e) == 0; }();      // (A/B: the any-geometry path instead)
    const bool strides_ok = nframes == 1 || (src_stride % 16 == 0 && dst_stride % 16 == 0);
    const int vec = ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0 && strides_ok
                        ? ((g.w % 16) == 0 ? 1 : ((g.w % 16) == 8 && g.w >= 24 && !no_half8 ? 2 : 0)) : 0;
#define MLV_DISPATCH_S(M, S)                                                                              \
    return packed ? (vec == 1 ? launch_frame_t<M, true, 1, S>(a, dev->num_cu, stream)                     \
                   : vec == 2 ? launch_frame_t<M, true, 2, S>(a, dev->num_cu, stream)                     \
                              : launch_frame_t<M, true, 0, S>(a, dev->num_cu, stream))                    \
                  : (vec == 1 ? launch_frame_t<M, false, 1, S>(a, dev->num_cu, stream)                    \
                   : vec == 2 ? launch_frame_t<M, false, 2, S>(a, dev->num_cu, stream)                    \
                              : launch_frame_t<M, false, 0, S>(a, dev->num_cu, stream))
#define MLV_DISPATCH(M)                                                                                   \
    if (spread && M != 0) { MLV_DISPATCH_S(M, true); }                                                    \
    MLV_DISPATCH_S(M, false)
    switch (method) {
        case 0: MLV_DISPATCH(0);
        case 2: MLV_DISPATCH(2);
        case 3: MLV_DISPATCH(3);
        case 5: MLV_DISPATCH(5);
        default: set_error("Unsupported chroma smooth method %d", method); return MLVFS_AMD_ERR_ARG;
    }
#undef MLV_DISPATCH
#undef MLV_DISPATCH_S
}

// ---------------------------------------------------------------- which T16 layout suits a clip
// every 7th pixel of every 5th row of one frame: how many lie 1 .. 511 above black (the range whose look-ups collide in the plain layout)
template <bool PACKED>
__global__ __launch_bounds__(256) void k_dark_share(const uint8_t *frame, int w, int h, int black, int *counts)
{
    const int nx = (w + 6) / 7, ny = (h + 4) / 5;
    int dark = 0, all = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nx * ny; i += gridDim.x * blockDim.x) {
        const int y = (i / nx) * 5, x = (i % nx) * 7;
        const int lin = (int)fetch_clamped<PACKED>(frame, w, h, x, y) - black;
        dark += lin >= 1 && lin < 512;
        all++;
    }
    for (int o = 32; o > 0; o >>= 1) { dark += __shfl_xor(dark, o); all += __shfl_xor(all, o); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&counts[0], dark); atomicAdd(&counts[1], all); }
}

int dark_share(bool packed, const void *d_frame, int w, int h, int black, hipStream_t stream, int *share_1024)
{
    int *d_counts = nullptr, hc[2] = { 0, 0 };
    MLV_HIP(hipMalloc(&d_counts, 2 * sizeof(int)));
    hipError_t e = hipMemsetAsync(d_counts, 0, 2 * sizeof(int), stream);
    if (e == hipSuccess) {
        if (packed) hipLaunchKernelGGL(k_dark_share<true>, dim3(64), dim3(256), 0, stream, (const uint8_t *)d_frame, w, h, black, d_counts);
        else hipLaunchKernelGGL(k_dark_share<false>, dim3(64), dim3(256), 0, stream, (const uint8_t *)d_frame, w, h, black, d_counts);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(hc, d_counts, sizeof(hc), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d_counts);
    if (e != hipSuccess) { set_error("dark_share: %s", hipGetErrorString(e)); return MLVFS_AMD_ERR_HIP; }
    *share_1024 = hc[1] > 0 ? (int)((long long)hc[0] * 1024 / hc[1]) : 0;
    return MLVFS_AMD_OK;
}

}  // namespace mlv
