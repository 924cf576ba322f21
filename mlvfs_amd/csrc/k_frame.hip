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
// Work decomposition (gfx950: 256 CUs, 8 XCDs, wave64, 160 KiB LDS/CU), round 4:
//   * tile = 64 x 15 Bayer cells (128 x 30 px) per 256-thread workgroup, halo 2 cells; 39.7 KiB of LDS and <= 128 VGPRs so that
//     FOUR workgroups (16 waves) share a CU and cover each other's barriers, LDS and HBM latencies
//   * a workgroup walks DOWN a column of tiles (tile list in column-major order, handed out in runs): the four plane rows
//     the next tile shares with this one -- and the pixels and green EVs of the two of them that are the next tile's first
//     rows -- stay in LDS, so every pixel's EV is computed once per column instead of 19/15 times, and only the first tile of a
//     run loads its upper halo
//   * LOADER: one thread = one "item" = 4 cells (2 rows x 8 px = 2 x 14 B of packed stream, fetched as 2 x 2 dwords per row);
//     15 rows x (16 items + 1 edge item for the halo columns) = 255 items: every wave converts, none waits.  It unpacks in
//     registers and writes, per cell, the EV triple {ge, dr = ev(R)-ge, db = ev(B)-ge} to LDS planes (the planes are what the
//     medians run on).  The packed dwords of the NEXT tile are prefetched into registers before the median phase, so HBM
//     latency hides behind the selection networks.
//   * MEDIANS: one thread = a strip of 4 horizontally adjacent cells; 5x5: 17 lanes per tile row (16 strips + the halo group),
//     neighbour sharing through DPP (see strip_chain_*)
//   * raw2ev lives in LDS as the 8192-entry mantissa-normalised 16-bit table (common.h),
//     the output pixel by EV is one 16-bit look-up in a per-black table served from L2
//   * pixel-map patches: per-tile lists of repaired cells (built once per clip on the host; values per frame: k_pixfix_cells)
//     recompute just the cells they touch
//   * persistent workgroups; the four residents of a CU draw runs of tiles from that CU's contiguous range of the tile list
//     (the ranges of an XCD's CUs adjacent, so halo re-reads hit its own L2), the last tiles of a range one by one
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
constexpr int TCH = FRAME_TCH;          // tile height in cells (15): 15 rows x 17 lanes = 255 of the 256 threads
constexpr int HC = FRAME_HC;            // halo in cells (2)
constexpr int PW = TCW + 2 * HC;        // plane width  (68)
constexpr int PH = TCH + 2 * HC;        // plane height (19)
constexpr int RH = TCH + HC;            // rows of interior pixels / green EVs kept: the tile's own and the first two of the tile below
constexpr int STRIP = 4;                // cells per thread in the median phase
constexpr int GROUPS = TCW / 4;         // loader items (8 px) per tile row (16)
constexpr int N_MAIN = TCH * GROUPS;    // threads 0..239: item (row t / 16, group t % 16); threads 240..254: edge item of row t - 240
constexpr int N_ITEMS = N_MAIN + TCH;
constexpr int N_TOP_MAIN = 2 * HC * GROUPS, N_TOP = N_TOP_MAIN + 2 * HC;      // first tile of a run: the four rows above, threads 0..67
#ifndef KF_DARK_ITEMS_MIN
#define KF_DARK_ITEMS_MIN 24
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
constexpr int PMAP_WORDS = 64;          // tiles per frame covered by the LDS patch bitmap: 2048 (3584x1320 has 1232)
static_assert(N_ITEMS <= 255 && TCH * 17 <= 256 && N_TOP <= 128, "one item per thread; 17 median lanes per tile row");

struct FrameArgs {
    const uint8_t *src;      // packed stream or u16 frames
    size_t src_stride;       // bytes between frames
    unsigned src_bytes;      // bytes of one frame (rounded up to a dword): the range the loader's buffer loads are checked against
    uint8_t *dst;
    size_t dst_stride;
    int w, h, black, white;
    int nframes;
    int tiles_x, tiles_y;
    const uint16_t *t16;
    const uint2 *e2d;        // the output pixel by EV: (uint16)(ev2raw[ev] + black), ev in [0, 14 * 32768), 32 entries per 8-byte record (E2D_RECORDS)
    // pixel map: per frame `n_rec` cell records {cell, R | G1 << 16, G2 | B << 16, -} (k_pixfix_cells), listed tile by tile (CSR)
    const int4 *cells;
    int n_rec;
    const int *tile_off;
    // stripes
    int coef[8];
    int coef_fast;           // all |coef - 65536| < 32768: 32-bit epilogue
    int coef_pk;             // additionally 14-bit input, white > black + 64, black <= 16384: packed 16-bit epilogue
    int patch, stripes;      // wave-uniform stage switches
    int *tickets;            // per group: [2g] tiles handed out in runs, [2g + 1] tiles handed out singly; [2 groups] workgroups done (the last one zeroes them all)
    int groups;              // workgroups b, b + groups, b + 2 groups, ... form a group (one CU's residents) and share a tile range
    int run, singles;        // tiles per run; tiles at the end of a group's range that go out one by one
#ifdef KF_DIAG_TIMES
    unsigned long long *times;
#endif
};

// Table look-ups as buffer loads with idxen: the address unit scales the index by the descriptor's stride, no VALU address arithmetic
// (tools/gather_probe.hip checks the semantics on gfx950).  The LLVM intrinsics are bound by name: hipcc has no builtin for
// the struct forms, and unlike inline asm the compiler counts these loads in its s_waitcnt bookkeeping.
typedef int mlv_i32x4 __attribute__((ext_vector_type(4)));
__device__ unsigned short mlv_sbl_u16(mlv_i32x4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.i16");
typedef unsigned mlv_tab_u32x2 __attribute__((ext_vector_type(2)));
__device__ mlv_tab_u32x2 mlv_sbl_x2(mlv_i32x4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.v2i32");
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
// The output table E2R[ev] = (uint16)(ev2raw[ev] + black), ev in [0, 14 * 32768), in 8 bytes per 32 entries:
//     record b = { E2R[32 b], bit j: E2R[32 b + j + 1] != E2R[32 b + j] }      E2R[ev] = base + popcount(bits & ((1 << (ev & 31)) - 1))
// (consecutive entries differ by at most one: d ev2raw / d ev < 0.35; checked when a table is built) -- v_bfe_u32 with the EV itself as
// the field width, v_bcnt_u32_b32 with the base as its addend: two operations more than a plain 16-bit table, which is what rounds
// 2-4 used (896 KiB per black level, a 128-byte line fetched per 2-byte entry: footage whose tiles span several EV missed the L1 on
// most look-ups).  112 KiB: same-box A/B (profiles/r04/ab_dense_kinds.log), us per frame plain -> dense: cs2x2 6.80 -> 6.18 on the
// benchmark's frames, 9.4 -> 7.0 in low light, 10.8 -> 8.75 on colour patches; cs5x5 8.35 -> 8.22 / 10.8 -> 10.55 / 12.4 -> 11.8.
constexpr int E2R_ENTRIES = 14 * MLV_EV_RES;
constexpr int E2D_RECORDS = E2R_ENTRIES / 32;
#ifdef KF_EXP_LEAN          // timing experiment: every rare path compiled out (results are wrong where one would have been taken)
#define KF_EXP_PKONLY
#define KF_EXP_NOFALLBACK
#define KF_EXP_FASTLOADER
#endif
#ifndef KF_SRC_AUX
#define KF_SRC_AUX 0          // cache policy of the loader's stream loads (experiments, same encoding)
#endif
#ifndef KF_E2R_AUX
#define KF_E2R_AUX 0          // cache policy of the output look-ups (experiments: 2 = nt, 16 = sc1, 17 = sc0 sc1)
#endif

// SPREAD: the T16 table with entry i at i + (i >> 7).  A pixel below 2^e above black uses only every 2^(13-e)-th entry, so
// the look-ups of dark footage crowd into a few LDS banks (below 128 DN: one); the spread form puts those entries into
// different banks for two more operations per pixel.  Chosen per clip from its first frame (launch_frame's `spread`).
constexpr int XCHG_WORDS = 28;          // what the first lane of a wave hands to the last lane of the wave before it (7 x 16 bytes)
template <bool SPREAD_, bool CHAIN_>
struct __align__(16) SmemT {
    static constexpr bool SPREAD = SPREAD_;
    static constexpr bool CHAIN = CHAIN_;               // 5x5: neighbour-sharing medians
    uint16_t raw[2 * RH][2 * TCW];      // interior pixels (post patch) + the four pixel rows below (the next tile's first), 8.5 KiB
    int dr[PH][PW];                     // 5 KiB
    int db[PH][PW];
    int ge[RH][TCW];                    // 4.25 KiB
    uint16_t t16[MLV_T16_N + (SPREAD_ ? 64 : 0)];   // mantissa-normalised raw2ev (common.h), 16 KiB
    uint32_t has_patch[PMAP_WORDS];     // one bit per tile of a frame: some pixel-map entry touches it
    uint32_t xchg[CHAIN_ ? 3 : 1][XCHG_WORDS];      // 5x5: sorted columns / pair list / rank window of the group held by lane 0 of waves 1..3
    uint8_t fb_queue[CHAIN_ ? 256 : 4]; // 5x5: strips whose packed medians are not certain (row * 16 + strip), settled densely
    int fb_count;
    int next_tile, next_end;            // the tile after this one and the end of the run it belongs to (thread 0 -> all)
    int dark_items[2];                  // loader items of the current / next tile that hold pixels at or below black (5x5 only)
};
static_assert(sizeof(SmemT<true, true>) <= 40 * 1024, "four workgroups per CU need <= 40 KiB of LDS each");

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

// The same for items that hold pixels AT or BELOW black (shadows of any real clip: with 7 DN of read noise a few per cent of a dark
// region's pixels), still without a branch or a select per pixel.  raw2ev there: lin < 0 -> 0 = raw2ev(1), lin == 0 -> INT_MIN
// (main.c:163-167), and the cell arithmetic wraps (chroma_smooth.c:32,54 on ints).  With f clamped to 1.0 a pixel's biased EV is
// that of lin = 1; the sign of |f| - 0.5 is set exactly for lin == 0 and goes into bit 31 of the biased EV: eb'' = eb + z * 2^31,
// i.e. ev = eb'' - bias (mod 2^32) for every pixel.  A green sum s = eb''(G1) + eb''(G2) is the true sum + 2 bias (mod 2^32); it
// has bit 31 set exactly when one of the two is INT_MIN, and C's truncating half of it AS A SIGNED number is ge + bias in every
// case (both INT_MIN: s = 2 bias, ge = 0, as the wrapped sum of the reference gives).  3 more operations per pixel and 2 per
// cell than the common path (the previous out-of-table path: compare + select per pixel, 3.4x the common path's time; it
// stays for what lies BEYOND the table, 16-bit input only).  Needs lin <= 16383 for every pixel.
template <int NC, bool SPREAD>
__device__ __forceinline__ void cell_multi_ev_dark(const uint32_t *p0, const uint32_t *p1, int black, const uint16_t *t,
                                                   int (&ge)[NC], int (&dr)[NC], int (&db)[NC])
{
    uint32_t fb[4 * NC], tv[4 * NC], ex[4 * NC], eb[4 * NC], z[4 * NC];
    const float fblack = (float)black;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const uint32_t px[4] = { p0[2 * c], p0[2 * c + 1], p1[2 * c], p1[2 * c + 1] };
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float f = (float)px[i] - fblack;
            z[4 * c + i] = __float_as_uint(fabsf(f) - 0.5f);
            fb[4 * c + i] = __float_as_uint(fmaxf(f, 1.0f));
        }
    }
#pragma unroll
    for (int i = 0; i < 4 * NC; i++) tv[i] = *(const uint16_t *)((const char *)t + t16_offset<SPREAD>(fb[i]));
#pragma unroll
    for (int i = 0; i < 4 * NC; i++) ex[i] = bfe_asm<23, 8>(fb[i]);
#pragma unroll
    for (int i = 0; i < 4 * NC; i += 8) {
        asm volatile("" :: "v"(ex[i]), "v"(ex[i + 1]), "v"(ex[i + 2]), "v"(ex[i + 3]), "v"(ex[i + 4]), "v"(ex[i + 5]), "v"(ex[i + 6]), "v"(ex[i + 7]));
    }
#pragma unroll
    for (int i = 0; i < 4 * NC; i += 8) {
        asm volatile("" :: "v"(tv[i]), "v"(tv[i + 1]), "v"(tv[i + 2]), "v"(tv[i + 3]), "v"(tv[i + 4]), "v"(tv[i + 5]), "v"(tv[i + 6]), "v"(tv[i + 7]));
    }
#pragma unroll
    for (int i = 0; i < 4 * NC; i++) eb[i] = (z[i] & 0x80000000u) | ((ex[i] << 15) + tv[i]);      // v_lshl_add_u32, v_and_or_b32
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const uint32_t gb = (uint32_t)half_trunc((int)(eb[4 * c + 1] + eb[4 * c + 2]));
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
// An item is 8 pixels on two rows = 4 Bayer cells: 2 x 14 bytes of the 14-bit stream (2 x 16 bytes of a 16-bit frame), fetched as
// two 8-byte loads per row into d[0..1] and d[2..3].
//   main item : the 8-pixel group at (x, y).  A group starts at an even byte of the stream: dword-aligned ("aligned": d = the 16
//               bytes from the group's first byte) or in the upper half of a dword ("mis": d = the 16 bytes from two bytes BEFORE
//               the group).  Which of the two depends on the group's number in its row and -- widths that are a multiple of 8 but
//               not of 16 (1736: the 3x crop of most APS-C bodies; 1880: the 5D2) -- on the row's parity.
//   edge item : d[0..1] = the 8 bytes that hold the four pixels RIGHT of the tile (the first 56 bits of their group), d[2..3] = the
//               8 bytes that hold the four pixels LEFT of it (the last 56 bits of theirs), each at its exact (even) byte address.
// Three v_perm selectors per lane turn d into the stream words S0..S3 (MSB-first) that hold the item's 112 bits, and the same
// eight bit-field extractions yield px 0..7 for every lane -- of an edge item px 0..3 = right halo, px 4..7 = left halo: one
// code path, no per-lane variant of the register layout.
typedef uint32_t mlv_u32x2 __attribute__((ext_vector_type(2)));
__device__ mlv_u32x2 mlv_rbl_x2(mlv_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2i32");
// one frame as a raw buffer: loads are range-checked by the address unit (an offset beyond the frame reads zeros, never faults)
__device__ __forceinline__ mlv_i32x4 frame_rsrc(const void *p, unsigned bytes)
{
    const unsigned long long a = (unsigned long long)p;
    mlv_i32x4 r;
    r.x = (int)(unsigned)a;
    r.y = (int)((unsigned)(a >> 32) & 0xFFFFu);
    r.z = (int)bytes;
    r.w = 0x00020000;
    return r;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// The kernel's third template argument, VEC, names the input's layout: 0 = any geometry (a load per pixel), 1 = rows of whole
// 8-pixel groups whose rows all start dword-aligned, 2 = the same with rows that start alternately dword-aligned and in the upper
// half of a dword (14-bit stream, w % 16 == 8) -- and, for the reduced bit depths of newer Magic Lantern builds, 3 = 12-bit and
// 4 = 10-bit stream, rows dword-aligned (round 4: such clips took an unpack pass to 16 bits first).  bits per pixel of a layout:
constexpr int bpp_of(bool packed, int vec) { return !packed ? 16 : (vec == 3 ? 12 : (vec == 4 ? 10 : 14)); }

constexpr uint32_t SEL_SWAP = 0x01000302u;      // v_perm_b32(nxt, d, .): the halves of d swapped
constexpr uint32_t SEL_MIS = 0x03020504u;       //                        (d & 0xFFFF0000) | (nxt & 0xFFFF)
constexpr uint32_t SEL_EDGE1 = 0x01000304u;     //                        halves of d swapped, lowest byte from nxt

// what a lane's item is, for the whole kernel (x offsets of the two loads relative to the tile; how their byte offsets are formed)
struct ItemLane {
    int xoff_a, xoff_b;      // main: 8 * group both; edge: 128 (right of the tile), -8 (the group whose last four pixels lie left of it)
    uint32_t amask;          // main: ~3 (loads start at the dword that holds the group's first byte); edge: ~0 (exact)
    uint32_t boff;           // second load: main: first + 8; edge: group b + 6
    uint32_t s0, s1, s23;    // v_perm selectors on an even row
    uint32_t flip;           // w % 16 == 8: what an odd row changes about them (main items: aligned <-> mis)
    bool edge;
};

template <bool PACKED, int VEC>
__device__ __forceinline__ ItemLane item_lane(int lk, bool edge)
{
    constexpr int BPP = bpp_of(PACKED, VEC);
    ItemLane L;
    L.edge = edge;
    L.xoff_a = edge ? 2 * TCW : 8 * lk;
    L.xoff_b = edge ? -8 : 8 * lk;
    L.amask = (PACKED && !edge) ? ~3u : ~0u;
    // edge item, second load: the last four 16-bit words of the group left of the tile (14 bit: words 3..6; 12: 2..5; 10: 2..5, the
    // group has five)
    L.boff = edge ? (BPP == 14 ? 6u : (BPP == 16 ? 8u : 4u)) : 8u;
    // rows of a tile start at a multiple of 128 pixels = 224 / 192 / 160 bytes: on an even row group lk of a 14- or 10-bit stream is
    // aligned when lk is even (w % 16 == 8: odd rows start two bytes into a dword, so there it is the other way round); 12-bit groups
    // (12 bytes) always are
    const bool mis = BPP != 12 && (lk & 1) != 0;
    L.s0 = edge ? SEL_SWAP : (mis ? SEL_MIS : SEL_SWAP);
    // edge item: S1 joins the right group's third word with the left group's (14 bit: the halo's boundary lies inside word 3, 10 bit:
    // inside word 2, 12 bit: between words 2 and 3); S2 (S3) are words of the left group
    L.s1 = edge ? (BPP == 14 ? SEL_EDGE1 : (BPP == 12 ? 0x01000706u : 0x01040706u)) : L.s0;
    L.s23 = edge ? (BPP == 14 ? SEL_MIS : (BPP == 12 ? 0x05040706u : 0x05040504u)) : L.s0;
    L.flip = (VEC == 2 && !edge) ? (SEL_SWAP ^ SEL_MIS) : 0u;
    return L;
}

// plane row p of the tile at (tx0, ty0): the item's two pixel rows into d0 / d1.  Rows are whole 8-pixel groups (w % 8 == 0), so the
// byte offset of the group at (x, y) is y * pitch + (x / 8) * gb, pitch = bytes per row, gb = bytes per group (14 / 16): 24-bit
// multiplies (v_mul_lo_u32 costs four issue slots), and a main item's second load follows from its first (amask / boff, item_lane)
template <int BPP, bool TOP = false>
__device__ __forceinline__ void issue_item(uint32_t (&d0)[4], uint32_t (&d1)[4], mlv_i32x4 rs, const ItemLane &L, int w, int h, int tx0,
                                           int ty0, int p)
{
    constexpr uint32_t GB = (uint32_t)BPP;              // bytes per 8-pixel group
    const int y = ty0 - 2 * HC + 2 * p;
    const uint32_t pitch = (uint32_t)(w >> 3) * GB;                                    // scalar
    const int gmax = (w >> 3) - 1, g0 = tx0 >> 3;
    const uint32_t ga = __umul24((uint32_t)min(g0 + (L.xoff_a >> 3), gmax), GB);
    const uint32_t gb = __umul24((uint32_t)max(min(g0 + (L.xoff_b >> 3), gmax), 0), GB);
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const int yy = TOP ? clampi(y + rr, 0, h - 1) : min(y + rr, h - 1);     // (only the rows above a tile's own can lie above the frame)
        const uint32_t row = __umul24((uint32_t)yy, pitch);                            // rows and row pitch < 2^24 (launcher)
        const uint32_t oa = (row + ga) & L.amask, ob = ((row + gb) & L.amask) + L.boff;
        const mlv_u32x2 a = mlv_rbl_x2(rs, (int)oa, 0, KF_SRC_AUX), b = mlv_rbl_x2(rs, (int)ob, 0, KF_SRC_AUX);
        uint32_t (&d)[4] = rr ? d1 : d0;
        d[0] = a.x; d[1] = a.y; d[2] = b.x; d[3] = b.y;
    }
}

// pixel K (0..7) of an item from its MSB-first stream words (B bits per pixel)
template <int K, int B>
__device__ __forceinline__ uint32_t pxk(const uint32_t (&s)[4])
{
    constexpr int bit = B * K, wi = bit >> 5, sh = bit & 31;
    constexpr uint32_t mask = (1u << B) - 1u;
    if constexpr (sh + B <= 32) return (s[wi] >> (32 - B - sh)) & mask;
    else return (uint32_t)((((uint64_t)s[wi] << 32) | s[wi + 1]) >> (64 - B - sh)) & mask;
}

template <int BPP>
__device__ __forceinline__ void unpack8(const uint32_t (&d)[4], uint32_t s0, uint32_t s1, uint32_t s23, uint32_t (&px)[8])
{
    if constexpr (BPP != 16) {
        uint32_t s[4];
        s[0] = __builtin_amdgcn_perm(d[1], d[0], s0);
        s[1] = __builtin_amdgcn_perm(d[2], d[1], s1);
        s[2] = __builtin_amdgcn_perm(d[3], d[2], s23);
        s[3] = BPP == 14 ? __builtin_amdgcn_perm(d[3], d[3], s23) : 0u;       // (96 / 80 bits of a 12- / 10-bit group end inside S2)
        px[0] = pxk<0, BPP>(s); px[1] = pxk<1, BPP>(s); px[2] = pxk<2, BPP>(s); px[3] = pxk<3, BPP>(s);
        px[4] = pxk<4, BPP>(s); px[5] = pxk<5, BPP>(s); px[6] = pxk<6, BPP>(s); px[7] = pxk<7, BPP>(s);
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) { px[2 * k] = d[k] & 0xFFFFu; px[2 * k + 1] = d[k] >> 16; }
    }
}

// any geometry: one pixel with clamped coordinates
template <int BPP>
__device__ __forceinline__ uint32_t fetch_clamped(const uint8_t *frame, int w, int h, int x, int y)
{
    x = clampi(x, 0, w - 1);
    y = clampi(y, 0, h - 1);
    const uint32_t i = (uint32_t)y * (uint32_t)w + (uint32_t)x;
    if (BPP != 16) {
        const uint16_t *s = (const uint16_t *)frame;
        const uint32_t bit = i * (uint32_t)BPP;          // < 2^28 pixels per frame (launcher): fits
        const uint32_t two = ((uint32_t)s[bit >> 4] << 16) | s[(bit >> 4) + 1];
        return (two >> (32 - BPP - (bit & 15))) & ((1u << BPP) - 1u);
    }
    return ((const uint16_t *)frame)[i];
}

// slow path (w % 8 != 0, unaligned buffers): fill the two pixel rows of an item pixel by pixel
template <int BPP>
__device__ __forceinline__ void fetch_rows(const uint8_t *frame, int w, int h, int tx0, int lk, int y, bool edge, uint32_t (&p0)[8],
                                           uint32_t (&p1)[8])
{
#pragma unroll 1
    for (int k = 0; k < 8; k++) {
        // edge items: px 0..3 = right halo, px 4..7 = left halo (same layout as the fast path)
        const int xx = edge ? (k < 4 ? tx0 + 2 * TCW + k : tx0 - 8 + k) : tx0 + 8 * lk + k;
        p0[k] = fetch_clamped<BPP>(frame, w, h, xx, y);
        p1[k] = fetch_clamped<BPP>(frame, w, h, xx, y + 1);
    }
}

// pixels of one item -> planes (+ the interior pixels and green EVs of the rows that have them).  p: plane row
template <int METHOD, class SM>
__device__ __forceinline__ void emit_item(SM &sm, int black, bool dark, bool slow, int p, int lk, bool edge, const uint32_t (&p0)[8],
                                          const uint32_t (&p1)[8])
{
    const int jj = p - HC;                               // >= 0: a row whose pixels are output by this tile or the one below
    const bool keep = !edge && jj >= 0;
    if (METHOD != 0) {
        int ge[4], dr[4], db[4];
        if (!dark) cell_multi_ev_fast<4, SM::SPREAD>(p0, p1, black, sm.t16, ge, dr, db);       // wave-uniform, all three
        else if (!slow) {
            // (two cells at a time: with all sixteen pixels in flight this branch set the kernel's register peak and the loop's
            // invariants were spilt for it)
#pragma unroll
            for (int c = 0; c < 4; c += 2) {
                int g2[2], r2[2], b2[2];
                cell_multi_ev_dark<2, SM::SPREAD>(p0 + 2 * c, p1 + 2 * c, black, sm.t16, g2, r2, b2);
                ge[c] = g2[0]; ge[c + 1] = g2[1]; dr[c] = r2[0]; dr[c + 1] = r2[1]; db[c] = b2[0]; db[c + 1] = b2[1];
            }
        }
        else {
#pragma unroll
            for (int c = 0; c < 4; c += 2) {
                int g2[2], r2[2], b2[2];
                cell_pair_ev<SM::SPREAD>(p0 + 2 * c, p1 + 2 * c, black, sm.t16, true, g2, r2, b2);
                ge[c] = g2[0]; ge[c + 1] = g2[1]; dr[c] = r2[0]; dr[c + 1] = r2[1]; db[c] = b2[0]; db[c + 1] = b2[1];
            }
        }
        // main item: plane columns HC + 4 lk ..; edge item: cells 0, 1 = right halo, cells 2, 3 = left halo
        const int ca = edge ? PW - HC : HC + 4 * lk, cb = edge ? 0 : ca + 2;
        const uint32_t prow = __umul24((uint32_t)p, (uint32_t)(PW * 4));              // (v_mul_lo_u32 otherwise)
        char *pdr = (char *)&sm.dr[0][0] + prow, *pdb = (char *)&sm.db[0][0] + prow;
        *(int2 *)(pdr + 4 * ca) = make_int2(dr[0], dr[1]);
        *(int2 *)(pdr + 4 * cb) = make_int2(dr[2], dr[3]);
        *(int2 *)(pdb + 4 * ca) = make_int2(db[0], db[1]);
        *(int2 *)(pdb + 4 * cb) = make_int2(db[2], db[3]);
        if (keep) *(int4 *)&sm.ge[jj][4 * lk] = make_int4(ge[0], ge[1], ge[2], ge[3]);
    }
    if (keep) {
        *(uint4 *)&sm.raw[2 * jj][8 * lk] = make_uint4(p0[0] | (p0[1] << 16), p0[2] | (p0[3] << 16), p0[4] | (p0[5] << 16), p0[6] | (p0[7] << 16));
        *(uint4 *)&sm.raw[2 * jj + 1][8 * lk] = make_uint4(p1[0] | (p1[1] << 16), p1[2] | (p1[3] << 16), p1[4] | (p1[5] << 16), p1[6] | (p1[7] << 16));
    }
}

// ---------------------------------------------------------------- pixel map
// A tile's repaired cells arrive as records {cell, R | G1 << 16, G2 | B << 16} (k_pixfix_cells): a record's EV triple goes into the
// planes -- with the four pixels into the interior pixel rows -- once the loader's stores are behind a barrier.  (A tile's list
// covers its halo too, so the rows a tile hands down to the one below are patched again there: same values.)
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
    if (have && i >= 0 && i < PW && j >= 0 && j < PH) { c.i = i; c.j = j; }
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
    if (ii >= 0 && ii < TCW && jj >= 0 && jj < RH) {
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
    // both lanes of a pair at once: t = v - lo (wraps), excess = t -sat span (unsigned saturating: 0 when inside), any excess -> unknown
    typedef unsigned short upk16 __attribute__((ext_vector_type(2)));
    const int lo_r = ar - 32767, span_r = 65533 - 2 * ar, lo_b = ab - 32767, span_b = 65533 - 2 * ab;
    const mlv_pk16 lo_pk = { (short)lo_r, (short)lo_b };
    const upk16 span_pk = { (unsigned short)max(span_r, 0), (unsigned short)max(span_b, 0) };
    bool unknown = ar >= 32767 || ab >= 32767;           // the references themselves are more than the 16-bit range apart
    upk16 excess = { 0, 0 };
#pragma unroll
    for (int c = 0; c < STRIP; c++) {
        const upk16 t = __builtin_bit_cast(upk16, o[c][0]) - __builtin_bit_cast(upk16, lo_pk);
        excess |= __builtin_elementwise_sub_sat(t, span_pk);
        mr[c] = (int)o[c][0].x + g.ref_r;
        mb[c] = (int)o[c][0].y + g.ref_b;
    }
    unknown |= __builtin_bit_cast(uint32_t, excess) != 0u;
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
    constexpr bool CHAIN = METHOD == 5;
    using Smem = SmemT<SPREAD, CHAIN>;
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

    // Persistent tile walk.  The tile list runs DOWN the columns of a frame (tile tt of a frame: column tt / tiles_y, row tt %
    // tiles_y), frame after frame.  Group g = blockIdx % groups: with 4 x CUs workgroups in the grid the dispatcher puts blocks
    // b, b + CUs, b + 2 CUs, b + 3 CUs on one CU (tools/hwid_probe.hip), so a group is one CU's four residents.  Each group
    // owns a contiguous range of the tile list -- the groups of an XCD (blocks b, b + 8, ... share one) next to each other,
    // so halo re-reads hit that XCD's L2 -- and its members draw RUNS of `run` consecutive tiles from it (one atomicAdd per
    // run, by one thread, issued a whole tile ahead): a tile that lies right below the tile its workgroup did before finds
    // its upper four plane rows in LDS.  The last `singles` tiles of a range go out one by one, so that the four workgroups
    // of a CU end together (with a fixed share each the slowest one ended up alone on its CU: the SIMD arbiter favours the
    // oldest waves).  Whatever the placement and whoever draws what, every tile is drawn exactly once.
    const int tiles_per_frame = a.tiles_x * a.tiles_y;
    const int total = tiles_per_frame * a.nframes;          // < 2^30 (checked by the launcher)
    const int nx = 8;
    const int grp = blockIdx.x % a.groups;
    const int gpx = (a.groups + nx - 1) / nx;                                    // groups per XCD
    const int grank = (a.groups % nx == 0) ? (grp % nx) * gpx + grp / nx : grp;  // position of the group's range in the tile list
    const int gq = total / a.groups, grem = total - gq * a.groups;
    const int band_start = grank * gq + min(grank, grem);
    const int band_end = band_start + gq + (grank < grem ? 1 : 0);
    const int run = max(a.run, 1);
    const int runs_len = max(band_end - band_start - a.singles, 0) / run * run;  // tiles of the range that go out in runs
    const int black16 = (int)(uint16_t)a.black, white16 = (int)(uint16_t)a.white;
    constexpr bool vec = VEC != 0;                       // w % 8 == 0, aligned buffers: vector loads and stores
    constexpr int BPP = bpp_of(PACKED, VEC);             // bits per pixel of the input
    constexpr int NEW0 = METHOD == 0 ? HC : 2 * HC;      // first plane row a tile loads itself (without chroma smoothing: no halo at all)

    const mlv_i32x4 rs_e2d = table_rsrc(a.e2d, 8, E2D_RECORDS);
    uint32_t r0[4], r1[4];                               // prefetch registers of this thread's item
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // loader: threads 0..239 own the main item (row t / 16, group t % 16) of the tile's new rows, threads 240..254 the edge items
    const bool l_edge = tid >= N_MAIN;
    const int l_row = l_edge ? tid - N_MAIN : tid >> 4, l_k = tid & 15;
    const ItemLane IL = item_lane<PACKED, VEC>(l_k, l_edge);
    // median phase: lane -> (row j, strip k).  5x5 (neighbour sharing): 17 consecutive lanes per tile row -- its 16 strips and, as
    // the 17th, the group of plane columns 64..67 (halo) that the last strip needs --, so that EVERY lane finds the group to its
    // right in the next lane; the plane reads are then linear in the thread number (16 bytes per lane: PW = 17 x 4) and free of
    // bank conflicts.  Other methods: 16 lanes per row.
    const int j_ = CHAIN ? (tid * 241) >> 12 : tid >> 4;                     // tid / 17 for tid < 256
    const int k = CHAIN ? tid - 17 * j_ : tid & 15;
    const int j = min(j_, TCH - 1);
    const bool is_strip = CHAIN ? (k < 16 && tid < 17 * TCH) : tid < N_MAIN;
    // rows handed down to the tile below (threads 0..67: dr rows TCH.. -> 0..3, 68..135: db, 136..199: pixel rows, 200..231: green EVs)
    constexpr int C_PL = 2 * HC * PW * 4 / 16, C_RAW = 2 * HC * 2 * TCW * 2 / 16, C_GE = HC * TCW * 4 / 16, C_ALL = 2 * C_PL + C_RAW + C_GE;
    static_assert(C_ALL <= 256, "one 16-byte piece per thread");
    int c_dst, c_delta;
    if (tid < C_PL) { c_dst = (int)offsetof(Smem, dr) + 16 * tid; c_delta = TCH * PW * 4; }
    else if (tid < 2 * C_PL) { c_dst = (int)offsetof(Smem, db) + 16 * (tid - C_PL); c_delta = TCH * PW * 4; }
    else if (tid < 2 * C_PL + C_RAW) { c_dst = (int)offsetof(Smem, raw) + 16 * (tid - 2 * C_PL); c_delta = 2 * TCH * 2 * TCW * 2; }
    else { c_dst = (int)offsetof(Smem, ge) + 16 * (tid - 2 * C_PL - C_RAW); c_delta = TCH * TCW * 4; }

    bool singly = false;                                 // thread 0: the runs of this group's range are all handed out
    auto draw = [&](int &nt, int &ne) {                  // thread 0: the next run of the group's range, or its next single tile
        if (!singly) {
            const int p = atomicAdd(&a.tickets[2 * grp], run);
            if (p < runs_len) { nt = band_start + p; ne = nt + run; return; }
            singly = true;
        }
        const int q = atomicAdd(&a.tickets[2 * grp + 1], 1);
        nt = min(band_start + runs_len + q, band_end);
        ne = nt + 1;
    };
    if (threadIdx.x == 0) {
        int nt, ne;
        draw(nt, ne);
        sm.next_tile = nt; sm.next_end = ne;
        sm.dark_items[0] = 0; sm.dark_items[1] = 0; sm.fb_count = 0;
    }
    __syncthreads();
    int t = __builtin_amdgcn_readfirstlane(sm.next_tile), t_end = __builtin_amdgcn_readfirstlane(sm.next_end);
    // Where a tile lies: frame, tile column, tile row.  Two divisions by run-time numbers -- forty scalar instructions and two
    // reciprocals that would live in scalar registers through the whole loop -- only where a run starts; inside a run the next tile
    // is one step further down (or at the top of the next column / frame).  The divisors are made opaque INSIDE the function, so
    // that the compiler does not hoist their reciprocals out of the loop.
    struct Pos { int f, tcol, trow; };
    auto pos_of = [&](int tt) {
        int tpf = tiles_per_frame, tys = a.tiles_y;
        asm volatile("" : "+s"(tpf), "+s"(tys));
        Pos p;
        p.f = tt / tpf;
        const int r = tt - p.f * tpf;
        p.tcol = r / tys;
        p.trow = r - p.tcol * tys;
        return p;
    };
    auto pos_below = [&](Pos p) {
        if (++p.trow == a.tiles_y) {
            p.trow = 0;
            if (++p.tcol == a.tiles_x) { p.tcol = 0; p.f++; }
        }
        return p;
    };
    // The prefetch is unconditional on purpose: threads without an item and the last
    // iteration re-load a valid item / tile.  A conditional load would need the old
    // register value on the other path, and the copies the compiler inserts for that
    // merge wait for the load right where it is issued.
    auto issue_tile = [&](const Pos &p) {
        const mlv_i32x4 rs = frame_rsrc(a.src + (size_t)p.f * a.src_stride, (unsigned)a.src_bytes);
        issue_item<BPP>(r0, r1, rs, IL, a.w, a.h, p.tcol * 2 * TCW, p.trow * 2 * TCH, NEW0 + l_row);
    };
    Pos cur = pos_of(min(t, max(total - 1, 0)));
    if (vec) issue_tile(cur);
    __syncthreads();                           // T16 copy complete

#ifdef KF_DIAG_TIMES
    const uint64_t rt0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long diag_n[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#endif
    int par = 0;                               // tile parity: which of the two dark_items counters this tile uses
    int fb_skip = 0, fb_wait = FB_WAIT_MIN;    // 5x5: tiles still to go straight to the 32-bit chain; how many after the next busy tile
    bool robust = false;                       // 5x5: rows of lanes agree on their references (robust_ref)
    int calm = 0;                              //      tiles in a row without an uncertain strip
    bool cont = false;                         // this tile lies right below the one this workgroup did before: its upper rows are in LDS
    while (t < band_end) {
        int nt = 0, ne = 0;
        if (threadIdx.x == 0) {                // the tile after this one: known, or drawn now and back long before it is needed
            if (t + 1 < t_end) { nt = t + 1; ne = t_end; }
            else draw(nt, ne);
        }
        const int f = cur.f, trow = cur.trow, tx0 = cur.tcol * 2 * TCW, ty0 = cur.trow * 2 * TCH;
        const int tr = cur.trow * a.tiles_x + cur.tcol;          // the tile's number in the (row-major) pixel-map lists
        const uint8_t *frame = a.src + (size_t)f * a.src_stride;
        uint16_t *out = (uint16_t *)(a.dst + (size_t)f * a.dst_stride);
        // ---- pixel-map entries of this tile (few tiles have any): list bounds now -- the wait that the uniform load implies is
        // for data the loader needs anyway --, the first 256 records themselves in flight while the loader phase runs
        const bool tile_patched = a.patch && (!pmap_ok || ((sm.has_patch[tr >> 5] >> (tr & 31)) & 1u));      // wave-uniform
        int pbeg = 0, pend = 0;
        int4 my_rec = make_int4(-1, 0, 0, 0);
        const int4 *cells = a.cells + (size_t)f * a.n_rec;
        if (tile_patched) {
            pbeg = a.tile_off[tr];
            pend = a.tile_off[tr + 1];
            if (pbeg + tid < pend) my_rec = cells[pbeg + tid];
        }
        // ---- loader: prefetched registers -> EV planes + interior raw pixels
#if !defined(KF_PRIO) || KF_PRIO == 1
        __builtin_amdgcn_s_setprio(0);
#elif KF_PRIO == 2
        __builtin_amdgcn_s_setprio(1);
#endif
        // Lane predicates and wave-uniform switches are re-derived per tile from opaque copies: hoisted out of the loop they
        // became 64-bit SGPR masks, two scalar registers each, of which the kernel kept more than it has -- they were spilt
        // to VGPR lanes and came back through v_readlane, VECTOR instructions (about 40 per tile and wave).
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
        // one item: stream words -> pixels -> EV triples -> planes.  d0 / d1: its two rows as loaded; p: plane row
        auto do_item = [&](const ItemLane &L, const uint32_t (&d0)[4], const uint32_t (&d1)[4], int p, int lk) {
            uint32_t p0[8], p1[8];
            if (vec) {
                unpack8<BPP>(d0, L.s0, L.s1, L.s23, p0);
                unpack8<BPP>(d1, L.s0 ^ L.flip, L.s1 ^ L.flip, L.s23 ^ L.flip, p1);
            } else fetch_rows<BPP>(frame, a.w, a.h, tx0, lk, ty0 - 2 * HC + 2 * p, L.edge, p0, p1);
            // Pixels at or below black (ev = INT_MIN / 0) or beyond the table need the fix-ups of cell_pair_ev: decided once
            // per item from the extremes of its 16 pixels (three-input min/max), wave-uniformly.
            // and, 16-bit input only, what lies beyond the table.
            bool odd = false, beyond = false;
            if (METHOD != 0) {
                uint32_t lo = min(p0[0], p1[0]), hi = max(p0[0], p1[0]);
#pragma unroll
                for (int i = 1; i < 8; i++) {
                    lo = min(min(lo, p0[i]), p1[i]);
                    if (!PACKED) hi = max(max(hi, p0[i]), p1[i]);
                }
                odd = (int)lo <= a.black;
                beyond = (!PACKED && (int)hi - a.black > 16383) || (PACKED && a.black < 0);
            }
#ifdef KF_EXP_FASTLOADER
            const bool slow = false, dark = false;
#else
            const bool slow = (!PACKED || a.black < 0) && __any(beyond);
            const bool dark = slow || __any(odd);
#endif
            if (METHOD == 5 && SPREAD && dark) {
                const unsigned long long who = __ballot(odd || beyond);
                if (lane == 0) atomicAdd(&sm.dark_items[par], __popcll(who));
            }
            emit_item<METHOD, Smem>(sm, a.black, dark, slow, p, lk, L.edge, p0, p1);
        };
        if (METHOD != 0 && !cont) {
            // the first tile of a run (or of a column): the four plane rows above the tile's own, straight from memory -- threads
            // 0..63 the main items of rows 0..3, 64..67 their edge items -- while the other waves convert the prefetched rows
            if (tid_o < N_TOP) {
                const bool te = tid_o >= N_TOP_MAIN;
                const int trw = te ? tid_o - N_TOP_MAIN : tid_o >> 4;
                const ItemLane TL = item_lane<PACKED, VEC>(l_k, te);
                uint32_t q0[4] = { 0, 0, 0, 0 }, q1[4] = { 0, 0, 0, 0 };
                if (vec) issue_item<BPP, true>(q0, q1, frame_rsrc(frame, (unsigned)a.src_bytes), TL, a.w, a.h, tx0, ty0, trw);
                do_item(TL, q0, q1, trw, l_k);
            }
        }
        const bool has_item = METHOD == 0 ? tid_o < N_MAIN : tid_o < N_ITEMS;
        if (has_item) do_item(IL, r0, r1, NEW0 + l_row, l_k);
        if (threadIdx.x == 0) { sm.next_tile = nt; sm.next_end = ne; }
        lds_barrier();
        const int t_next = __builtin_amdgcn_readfirstlane(sm.next_tile), t_end_next = __builtin_amdgcn_readfirstlane(sm.next_end);
        // the tile after this one continues it when it is the next of the list and not the top of a column
        const bool cont_next = METHOD != 0 && t_next == t + 1 && trow + 1 < a.tiles_y && t_next < band_end;       // scalar
        if (tile_patched) {
            PatchCell c = patch_cell<METHOD, PACKED, Smem>(sm, a, my_rec, tx0, ty0);
            patch_store<METHOD, Smem>(sm, c);
            for (int base = pbeg + 256; base < pend; base += 256) {         // a dense map (focus pixels): the rest
                const int4 rec = base + tid < pend ? cells[base + tid] : make_int4(-1, 0, 0, 0);
                c = patch_cell<METHOD, PACKED, Smem>(sm, a, rec, tx0, ty0);
                patch_store<METHOD, Smem>(sm, c);
            }
            lds_barrier();
        }
        // Waves that are past the loader issue ahead of waves (of the CU's other workgroups) that are still in it: a tile that
        // is about to finish finishes sooner, its workgroup's barrier opens sooner, and the loader instructions of the others
        // fill the gaps.  Measured (tools/kbench.py): cs5x5 11.2-11.5 -> 10.2-10.4 us per frame, cs2x2 8.2 -> 7.8; which of the
        // levels 1..3 is used, and a third level for the output stage, make no difference.
#if !defined(KF_PRIO) || KF_PRIO == 1
        if (METHOD != 0) __builtin_amdgcn_s_setprio(1);       // (without chroma smoothing the kernel is load-bound and this costs 8 %)
#elif KF_PRIO == 2
        if (METHOD != 0) __builtin_amdgcn_s_setprio(0);
#endif
        // ---- prefetch the next tile while the medians run
        // (scalar branch: the first tile of another run -- or nothing left: the prefetch is unconditional, so it gets a valid tile;
        // the tile "below" the list's last one would lie in a frame behind the buffer)
        Pos nxt = pos_below(cur);
        if (t_next != t + 1 || t_next >= band_end) nxt = pos_of(min(t_next, band_end - 1));
        if (vec) issue_tile(nxt);

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
                // the output pixel by EV (112 KiB table, one 8-byte record per 32 EV steps): all 8 look-ups issued before the first use
                int er[STRIP], eb[STRIP], ur[STRIP], ub[STRIP], cr[STRIP], cb[STRIP];
                mlv_tab_u32x2 dr2[STRIP], db2[STRIP];
#pragma unroll
                for (int c = 0; c < STRIP; c++) {
                    er[c] = wadd(gev[c], mr[c]);
                    eb[c] = wadd(gev[c], mb[c]);
                    cr[c] = min(max(er[c], 0), MLV_EV_MAX); cb[c] = min(max(eb[c], 0), MLV_EV_MAX);
#ifdef KF_EXP_NOLOOKUP
                    dr2[c].x = cr[c]; dr2[c].y = 0; db2[c].x = cb[c]; db2[c].y = 0;
#else
                    dr2[c] = mlv_sbl_x2(rs_e2d, cr[c] >> 5, 0, 0, KF_E2R_AUX);
                    db2[c] = mlv_sbl_x2(rs_e2d, cb[c] >> 5, 0, 0, KF_E2R_AUX);
#endif
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
                asm volatile("" :: "v"(dr2[0]), "v"(dr2[1]), "v"(dr2[2]), "v"(dr2[3]), "v"(db2[0]), "v"(db2[1]), "v"(db2[2]), "v"(db2[3]));
#pragma unroll
                for (int c = 0; c < STRIP; c++) {             // v_bfe_u32 (the width operand's low five bits count), v_bcnt_u32_b32
                    ur[c] = (int)(__builtin_popcount(__builtin_amdgcn_ubfe(dr2[c].y, 0u, (unsigned)cr[c] & 31u)) + dr2[c].x);
                    ub[c] = (int)(__builtin_popcount(__builtin_amdgcn_ubfe(db2[c].y, 0u, (unsigned)cb[c] & 31u)) + db2[c].x);
                }
#pragma unroll
                for (int c = 0; c < STRIP; c++) {
                    const bool ok = okc[c];
                    // R = the lower half of top, B = the upper half of bot: one v_perm_b32 each (selector: new value or the word as it is)
                    top[c] = __builtin_amdgcn_perm((uint32_t)ur[c], top[c], ok ? 0x03020504u : 0x03020100u);
                    bot[c] = __builtin_amdgcn_perm((uint32_t)ub[c], bot[c], ok ? 0x05040100u : 0x03020100u);
                }
            }
            if (stripe_mode != 0) {
                // a strip starts at an x that is a multiple of 8, so pixel n of the strip has column phase n
                int co[8];
#pragma unroll
                for (int i = 0; i < 8; i++) { co[i] = a.coef[i]; asm volatile("" : "+s"(co[i])); }
#ifdef KF_EXP_PKONLY
                stripe_strip_pk(top, bot, co, black16, white16);
#else
                if (stripe_mode == 1) stripe_strip_pk(top, bot, co, black16, white16);
                else if (stripe_mode == 2) stripe_strip<true>(top, bot, co, black16, white16);
                else stripe_strip<false>(top, bot, co, black16, white16);
#endif
            }
            if (store && y < a.h) {
                if (vec) {
                    if (x < a.w) {
                        const uint32_t o = (uint32_t)y * (uint32_t)a.w + (uint32_t)x;           // < 2^28 pixels per frame
                        // non-temporal stores: the output is not read again by this launch (same-box A/B, profiles/r04/ab_cache_policy.log: cs2x2
                        // -1.3 ... -3.5 %, cs5x5 -1 ... -2.6 %; the stream LOADS marked nt or sc1 are 3-9 % slower: a line serves two loads)
                        typedef unsigned kf_u4 __attribute__((ext_vector_type(4)));
                        const kf_u4 vt = { top[0], top[1], top[2], top[3] }, vb = { bot[0], bot[1], bot[2], bot[3] };
                        __builtin_nontemporal_store(vt, (kf_u4 *)(out + o));
                        if (y + 1 < a.h) __builtin_nontemporal_store(vb, (kf_u4 *)(out + o + (uint32_t)a.w));
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
#ifdef KF_EXP_NOFALLBACK
            skip_packed = false;
#endif
            if (fb_skip > 0) fb_skip--;
            if (SPREAD && tid == 0) sm.dark_items[par ^ 1] = 0;
            bool unknown = true;
            // a wave's first lane holds the group that the last lane of the wave before it needs: through LDS
            const bool publishes = lane == 0 && tid != 0, collects = lane == 63 && tid < 192;
            const int wv = tid >> 6;
            if (skip_packed) {
                // a tile that skips the packed attempt: both planes through the 32-bit chain, one after the other (the
                // exchange records are reused)
#pragma unroll 1
                for (int pln = 0; pln < 2; pln++) {
                    Chain32 g;
                    chain32_group(pln ? sm.db : sm.dr, j, STRIP * k, g);
                    if (publishes) chain32_publish(g, sm.xchg[wv - 1]);
                    lds_barrier();
                    Next32 n;
                    chain32_fetch(g, n);
                    if (collects) chain32_collect(sm.xchg[wv], n);
                    if (pln) chain32_finish(g, n, mb);
                    else {
                        chain32_finish(g, n, mr);
                        lds_barrier();                         // every record has been read: the second plane may overwrite it
                    }
                }
                unknown = false;
            } else {
                ChainGroup g;
                chain_group(sm.dr, sm.db, j, STRIP * k, robust, g);    // a strip's own group, or (k = 16) a row's halo group
                const bool early = tid >= 64;                          // the waves that publish (uniform): their rank window first
                if (early) {
                    chain_group_window(g);
                    if (publishes) chain_publish(g, sm.xchg[wv - 1]);
                }
                lds_barrier();                                         // the records are in LDS
                ChainNext n;
                chain_fetch_lists(g, n);
                if (collects) chain_collect_lists(sm.xchg[wv], n);
                if (!early) chain_group_window(g);                     // the lane's own rank window, while those reads are under way
                chain_fetch_window(g, n);
                if (collects) chain_collect_window(sm.xchg[wv], n);
                unknown = chain_finish(g, n, mr, mb);
            }
            unknown = unknown && is_strip && smooth_row;
#ifdef KF_EXP_NOFALLBACK
            unknown = false;
#endif
            if (unknown) sm.fb_queue[atomicAdd(&sm.fb_count, 1)] = (uint8_t)(j * 16 + k);      // settled below, densely
            if (is_strip && !unknown) finish_strip(j, k, smooth_row, mr, mb, true);
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
            if (is_strip) finish_strip(j, k, smooth_row, mr, mb, true);
        }
        // ---- the rows the tile below shares with this one: read before the barrier that ends the tile, stored behind it
        int4 carry = make_int4(0, 0, 0, 0);
        int tid_c = tid;
        asm volatile("" : "+v"(tid_c));
        const bool do_carry = cont_next && tid_c < C_ALL;
        if (do_carry) carry = *(const int4 *)((const char *)&sm + c_dst + c_delta);
        if (CHAIN) {
            // Strips whose packed medians were not certain: all of the tile's, gathered in LDS, go through the 32-bit networks
            // one per lane -- as many waves as ceil(count / 64) run them, instead of every wave that had one such strip.
            lds_barrier();
#if defined(KF_EXP_LEAN) || defined(KF_EXP_NOSETTLE)
            const int nfb = 0;
#else
            const int nfb = __builtin_amdgcn_readfirstlane(sm.fb_count);
#endif
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
            // tiles, tiles that skipped the packed networks, strips settled densely, tiles that continued the one above (summed per
            // workgroup, added once at its end: an atomic per tile on one address serialised the whole launch)
            diag_n[0]++; diag_n[1] += skip_packed ? 1 : 0; diag_n[2] += nfb; diag_n[3] += cont ? 1 : 0;
            diag_n[4] += nfb > 0 ? 1 : 0; diag_n[5] += robust ? 1 : 0; diag_n[6] += nfb > 64 ? 1 : 0; diag_n[7] += (nfb > 0 && nfb <= 4) ? 1 : 0;
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
        if (do_carry) *(int4 *)((char *)&sm + c_dst) = carry;
        t = t_next;
        t_end = t_end_next;
        cont = cont_next;
        cur = nxt;
        if (SPREAD) par ^= 1;
    }
    if (threadIdx.x == 0 && atomicAdd(&a.tickets[2 * a.groups], 1) == (int)gridDim.x - 1) {
        for (int i = 0; i <= 2 * a.groups; i++) a.tickets[i] = 0;   // last workgroup out: ready for the next launch on this stream
    }
#ifdef KF_DIAG_TIMES
    if (threadIdx.x == 0 && a.times) {
        a.times[2 * blockIdx.x] = rt0; a.times[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 8; i++) atomicAdd(&a.times[4096 + i], diag_n[i]);
    }
#endif
}

// ---------------------------------------------------------------- host launcher
// zeroed counters per stream (launches on one stream run one after the other and leave the counters zeroed)
constexpr int MAX_GROUPS = 1024;
#ifndef KF_RUN_MAX
#define KF_RUN_MAX 22
#endif
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
        if (hipMalloc(&p, (2 * MAX_GROUPS + 1) * sizeof(int)) != hipSuccess ||
            hipMemsetAsync(p, 0, (2 * MAX_GROUPS + 1) * sizeof(int), stream) != hipSuccess) {
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
// E2R[ev] = (uint16)(ev2raw[ev] + black) for ev in [0, 14 * 32768): exactly what chroma_smooth.c:67-68 stores for a clamped EV, in
// the dense form described at E2D_RECORDS.  One look-up replaces mask, address, quotient, two shifts, add and mask per output pixel
// (cs5x5 -2.5 %, A/B in profiles/r02/ab_table_gathers_kbench.log: "e1").  Built on the device from the exact 16-bit re-encoding U16
// (common.h) the first time a black level is seen on a device; *bad counts entries that rise by more than one (never).
// (The same trick for the loader -- raw2ev by pixel value from a 256 KiB table, no conversion arithmetic at all, results
// identical -- makes the kernel wait for the texture addresser instead: cs5x5 +11 %, cs2x2 +27 %, "e2" in the same log.)
__global__ __launch_bounds__(256) void k_build_e2d(const uint16_t *__restrict__ u16, int black, uint2 *__restrict__ e2d, int *__restrict__ bad)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= E2D_RECORDS) return;
    auto entry = [&](int i) { return (uint16_t)((((int)u16[i & 32767]) >> (13 - (i >> 15))) + black); };
    unsigned bits = 0;
    uint16_t prev = entry(32 * b);
    const uint16_t base = prev;
    for (int j = 0; j < 31; j++) {
        const uint16_t next = entry(32 * b + j + 1);
        const unsigned d = (uint16_t)(next - prev);
        if (d > 1) atomicAdd(bad, 1);
        bits |= (d & 1u) << j;
        prev = next;
    }
    e2d[b] = make_uint2(base, bits);
}
// The black level comes from file headers (and dual ISO multiplies it by 4): a long-running host that serves many clips would
// otherwise collect one table per level it has ever seen.  At most E2R_CACHE tables per device stay; the least recently used one
// goes -- not at once: a thread that fetched its pointer a moment ago may not have launched yet, so an evicted table is parked and
// freed at the NEXT eviction on that device, after the device has drained (the same rule as DeviceTables::retire in dualiso.cpp).  The table of a new level is built OUTSIDE the lock, so first launches of
// different clips do not queue behind each other's synchronisation.
namespace {
constexpr size_t E2R_CACHE = 8;
struct E2rEntry { uint2 *table; unsigned long long used; };
std::mutex g_tables_mu;
std::map<std::pair<int, int>, E2rEntry> g_e2r;               // (device, black)
std::map<int, uint2 *> g_e2r_parked;                       // per device: the table evicted last
unsigned long long g_e2r_clock = 0;
}
static int e2r_table(const Device *dev, int black, const uint2 **out, hipStream_t stream)
{
    {
        std::lock_guard<std::mutex> lk(g_tables_mu);
        auto it = g_e2r.find({ dev->id, black });
        if (it != g_e2r.end()) { it->second.used = ++g_e2r_clock; *out = it->second.table; return MLVFS_AMD_OK; }
    }
    uint2 *t = nullptr;
    MLV_HIP(hipMalloc(&t, sizeof(uint2) * (E2D_RECORDS + 1)));          // the records, then the counter of the check
    int *d_bad = (int *)(t + E2D_RECORDS);
    int bad = -1;
    (void)hipMemsetAsync(d_bad, 0, sizeof(int), stream);
    hipLaunchKernelGGL(k_build_e2d, dim3((E2D_RECORDS + 255) / 256), dim3(256), 0, stream, dev->luts.u16, black, t, d_bad);
    (void)hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, stream);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(stream) != hipSuccess || bad != 0) {       // other streams use the table from now on
        (void)hipFree(t);
        set_error("building the output table for black level %d failed", black);
        return MLVFS_AMD_ERR_HIP;
    }
    uint2 *victim = nullptr;
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
    auto kern = k_frame<METHOD, PACKED, VEC, SPREAD>;
    FrameArgs a = a_in;
    a.tickets = ticket_counters(stream);
    if (!a.tickets) return MLVFS_AMD_ERR_HIP;
    // Groups: workgroups that draw from one range of the tile list.  Until round 4 a group was one CU's four residents; the CUs
    // of a chip do not run at one speed (their workgroups ended between 780 and 835 us of an 844-us launch: 5.5 % of the launch
    // was its tail, -DKF_DIAG_TIMES), and drawing runs instead of single tiles made the atomics rare enough for larger groups:
    // eight CUs (a quarter of an XCD: blocks b, b + groups, ... share b % 8, i.e. their XCD, as long as groups is a multiple of 8)
    // share a range, 123.0 -> 128.4 k fps; 16 / 24 / 32 / 64 groups and runs of 11 / 22 / 44 tiles are within 0.5 % of each other,
    // one group per XCD (8) loses the gain to its 1 408 single tiles (profiles/r04/ab_groups.log).  MLVFS_AMD_KF_GROUPS overrides.
    const int per_cu = std::min(std::max(grid / 4, 1), MAX_GROUPS);
    a.groups = per_cu >= 64 ? per_cu / 8 / 8 * 8 : per_cu;
    static const int env_groups = [] { const char *e = getenv("MLVFS_AMD_KF_GROUPS"); return e ? atoi(e) : 0; }();
    if (env_groups > 0 && env_groups <= per_cu) a.groups = env_groups;
    // tiles per run and tiles that go out one by one at the end of a group's range (tools/kbench.py sweeps them: KB_RUN / KB_SINGLES)
    static const int env_run = [] { const char *e = getenv("MLVFS_AMD_KF_RUN"); return e ? atoi(e) : 0; }();
    static const int env_singles = [] { const char *e = getenv("MLVFS_AMD_KF_SINGLES"); return e ? atoi(e) : -1; }();
    // default: at most half a column of the benchmark's geometry per run (same-box sweep with one CU per group: 4 / 8 / 11 / 22 / 44
    // tiles per run -> 117.2 / 117.5 / 117.7 / 118.0-121.0 / 120.5 k fps), a sixteenth of the range for short launches
    const int band = (int)(total / a.groups);
    a.run = env_run > 0 ? env_run : std::min(std::max(band / 16, 1), KF_RUN_MAX);
    a.singles = env_singles >= 0 ? env_singles : (a.run > 1 ? 8 * std::max(grid / a.groups, 1) : 0);      // eight per workgroup of the group
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
            unsigned long long fbc[8];
            hipMemcpy(fbc, d_times + 4096, sizeof(fbc), hipMemcpyDeviceToHost);
            fprintf(stderr, "KF_TIMES tiles %llu, of which %llu skipped the packed networks and %llu continued the tile above; strips settled densely %llu (%.1f %% of all)\n", fbc[0], fbc[1],
                    fbc[3], fbc[2], fbc[0] ? 100.0 * fbc[2] / (240.0 * fbc[0]) : 0.0);
            fprintf(stderr, "KF_TIMES tiles with uncertain strips %llu (1..4 strips: %llu, more than 64: %llu); tiles on shared references %llu\n", fbc[4], fbc[7], fbc[6], fbc[5]);
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

// Which packed streams the fused kernel reads itself: 14 bits in any geometry; 12 bits (8 pixels = 12 bytes: every group
// dword-aligned) with rows of whole groups; 10 bits (8 pixels = 10 bytes) with rows of whole 16-pixel groups (every row then
// starts dword-aligned) -- the last two on 16-byte aligned buffers.  Everything else takes an unpack pass to 16 bits first.
bool frame_kernel_takes(const Geom &g, const void *src, size_t src_stride, const void *dst, size_t dst_stride, int nframes)
{
    if (g.bpp == 14) return true;
    if (g.bpp != 12 && g.bpp != 10) return false;
    const bool aligned = ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0 && (nframes == 1 || (src_stride % 16 == 0 && dst_stride % 16 == 0));
    return aligned && g.w >= 16 && g.w % (g.bpp == 12 ? 8 : 16) == 0;
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
    if (packed && !frame_kernel_takes(g, src, src_stride, dst, dst_stride, nframes)) {
        set_error("fused path: %d-bit input at %dx%d needs an unpack pass first", g.bpp, g.w, g.h);
        return MLVFS_AMD_ERR_ARG;
    }
    FrameArgs a{};
    a.src = (const uint8_t *)src; a.src_stride = src_stride;
    a.src_bytes = (unsigned)(((size_t)g.w * g.h * (packed ? g.bpp : 16) / 8 + 3) / 4 * 4);  // one frame as the loader's range-checked buffer
    a.dst = (uint8_t *)dst; a.dst_stride = dst_stride;
    a.w = g.w; a.h = g.h; a.black = g.black; a.white = g.white;
    a.nframes = nframes;
    a.tiles_x = frame_tiles_x(g.w);
    a.tiles_y = frame_tiles_y(g.h, geo);
    a.t16 = dev->luts.t16;
    if (method != 0) {
        int rc = e2r_table(dev, g.black, &a.e2d, stream);
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
    // vector path: rows are whole 8-pixel groups (14 bytes of stream) and the buffers 16-byte aligned: every row of a 16-pixel-multiple
    // width starts dword-aligned (1); widths that are 8 mod 16 alternate between dword-aligned rows and rows that start in the upper
    // half of a dword (2; an even height keeps the frame's last group off the end of the buffer)
    static const bool no_half8 = [] { const char *e = getenv("MLVFS_AMD_KF_HALF8"); return e && atoi(e) == 0; }();      // (A/B: the any-geometry path instead)
    const bool strides_ok = nframes == 1 || (src_stride % 16 == 0 && dst_stride % 16 == 0);
    int vec = ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0 && strides_ok && g.w >= 16
                        ? ((g.w % 16) == 0 ? 1 : ((g.w % 16) == 8 && (g.h % 2) == 0 && !no_half8 ? 2 : 0)) : 0;
    if (packed && g.bpp == 12) vec = 3;                  // (frame_kernel_takes has checked the geometry)
    if (packed && g.bpp == 10) vec = 4;
#define MLV_DISPATCH_S(M, S)                                                                              \
    return packed ? (vec == 1 ? launch_frame_t<M, true, 1, S>(a, dev->num_cu, stream)                     \
                   : vec == 2 ? launch_frame_t<M, true, 2, S>(a, dev->num_cu, stream)                     \
                   : vec == 3 ? launch_frame_t<M, true, 3, S>(a, dev->num_cu, stream)                     \
                   : vec == 4 ? launch_frame_t<M, true, 4, S>(a, dev->num_cu, stream)                     \
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
template <int BPP>
__global__ __launch_bounds__(256) void k_dark_share(const uint8_t *frame, int w, int h, int black, int *counts)
{
    const int nx = (w + 6) / 7, ny = (h + 4) / 5;
    int dark = 0, all = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nx * ny; i += gridDim.x * blockDim.x) {
        const int y = (i / nx) * 5, x = (i % nx) * 7;
        const int lin = (int)fetch_clamped<BPP>(frame, w, h, x, y) - black;
        dark += lin >= 1 && lin < 512;
        all++;
    }
    for (int o = 32; o > 0; o >>= 1) { dark += __shfl_xor(dark, o); all += __shfl_xor(all, o); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&counts[0], dark); atomicAdd(&counts[1], all); }
}

// packed_bpp: 0 = 16-bit frames, else the bits per pixel of the packed stream (14, 12, 10)
int dark_share(int packed_bpp, const void *d_frame, int w, int h, int black, hipStream_t stream, int *share_1024)
{
    if (packed_bpp != 0 && packed_bpp != 14 && packed_bpp != 12 && packed_bpp != 10) { set_error("dark_share: %d-bit stream", packed_bpp); return MLVFS_AMD_ERR_ARG; }
    // the two counters live as long as the thread (an allocation and a release per clip were a fifth of this call's time)
    struct Counts {
        std::map<int, int *> m;
        ~Counts() { for (auto &kv : m) if (kv.second) (void)hipFree(kv.second); }
    };
    static thread_local Counts t_counts;
    int dev = 0, hc[2] = { 0, 0 };
    MLV_HIP(hipGetDevice(&dev));
    int *&d_counts = t_counts.m[dev];
    if (!d_counts) MLV_HIP(hipMalloc(&d_counts, 2 * sizeof(int)));
    hipError_t e = hipMemsetAsync(d_counts, 0, 2 * sizeof(int), stream);
    if (e == hipSuccess) {
        auto kern = packed_bpp == 14 ? k_dark_share<14> : packed_bpp == 12 ? k_dark_share<12> : packed_bpp == 10 ? k_dark_share<10> : k_dark_share<16>;
        hipLaunchKernelGGL(kern, dim3(64), dim3(256), 0, stream, (const uint8_t *)d_frame, w, h, black, d_counts);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(hc, d_counts, sizeof(hc), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) { set_error("dark_share: %s", hipGetErrorString(e)); return MLVFS_AMD_ERR_HIP; }
    *share_1024 = hc[1] > 0 ? (int)((long long)hc[0] * 1024 / hc[1]) : 0;
    return MLVFS_AMD_OK;
}


// the first launch of any kernel of this file loads the file's code object (HIP loads them lazily): the device context asks for a
// kernel's attributes when it is created, so that a clip's first frame does not pay for it (runtime.cpp: get_device)
void preload_k_frame() { hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, (const void *)k_build_e2d); (void)hipGetLastError(); }

}  // namespace mlv
