// k_frame_dev.h -- device code shared by the two kernels of the fused pass (k_frame.hip, k_frame_p.hip): arguments, LDS layouts,
// loader, selection networks, stripes epilogue.  See k_frame.hip for the description of the pass.
#pragma once
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
    // work list between the two kernels of a launch (round 5): k_frame_p pushes the runs of tiles it does not settle itself
    // {first tile, count}; k_frame in list mode draws its runs from the list instead of from the groups' ranges
    int2 *wl;
    int *wl_ctl;             // [0] entries pushed, [1] entries drawn, [2] workgroups of the list-mode launch done (the last one zeroes all three), [3] tiles listed so far (statistics)
    int list_mode;
    int *wl_stat;            // page-locked host word: tiles listed so far on this stream, written by the list-mode launch when it ends
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
constexpr int E2D_RECORDS_EXT = 2 * E2D_RECORDS;       // the table goes on with its last value up to 28 stops (k_frame_p looks up unclamped EVs)
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
    int walk[5];                        // thread 0's: first tile of the group's range, tiles that go out in runs, tiles per run, group, runs all out
    int low[2];                         // some pixel this tile / the tile before loaded lies at most 64 above black (by tile parity)
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
// Round 5: LDS holds T'[m] = T16[m] - 4 m (load_t16_rel; T16[m] >= 4 m since log2(1 + x) >= x on [0, 1], checked where the host
// builds the table).  The float of lin is (e + 127) << 23 | m << 10, so its bits >> 8 are (e + 127) << 15 | 4 m and
//   ev(lin) + (127 << 15) = (float bits >> 8) + T'[m]
// -- a shift and an add per pixel where exponent and table value took a bit-field extract and a shift-add (and the shift issues
// faster than the extract: tools/valu_rate4.hip).
// Pixels at or below black (ev = INT_MIN / 0) or beyond the table are the rare case: a
// wave-wide vote picks the branch-free fast path unless some lane needs the fix-up.
// Exponent and 13-bit mantissa fraction come out of the float conversion (exact for l < 2^24): one v_cvt + one v_bfe
// instead of count-leading-zeros, variable shift and mask.
template <bool SPREAD>
__device__ __forceinline__ void load_t16_rel(uint16_t *dst, const uint16_t *src, int tid)
{
    if (SPREAD) {
        for (int i = tid; i < MLV_T16_N; i += 256) dst[i + (i >> 7)] = (uint16_t)(src[i] - 4 * i);
    } else {
        const uint4 *s4 = (const uint4 *)src;
        uint4 *d4 = (uint4 *)dst;
        for (int i = tid; i < MLV_T16_N * 2 / 16; i += 256) {       // eight entries: 4 m = 32 i, 32 i + 4, ... (no borrow between the halves: T16[m] >= 4 m)
            uint4 v = s4[i];
            const uint32_t b = (uint32_t)(32 * i) * 0x10001u + 0x40000u;
            v.x -= b; v.y -= b + 0x00080008u; v.z -= b + 0x00100010u; v.w -= b + 0x00180018u;
            d4[i] = v;
        }
    }
}
__device__ __forceinline__ int ev_index(int l) { return (int)((__float_as_uint((float)(unsigned)l) >> 10) & 8191u); }
__device__ __forceinline__ int ev_value(int l, int tv)
{
    return tv + (int)(__float_as_uint((float)(unsigned)l) >> 8) - (127 << 15);        // (tv: an entry of the relative table)
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

// v_bfe_u32 with a register width: the instruction itself takes the width's low five bits (the builtin's semantics make the
// compiler mask it first)
__device__ __forceinline__ uint32_t bfe_low_bits(uint32_t v, uint32_t width)
{
    uint32_t r;
    asm("v_bfe_u32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "v"(width));
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
        for (int i = 0; i < 8; i++) ex[i] = fb[i] >> 8;                                               // while the reads are in flight
        asm volatile("" :: "v"(ex[0]), "v"(ex[1]), "v"(ex[2]), "v"(ex[3]), "v"(ex[4]), "v"(ex[5]), "v"(ex[6]), "v"(ex[7]));
        asm volatile("" :: "v"(tv[0]), "v"(tv[1]), "v"(tv[2]), "v"(tv[3]), "v"(tv[4]), "v"(tv[5]), "v"(tv[6]), "v"(tv[7]));
#pragma unroll
        for (int i = 0; i < 8; i++) eb[i] = ex[i] + (uint32_t)tv[i];
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
    // px | 0x4B000000 IS the float 2^23 + px; minus (2^23 + black) it is float(px - black), exactly (integers below 2^24): a v_or_b32
    // with a literal and a v_sub_f32, both at twice the rate of the v_cvt_f32_u32 of rounds 3-4 (tools/valu_rate4.hip: 0.456 / 0.45
    // against 0.245 per clock and SIMD)
    const float fmagic = 8388608.0f + (float)black;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const uint32_t px[4] = { p0[2 * c], p0[2 * c + 1], p1[2 * c], p1[2 * c + 1] };
#pragma unroll
        for (int i = 0; i < 4; i++) fb[4 * c + i] = __float_as_uint(__uint_as_float(px[i] | 0x4B000000u) - fmagic);
    }
#pragma unroll
    for (int i = 0; i < 4 * NC; i++) tv[i] = *(const uint16_t *)((const char *)t + t16_offset<SPREAD>(fb[i]));
#pragma unroll
    for (int i = 0; i < 4 * NC; i++) ex[i] = fb[i] >> 8;
#pragma unroll
    for (int i = 0; i < 4 * NC; i += 8) {                // opaque uses: the reads stay unconditional and back to back
        asm volatile("" :: "v"(ex[i]), "v"(ex[i + 1]), "v"(ex[i + 2]), "v"(ex[i + 3]), "v"(ex[i + 4]), "v"(ex[i + 5]), "v"(ex[i + 6]), "v"(ex[i + 7]));
    }
#pragma unroll
    for (int i = 0; i < 4 * NC; i += 8) {
        asm volatile("" :: "v"(tv[i]), "v"(tv[i + 1]), "v"(tv[i + 2]), "v"(tv[i + 3]), "v"(tv[i + 4]), "v"(tv[i + 5]), "v"(tv[i + 6]), "v"(tv[i + 7]));
    }
#pragma unroll
    for (int i = 0; i < 4 * NC; i++) eb[i] = ex[i] + tv[i];
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
    const float fmagic = 8388608.0f + (float)black;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const uint32_t px[4] = { p0[2 * c], p0[2 * c + 1], p1[2 * c], p1[2 * c + 1] };
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float f = __uint_as_float(px[i] | 0x4B000000u) - fmagic;
            z[4 * c + i] = __float_as_uint(fabsf(f) - 0.5f);
            fb[4 * c + i] = __float_as_uint(fmaxf(f, 1.0f));
        }
    }
#pragma unroll
    for (int i = 0; i < 4 * NC; i++) tv[i] = *(const uint16_t *)((const char *)t + t16_offset<SPREAD>(fb[i]));
#pragma unroll
    for (int i = 0; i < 4 * NC; i++) ex[i] = fb[i] >> 8;
#pragma unroll
    for (int i = 0; i < 4 * NC; i += 8) {
        asm volatile("" :: "v"(ex[i]), "v"(ex[i + 1]), "v"(ex[i + 2]), "v"(ex[i + 3]), "v"(ex[i + 4]), "v"(ex[i + 5]), "v"(ex[i + 6]), "v"(ex[i + 7]));
    }
#pragma unroll
    for (int i = 0; i < 4 * NC; i += 8) {
        asm volatile("" :: "v"(tv[i]), "v"(tv[i + 1]), "v"(tv[i + 2]), "v"(tv[i + 3]), "v"(tv[i + 4]), "v"(tv[i + 5]), "v"(tv[i + 6]), "v"(tv[i + 7]));
    }
#pragma unroll
    for (int i = 0; i < 4 * NC; i++) eb[i] = (z[i] & 0x80000000u) | (ex[i] + tv[i]);      // v_add_u32, v_and_or_b32
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
typedef uint32_t mlv_u32x4 __attribute__((ext_vector_type(4)));
__device__ void mlv_rbs_x4(mlv_u32x4 data, mlv_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4i32");
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
__device__ __forceinline__ PatchCell patch_cell(const SM &sm, int black, int4 rec, int tx0, int ty0)
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
            for (int q = 0; q < 4; q++) odd = odd || (unsigned)(px[q] - black - 1) >= 16383u;
        }
        cell_ev<SM::SPREAD>(px[0], px[1], px[2], px[3], black, sm.t16, __any(odd), c.ge, c.dr, c.db);
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
// MASK = false: the caller knows that every pixel of the strip lies more than 64 above black (the mask would be all ones)
template <bool MASK = true>
__device__ __forceinline__ uint32_t stripe_pair(uint32_t x, int d0, int d1, uint32_t black_pk, uint32_t white_pk)
{
    typedef unsigned short upk16 __attribute__((ext_vector_type(2)));
    const mlv_pk16 a = __builtin_bit_cast(mlv_pk16, x) - __builtin_bit_cast(mlv_pk16, black_pk);
    const mlv_pk16 c64 = { 64, 64 }, s15 = { 15, 15 };
    const int p0 = __mul24((int)a.x, d0), p1 = __mul24((int)a.y, d1);                         // |.| < 2^29
    const uint32_t delta = __builtin_amdgcn_perm((uint32_t)p1, (uint32_t)p0, 0x07060302u);    // {p1 >> 16, p0 >> 16}
    uint32_t dm = delta;
    if (MASK) dm &= __builtin_bit_cast(uint32_t, (c64 - a) >> s15);                           // -1 where a > 64
    const upk16 v = __builtin_bit_cast(upk16, x) + __builtin_bit_cast(upk16, dm);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(v, __builtin_bit_cast(upk16, white_pk)));
}

template <bool MASK = true>
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
        top[c] = stripe_pair<MASK>(top[c], d0, d1, black_pk, white_pk);
        bot[c] = stripe_pair<MASK>(bot[c], d0, d1, black_pk, white_pk);
    }
}



// Arguments the common path touches once per tile or less are NOT kept in scalar registers for the life of the kernel: they are read
// where they are used, through a copy of the kernel-argument pointer the compiler cannot see through (scalar loads that hit the
// scalar cache).  Until round 5 k_frame kept all of FrameArgs live and spilt 45-53 scalar registers to vector lanes -- every reload a
// v_readlane_b32 in the vector pipe of a kernel that is bound by vector issue.
typedef const __attribute__((address_space(4))) FrameArgs *KArgs;
__device__ __forceinline__ KArgs cold_args()
{
    KArgs p = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
#ifndef KF_EXP_WARM_ARGS           // (A/B: the compiler sees through, hoists every load out of the tile loop and keeps the arguments live)
    asm volatile("" : "+s"(p));
#endif
    return p;
}


// ---------------------------------------------------------------- prefetch of a tile's new rows (both kernels)
// the item of lane L (its row of the tile: l_row) of the tile at (tx0, ty0); new0: first plane row the tile loads itself.
// The two forms below differ in how the four byte offsets (and the descriptor) come about; the LOADS are issued once, behind the
// branch: loads in both arms would be merged by copies, and those wait for the data right where the loads are issued (the same trap
// as a conditional prefetch, see k_frame) -- measured: +54 % of SQ_WAIT_ANY, the kernel 15 % slower with 2 % fewer instructions
// (profiles/r05/w8_pmc.log, w9_pmc.log).
// what of a lane's item the interior form needs: its two byte offsets relative to the tile (a constant of the lane for the launch)
template <int BPP>
__device__ __forceinline__ void item_tile_offsets(const ItemLane &L, int l_row, int w, uint32_t &pre_a, uint32_t &pre_b)
{
    constexpr uint32_t GB = (uint32_t)BPP;
    const uint32_t pitch = (uint32_t)(w >> 3) * GB;
    const uint32_t rowoff = __umul24((uint32_t)(2 * l_row), pitch) + 16u;
    pre_a = rowoff + ((uint32_t)((L.xoff_a >> 3) * (int)GB) & L.amask);
    pre_b = rowoff + ((uint32_t)((L.xoff_b >> 3) * (int)GB) & L.amask) + L.boff;
}
// LANE: a callable that yields the lane's ItemLane and row (called in the general form only: k_frame_p keeps just the two offsets)
template <int BPP, int VEC, class LANE>
__device__ __forceinline__ void issue_tile_rows_pre(uint32_t (&r0)[4], uint32_t (&r1)[4], const uint8_t *frame, unsigned bytes, uint32_t pre_a, uint32_t pre_b,
                                                    LANE lane_of, int w, int h, int tx0, int ty0, int new0)
{
    constexpr uint32_t GB = (uint32_t)BPP;              // bytes per 8-pixel group
    const uint32_t pitch = (uint32_t)(w >> 3) * GB;
    const int ybase = ty0 - 2 * HC + 2 * new0;
    uint32_t oa0, ob0, oa1, ob1, S = 0;
    if (VEC != 2 && tx0 >= 8 && tx0 + 2 * TCW + 8 <= w && ybase + 2 * TCH <= h) {           // (scalar)
        // A tile whose new rows and halo columns lie inside the frame -- nine in ten -- clamps nothing, and with rows that are
        // whole dwords (every layout but VEC 2) the dword alignment of a main item commutes with the row and tile offsets: what
        // is left per lane is a constant, the tile's part goes into the buffer descriptor (16 bytes early: the left halo's
        // group lies before the tile).  issue_item's general form costs 45 vector instructions per tile, this one a dozen.
        S = (uint32_t)ybase * pitch + (uint32_t)(tx0 >> 3) * GB - 16u;
        oa0 = pre_a;
        ob0 = pre_b;
        oa1 = oa0 + pitch;
        ob1 = ob0 + pitch;
    } else {
        // (issue_item's arithmetic: rows and groups clamped to the frame)
        int l_row;
        const ItemLane L = lane_of(l_row);
        const int y = ybase + 2 * l_row;
        const int gmax = (w >> 3) - 1, g0 = tx0 >> 3;
        const uint32_t ga = __umul24((uint32_t)min(g0 + (L.xoff_a >> 3), gmax), GB);
        const uint32_t gb = __umul24((uint32_t)max(min(g0 + (L.xoff_b >> 3), gmax), 0), GB);
        const uint32_t row0 = __umul24((uint32_t)min(y, h - 1), pitch), row1 = __umul24((uint32_t)min(y + 1, h - 1), pitch);
        oa0 = (row0 + ga) & L.amask; ob0 = ((row0 + gb) & L.amask) + L.boff;
        oa1 = (row1 + ga) & L.amask; ob1 = ((row1 + gb) & L.amask) + L.boff;
        asm volatile("" : "+v"(oa0), "+v"(ob0));       // (keeps this arm a branch target: as plain arithmetic the compiler runs BOTH arms on every tile and selects)
    }
    const mlv_i32x4 rs = frame_rsrc(frame + S, bytes - S);
    const mlv_u32x2 a0 = mlv_rbl_x2(rs, (int)oa0, 0, KF_SRC_AUX), b0 = mlv_rbl_x2(rs, (int)ob0, 0, KF_SRC_AUX);
    const mlv_u32x2 a1 = mlv_rbl_x2(rs, (int)oa1, 0, KF_SRC_AUX), b1 = mlv_rbl_x2(rs, (int)ob1, 0, KF_SRC_AUX);
    r0[0] = a0.x; r0[1] = a0.y; r0[2] = b0.x; r0[3] = b0.y;
    r1[0] = a1.x; r1[1] = a1.y; r1[2] = b1.x; r1[3] = b1.y;
}
template <int BPP, int VEC>
__device__ __forceinline__ void issue_tile_rows(uint32_t (&r0)[4], uint32_t (&r1)[4], const uint8_t *frame, unsigned bytes, const ItemLane &L, int l_row,
                                                int w, int h, int tx0, int ty0, int new0)
{
    uint32_t pre_a, pre_b;
    item_tile_offsets<BPP>(L, l_row, w, pre_a, pre_b);
    issue_tile_rows_pre<BPP, VEC>(r0, r1, frame, bytes, pre_a, pre_b, [&](int &row) { row = l_row; return L; }, w, h, tx0, ty0, new0);
}

// ---------------------------------------------------------------- output stage (both kernels)
// What the stage needs of the cold arguments, fetched in one go (the loads go out together and are waited for once)
struct OutArgs {
    const uint2 *e2d;
    uint8_t *dst;
    size_t dst_stride;
    int stripes, coef_pk, coef_fast, white;
    int co[8];
};
__device__ __forceinline__ OutArgs out_args(KArgs kt)
{
    OutArgs o;
    o.e2d = kt->e2d; o.dst = kt->dst; o.dst_stride = kt->dst_stride;
    o.stripes = kt->stripes; o.coef_pk = kt->coef_pk; o.coef_fast = kt->coef_fast; o.white = kt->white;
#pragma unroll
    for (int i = 0; i < 8; i++) o.co[i] = kt->coef[i];
    return o;
}

// R is the lower half of a cell's top word, B the upper half of its bottom word: in the lanes of mask `m` they take the looked-up
// values.  One v_cndmask_b32_sdwa each (the condition in VCC, the other half of the destination preserved) -- rounds 2-4 chose a
// v_perm_b32 selector with a v_cndmask_b32 and permuted: two instructions per word.
__device__ __forceinline__ void put_rb(uint32_t &top, uint32_t &bot, uint32_t ur, uint32_t ub, unsigned long long m)
{
    asm("s_mov_b64 vcc, %4\n\t"
        "v_cndmask_b32_sdwa %0, %0, %2, vcc dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n\t"
        "v_cndmask_b32_sdwa %1, %1, %3, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_0"
        : "+v"(top), "+v"(bot) : "v"(ur), "v"(ub), "s"(m) : "vcc");
}
// lane masks of signed compares (v_cmp into a scalar register pair; the conditions of a cell are ANDed there, in the scalar unit:
// as `bool`s the compiler turned them into 0 / 1 registers and back -- eight vector instructions per cell)
__device__ __forceinline__ unsigned long long lanes_gt(int a, int b) { return __builtin_amdgcn_sicmp(a, b, 38); }       // a > b
__device__ __forceinline__ unsigned long long lanes_ge(int a, int b) { return __builtin_amdgcn_sicmp(a, b, 39); }       // a >= b
__device__ __forceinline__ unsigned long long lanes_lt(int a, int b) { return __builtin_amdgcn_sicmp(a, b, 40); }       // a < b

// The rest of a strip once the EVs of its smoothed R and B are known (er = ge + median(dr), eb = ge + median(db), wrapping like the
// reference's ints): look-up, R / B replacement (chroma_smooth.c:28, 35, 64-68), stripes (stripes.c:250-266), store.
//   msmooth: the lanes whose row takes chroma smoothing at all (chroma_smooth.c:25).
//   CLAMP = false: the caller knows that no pixel of the rows the tile holds lies at most 64 above black (and the tile holds no
//   pixel-map cells).  Then every EV involved lies in [6, 14] stops: er and eb stay below 28 stops -- where the table goes on with
//   its last value (E2D_RECORDS_EXT) -- so the clamp of chroma_smooth.c:67 is the table's own, and one below zero belongs to a cell
//   that keeps its pixel (er > 1 stop fails; its look-up is out of range and reads 0).  And the stripes epilogue's "more than 64
//   above black" holds for every pixel: no mask.
//   XM: the tile touches the frame's left or right margin (chroma_smooth.c:27 leaves columns 0..3 and w-4.. alone).
//   VECST: rows of whole 8-pixel groups on 16-byte aligned buffers (two 16-byte stores); else pixel by pixel.
// The variants are separate instantiations chosen by scalar branches: written as conditions inside one body the compiler computed
// both sides for every strip and selected (profiles/r05: 207 instead of 178 vector instructions per strip).
//   ANYSTRIPES: the stripes epilogue in all its forms (else: the packed 16-bit one or none -- what a launch has is a scalar of the launch).
//   RAWREG (k_frame_s): the strip's pixels come in registers (rtop / rbot) instead of from the tile's LDS rows, and go back there
//   (store = false: the caller stores them).
struct NoSmem {};
// (frame_stream_colw / _cols / _fold -- the streaming kernels' launch plan, host arithmetic -- live in common.h)
template <int METHOD, bool PACKED, bool VECST, bool CLAMP, bool XM, bool ANYSTRIPES, bool BRIGHT, class SM, bool RAWREG = false>
__device__ __forceinline__ void strip_output_t(const SM &sm, const OutArgs &oa, int w, int h, int black, int f, int tx0, int ty0, int jj, int kk,
                                               unsigned long long msmooth, const int (&gev)[STRIP], int gev_off, const int (&er)[STRIP], const int (&eb)[STRIP], bool store,
                                               uint32_t *rtop = nullptr, uint32_t *rbot = nullptr)
{
    const int y = ty0 + 2 * jj, x = tx0 + 2 * STRIP * kk;
    uint32_t top[STRIP], bot[STRIP];        // (R | G1<<16), (G2 | B<<16)
    auto read_raw = [&]() {
        if constexpr (RAWREG) {
#pragma unroll
            for (int c = 0; c < STRIP; c++) { top[c] = rtop[c]; bot[c] = rbot[c]; }
        } else {
            const uint4 v0 = *(const uint4 *)&sm.raw[2 * jj][2 * STRIP * kk];
            const uint4 v1 = *(const uint4 *)&sm.raw[2 * jj + 1][2 * STRIP * kk];
            top[0] = v0.x; top[1] = v0.y; top[2] = v0.z; top[3] = v0.w;
            bot[0] = v1.x; bot[1] = v1.y; bot[2] = v1.z; bot[3] = v1.w;
        }
    };
    if (METHOD == 0) read_raw();
    if (METHOD != 0) {
        const mlv_i32x4 rs_e2d = table_rsrc(oa.e2d, 8, E2D_RECORDS_EXT);
        // the output pixel by EV (one 8-byte record per 32 EV steps): all 8 look-ups issued before the first use
        int ur[STRIP], ub[STRIP], cr[STRIP], cb[STRIP];
        mlv_tab_u32x2 dr2[STRIP], db2[STRIP];
#pragma unroll
        for (int c = 0; c < STRIP; c++) {
            cr[c] = CLAMP ? min(max(er[c], 0), MLV_EV_MAX) : er[c];
            cb[c] = CLAMP ? min(max(eb[c], 0), MLV_EV_MAX) : eb[c];
#ifdef KF_EXP_NOLOOKUP
            dr2[c].x = cr[c]; dr2[c].y = 0; db2[c].x = cb[c]; db2[c].y = 0;
#else
            dr2[c] = mlv_sbl_x2(rs_e2d, cr[c] >> 5, 0, 0, KF_E2R_AUX);
            db2[c] = mlv_sbl_x2(rs_e2d, cb[c] >> 5, 0, 0, KF_E2R_AUX);
#endif
        }
        read_raw();
        // (the fence keeps the eight look-ups together)
        asm volatile("" :: "v"(dr2[0]), "v"(dr2[1]), "v"(dr2[2]), "v"(dr2[3]), "v"(db2[0]), "v"(db2[1]), "v"(db2[2]), "v"(db2[3]));
#pragma unroll
        for (int c = 0; c < STRIP; c++) {             // v_bfe_u32 (the width operand's low five bits count), v_bcnt_u32_b32
            ur[c] = (int)(__builtin_popcount(bfe_low_bits(dr2[c].y, (uint32_t)cr[c])) + dr2[c].x);
            ub[c] = (int)(__builtin_popcount(bfe_low_bits(db2[c].y, (uint32_t)cb[c])) + db2[c].x);
        }
#pragma unroll
        for (int c = 0; c < STRIP; c++) {             // which cells take the smoothed values (one mask at a time: four of them live cost eight scalar registers)
            unsigned long long okm = msmooth;
            // (gev holds the green EV + gev_off, wrapping: k_frame_p keeps it with its reference added; the subtraction lives here so that the
            // bright variant, which tests nothing, does not pay for it)
            if (!BRIGHT) okm &= lanes_ge((int)((uint32_t)gev[c] - (uint32_t)gev_off), 2 * MLV_EV_RES) & lanes_gt(er[c], MLV_EV_RES) & lanes_gt(eb[c], MLV_EV_RES);
            if (XM) okm &= lanes_ge(x + 2 * c, 4) & lanes_lt(x + 2 * c, w - 4);
            put_rb(top[c], bot[c], (uint32_t)ur[c], (uint32_t)ub[c], okm);
        }
    }
    if (oa.stripes) {
        // a strip starts at an x that is a multiple of 8, so pixel n of the strip has column phase n
        const int black16 = (int)(uint16_t)black, white16 = (int)(uint16_t)oa.white;
        if (!ANYSTRIPES) stripe_strip_pk<CLAMP>(top, bot, oa.co, black16, white16);
        else if (PACKED && oa.coef_pk) stripe_strip_pk<true>(top, bot, oa.co, black16, white16);
        else if (oa.coef_fast) stripe_strip<true>(top, bot, oa.co, black16, white16);
        else stripe_strip<false>(top, bot, oa.co, black16, white16);
    }
    if constexpr (RAWREG) {
#pragma unroll
        for (int c = 0; c < STRIP; c++) { rtop[c] = top[c]; rbot[c] = bot[c]; }
    }
    if (store && y < h) {
        if (VECST) {
            if (x < w) {
                // the frame as a buffer whose base is the tile's first pixel: the lane's part of the address is (2 j w + 8 k) pixels.
                // Non-temporal stores: the output is not read again by this launch (profiles/r04/ab_cache_policy.log: -1 ... -3.5 %)
                const uint32_t T = ((uint32_t)ty0 * (uint32_t)w + (uint32_t)tx0) * 2u;
                const mlv_i32x4 rs_out = frame_rsrc(oa.dst + (size_t)f * oa.dst_stride + T, (uint32_t)w * (uint32_t)h * 2u - T);
                const uint32_t vo = (__umul24((uint32_t)(2 * jj), (uint32_t)w) + 2 * STRIP * (uint32_t)kk) * 2u;
                const mlv_u32x4 vt = { top[0], top[1], top[2], top[3] }, vb = { bot[0], bot[1], bot[2], bot[3] };
                mlv_rbs_x4(vt, rs_out, (int)vo, 0, 2);                                     // (2: non-temporal)
                if (y + 1 < h) mlv_rbs_x4(vb, rs_out, (int)vo, w * 2, 2);
            }
        } else {
            uint16_t *out = (uint16_t *)(oa.dst + (size_t)f * oa.dst_stride);
#pragma unroll 1
            for (int c = 0; c < STRIP; c++) {
                const int xc = x + 2 * c;
                if (xc < w) out[(size_t)y * w + xc] = (uint16_t)top[c];
                if (xc + 1 < w) out[(size_t)y * w + xc + 1] = (uint16_t)(top[c] >> 16);
                if (y + 1 < h) {
                    if (xc < w) out[(size_t)(y + 1) * w + xc] = (uint16_t)bot[c];
                    if (xc + 1 < w) out[(size_t)(y + 1) * w + xc + 1] = (uint16_t)(bot[c] >> 16);
                }
            }
        }
    }
}

// low_any (scalar): some pixel of the rows this tile holds lies at most 64 above black, or the tile holds pixel-map cells
// bright (scalar; implies !low_any): every pixel of those rows lies at least 256 above black (and below 2^14).  Then every EV lies in
// [8, 14) stops, every green EV too, every colour difference in (-6, 6) and every smoothed EV above 8 - 6 = 2 stops: the three
// conditions of chroma_smooth.c:64-66 (green >= 2 stops, smoothed R and B > 1 stop) hold for every cell and are not evaluated.
template <int METHOD, bool PACKED, bool VECST, class SM>
__device__ __forceinline__ void strip_output(const SM &sm, const OutArgs &oa, int w, int h, int black, int f, int tx0, int ty0, int jj, int kk,
                                             unsigned long long msmooth, const int (&gev)[STRIP], int gev_off, const int (&er)[STRIP], const int (&eb)[STRIP], bool low_any, bool bright,
                                             bool store)
{
    const bool xm = tx0 < 4 || tx0 + 2 * TCW > w - 4;                   // scalar
    if (oa.stripes && !(PACKED && oa.coef_pk)) {        // (gains beyond the packed form's range, 16-bit input: one variant, everything tested)
        strip_output_t<METHOD, PACKED, VECST, true, true, true, false, SM>(sm, oa, w, h, black, f, tx0, ty0, jj, kk, msmooth, gev, gev_off, er, eb, store);
    } else if (low_any) {
        if (xm) strip_output_t<METHOD, PACKED, VECST, true, true, false, false, SM>(sm, oa, w, h, black, f, tx0, ty0, jj, kk, msmooth, gev, gev_off, er, eb, store);
        else strip_output_t<METHOD, PACKED, VECST, true, false, false, false, SM>(sm, oa, w, h, black, f, tx0, ty0, jj, kk, msmooth, gev, gev_off, er, eb, store);
    } else if (xm) {
        strip_output_t<METHOD, PACKED, VECST, false, true, false, false, SM>(sm, oa, w, h, black, f, tx0, ty0, jj, kk, msmooth, gev, gev_off, er, eb, store);
    } else {
#ifndef KF_EXP_NO_BRIGHT
        if (bright) strip_output_t<METHOD, PACKED, VECST, false, false, false, true, SM>(sm, oa, w, h, black, f, tx0, ty0, jj, kk, msmooth, gev, gev_off, er, eb, store);
        else
#endif
        strip_output_t<METHOD, PACKED, VECST, false, false, false, false, SM>(sm, oa, w, h, black, f, tx0, ty0, jj, kk, msmooth, gev, gev_off, er, eb, store);
    }
}

}  // namespace mlv
