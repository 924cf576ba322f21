// k_amaze_rows.hip -- the AMaZE demosaic (mlvfs/amaze_demosaic_RT.c:113-1487, SSE2 variant) for COMPLETE tiles, row-streamed
// through LDS.
//
// k_amaze.hip keeps the 26 planes of a 160x160 tile in a 2 MB block of HBM and runs the ~20 passes one after the other over the
// whole tile: in bulk every plane goes to HBM and comes back once or twice (6 MB of traffic per tile), and the
// waves wait for it two thirds of their time.  Here the passes run as a software pipeline down the rows of the tile instead: in
// step s pass P works on the row pair s - lag(P), every plane is a ring of just the rows that are still needed (2 ... 38 of
// them), all rings together are 151 KiB of LDS, and HBM sees the tile once on the way in and its three planes once on the way out.
//
//   * A workgroup (16 waves) owns a list of tiles and streams through them without draining: the pair counter simply runs on
//     into the next tile (no pass reads across a tile's first or last row).
//   * The waves are specialised: an item is (pass, 64 lanes' worth of the row pair), each wave owns a fixed list of items per
//     phase, one barrier ends a phase, two phases make a step.  A consumer in phase B may read what phase A of the same step
//     wrote; otherwise it lags its producer by the rows it looks ahead plus one step (table below).
//   * The passes that are sequential inside a tile are sequential down the rows only: the vcd refinement reads the refined
//     row two above (the previous pair), the hvwt / pmwt sweeps read the updated row above (two sub-steps of one wave), and the
//     raster-order Nyquist vote is a recurrence along the row on ONE bit -- new[k] = f_k(new[k-1]) with f_k one of the four
//     boolean functions -- evaluated as a prefix composition by one wave.
//   * Lanes of one item cover both rows of the pair ("row 1" lanes address one row pitch further); every ring has one guard
//     row behind its last that mirrors row 0, so that a lane pair never sees the wrap.
//   * "Never written reads as zero" (the reference's calloc'ed block, for complete tiles) is explicit: the first producer of a
//     ring row writes the whole row, zeros outside the pass's range.
// What stays in k_amaze.hip: tiles whose right/bottom apron leaves the image (mirrored fills that run off their rows, stale
// planes of the tile before them) and the workgroups that chain such tiles -- 15 % of the tiles at 3584x1320.
// Arithmetic: amaze_math.h, the same expressions in the same order as k_amaze.hip; dw0/dw1 are recomputed from cfa in the
// last green pass (same expression, same bits) instead of being kept for 28 rows.
#include "amaze_math.h"
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>
#include <algorithm>

namespace mlv {

using namespace amz;

namespace {

constexpr int HT = T / 2, NP = T / 2;      // half-plane row, row pairs per tile

// ---- rings (rows without the guard row; even).  Sizes follow from the schedule: R/2 >= lag(last consumer) - lag(producer) + 1
// + ceil(rows looked back / 2).
enum { R_C, R_DW0, R_DW1, R_VCD, R_HCD, R_HCDALT, R_VCDALT, R_HCD2, R_CDSQ, R_DELSQ, R_DGV, R_DGH, NFULL,
       R_HVWT = NFULL, R_HCRB, R_VCRB, R_DELP, R_DELM, R_SQP, R_SQM, R_RBM, R_RBP, R_PMWT, R_RBINT, R_CURVH, R_CURVV, R_DGRB0, R_DGRB1,
       R_GRB, NRINGS };
constexpr int RR[NRINGS] = { 38, 16, 14, 14, 2, 2, 6, 10, 8, 8, 12, 10,
                             26, 16, 16, 6, 6, 6, 6, 6, 6, 10, 8, 6, 6, 12, 12, 10 };
constexpr int NYQ_ROWS = 18;
constexpr int pitch_of(int k) { return k < NFULL ? T : HT; }
constexpr int LDS_PAD = 16;                                       // floats in front of the first ring (reads left of column 0)
constexpr int ring_off(int k) { int o = LDS_PAD; for (int j = 0; j < k; j++) o += (RR[j] + 1) * pitch_of(j); return o; }
constexpr int NYQ_OFF = ring_off(NRINGS);                         // floats; the flags are bytes, HT per row
constexpr int AREA_LIST_FLOATS = 3 * T / 4;                      // three byte lists of up to 160 flagged sites (P_AREA)
constexpr int RINGS_FLOATS = NYQ_OFF + ((NYQ_ROWS + 1) * HT + 3) / 4 + AREA_LIST_FLOATS + LDS_PAD;           // everything that starts as zero


// ---- schedule: lag in row pairs, phase (0 = A, 1 = B)
enum { P_LOAD, P_DIRDIFF, P_NYQTEST, P_HVWT, P_AREA, P_CURV, P_CGRAD, P_PMSWEEP, P_CHROMA,          // phase A
       P_GRAD, P_HREF, P_VWALK, P_VOTE, P_HVSWEEP, P_DIAG, P_RBINT, P_GFINAL, P_OUTPUT, NPASS };      // phase B
constexpr int LAG[NPASS] = { 0, 3, 6, 7, 11, 13, 9, 12, 17,   1, 3, 4, 7, 12, 10, 12, 14, 18 };
constexpr int LAG_MAX = 18;
#define IT(p, c) ((p) * 8 + (c))
// Which wave runs which items is decided on the host (amaze_rows_launch): the items of a phase, longest first, each to the wave with
// the least work so far (measured cycles per item below, MLVFS_AMD_AMAZE_ROWS_PROF=2; taking items from a counter in LDS balanced
// no better and cost 54 same-address atomics per step).  AREA is priced as if idle: where the Nyquist test fired it is by far the
// longest item, elsewhere it returns at once.  LOAD keeps the next row pair in registers across steps: it must stay with one wave.
constexpr int NCHUNKS[NPASS] = { 1, 5, 3, 3, 3, 3, 3, 1, 3,   5, 5, 5, 1, 1, 3, 3, 3, 4 };
constexpr int COST[NPASS] = { 1000, 1000, 1000, 1000, 1000, 1000, 1000, 1000, 1000,   1000, 1000, 1000, 1000, 1000, 1000, 1000, 1000, 1000 };
// The deal in use: found by local search (moves and swaps of items between waves; tools/amaze_rows_assign_search.py), first with the
// kernel's time on 2 254 unflagged tiles as the objective (3.26 ms where the deal by item count takes 3.71 and the one by measured
// cycles per item 4.07), then with a batch of 8 dual-ISO conversions (9.15 -> 8.92 ms, best of 5; after the analysis kernel's change 8.78 -> 8.67; 760 candidates in all).  What an item costs depends on what runs
// beside it on its SIMD, and no cost table captures that.
constexpr unsigned long long TUNED[2][16] = {
    { 0xffffffff06100c38ull, 0xffffffff030a0612ull, 0xffffffff03090d28ull, 0xffffffffffff030cull, 0xffffffffffff030bull, 0xffffffff00000931ull, 0xffffffff07181140ull, 0xffffffff06110b21ull, 0xffffffff11420d29ull, 0xffffffff09320b22ull, 0xffffffffffff0308ull, 0xffffffff0d2a071aull, 0xffffffffffff1141ull, 0xffffffffffff0b20ull, 0xffffffffffff0930ull, 0xffffffffffff0719ull },
    { 0xffffffff0c680e82ull, 0xffffffff0a700352ull, 0xffffffff0a71014aull, 0xffffffffffff0a72ull, 0xffff0148014b0351ull, 0xffffffff0c790e80ull, 0xffff045a0c7a0354ull, 0xffffffff03500353ull, 0xffffffffffff0e81ull, 0xffffffff0c780149ull, 0xffffffff12880458ull, 0xffffffff12890459ull, 0xffffffffffff128aull, 0xffffffff0760128bull, 0xffffffff014c045cull, 0xffffffffffff045bull } };
__constant__ unsigned long long c_desc[2][16];      // per phase and wave: up to four items, 16 bits each: item code | lag of its pass << 8
bool g_tab_ready[64] = {};                          // per device: the kernels' LDS attribute is set

// legacy plane numbering of k_amaze.hip's block, for the debug dump: 13 full planes, then 13 half planes
enum { D_CFA, D_GREEN, D_DELSQ, D_DW0, D_DW1, D_VCD, D_HCD, D_VCDALT, D_HCDALT, D_CDSQ, D_DGV, D_DGH, D_HCD2,
       D_HVWT, D_DGRB0, D_DGRB1, D_DELP, D_DELM, D_RBINT, D_CURVH, D_CURVV, D_SQM, D_SQP, D_PMWT, D_RBM, D_RBP };
__device__ __forceinline__ size_t dbg_off(int plane) { return plane < 13 ? (size_t)plane * TT : (size_t)13 * TT + (size_t)(plane - 13) * HALF; }

// ---- row addresses.  Every pass reads and writes a fixed set of (ring, row offset) pairs; where a pair's row lies in LDS is
// base + ((2 q + d) mod R) * pitch.  Computing that on the scalar unit costs 6-7 instructions per pair -- 8 400 per step and
// workgroup, more than the vector work, and an instruction of ANY kind occupies its wave's issue slot for 4 cycles.  Instead each
// pass has a table of its pairs (at most 32, TAB_* below); an item's lane j evaluates entry j (5 vector instructions for all pairs
// at once: multiply-high by a reciprocal, no division), and a pair's byte offset is one v_readlane away.
constexpr int R_NYQ = NRINGS;                                     // the flags' byte ring in the tables
struct RD { short K, dlo, dhi; };                                // ring, row offsets dlo..dhi (consecutive entries)
constexpr int NTAB = 32;
constexpr RD TAB_LOAD[] = { { R_C, 0, 0 } };
constexpr RD TAB_GRAD[] = { { R_C, -2, 2 }, { R_DW1, 0, 0 }, { R_DW0, 0, 0 } };
constexpr RD TAB_DIRDIFF[] = { { R_C, -2, 2 }, { R_DW0, -2, 2 }, { R_DW1, 0, 0 }, { R_HCDALT, 0, 0 }, { R_VCDALT, 0, 0 }, { R_VCD, 0, 0 },
                               { R_HCD, 0, 0 }, { R_DGV, 0, 0 }, { R_DGH, 0, 0 } };
constexpr RD TAB_HREF[] = { { R_HCD, 0, 0 }, { R_HCDALT, 0, 0 }, { R_C, 0, 0 }, { R_HCD2, 0, 0 } };
constexpr RD TAB_VWALK[] = { { R_VCD, -2, 2 }, { R_VCDALT, -2, 2 }, { R_C, -1, 1 }, { R_HCD2, 0, 0 }, { R_CDSQ, 0, 0 }, { R_DELSQ, 0, 0 } };
constexpr RD TAB_NYQTEST[] = { { R_CDSQ, -2, 2 }, { R_DELSQ, -2, 2 }, { R_NYQ, 0, 0 } };
constexpr RD TAB_HVWT[] = { { R_HCD2, 0, 0 }, { R_DGH, 0, 0 }, { R_DW1, 0, 0 }, { R_VCD, -3, 3 }, { R_DW0, -1, 1 }, { R_DGV, -2, 2 }, { R_HVWT, 0, 0 },
                            { R_HCRB, 0, 0 }, { R_VCRB, 0, 0 } };
constexpr RD TAB_VOTE[] = { { R_NYQ, -2, 3 } };
constexpr RD TAB_AREA[] = { { R_NYQ, -6, 6 }, { R_C, -7, 7 }, { R_HVWT, 0, 0 } };
constexpr RD TAB_HVSWEEP[] = { { R_HVWT, -1, 2 } };
constexpr RD TAB_CURV[] = { { R_HVWT, 0, 0 }, { R_HCRB, 0, 0 }, { R_VCRB, 0, 0 }, { R_C, -1, 1 }, { R_NYQ, 0, 0 }, { R_CURVH, 0, 0 }, { R_CURVV, 0, 0 } };
constexpr RD TAB_CGRAD[] = { { R_C, -1, 1 }, { R_DELP, 0, 0 }, { R_DELM, 0, 0 }, { R_SQP, 0, 0 }, { R_SQM, 0, 0 } };
constexpr RD TAB_DIAG[] = { { R_C, -2, 2 }, { R_DELM, -2, 2 }, { R_DELP, -2, 2 }, { R_SQM, -2, 2 }, { R_SQP, -2, 2 }, { R_RBM, 0, 0 }, { R_RBP, 0, 0 },
                            { R_PMWT, 0, 0 } };
constexpr RD TAB_PMSWEEP[] = { { R_PMWT, -1, 2 } };
constexpr RD TAB_RBINT[] = { { R_PMWT, 0, 0 }, { R_C, 0, 0 }, { R_RBM, 0, 0 }, { R_RBP, 0, 0 }, { R_RBINT, 0, 0 } };
constexpr RD TAB_GFINAL[] = { { R_C, -3, 3 }, { R_HVWT, 0, 0 }, { R_HCRB, 0, 0 }, { R_VCRB, 0, 0 }, { R_NYQ, 0, 0 }, { R_CURVH, -2, 2 }, { R_CURVV, -2, 2 },
                              { R_PMWT, 0, 0 }, { R_RBINT, -2, 2 }, { R_DGRB0, 0, 0 }, { R_DGRB1, 0, 0 }, { R_GRB, 0, 0 } };
constexpr RD TAB_CHROMA[] = { { R_DGRB0, -3, 3 } };
constexpr RD TAB_OUTPUT[] = { { R_C, 0, 0 }, { R_HVWT, -1, 1 }, { R_DGRB0, -1, 1 }, { R_DGRB1, -1, 1 }, { R_GRB, 0, 0 } };
// index of (K, d) in a pass's table; -1: not there (a compile error where it is used)
template <int N> constexpr int tfind(const RD (&t)[N], int K, int d)
{
    int idx = 0;
    for (int k = 0; k < N; k++) {
        if (t[k].K == K && d >= t[k].dlo && d <= t[k].dhi) return idx + d - t[k].dlo;
        idx += t[k].dhi - t[k].dlo + 1;
    }
    return -1;
}
template <int N> constexpr int tcount(const RD (&t)[N]) { int n = 0; for (int k = 0; k < N; k++) n += t[k].dhi - t[k].dlo + 1; return n; }
static_assert(tcount(TAB_GFINAL) <= NTAB && tcount(TAB_DIAG) <= NTAB && tcount(TAB_AREA) <= NTAB, "table size");
// the entry a lane evaluates: x = 2 s + e (e = (d - 2 lag) mod R >= 0), row = x mod R by multiply-high, offset = base + row * pitch
struct TabEntry { unsigned packed, magic, base; };               // packed: e | R << 8 | pitch bytes << 16
constexpr int ring_rows(int K) { return K == R_NYQ ? NYQ_ROWS : RR[K]; }
constexpr unsigned ring_pitch_bytes(int K) { return K == R_NYQ ? HT : 4u * pitch_of(K); }
constexpr unsigned ring_base_bytes(int K) { return K == R_NYQ ? 4u * NYQ_OFF : 4u * ring_off(K); }
struct PassTables { TabEntry e[NPASS][NTAB]; };
template <int N> constexpr void tab_fill(PassTables &pt, int pass, const RD (&t)[N])
{
    int idx = 0;
    for (int k = 0; k < N; k++)
        for (int d = t[k].dlo; d <= t[k].dhi; d++, idx++) {
            const int R = ring_rows(t[k].K);
            const int e = (((d - 2 * LAG[pass]) % R) + R) % R;
            pt.e[pass][idx] = TabEntry{ (unsigned)e | (unsigned)R << 8 | ring_pitch_bytes(t[k].K) << 16, (unsigned)(0x100000000ull / (unsigned)R + 1), ring_base_bytes(t[k].K) };
        }
}
constexpr PassTables make_tables()
{
    PassTables pt{};
    tab_fill(pt, P_LOAD, TAB_LOAD); tab_fill(pt, P_GRAD, TAB_GRAD); tab_fill(pt, P_DIRDIFF, TAB_DIRDIFF); tab_fill(pt, P_HREF, TAB_HREF);
    tab_fill(pt, P_VWALK, TAB_VWALK); tab_fill(pt, P_NYQTEST, TAB_NYQTEST); tab_fill(pt, P_HVWT, TAB_HVWT); tab_fill(pt, P_VOTE, TAB_VOTE);
    tab_fill(pt, P_AREA, TAB_AREA); tab_fill(pt, P_HVSWEEP, TAB_HVSWEEP); tab_fill(pt, P_CURV, TAB_CURV); tab_fill(pt, P_CGRAD, TAB_CGRAD);
    tab_fill(pt, P_DIAG, TAB_DIAG); tab_fill(pt, P_PMSWEEP, TAB_PMSWEEP); tab_fill(pt, P_RBINT, TAB_RBINT); tab_fill(pt, P_GFINAL, TAB_GFINAL);
    tab_fill(pt, P_CHROMA, TAB_CHROMA); tab_fill(pt, P_OUTPUT, TAB_OUTPUT);
    return pt;
}
constexpr PassTables H_TABLES = make_tables();
__constant__ PassTables c_tables;
constexpr int TAB_FLOATS = NPASS * NTAB * 3;                      // the tables' copy in LDS, behind the rings
constexpr int LDS_FLOATS = RINGS_FLOATS + TAB_FLOATS;
static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");

}  // namespace

// row d of ring K for this item's pair (lanes of the pair's second row add one pitch); PT is the table of the pass being compiled
#define TIDX(K, d) ([] { constexpr int i0 = tfind(PT, K, 0) >= 0 ? tfind(PT, K, 0) : -1; static_assert(i0 >= 0, "(ring, row 0) is not in this pass's table"); return i0; }() + (d))
#define FOFF(K, d) ((unsigned)__builtin_amdgcn_readlane((int)tabv, TIDX(K, d)))
#define FP(K, d) ((float *)((char *)sm + FOFF(K, d)))
#define FPN(d) ((unsigned char *)sm + FOFF(R_NYQ, d))
#define ST(K, lo, v) do { const unsigned o_ = FOFF(K, 0); float *p_ = (float *)((char *)sm + o_); p_[lo] = (v);                                   \
                          if (o_ == 4u * ring_off(K) && (lo) < pitch_of(K)) p_[RR[K] * pitch_of(K) + (lo)] = (v); } while (0)      /* row 0 has a mirror behind the last row */
#define NST(lo, v) do { const unsigned o_ = FOFF(R_NYQ, 0); unsigned char *p_ = (unsigned char *)sm + o_; p_[lo] = (v);                           \
                        if (o_ == 4u * NYQ_OFF && (lo) < HT) p_[NYQ_ROWS * HT + (lo)] = (v); } while (0)
#define HX(dc) (((dc) & 1) ? h1 + ((dc) - 1) / 2 : h0 + (dc) / 2)
#define DBGF(plane, val) do { if (DBG) dbgt[dbg_off(plane) + (size_t)r * T + col] = (val); } while (0)
#define DBGH(plane, val) do { if (DBG) dbgt[dbg_off(plane) + (size_t)r * HT + hj] = (val); } while (0)
// decode of an item's lanes.  Full-width passes: 320 pixels of the row pair in 5 chunks; half-width passes (one lane per R/B
// site, or per column pair): 160 sites in 3 chunks, site hj of row r sits in column 2 * hj + (r & 1).
#define FULL_LANES() const int n = ck * 64 + lane, rho = n >= T, col = n - T * rho, r = 2 * p + rho; (void)col; (void)r
#define HALF_LANES() const int n = ck * 64 + lane, act = n < T, rho = n >= HT, hj = n - HT * rho, col = 2 * hj + rho, r = 2 * p + rho, \
                               f0 = T * rho + col, h0 = n, h1 = n + rho; (void)act; (void)f0; (void)h0; (void)h1; (void)col; (void)r

template <bool DBG>
__global__ __launch_bounds__(1024) void k_amaze_rows(const float *__restrict__ raw, int w, int h, float *__restrict__ red,
                                                      float *__restrict__ green_out, float *__restrict__ blue, int nfx, int ntiles, int nrect,
                                                      size_t plane_stride, const int *__restrict__ h_of, int h_stride,
                                                      float *__restrict__ dbg, unsigned long long *__restrict__ prof, unsigned skip_mask, int *__restrict__ tile_ctr,
                                                      const int *__restrict__ r2e, int ev_black, int *__restrict__ gray)
{
    __shared__ float sm[LDS_FLOATS];                          // static: row offsets are compile-time constants (a dynamic array costs an add each)
    {
        const size_t f = blockIdx.y;
        if (h_of && h_of[f * (size_t)h_stride] != h) return;
        raw += f * plane_stride; red += f * plane_stride; green_out += f * plane_stride; blue += f * plane_stride;
        if (r2e) gray += f * plane_stride;
        if (DBG) dbg += f * (size_t)ntiles * AMAZE_TILE_FLOATS;
    }
    if ((int)blockIdx.x >= ntiles) return;
    // Tiles come from a counter per frame (tile_ctr): a workgroup that starts late -- the CUs are shared with k_amaze.hip's launches
    // for the incomplete tiles, on another stream -- simply takes fewer.  ctl[0..3]: the numbers of the tiles in flight (tile k of
    // this workgroup in ctl[k & 3]), ctl[4]: the pair counter at which the tiles ran out (INT_MAX until then).
    tile_ctr += blockIdx.y;
    int *const ctl = (int *)(sm + RINGS_FLOATS - 8);
    for (int n = threadIdx.x; n < RINGS_FLOATS; n += blockDim.x) sm[n] = 0.0f;
    for (int n = threadIdx.x; n < TAB_FLOATS; n += blockDim.x) ((unsigned *)(sm + RINGS_FLOATS))[n] = ((const unsigned *)&c_tables)[n];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;

    // tile of pair counter q: its number and its origin in the image
    // Tiles 0 .. nrect - 1 are the nfx x nfy block of the tile grid; tiles nrect .. ntiles - 1 are the tiles of column nfx in rows
    // 0, 1, ... (amaze_rows_extra: the last column whose tiles have all their rows and columns, where the image's right edge is
    // mirrored into their last 16 columns and nothing downstream reads the block they would have left behind in k_amaze.hip)
    auto origin = [&](int id, int &top, int &left) {
        const int ty = id < nrect ? id / nfx : id - nrect, tx = id < nrect ? id % nfx : nfx;
        top = -16 + ty * (T - 32);
        left = -16 + tx * (T - 32);
    };
    auto tile_of = [&](int q, int &top, int &left) -> int {
        const int id = __builtin_amdgcn_readfirstlane(ctl[(q / NP) & 3]);
        origin(id, top, left);
        return id;
    };
    // cfa source of tile pixel (r, col): amaze_demosaic_RT.c:361-469 for a tile without bottom apron; the top and left aprons of
    // the image's first tile row / column are mirrored, and so is the right apron of the extra tiles (:399-410 and, in the first
    // tile row, the corner :441-450 -- which, like the top-left one, counts its rows from 32, not from 32 + top)
    auto source = [&](int top, int left, int r, int col) -> size_t {
        const int rrmin = top < 0 ? 16 : 0, ccmin = left < 0 ? 16 : 0, ccmax = w - left;      // ccmax >= T: no right apron
        int y, x;
        if (col >= ccmax) { y = r >= rrmin ? r + top : 32 - r; x = w - (col - ccmax) - 2; }
        else if (r >= rrmin && col >= ccmin) { y = r + top; x = col + left; }
        else if (r < rrmin && col >= ccmin) { y = 32 - r + top; x = col + left; }
        else if (r >= rrmin) { y = r + top; x = 32 - col + left; }
        else { y = 32 - r; x = 32 - (col & ~3) + (col & 3); }
        return (size_t)y * w + x;
    };
    float pf[5] = { 0, 0, 0, 0, 0 };                              // the loader's row pair, one step ahead
    bool loader = false;
    for (int it = 0; it < 4; it++) loader = loader || ((c_desc[0][wave] >> (16 * it)) & 255ull) == IT(P_LOAD, 0);
    __syncthreads();                                               // (the rings are zero)
    auto next_tile = [&]() { int id = 0; if (lane == 0) id = atomicAdd(tile_ctr, 1); return __builtin_amdgcn_readfirstlane(id); };
    if (loader) {
        const int id = next_tile();
        if (lane == 0) { ctl[0] = id; ctl[4] = id < ntiles ? 0x7FFFFFFF : 0; }
        if (id < ntiles) {
            int top, left;
            origin(id, top, left);
#pragma unroll
            for (int ck = 0; ck < 5; ck++) {
                const int n = ck * 64 + lane, rho = n >= T, col = n - T * rho;
                pf[ck] = raw[source(top, left, rho, col)];
            }
        }
    }
    __syncthreads();

    const unsigned long long desc_a = c_desc[0][wave], desc_b = c_desc[1][wave];       // this wave's items
    unsigned long long prof_busy[2] = { 0, 0 }, prof_wait[2] = { 0, 0 }, prof_items[2] = { 0, 0 }, prof_n = 0, t_phase = 0;
    const bool prof_detail = prof && prof[2 * NPASS + 5 * 16] != 0;
    for (int s = 0;; s++) {
        const int nq = __builtin_amdgcn_readfirstlane(ctl[4]);         // (written in phase A of an earlier step, if at all)
        if (s - LAG_MAX >= nq) break;
#pragma unroll 1
        for (int phase = 0; phase < 2; phase++) {
#pragma unroll 1
            for (int it = 0; it < 4; it++) {
                const unsigned d16 = (unsigned)((phase ? desc_b : desc_a) >> (16 * it)) & 0xFFFFu;
                if (d16 == 0xFFFFu) break;
                const int pass = (d16 & 255u) >> 3, ck = d16 & 7u;
                if (skip_mask >> pass & 1u) continue;                              // (timing experiments only: MLVFS_AMD_AMAZE_ROWS_SKIP)
                const int q = s - (int)(d16 >> 8);
                if (q < 0 || q >= nq) continue;
                const unsigned long long t_item = prof_detail ? __builtin_amdgcn_s_memtime() : 0;
                const int p = q % NP;
                unsigned tabv;                                                    // lane j: byte offset of pair j of this pass's table (rows of pair q)
                {
                    const unsigned *te = (const unsigned *)(sm + RINGS_FLOATS) + (pass * NTAB + (lane & (NTAB - 1))) * 3;
                    const unsigned pk = te[0], x = 2u * (unsigned)s + (pk & 255u), R = (pk >> 8) & 255u;
                    tabv = te[2] + (x - __umulhi(x, te[1]) * R) * (pk >> 16);
                }
                float *dbgt = nullptr;
                if (DBG) { int t0, l0; dbgt = dbg + (size_t)tile_of(q, t0, l0) * AMAZE_TILE_FLOATS; }
                (void)dbgt;
                switch (pass) {
                // ---------------------------------------------------------------- tile rows in (:361-469)
                case P_LOAD: {
#undef PT
#define PT TAB_LOAD
#pragma unroll
                    for (int c5 = 0; c5 < 5; c5++) {
                        const int n = c5 * 64 + lane;
                        const float v = pf[c5] / 65535.0f;
                        ST(R_C, n, v);
                        if (DBG) { const int rho = n >= T, col = n - T * rho, r = 2 * p + rho; DBGF(D_CFA, v); }
                    }
                    const int p1 = (q + 1) % NP;
                    int id1 = __builtin_amdgcn_readfirstlane(ctl[((q + 1) / NP) & 3]);
                    if (p1 == 0) {                                                 // the next pair opens a tile: which one, if any
                        id1 = next_tile();
                        if (lane == 0) { if (id1 < ntiles) ctl[((q + 1) / NP) & 3] = id1; else ctl[4] = q + 1; }
                    }
                    if (id1 < ntiles) {
                        int top, left;
                        origin(id1, top, left);
#pragma unroll
                        for (int c5 = 0; c5 < 5; c5++) {
                            const int n = c5 * 64 + lane, rho = n >= T, col = n - T * rho;
                            pf[c5] = raw[source(top, left, 2 * p1 + rho, col)];
                        }
                    }
                } break;
                // ---------------------------------------------------------------- gradients (:537-613): dw1, dw0
                case P_GRAD: {
#undef PT
#define PT TAB_GRAD
                    FULL_LANES();
                    float d1v = 0.0f, d0v = 0.0f;
                    if (r >= 2 && r < T - 2) {
                        const float *c0 = FP(R_C, 0) + n;
                        const float cc = c0[0];
                        const float dh = fabsf(c0[1] - c0[-1]), dv = fabsf(FP(R_C, 1)[n] - FP(R_C, -1)[n]);
                        d1v = EPS + fabsf(c0[2] - cc) + fabsf(cc - c0[-2]) + dh;
                        d0v = EPS + fabsf(FP(R_C, 2)[n] - cc) + fabsf(cc - FP(R_C, -2)[n]) + dv;
                    }
                    ST(R_DW1, n, d1v); ST(R_DW0, n, d0v);
                    DBGF(D_DW1, d1v); DBGF(D_DW0, d0v);
                } break;
                // ---------------------------------------------------------------- directional colour differences (:622-675)
                case P_DIRDIFF: {
#undef PT
#define PT TAB_DIRDIFF
                    // no branches: one block for the scheduler.  (Two chunks per item -- two independent streams in one block -- took
                    // 1.49x the time of one: a gain per chunk, a loss for the balance of the phase, whose longest item it became.)
                    struct Out { float halt, valt, vcd, hcd, dgv, dgh; };
                    auto body = [&](const int ck) -> Out {
                        FULL_LANES();
                        const float *c0 = FP(R_C, 0) + n, *d1 = FP(R_DW1, 0) + n;
                        const float ci = c0[0], cu1 = FP(R_C, -1)[n], cu2 = FP(R_C, -2)[n], cd1 = FP(R_C, 1)[n], cd2 = FP(R_C, 2)[n];
                        const float d0c = FP(R_DW0, 0)[n], d0u1 = FP(R_DW0, -1)[n], d0u2 = FP(R_DW0, -2)[n], d0d1 = FP(R_DW0, 1)[n], d0d2 = FP(R_DW0, 2)[n];
                        const float sgn = ((r + col) & 1) ? -1.0f : 1.0f;
                        const float cru = cu1 * (d0u2 + d0c) / (d0u2 * (EPS + ci) + d0c * (EPS + cu2));
                        const float crd = cd1 * (d0d2 + d0c) / (d0d2 * (EPS + ci) + d0c * (EPS + cd2));
                        const float crl = c0[-1] * (d1[-2] + d1[0]) / (d1[-2] * (EPS + ci) + d1[0] * (EPS + c0[-2]));
                        const float crr = c0[1] * (d1[2] + d1[0]) / (d1[2] * (EPS + ci) + d1[0] * (EPS + c0[2]));
                        const float guha = cu1 + 0.5f * (ci - cu2), gdha = cd1 + 0.5f * (ci - cd2);
                        const float glha = c0[-1] + 0.5f * (ci - c0[-2]), grha = c0[1] + 0.5f * (ci - c0[2]);
                        float guar = fabsf(1.0f - cru) < ARTHRESH ? ci * cru : guha, gdar = fabsf(1.0f - crd) < ARTHRESH ? ci * crd : gdha;
                        float glar = fabsf(1.0f - crl) < ARTHRESH ? ci * crl : glha, grar = fabsf(1.0f - crr) < ARTHRESH ? ci * crr : grha;
                        const float hwt = d1[-1] / (d1[-1] + d1[1]), vwt = d0u1 / (d0d1 + d0u1);
                        const float ginth = hwt * grha + (1.0f - hwt) * glha, gintv = vwt * gdha + (1.0f - vwt) * guha;
                        const float halt = sgn * (ginth - ci), valt = sgn * (gintv - ci);
                        const bool clip = ci > CLIP_PT8 || gintv > CLIP_PT8 || ginth > CLIP_PT8;
                        guar = clip ? guha : guar; gdar = clip ? gdha : gdar; glar = clip ? glha : glar; grar = clip ? grha : grar;
                        const bool in = r >= 4 && r < T - 4 && col >= 4 && col < T - 4;
                        Out o;
                        o.halt = in ? halt : 0.0f; o.valt = in ? valt : 0.0f;
                        o.vcd = in ? (clip ? valt : sgn * ((vwt * gdar + (1.0f - vwt) * guar) - ci)) : 0.0f;
                        o.hcd = in ? (clip ? halt : sgn * ((hwt * grar + (1.0f - hwt) * glar) - ci)) : 0.0f;
                        o.dgv = in ? fminv(sq(guha - gdha), sq(guar - gdar)) : 0.0f;
                        o.dgh = in ? fminv(sq(glha - grha), sq(glar - grar)) : 0.0f;
                        return o;
                    };
                    const Out oa = body(ck);
                    {
                        const Out &o = oa;
                        const int n = ck * 64 + lane;
                        {
                            const int rho = n >= T, col = n - T * rho, r = 2 * p + rho; (void)col; (void)r;
                            ST(R_HCDALT, n, o.halt); ST(R_VCDALT, n, o.valt); ST(R_VCD, n, o.vcd);
                            ST(R_HCD, n, o.hcd); ST(R_DGV, n, o.dgv); ST(R_DGH, n, o.dgh);
                            DBGF(D_HCDALT, o.halt); DBGF(D_VCDALT, o.valt); DBGF(D_HCD, o.hcd); DBGF(D_DGV, o.dgv); DBGF(D_DGH, o.dgh);
                        }
                    }
                } break;
                // ---------------------------------------------------------------- refinement, horizontal (:766-801): hcd -> hcd2
                // lanes 2,3 of a 4-lane group read unrefined neighbours only; lanes 0,1 need the refined value of the previous
                // group's lanes 2,3, which they derive themselves (k_amaze.hip).  hcd is zero outside columns [4, 156).
                case P_HREF: {
#undef PT
#define PT TAB_HREF
                    FULL_LANES();
                    float v = 0.0f;
                    if (r >= 4 && r < T - 4 && col >= 4 && col < T - 4) {
                        const float *H = FP(R_HCD, 0) + n, *A = FP(R_HCDALT, 0) + n, *c0 = FP(R_C, 0) + n;
                        const float sgn = ((r + col) & 1) ? -1.0f : 1.0f;
                        auto refined = [&](int o, float leftval) {
                            const float hv = var3(leftval, H[o], H[o + 2]), hav = var3(A[o - 2], A[o], A[o + 2]);
                            return bound_difference(hav < hv ? A[o] : H[o], sgn, c0[o], c0[o - 1], c0[o + 1]);
                        };
                        float leftval = H[-2];
                        if (((col - 4) & 3) < 2 && col >= 8) leftval = refined(-2, H[-4]);
                        v = refined(0, leftval);
                    }
                    ST(R_HCD2, n, v);
                    DBGF(D_HCD2, v);
                } break;
                // ---------------------------------------------------------------- refinement, vertical: vcd in place (reads the
                // refined row two above = the previous pair), the squared difference of the two, and the gradient magnitude
                case P_VWALK: {
#undef PT
#define PT TAB_VWALK
                    FULL_LANES();
                    float o_cdsq = 0.0f, o_delsq = 0.0f;
                    if (r >= 4 && r < T - 4 && col >= 4 && col < T - 4) {
                        const float up = FP(R_VCD, -2)[n], v0 = FP(R_VCD, 0)[n], v1 = FP(R_VCD, 2)[n];
                        const float a_1 = FP(R_VCDALT, -2)[n], a0 = FP(R_VCDALT, 0)[n], a1 = FP(R_VCDALT, 2)[n];
                        const float *c0 = FP(R_C, 0) + n;
                        const float cm = FP(R_C, -1)[n], cp = FP(R_C, 1)[n], h2 = FP(R_HCD2, 0)[n];
                        const float sgn = ((r + col) & 1) ? -1.0f : 1.0f;
                        const float vv = var3(up, v0, v1), vav = var3(a_1, a0, a1);
                        const float v = bound_difference(vav < vv ? a0 : v0, sgn, c0[0], cm, cp);
                        ST(R_VCD, n, v);
                        DBGF(D_VCD, v);
                        o_cdsq = sq(v - h2);
                        const float dh = fabsf(c0[1] - c0[-1]), dv = fabsf(cp - cm);
                        o_delsq = dh * dh + dv * dv;
                    } else DBGF(D_VCD, 0.0f);
                    ST(R_CDSQ, n, o_cdsq); ST(R_DELSQ, n, o_delsq);
                    DBGF(D_CDSQ, o_cdsq); DBGF(D_DELSQ, o_delsq);
                } break;
                // ---------------------------------------------------------------- Nyquist texture test (:969-996)
                case P_NYQTEST: {
#undef PT
#define PT TAB_NYQTEST
                    HALF_LANES();
                    unsigned char flag = 0;
                    if (act && r >= 6 && r < T - 6 && hj >= 3 && hj < 77) {
                        const float G_ODD[4] = { 0.14659727707323927f, 0.103592713382435f, 0.0732036125103057f, 0.0365543548389495f };
                        const float G_GRAD[6] = { 0.07384411893421103f, 0.06207511968171489f, 0.0521818194747806f,
                                                  0.03687419286733595f, 0.03099732204057846f, 0.018413194161458882f };
#define Q(dr, dc) FP(R_CDSQ, dr)[f0 + (dc)]
#define D(dr, dc) FP(R_DELSQ, dr)[f0 + (dc)]
                        float test = (G_ODD[0] * Q(0, 0) + G_ODD[1] * (Q(-1, -1) + Q(-1, 1) + Q(1, -1) + Q(1, 1)) +
                                      G_ODD[2] * (Q(-2, 0) + Q(0, -2) + Q(0, 2) + Q(2, 0)) + G_ODD[3] * (Q(-2, -2) + Q(-2, 2) + Q(2, -2) + Q(2, 2)));
                        test -= NYQTHRESH * (G_GRAD[0] * D(0, 0) + G_GRAD[1] * (D(-1, 0) + D(0, 1) + D(0, -1) + D(1, 0)) +
                                             G_GRAD[2] * (D(-1, -1) + D(-1, 1) + D(1, -1) + D(1, 1)) + G_GRAD[3] * (D(-2, 0) + D(0, -2) + D(0, 2) + D(2, 0)) +
                                             G_GRAD[4] * (D(-2, -1) + D(-2, 1) + D(-1, -2) + D(-1, 2) + D(1, -2) + D(1, 2) + D(2, -1) + D(2, 1)) +
                                             G_GRAD[5] * (D(-2, -2) + D(-2, 2) + D(2, -2) + D(2, 2)));
#undef Q
#undef D
                        flag = test > 0 ? 1 : 0;
                    }
                    if (act) NST(n, flag);
                } break;
                // ---------------------------------------------------------------- horizontal / vertical weight (:881-925)
                case P_HVWT: {
#undef PT
#define PT TAB_HVWT
                    HALF_LANES();
                    float o_w = 0.0f, o_hc = 0.0f, o_vc = 0.0f;
                    if (act && r >= 6 && r < T - 6 && hj >= 3 && hj < 79) {
                        const float *hc = FP(R_HCD2, 0) + f0, *dgh = FP(R_DGH, 0) + f0, *dw1 = FP(R_DW1, 0) + f0;
                        const float vc0 = FP(R_VCD, 0)[f0], vcu1 = FP(R_VCD, -1)[f0], vcu2 = FP(R_VCD, -2)[f0], vcu3 = FP(R_VCD, -3)[f0];
                        const float vcd1 = FP(R_VCD, 1)[f0], vcd2 = FP(R_VCD, 2)[f0], vcd3 = FP(R_VCD, 3)[f0];
                        const float uave = vc0 + vcu1 + vcu2 + vcu3, dave = vc0 + vcd1 + vcd2 + vcd3;
                        float vu = sq(vc0 - uave) + sq(vcu1 - uave) + sq(vcu2 - uave) + sq(vcu3 - uave);
                        float vd = sq(vc0 - dave) + sq(vcd1 - dave) + sq(vcd2 - dave) + sq(vcd3 - dave);
                        const float hwt = dw1[-1] / (dw1[-1] + dw1[1]);
                        const float dw0u = FP(R_DW0, -1)[f0], dw0d = FP(R_DW0, 1)[f0];
                        const float vwt = dw0u / (dw0d + dw0u);
                        const float lave = hc[0] + hc[-1] + hc[-2] + hc[-3], rave = hc[0] + hc[1] + hc[2] + hc[3];
                        float hl = sq(hc[0] - lave) + sq(hc[-1] - lave) + sq(hc[-2] - lave) + sq(hc[-3] - lave);
                        float hr = sq(hc[0] - rave) + sq(hc[1] - rave) + sq(hc[2] - rave) + sq(hc[3] - rave);
                        const float vcdvar = EPSSQ + vwt * vd + (1.0f - vwt) * vu, hcdvar = EPSSQ + hwt * hr + (1.0f - hwt) * hl;
                        const float gv0 = FP(R_DGV, 0)[f0];
                        vu = gv0 + FP(R_DGV, -1)[f0] + FP(R_DGV, -2)[f0];
                        vd = gv0 + FP(R_DGV, 1)[f0] + FP(R_DGV, 2)[f0];
                        hl = dgh[0] + dgh[-1] + dgh[-2];
                        hr = dgh[0] + dgh[1] + dgh[2];
                        const float vcdvar1 = EPSSQ + vwt * vd + (1.0f - vwt) * vu, hcdvar1 = EPSSQ + hwt * hr + (1.0f - hwt) * hl;
                        const float varwt = hcdvar / (vcdvar + hcdvar), diffwt = hcdvar1 / (vcdvar1 + hcdvar1);
                        const bool agree = (0.5f - varwt) * (0.5f - diffwt) > 0.0f && fabsf(0.5f - diffwt) < fabsf(0.5f - varwt);
                        o_w = agree ? varwt : diffwt;
                        o_hc = hc[0]; o_vc = vc0;
                    }
                    if (act) {
                        ST(R_HVWT, n, o_w); ST(R_HCRB, n, o_hc); ST(R_VCRB, n, o_vc);
                        DBGH(D_HVWT, o_w);
                    }
                } break;
                // ---------------------------------------------------------------- majority vote in raster order (:998-1010)
                // new[k] = f_k(new[k-1]): with s' the sum of the eight other neighbours (rows above final, rows below and the right
                // neighbour as tested), f_k(x) = 1 if s' + x > 4, 0 if s' + x < 4, else the site's own flag.  Prefix composition.
                case P_VOTE: {
#undef PT
#define PT TAB_VOTE
#pragma unroll 1
                    for (int sub = 0; sub < 2; sub++) {
                        const int r = 2 * p + sub, par = sub;
                        if (r < 8 || r >= T - 8) continue;
                        const unsigned char *u2 = FPN(sub - 2), *u1 = FPN(sub - 1), *d1 = FPN(sub + 1), *d2 = FPN(sub + 2);
                        unsigned char *cur = FPN(sub);
                        const bool row0 = cur == (unsigned char *)sm + 4u * NYQ_OFF;
                        unsigned any = 0;
                        if (lane < HT / 4) any = ((const unsigned *)u2)[lane] | ((const unsigned *)u1)[lane] | ((const unsigned *)cur)[lane] |
                                                 ((const unsigned *)d1)[lane] | ((const unsigned *)d2)[lane];
                        if (!__any(any != 0)) continue;
                        unsigned x_in = cur[3];                                           // the site left of the first voted one keeps its flag
#pragma unroll 1
                        for (int blk = 0; blk < 2; blk++) {
                            const int k = blk * 64 + lane, hj = 4 + k;
                            const bool on = k < 72;
                            unsigned own = 0, sp = 0;
                            if (on) {
                                own = cur[hj];
                                sp = u2[hj] + u1[hj - 1 + par] + u1[hj + par] + own + cur[hj + 1] + d1[hj - 1 + par] + d1[hj + par] + d2[hj];
                            }
                            unsigned f_0 = sp > 4 ? 1u : (sp < 4 ? 0u : own), f_1 = sp + 1 > 4 ? 1u : (sp + 1 < 4 ? 0u : own);
                            if (!on) { f_0 = 0; f_1 = 1; }                                // identity
#pragma unroll
                            for (int o = 1; o < 64; o <<= 1) {
                                const unsigned p0 = __shfl_up(f_0, o), p1 = __shfl_up(f_1, o);
                                if (lane >= o) { const unsigned n0 = p0 ? f_1 : f_0, n1 = p1 ? f_1 : f_0; f_0 = n0; f_1 = n1; }
                            }
                            const unsigned nv = x_in ? f_1 : f_0;
                            __builtin_amdgcn_wave_barrier();
                            if (on) {
                                cur[hj] = (unsigned char)nv;
                                if (row0) cur[NYQ_ROWS * HT + hj] = (unsigned char)nv;
                            }
                            x_in = __shfl(nv, 63);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    }
                } break;
                // ---------------------------------------------------------------- area interpolation in Nyquist regions (:1016-1044)
                case P_AREA: {
#undef PT
#define PT TAB_AREA
                    // The flagged sites of the pair are few where they exist at all, and a wave pays for the 49-tap loop whether one
                    // of its lanes works or all: each of the three items lists the pair's flagged sites (ballots, ranks, a byte list in
                    // LDS of its own) and takes the 64 * ck-th ... of them -- one loop per 64 flagged sites, not per 64 sites.
                    unsigned char *list = (unsigned char *)(sm + RINGS_FLOATS - LDS_PAD - AREA_LIST_FLOATS) + ck * T;
                    int nflag = 0;
#pragma unroll
                    for (int c3 = 0; c3 < 3; c3++) {
                        const int n3 = c3 * 64 + lane, rho3 = n3 >= HT, hj3 = n3 - HT * rho3, r3 = 2 * p + rho3;
                        const bool fl3 = n3 < T && r3 >= 8 && r3 < T - 8 && hj3 >= 4 && hj3 < 76 && FPN(0)[n3] != 0;
                        const unsigned long long m = __ballot(fl3);
                        if (fl3) list[nflag + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned char)n3;
                        nflag += __popcll(m);
                    }
                    if (nflag <= 64 * ck) break;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    const bool flag = 64 * ck + lane < nflag;
                    const int n = flag ? list[64 * ck + lane] : 0;
                    const int rho = n >= HT, hj = n - HT * rho, col = 2 * hj + rho, r = 2 * p + rho, f0 = T * rho + col; (void)col; (void)r;
                    if (flag) {
                        // a row of the 7x7 window at a time: its seven flags and its 29 cfa values are loaded together, whatever the flags
                        // say (one LDS round trip per row instead of two per tap), and a tap that is not flagged adds nothing -- by
                        // select, in the reference's order of additions
                        float sumh = 0, sumv = 0, sumsqh = 0, sumsqv = 0, area = 0;
#pragma unroll 1
                        for (int a = -6; a < 7; a += 2) {
                            const unsigned char *ny = FPN(a) + n;
                            const float *ca = FP(R_C, a) + f0, *cu = FP(R_C, a - 1) + f0, *cd = FP(R_C, a + 1) + f0;
                            unsigned char fl[7];
                            float vc_[15], vu_[7], vd_[7];
#pragma unroll
                            for (int k = 0; k < 7; k++) { fl[k] = ny[k - 3]; vu_[k] = cu[2 * k - 6]; vd_[k] = cd[2 * k - 6]; }
#pragma unroll
                            for (int k = 0; k < 15; k++) vc_[k] = ca[k - 7];
#pragma unroll
                            for (int k = 0; k < 7; k++) {
                                const bool on = fl[k] != 0;
                                const float cj = vc_[2 * k + 1], cl = vc_[2 * k], cr = vc_[2 * k + 2];
                                const float th = cj - half_exp(cl + cr), tv = cj - half_exp(vu_[k] + vd_[k]);
                                const float qh = half_exp(sq(cj - cl) + sq(cj - cr)), qv = half_exp(sq(cj - vu_[k]) + sq(cj - vd_[k]));
                                sumh = on ? sumh + th : sumh;
                                sumv = on ? sumv + tv : sumv;
                                sumsqh = on ? sumsqh + qh : sumsqh;
                                sumsqv = on ? sumsqv + qv : sumsqv;
                                area = on ? area + 1.0f : area;
                            }
                        }
                        const float hvar = EPSSQ + fabsf(area * sumsqh - sumh * sumh), vvar = EPSSQ + fabsf(area * sumsqv - sumv * sumv);
                        const float v = hvar / (vvar + hvar);
                        ST(R_HVWT, n, v);
                        DBGH(D_HVWT, v);
                    }
                } break;
                // ---------------------------------------------------------------- hvwt asks its neighbours (:1049-1056): a row reads
                // the updated row above, the rows of a pair one after the other in this wave
                case P_HVSWEEP: {
#undef PT
#define PT TAB_HVSWEEP
#pragma unroll 1
                    for (int sub = 0; sub < 2; sub++) {
                        const int r = 2 * p + sub, par = sub;
                        if (r < 8 || r >= T - 8) continue;
                        const float *up = FP(R_HVWT, sub - 1), *dn = FP(R_HVWT, sub + 1);
                        float *cur = FP(R_HVWT, sub);
                        const bool row0 = cur == sm + ring_off(R_HVWT);
                        float nv[2]; bool wr[2];
#pragma unroll
                        for (int blk = 0; blk < 2; blk++) {
                            const int k = blk * 64 + lane, hj = 4 + k;
                            wr[blk] = false; nv[blk] = 0;
                            if (k < 72) {
                                const float alt = quarter_exp(up[hj - 1 + par] + up[hj + par] + dn[hj - 1 + par] + dn[hj + par]);
                                if (fabsf(0.5f - cur[hj]) < fabsf(0.5f - alt)) { wr[blk] = true; nv[blk] = alt; }
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int blk = 0; blk < 2; blk++) {
                            const int hj = 4 + blk * 64 + lane;
                            if (wr[blk]) {
                                cur[hj] = nv[blk];
                                if (row0) cur[RR[R_HVWT] * HT + hj] = nv[blk];
                                if (DBG) dbgt[dbg_off(D_HVWT) + (size_t)r * HT + hj] = nv[blk];
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    }
                } break;
                // ---------------------------------------------------------------- G at R/B sites, first estimate, and its curvature
                // where the Nyquist test fired (:1046-1073); the estimate itself is recomputed by P_GFINAL
                case P_CURV: {
#undef PT
#define PT TAB_CURV
                    HALF_LANES();
                    float ch = 0.0f, cv = 0.0f;
                    if (act && r >= 8 && r < T - 8 && hj >= 4 && hj < 76 && FPN(0)[n]) {
                        const float hw = FP(R_HVWT, 0)[n], hc = FP(R_HCRB, 0)[n], vc = FP(R_VCRB, 0)[n];
                        const float *c0 = FP(R_C, 0) + f0;
                        const float dg = hc * (1.0f - hw) + vc * hw;
                        const float gi = c0[0] + dg;
                        ch = sq(gi - half_exp(c0[-1] + c0[1]));
                        cv = sq(gi - half_exp(FP(R_C, -1)[f0] + FP(R_C, 1)[f0]));
                    }
                    if (act) { ST(R_CURVH, n, ch); ST(R_CURVV, n, cv); DBGH(D_CURVH, ch); DBGH(D_CURVV, cv); }
                } break;
                // ---------------------------------------------------------------- diagonal gradients (:596-612): per column pair,
                // delp / delm at the R/B pixel, sqp / sqm at the green one
                case P_CGRAD: {
#undef PT
#define PT TAB_CGRAD
                    HALF_LANES();
                    float o_dp = 0, o_dm = 0, o_sp = 0, o_sm = 0;
                    if (act && r >= 6 && r < T - 6 && hj >= 3 && hj < 79) {
                        const int fg = T * rho + 2 * hj + 1 - rho;
                        const float *cu = FP(R_C, -1), *c0 = FP(R_C, 0), *cd = FP(R_C, 1);
                        o_dp = fabsf(cu[f0 + 1] - cd[f0 - 1]);
                        o_dm = fabsf(cd[f0 + 1] - cu[f0 - 1]);
                        const float cg = c0[fg];
                        o_sp = sq(cg - cd[fg - 1]) + sq(cg - cu[fg + 1]);
                        o_sm = sq(cg - cu[fg - 1]) + sq(cg - cd[fg + 1]);
                    }
                    if (act) {
                        ST(R_DELP, n, o_dp); ST(R_DELM, n, o_dm); ST(R_SQP, n, o_sp); ST(R_SQM, n, o_sm);
                        DBGH(D_DELP, o_dp); DBGH(D_DELM, o_dm); DBGH(D_SQP, o_sp); DBGH(D_SQM, o_sm);
                    }
                } break;
                // ---------------------------------------------------------------- diagonal interpolation (:1112-1262)
                case P_DIAG: {
#undef PT
#define PT TAB_DIAG
                    HALF_LANES();
                    float o_rbm = 0, o_rbp = 0, o_pm = 0;
                    if (act && r >= 8 && r < T - 8 && hj >= 4 && hj < 76) {
                        const float G_EVEN[2] = { 0.13719494435797422f, 0.05640252782101291f };
#define C(dr, dc) FP(R_C, dr)[f0 + (dc)]
#define HP(K, dr, dc) FP(K, dr)[HX(dc)]
                        const float ci = C(0, 0);
                        const float se = diag_estimate(ci, C(1, 1), C(2, 2)), nw = diag_estimate(ci, C(-1, -1), C(-2, -2));
                        const float base_m = EPS + HP(R_DELM, 0, 0);
                        const float wse = base_m + HP(R_DELM, 1, 1) + HP(R_DELM, 2, 2);
                        const float wnw = base_m + HP(R_DELM, -1, -1) + HP(R_DELM, -2, -2);
                        o_rbm = diag_bound((wse * nw + wnw * se) / (wse + wnw), ci, C(-1, -1), C(1, 1));
                        const float ne = diag_estimate(ci, C(-1, 1), C(-2, 2)), sw = diag_estimate(ci, C(1, -1), C(2, -2));
                        const float base_p = EPS + HP(R_DELP, 0, 0);
                        const float wne = base_p + HP(R_DELP, -1, 1) + HP(R_DELP, -2, 2);
                        const float wsw = base_p + HP(R_DELP, 1, -1) + HP(R_DELP, 2, -2);
                        o_rbp = diag_bound((wne * sw + wsw * ne) / (wne + wsw), ci, C(1, -1), C(-1, 1));
#define EVEN_RING(K) (EPSSQ + (G_EVEN[0] * (HP(K, -1, 0) + HP(K, 0, -1) + HP(K, 0, 1) + HP(K, 1, 0)) +                                       \
                               G_EVEN[1] * (HP(K, -2, -1) + HP(K, -2, 1) + HP(K, -1, -2) + HP(K, -1, 2) + HP(K, 1, -2) + HP(K, 1, 2) + HP(K, 2, -1) + \
                                            HP(K, 2, 1))))
                        const float varm = EVEN_RING(R_SQM);
                        o_pm = varm / (EVEN_RING(R_SQP) + varm);
#undef EVEN_RING
                    }
                    if (act) {
                        ST(R_RBM, n, o_rbm); ST(R_RBP, n, o_rbp); ST(R_PMWT, n, o_pm);
                        DBGH(D_RBM, o_rbm); DBGH(D_RBP, o_rbp); DBGH(D_PMWT, o_pm);
                    }
                } break;
                // ---------------------------------------------------------------- pmwt asks its neighbours (:1266-1276)
                case P_PMSWEEP: {
#undef PT
#define PT TAB_PMSWEEP
#pragma unroll 1
                    for (int sub = 0; sub < 2; sub++) {
                        const int r = 2 * p + sub, par = sub;
                        if (r < 10 || r >= T - 10) continue;
                        const float *up = FP(R_PMWT, sub - 1), *dn = FP(R_PMWT, sub + 1);
                        float *cur = FP(R_PMWT, sub);
                        const bool row0 = cur == sm + ring_off(R_PMWT);
                        float nv[2];
#pragma unroll
                        for (int blk = 0; blk < 2; blk++) {
                            const int k = blk * 64 + lane, hj = 5 + k;
                            nv[blk] = 0;
                            if (k < 72) {
                                const float alt = 0.25f * (up[hj - 1 + par] + up[hj + par] + dn[hj - 1 + par] + dn[hj + par]);
                                const float c1 = cur[hj];
                                nv[blk] = fabsf(0.5f - c1) < fabsf(0.5f - alt) ? alt : c1;
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int blk = 0; blk < 2; blk++) {
                            const int k = blk * 64 + lane, hj = 5 + k;
                            if (k < 72) {
                                cur[hj] = nv[blk];
                                if (row0) cur[RR[R_PMWT] * HT + hj] = nv[blk];
                                if (DBG) dbgt[dbg_off(D_PMWT) + (size_t)r * HT + hj] = nv[blk];
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    }
                } break;
                // ---------------------------------------------------------------- R/B at the other colour's sites from the two
                // diagonals and the site's final weight (:1278-1282 as restructured in k_amaze.hip)
                case P_RBINT: {
#undef PT
#define PT TAB_RBINT
                    HALF_LANES();
                    float v = 0.0f;
                    if (act && r >= 10 && r < T - 10 && hj >= 5 && hj < 77) {
                        const float nw = FP(R_PMWT, 0)[n];
                        v = 0.5f * (FP(R_C, 0)[f0] + FP(R_RBM, 0)[n] * (1.0f - nw) + FP(R_RBP, 0)[n] * nw);
                    }
                    if (act) { ST(R_RBINT, n, v); DBGH(D_RBINT, v); }
                } break;
                // ---------------------------------------------------------------- G at R/B sites, final: the first estimate
                // (:1057-1073), its Nyquist refinement (:1081-1101), the estimate from the diagonal interpolation where that is the
                // more decisive one (:1284-1338), and the move of G-B to its own plane at B sites (:1345-1352)
                case P_GFINAL: {
#undef PT
#define PT TAB_GFINAL
                    HALF_LANES();
                    float dg = 0.0f, gi = 0.0f;
                    const float *c0 = FP(R_C, 0) + f0;
                    const float ci = act ? c0[0] : 0.0f;
                    if (act && r >= 8 && r < T - 8 && hj >= 4 && hj < 76) {
                        const float hw = FP(R_HVWT, 0)[n], hc = FP(R_HCRB, 0)[n], vc = FP(R_VCRB, 0)[n];
                        dg = hc * (1.0f - hw) + vc * hw;
                        if (FPN(0)[n]) {
                            const float G_QUINC[4] = { 0.169917f, 0.108947f, 0.069855f, 0.0287182f };
#define QRING(K) (G_QUINC[0] * HP(K, 0, 0) + G_QUINC[1] * (HP(K, -1, -1) + HP(K, -1, 1) + HP(K, 1, -1) + HP(K, 1, 1)) +      \
                  G_QUINC[2] * (HP(K, -2, 0) + HP(K, 0, -2) + HP(K, 0, 2) + HP(K, 2, 0)) +                                    \
                  G_QUINC[3] * (HP(K, -2, -2) + HP(K, -2, 2) + HP(K, 2, -2) + HP(K, 2, 2)))
                            const float gvarh = EPSSQ + QRING(R_CURVH), gvarv = EPSSQ + QRING(R_CURVV);
#undef QRING
                            dg = (hc * gvarv + vc * gvarh) / (gvarv + gvarh);
                        }
                        gi = ci + dg;
                    }
                    if (act && r >= 12 && r < T - 12 && hj >= 6 && hj < 74) {
                        const float hw = FP(R_HVWT, 0)[n];
                        if (!(fabsf(0.5f - FP(R_PMWT, 0)[n]) < fabsf(0.5f - hw))) {
                            // sic: the half-width rbint plane is offset by a FULL row in the reference (indx1 - v1), :1289-1290
                            const float rb = FP(R_RBINT, 0)[n], rbu = FP(R_RBINT, -2)[n], rbd = FP(R_RBINT, 2)[n], rbl = FP(R_RBINT, 0)[n - 1], rbr = FP(R_RBINT, 0)[n + 1];
                            const float cu1 = C(-1, 0), cd1 = C(1, 0), cl1 = c0[-1], cr1 = c0[1];
                            // the reference divides in double (a 2.0 literal) and stores a float: for float operands that IS the correctly rounded
                            // float quotient (53 >= 2 * 24 + 2 bits: the second rounding is innocuous), so binary32 division gives the same bits
                            const float cru = (cu1 * 2.0f) / (EPS + rb + rbu), crd = (cd1 * 2.0f) / (EPS + rb + rbd);
                            const float crl = (cl1 * 2.0f) / (EPS + rb + rbl), crr = (cr1 * 2.0f) / (EPS + rb + rbr);
                            const float gu = fabsf(1.0f - cru) < ARTHRESH ? rb * cru : cu1 + half_exp(rb - rbu);
                            const float gd = fabsf(1.0f - crd) < ARTHRESH ? rb * crd : cd1 + half_exp(rb - rbd);
                            const float gl = fabsf(1.0f - crl) < ARTHRESH ? rb * crl : cl1 + half_exp(rb - rbl);
                            const float gr = fabsf(1.0f - crr) < ARTHRESH ? rb * crr : cr1 + half_exp(rb - rbr);
                            // dw0 one row up / down and dw1 one column left / right, from cfa (the expression of P_GRAD)
                            const float cu2 = C(-2, 0), cu3 = C(-3, 0), cd2 = C(2, 0), cd3 = C(3, 0);
                            const float dw0u = EPS + fabsf(cd1 - cu1) + fabsf(cu1 - cu3) + fabsf(ci - cu2);
                            const float dw0d = EPS + fabsf(cd3 - cd1) + fabsf(cd1 - cu1) + fabsf(cd2 - ci);
                            const float dw1l = EPS + fabsf(cr1 - cl1) + fabsf(cl1 - c0[-3]) + fabsf(ci - c0[-2]);
                            const float dw1r = EPS + fabsf(c0[3] - cr1) + fabsf(cr1 - cl1) + fabsf(c0[2] - ci);
                            float gv = (dw0u * gd + dw0d * gu) / (dw0d + dw0u);
                            float gh = (dw1l * gr + dw1r * gl) / (dw1l + dw1r);
                            if (gv < rb) {
                                if (2.0f * gv < rb) gv = ulim(gv, cu1, cd1);
                                else { const float wt = (2.0f * (rb - gv)) / (EPS + gv + rb); gv = wt * gv + (1.0f - wt) * ulim(gv, cu1, cd1); }
                            }
                            if (gh < rb) {
                                if (2.0f * gh < rb) gh = ulim(gh, cl1, cr1);
                                else { const float wt = (2.0f * (rb - gh)) / (EPS + gh + rb); gh = wt * gh + (1.0f - wt) * ulim(gh, cl1, cr1); }
                            }
                            if (gh > CLIP_PT) gh = ulim(gh, cl1, cr1);
                            if (gv > CLIP_PT) gv = ulim(gv, cu1, cd1);
                            gi = gh * (1.0f - hw) + gv * hw;
                            dg = gi - ci;
                        }
                    }
                    if (act) {
                        const bool bsite = (r & 1) && r >= 13 && r < T - 12 && hj >= 6 && hj < 74;
                        const float d0 = bsite ? 0.0f : dg, d1 = bsite ? dg : 0.0f;
                        ST(R_DGRB0, n, d0); ST(R_DGRB1, n, d1); ST(R_GRB, n, gi);
                        DBGH(D_DGRB0, d0); DBGH(D_DGRB1, d1);
                        if (DBG && r >= 8 && r < T - 8 && hj >= 4 && hj < 76) dbgt[dbg_off(D_GREEN) + (size_t)r * T + col] = gi;
                    }
                } break;
                // ---------------------------------------------------------------- chrominance at the other colour's sites
                // (:1354-1395): even rows work on G-B (dgrb1), odd rows on G-R (dgrb0); the two rings lie back to back
                case P_CHROMA: {
#undef PT
#define PT TAB_CHROMA
                    HALF_LANES();
                    static_assert(RR[R_DGRB0] == RR[R_DGRB1] && R_DGRB1 == R_DGRB0 + 1, "the two chrominance rings share their addressing");
                    if (act && r >= 14 && r < T - 14 && hj >= 7 && hj < 75) {
                        const int dsel = rho ? 0 : (RR[R_DGRB0] + 1) * HT;
#define DD(dr, dc) FP(R_DGRB0, dr)[dsel + HX(dc)]
                        // flat offsets of k_amaze.hip as (row, column): -M1 (-1,-1)  M1 (1,1)  -M3 (-3,-3)  M3 (3,3)  P1 (-1,1)  -P1 (1,-1)
                        // P3 (-3,3)  -P3 (3,-3)  -M1-2 (-1,-3)  -M1-V2 (-3,-1)  P1+2 (-1,3)  P1+V2 (1,1)  -P1-2 (1,-3)  -P1-V2 (-1,-1)
                        // M1+2 (1,3)  M1+V2 (3,1)
                        const float nwv = DD(-1, -1), sev = DD(1, 1), nev = DD(-1, 1), swv = DD(1, -1);
                        const float nw3 = DD(-3, -3), se3 = DD(3, 3), ne3 = DD(-3, 3), sw3 = DD(3, -3);
                        const float wnw = 1.0f / (EPS + fabsf(nwv - sev) + fabsf(nwv - nw3) + fabsf(sev - nw3));
                        const float wne = 1.0f / (EPS + fabsf(nev - swv) + fabsf(nev - ne3) + fabsf(swv - ne3));
                        const float wsw = 1.0f / (EPS + fabsf(swv - nev) + fabsf(swv - se3) + fabsf(nev - sw3));
                        const float wse = 1.0f / (EPS + fabsf(sev - nwv) + fabsf(sev - sw3) + fabsf(nwv - se3));
                        const float v = (wnw * (1.325f * nwv - 0.175f * nw3 - 0.075f * DD(-1, -3) - 0.075f * DD(-3, -1)) +
                                         wne * (1.325f * nev - 0.175f * ne3 - 0.075f * DD(-1, 3) - 0.075f * DD(1, 1)) +
                                         wsw * (1.325f * swv - 0.175f * sw3 - 0.075f * DD(1, -3) - 0.075f * DD(-1, -1)) +
                                         wse * (1.325f * sev - 0.175f * se3 - 0.075f * DD(1, 3) - 0.075f * DD(3, 1))) /
                                        (wnw + wne + wsw + wse);
#undef DD
                        float *pr = FP(R_DGRB0, 0) + dsel;
                        pr[n] = v;
                        if (FOFF(R_DGRB0, 0) == 4u * ring_off(R_DGRB0) && n < HT) pr[RR[R_DGRB0] * HT + n] = v;
                        if (DBG) dbgt[dbg_off(rho ? D_DGRB0 : D_DGRB1) + (size_t)r * HT + hj] = v;
                    }
                } break;
                // ---------------------------------------------------------------- the three planes of the tile interior (:1397-1470)
                case P_OUTPUT: {
#undef PT
#define PT TAB_OUTPUT
                    const int n = ck * 64 + lane, rho = n >> 7, col = 16 + (n & 127), r = 2 * p + rho;
                    if (r < 16 || r >= T - 16) break;
                    int top, left;
                    tile_of(q, top, left);
                    const size_t o = (size_t)(r + top) * w + (col + left);
                    const int hb = HT * rho;
                    float g, rv, bv;
                    if ((r + col) & 1) {
                        const int hu = hb + (col >> 1), hl = hb + ((col - 1) >> 1), hr = hb + ((col + 1) >> 1);
                        g = FP(R_C, 0)[T * rho + col];
                        const float wu = FP(R_HVWT, -1)[hu], wr = 1.0f - FP(R_HVWT, 0)[hr], wl = 1.0f - FP(R_HVWT, 0)[hl], wd = FP(R_HVWT, 1)[hu];
                        const float inv = 1.0f / (wu + wr + wl + wd);
                        rv = 65535.0f * (g - (wu * FP(R_DGRB0, -1)[hu] + wr * FP(R_DGRB0, 0)[hr] + wl * FP(R_DGRB0, 0)[hl] + wd * FP(R_DGRB0, 1)[hu]) * inv);
                        bv = 65535.0f * (g - (wu * FP(R_DGRB1, -1)[hu] + wr * FP(R_DGRB1, 0)[hr] + wl * FP(R_DGRB1, 0)[hl] + wd * FP(R_DGRB1, 1)[hu]) * inv);
                    } else {
                        const int hc = hb + (col >> 1);
                        g = FP(R_GRB, 0)[hc];
                        rv = 65535.0f * (g - FP(R_DGRB0, 0)[hc]);
                        bv = 65535.0f * (g - FP(R_DGRB1, 0)[hc]);
                    }
                    if (r2e) {                                                    // (k_amaze.hip: the conversion's look-ups instead of the planes)
                        int er, eg, eb, ey;
                        amz::ev_of_planes(r2e, ev_black, rv, g * 65535.0f, bv, er, eg, eb, ey);
                        ((int *)red)[o] = er; ((int *)green_out)[o] = eg; ((int *)blue)[o] = eb; gray[o] = ey;
                    } else {
                        red[o] = rv;
                        blue[o] = bv;
                        green_out[o] = g * 65535.0f;
                    }
                } break;
                default: break;
                }
                if (prof_detail && s >= 2 * LAG_MAX && s < nq && blockIdx.x == 0 && blockIdx.y == 0) {   // steady state: cycles per item
                    const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_item;
                    prof_items[phase] += dt;
                    if (lane == 0) { atomicAdd(&prof[2 * pass], dt); atomicAdd(&prof[2 * pass + 1], 1ull); }
                }
            }
            const unsigned long long t_bar = prof ? __builtin_amdgcn_s_memtime() : 0;
            if (!(skip_mask >> 31)) __syncthreads();                               // (bit 31: timing of the empty loop without its barriers)
            if (prof && s >= 2 * LAG_MAX && s < nq) {
                const unsigned long long t_now = __builtin_amdgcn_s_memtime();
                prof_busy[phase] += t_bar - t_phase; prof_wait[phase] += t_now - t_bar; prof_n++;
                t_phase = t_now;
            } else if (prof) t_phase = __builtin_amdgcn_s_memtime();
        }
    }
    if (prof && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && prof_n) {     // per wave of workgroup 0: busy A, wait A, busy B, wait B, phases
        unsigned long long *o = prof + 2 * NPASS + 5 * wave;
        o[0] = prof_busy[0]; o[1] = prof_wait[0]; o[2] = prof_busy[1]; o[3] = prof_wait[1];
        if (prof_detail) { o[1] = prof_items[0]; o[3] = prof_items[1]; }                // (detail mode: the items' own time in place of the waits)
        o[4] = prof_n | (unsigned long long)__builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11)) << 56;       // HW_ID.SIMD_ID
    }
}
#undef C
#undef HP

int g_amaze_rows_mode = -1;        // -1: MLVFS_AMD_AMAZE_ROWS decides (default on), 0 / 1: forced (the debug entry point)

// Complete tiles of a w x h plane that this kernel takes: the first nfx columns and nfy rows of the tile grid -- tiles whose
// 160 rows and columns lie inside the image (no right / bottom apron) and that do not head a chain of incomplete tiles.
void amaze_rows_extent(int w, int h, int *nfx, int *nfy)
{
    static const bool off = [] { const char *e = getenv("MLVFS_AMD_AMAZE_ROWS"); return e && atoi(e) == 0; }();
    *nfx = *nfy = 0;
    if (g_amaze_rows_mode == 0 || (g_amaze_rows_mode < 0 && off)) return;
    const int step = T - 32;
    const int tiles_x = (w + 16 + step - 1) / step, tiles_y = (h + 16 + step - 1) / step;
    const int cc1_last = w + 16 - (-16 + (tiles_x - 1) * step), rr1_last = h + 16 - (-16 + (tiles_y - 1) * step);
    const int incomplete_x = cc1_last >= T ? 0 : (cc1_last < 32 ? 2 : 1), incomplete_y = rr1_last >= T ? 0 : (rr1_last < 32 ? 2 : 1);
    if (tiles_x < incomplete_x + 1 || tiles_x < 3) return;                          // small image: one workgroup walks it in the reference's order
    const int wgs_per_row = tiles_x - incomplete_x, rows_a = tiles_y - incomplete_y;
    int fx = w >= T - 16 ? (w - (T - 16)) / step + 1 : 0, fy = h >= T - 16 ? (h - (T - 16)) / step + 1 : 0;   // -16 + 128 t + 160 <= w
    fx = fx < wgs_per_row - (incomplete_x ? 1 : 0) ? fx : wgs_per_row - (incomplete_x ? 1 : 0);
    fy = fy < rows_a ? fy : rows_a;
    if (fx <= 0 || fy <= 0) return;
    *nfx = fx; *nfy = fy;
}

// Tiles of column nfx (the heads of k_amaze.hip's chains) that this kernel takes as well, rows 0 .. n - 1.  Where the width is a multiple
// of 128 the one incomplete tile of a row is 32 columns wide -- all apron, no output pixel (amaze_demosaic_RT.c:1459: the interior
// starts 16 columns in and ends 16 before the end) -- and the block it leaves behind is read by nobody except, after the LAST
// row of complete tiles, the launch for the incomplete bottom row (k_amaze.hip: amaze_launch, `src`).  The head of such a chain is a
// tile like any other but for its last 16 columns, which mirror the image's right edge; k_amaze.hip then skips the whole chain.
// Measured (tools/ab_di_bench.sh, batch of 8 at 3584x1320, three rounds round-robin): 8.00 ms per batch without, 8.09 with -- the 90 chains
// ran beside the row kernel on an otherwise idle part of the chip, their 9 heads per frame lengthen the row kernel's last round
// (1 116 instead of 1 080 tiles over 256 workgroups per half batch), which is what the batch waits for.  A conversion ON ITS OWN is the
// other way round: its one chain launch is the critical path and the row kernel has room in its second round -- 1.71 -> 1.645 ms
// (two rounds, same box).  So: on for launches of one frame, off for batches; MLVFS_AMD_AMAZE_ROWS_EXTRA=0 / 1 (or the test hook
// mlvfs_amd_amaze_rows_extra_mode) forces it; bit-identical either way (tests/test_gpu_amaze_rows.py).
int g_amaze_rows_extra_mode = -1;  // -1: the environment decides, 0 / 1: forced
int amaze_rows_extra(int w, int h, int nframes)
{
    static const int env = [] { const char *e = getenv("MLVFS_AMD_AMAZE_ROWS_EXTRA"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
    int nfx, nfy;
    amaze_rows_extent(w, h, &nfx, &nfy);
    const int mode = g_amaze_rows_extra_mode >= 0 ? g_amaze_rows_extra_mode : (env >= 0 ? env : (nframes == 1 ? 1 : 0));
    if (!nfx || !mode) return 0;
    const int step = T - 32;
    const int tiles_x = (w + 16 + step - 1) / step, tiles_y = (h + 16 + step - 1) / step;
    const int cc1_last = w + 16 - (-16 + (tiles_x - 1) * step), rr1_last = h + 16 - (-16 + (tiles_y - 1) * step);
    if (cc1_last != 32 || nfx != tiles_x - 2) return 0;                          // (cc1_last == 32 <=> w % 128 == 0)
    const int incomplete_y = rr1_last >= T ? 0 : (rr1_last < 32 ? 2 : 1), rows_a = tiles_y - incomplete_y;
    if (rr1_last > 32 && rr1_last < 48) return 0;                                // (amaze_launch's `garbage` case walks the last row in one block)
    const int n = incomplete_y ? rows_a - 1 : rows_a;                            // the last row's chain feeds the bottom row's launch
    return n < 0 ? 0 : (n < nfy ? n : nfy);
}

int amaze_rows_launch(const float *d_raw, int w, int h, float *d_red, float *d_green, float *d_blue, hipStream_t s, int nframes,
                      size_t plane_stride, const int *h_of, int h_stride, float *d_dbg, int *d_ctr, const int *d_r2e, int ev_black, int *d_gray)
{
    // MLVFS_AMD_AMAZE_ROWS_PROF=1: cycles per item of each pass and per wave at the barriers (workgroup 0, steady state), printed at exit
    static unsigned long long *d_prof = [] {
        unsigned long long *p = nullptr;
        const char *e = getenv("MLVFS_AMD_AMAZE_ROWS_PROF");
        if (e && atoi(e)) {
            constexpr int N = 2 * NPASS + 5 * 16 + 1;
            if (hipMalloc(&p, 8 * N) != hipSuccess) return (unsigned long long *)nullptr;
            unsigned long long init[N] = {};
            init[N - 1] = atoi(e) > 1;                                     // 2: cycles per item of each pass as well (atomics: slower)
            (void)hipMemcpy(p, init, sizeof init, hipMemcpyHostToDevice);
            static unsigned long long *keep; keep = p;
            atexit([] {
                unsigned long long hst[N];
                if (hipMemcpy(hst, keep, sizeof hst, hipMemcpyDeviceToHost) != hipSuccess) return;
                const char *names[NPASS] = { "LOAD", "DIRDIFF", "NYQTEST", "HVWT", "AREA", "CURV", "CGRAD", "PMSWEEP", "CHROMA", "GRAD", "HREF", "VWALK", "VOTE",
                                             "HVSWEEP", "DIAG", "RBINT", "GFINAL", "OUTPUT" };
                for (int k = 0; k < NPASS; k++)
                    if (hst[2 * k + 1]) fprintf(stderr, "AMAZE_ROWS_PROF %-8s %8.0f cycles per item (%llu items)\n", names[k], (double)hst[2 * k] / hst[2 * k + 1], hst[2 * k + 1]);
                for (int k = 0; k < 16; k++) {
                    const unsigned long long *o = hst + 2 * NPASS + 5 * k;
                    const unsigned long long n = o[4] & 0xFFFFFFFFFFFFFFull;
                    if (n) fprintf(stderr, "AMAZE_ROWS_PROF wave %2d (SIMD %d): phase A busy %6.0f wait %6.0f, phase B busy %6.0f wait %6.0f cycles per step\n", k,
                                   (int)(o[4] >> 56), 2.0 * o[0] / n, 2.0 * o[1] / n, 2.0 * o[2] / n, 2.0 * o[3] / n);
                }
            });
        }
        return p;
    }();
    int nfx, nfy;
    amaze_rows_extent(w, h, &nfx, &nfy);
    if (!nfx) return MLVFS_AMD_OK;
    int dev = 0;
    MLV_HIP(hipGetDevice(&dev));
    static std::mutex mu;
    static int cus[64] = {};
    {
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 64 && !g_tab_ready[dev]) {
            MLV_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_tables), &H_TABLES, sizeof H_TABLES));
            {   // items to waves: longest first, each to the wave with the least work so far
                int cost[NPASS];
                for (int k = 0; k < NPASS; k++) cost[k] = COST[k];
                if (const char *e = getenv("MLVFS_AMD_AMAZE_ROWS_COSTS")) {          // tuning: NPASS comma separated cycle counts
                    int k = 0;
                    for (const char *q = e; *q && k < NPASS; k++) { cost[k] = atoi(q); while (*q && *q != ',') q++; if (*q) q++; }
                }
                unsigned long long desc[2][16];
                for (int ph = 0; ph < 2; ph++) {
                    struct Item { int code, cost; };
                    std::vector<Item> items;
                    for (int pass = ph ? P_GRAD : 0; pass < (ph ? NPASS : P_GRAD); pass++)
                        for (int ck = 0; ck < NCHUNKS[pass]; ck++) items.push_back({ IT(pass, ck), cost[pass] });
                    std::stable_sort(items.begin(), items.end(), [](const Item &x, const Item &y) { return x.cost > y.cost; });
                    int load[16] = {}, cnt[16] = {}, simd[4] = {};                     // waves w, w + 4, w + 8, w + 12 share a SIMD: its sum counts too
                    for (int w2 = 0; w2 < 16; w2++) desc[ph][w2] = ~0ull;
                    for (const Item &it : items) {
                        int best = -1;
                        for (int w2 = 0; w2 < 16; w2++)
                            if (cnt[w2] < 4 && (best < 0 || 2 * load[w2] + simd[w2 & 3] < 2 * load[best] + simd[best & 3])) best = w2;
                        simd[best & 3] += it.cost;
                        const unsigned long long v = (unsigned)it.code | (unsigned)LAG[it.code >> 3] << 8;
                        desc[ph][best] = (desc[ph][best] & ~(0xFFFFull << (16 * cnt[best]))) | v << (16 * cnt[best]);
                        cnt[best]++; load[best] += it.cost;
                    }
                }
                if (!getenv("MLVFS_AMD_AMAZE_ROWS_COSTS"))
                    for (int ph = 0; ph < 2; ph++)
                        for (int w2 = 0; w2 < 16; w2++) desc[ph][w2] = TUNED[ph][w2];
                if (const char *e = getenv("MLVFS_AMD_AMAZE_ROWS_ASSIGN")) {         // tuning: the 32 descriptors themselves, hex, comma separated
                    int k = 0;
                    for (const char *q = e; *q && k < 32; k++) { desc[k / 16][k % 16] = strtoull(q, nullptr, 16); while (*q && *q != ',') q++; if (*q) q++; }
                }
                if (getenv("MLVFS_AMD_AMAZE_ROWS_SHOW")) {
                    for (int ph = 0; ph < 2; ph++)
                        for (int w2 = 0; w2 < 16; w2++) fprintf(stderr, "%016llx%s", desc[ph][w2], ph == 1 && w2 == 15 ? "\n" : ",");
                }
                MLV_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_desc), desc, sizeof desc));
            }
            hipDeviceProp_t pr;
            MLV_HIP(hipGetDeviceProperties(&pr, dev));
            cus[dev] = pr.multiProcessorCount;
            g_tab_ready[dev] = true;
        }
    }
    static const unsigned skip = [] { const char *e = getenv("MLVFS_AMD_AMAZE_ROWS_SKIP"); return e ? (unsigned)strtoul(e, nullptr, 16) : 0u; }();
    const int nrect = nfx * nfy, ntiles = nrect + (d_dbg ? 0 : amaze_rows_extra(w, h, nframes));       // (the debug dump is laid out for the block only)
    // one workgroup per CU (the rings fill its LDS); the frames of a batch share the CUs
    int per_frame = (dev < 64 && cus[dev] ? cus[dev] : 256) / (nframes > 0 ? nframes : 1);
    static const int cap = [] { const char *e = getenv("MLVFS_AMD_AMAZE_ROWS_WGS"); return e ? atoi(e) : 0; }();    // tests: few workgroups, many tiles each
    if (cap > 0 && per_frame > cap) per_frame = cap;
    per_frame = per_frame < 1 ? 1 : (per_frame > ntiles ? ntiles : per_frame);
    if (d_dbg)
        hipLaunchKernelGGL(k_amaze_rows<true>, dim3(per_frame, nframes), dim3(1024), 0, s, d_raw, w, h, d_red, d_green, d_blue, nfx,
                           ntiles, nrect, plane_stride, h_of, h_stride, d_dbg, d_prof, skip, d_ctr, d_r2e, ev_black, d_gray);
    else
        hipLaunchKernelGGL(k_amaze_rows<false>, dim3(per_frame, nframes), dim3(1024), 0, s, d_raw, w, h, d_red, d_green, d_blue, nfx,
                           ntiles, nrect, plane_stride, h_of, h_stride, d_dbg, d_prof, skip, d_ctr, d_r2e, ev_black, d_gray);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

}  // namespace mlv
