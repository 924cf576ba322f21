// k_frame_p.hip -- the packed-once kernel of the fused pass (round 5); launched by k_frame.hip's launch_frame_t before the
// list-mode k_frame.  Its own translation unit: its own scheduler flags (Makefile), and the two kernels compile in parallel.
#include "k_frame_dev.h"

namespace mlv {

#ifndef KF_P_WAIT_MIN
#define KF_P_WAIT_MIN 1
#endif
#ifndef KF_P_WAIT_MAX
#define KF_P_WAIT_MAX 15
#endif
// ================================================================ packed-once kernel (round 5)
// k_frame_p: the same pass with the colour differences packed ONCE, by the loader.
//
// k_frame keeps (dr, db) of a cell as two int32 planes and every median lane packs what it reads -- each cell five times (the
// five lanes whose windows hold it), relative to a reference of the lane's own, and what a lane takes from its neighbour is rebased
// by the difference of their references: 181 of the 417 vector instructions of a strip were packing, rebasing, hand-over and
// certainty tests.  Here a TILE has one reference (ref_r, ref_b) and the loader stores
//     pk = { sat16(dr - ref_r), sat16(db - ref_b) }                                      one dword per cell, one plane
// so a median lane reads half the LDS, packs nothing and rebases nothing.  Exactness is the rule k_frame derives above ChainGroup
// with D = 0: both saturations are monotone, so the packed median is sat16(true median - ref), which IS the true median - ref
// whenever it comes out strictly inside (-32768, 32767); a strip where it does not is "uncertain".
//
// This kernel never settles an uncertain strip itself (that takes the int32 planes it no longer has): a tile with one is pushed on
// the launch's work list and done again, whole, by k_frame in list mode -- the launch that follows on the same stream.  Tiles after
// such a tile go to the list unseen (1, 3, 7, 15 tiles, doubling while the tile after them is no better: footage made of hard colour
// edges costs this kernel a sixteenth of its tiles), a stray one costs itself and nothing else.
//
// The reference.  Any reference gives exact medians; it only decides how many strips are certain.  A run's first tile (and every
// tile that does not continue the one above) takes the median of three cells at the tile's centre, fetched and converted by every
// wave for itself (16 lanes, one pixel each; scalar arithmetic from there: no exchange, no barrier); the tiles below it keep that
// reference, so the four plane rows they inherit stay valid as they are.
template <bool SPREAD_, bool CHAIN_>
struct __align__(16) SmemP {
    static constexpr bool SPREAD = SPREAD_;
    uint16_t raw[2 * RH][2 * TCW];      // interior pixels (post patch) + the four pixel rows below (the next tile's first), 8.5 KiB
    uint32_t pk[PH][PW];                // 5 KiB
    int ge[RH][TCW];                    // 4.25 KiB
    uint16_t t16[MLV_T16_N + (SPREAD_ ? 64 : 0)];
    uint32_t has_patch[PMAP_WORDS];
    uint32_t xchg[CHAIN_ ? 3 : 1][XCHG_WORDS];
    int unc[2];                         // some strip of this tile / of the tile before is uncertain (by tile parity)
    int low[2];                         // some pixel this tile / the tile before loaded lies at most 64 above black
    int dim[2];                         // ... lies less than 256 above black (or the tile took the loader's slow form)
    int next_tile, next_end;
    int walk[5];                        // thread 0's: first tile of the group's range, tiles that go out in runs, tiles per run, group, runs all out
    int2 carry_at[256];                 // per thread: byte offsets {from, to} of the 16-byte piece it hands down to the tile below (2 KiB)
    uint4 carry_spare;                  // (what threads without a piece copy)
#ifdef KFP_EXP_LDSPAD
    char exp_pad[KFP_EXP_LDSPAD];       // (occupancy experiment: fewer workgroups per CU)
#endif
};

struct PGroup {
    mlv_pk16 s[4][5];        // sorted columns
    mlv_pk16 p0[10], p1[10]; // columns 0+1 and 2+3 merged
    mlv_pk16 q[6];           // ranks 8..13 of the 20
};
struct PNext { mlv_pk16 s0[5], s2[5], p0[10], q[6]; };

__device__ __forceinline__ mlv_pk16 as_pk(uint32_t v) { return __builtin_bit_cast(mlv_pk16, v); }
__device__ __forceinline__ uint32_t as_u(mlv_pk16 v) { return __builtin_bit_cast(uint32_t, v); }

__device__ __forceinline__ void pchain_group(const uint32_t (*pk)[PW], int row_top, int col_left, PGroup &g)
{
    mlv_pk16 col[4][5];
#pragma unroll
    for (int r = 0; r < 5; r++) {
        const uint4 v = *(const uint4 *)&pk[row_top + r][col_left];
        col[0][r] = as_pk(v.x); col[1][r] = as_pk(v.y); col[2][r] = as_pk(v.z); col[3][r] = as_pk(v.w);
    }
#pragma unroll
    for (int c = 0; c < 4; c++) mlv_sort5(col[c], g.s[c]);
    mlv_merge55(g.s[0], g.s[1], g.p0);
    mlv_merge55(g.s[2], g.s[3], g.p1);
}
__device__ __forceinline__ void pchain_fetch_lists(const PGroup &g, PNext &n)
{
#pragma unroll
    for (int i = 0; i < 5; i++) { n.s0[i] = dpp_next(g.s[0][i]); n.s2[i] = dpp_next(g.s[2][i]); }
#pragma unroll
    for (int i = 0; i < 10; i++) n.p0[i] = dpp_next(g.p0[i]);
}
__device__ __forceinline__ void pchain_fetch_window(const PGroup &g, PNext &n)
{
#pragma unroll
    for (int i = 0; i < 6; i++) n.q[i] = dpp_next(g.q[i]);
}
// the halo group's lane (and the first lane of waves 1..3) -> LDS -> the last lane of the wave before
__device__ __forceinline__ void pchain_publish(const PGroup &g, uint32_t *x)
{
    uint4 *o = (uint4 *)x;
    o[0] = make_uint4(as_u(g.s[0][0]), as_u(g.s[0][1]), as_u(g.s[0][2]), as_u(g.s[0][3]));
    o[1] = make_uint4(as_u(g.s[0][4]), as_u(g.s[2][0]), as_u(g.s[2][1]), as_u(g.s[2][2]));
    o[2] = make_uint4(as_u(g.s[2][3]), as_u(g.s[2][4]), as_u(g.p0[0]), as_u(g.p0[1]));
    o[3] = make_uint4(as_u(g.p0[2]), as_u(g.p0[3]), as_u(g.p0[4]), as_u(g.p0[5]));
    o[4] = make_uint4(as_u(g.p0[6]), as_u(g.p0[7]), as_u(g.p0[8]), as_u(g.p0[9]));
    o[5] = make_uint4(as_u(g.q[0]), as_u(g.q[1]), as_u(g.q[2]), as_u(g.q[3]));
    *(uint2 *)&o[6] = make_uint2(as_u(g.q[4]), as_u(g.q[5]));
}
__device__ __forceinline__ void pchain_collect_lists(const uint32_t *x, PNext &n)
{
    const uint4 *o = (const uint4 *)x;
    const uint4 a0 = o[0], a1 = o[1], a2 = o[2], a3 = o[3], a4 = o[4];
    n.s0[0] = as_pk(a0.x); n.s0[1] = as_pk(a0.y); n.s0[2] = as_pk(a0.z); n.s0[3] = as_pk(a0.w); n.s0[4] = as_pk(a1.x);
    n.s2[0] = as_pk(a1.y); n.s2[1] = as_pk(a1.z); n.s2[2] = as_pk(a1.w); n.s2[3] = as_pk(a2.x); n.s2[4] = as_pk(a2.y);
    n.p0[0] = as_pk(a2.z); n.p0[1] = as_pk(a2.w); n.p0[2] = as_pk(a3.x); n.p0[3] = as_pk(a3.y); n.p0[4] = as_pk(a3.z);
    n.p0[5] = as_pk(a3.w); n.p0[6] = as_pk(a4.x); n.p0[7] = as_pk(a4.y); n.p0[8] = as_pk(a4.z); n.p0[9] = as_pk(a4.w);
}
__device__ __forceinline__ void pchain_collect_window(const uint32_t *x, PNext &n)
{
    const uint4 a5 = *(const uint4 *)(x + 20);
    const uint2 a6 = *(const uint2 *)(x + 24);
    n.q[0] = as_pk(a5.x); n.q[1] = as_pk(a5.y); n.q[2] = as_pk(a5.z); n.q[3] = as_pk(a5.w); n.q[4] = as_pk(a6.x); n.q[5] = as_pk(a6.y);
}
// medians of the strip's four cells (relative to the tile's reference) from its own group and the neighbour's
__device__ __forceinline__ void pchain_finish(const PGroup &g, const PNext &n, mlv_pk16 (&o)[STRIP])
{
    mlv_pk16 q1[6], t[1];
    mlv_quad_mid6(g.p1, n.p0, q1);
    mlv_final6of11(g.q, n.s0, t);  o[0] = t[0];      // window columns 0..3 | 4
    mlv_final6of11(q1, g.s[1], t); o[1] = t[0];      // 2..5 | 1
    mlv_final6of11(q1, n.s2, t);   o[2] = t[0];      // 2..5 | 6
    mlv_final6of11(n.q, g.s[3], t); o[3] = t[0];     // 4..7 | 3
}
// plus-shaped 5 (chroma_smooth.c:44-47 with CHROMA_SMOOTH_2X2) and 3x3 on the packed plane: the strip's cells are plane columns
// col0 + 2 .. col0 + 5 (col0 = 4 k: the halo is two cells), their rows jrow + 1 .. jrow + 3
__device__ __forceinline__ void pstrip_median5(const uint32_t (*pk)[PW], int jrow, int col0, mlv_pk16 (&o)[STRIP])
{
    const uint2 u0 = *(const uint2 *)&pk[jrow + 1][col0 + 2], u1 = *(const uint2 *)&pk[jrow + 1][col0 + 4];
    const uint4 c0 = *(const uint4 *)&pk[jrow + 2][col0], c1 = *(const uint4 *)&pk[jrow + 2][col0 + 4];
    const uint2 d0 = *(const uint2 *)&pk[jrow + 3][col0 + 2], d1 = *(const uint2 *)&pk[jrow + 3][col0 + 4];
    const uint32_t up[4] = { u0.x, u0.y, u1.x, u1.y }, dn[4] = { d0.x, d0.y, d1.x, d1.y };
    const uint32_t ce[8] = { c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w };
#pragma unroll
    for (int c = 0; c < STRIP; c++) {
        const mlv_pk16 v[5] = { as_pk(up[c]), as_pk(ce[c + 1]), as_pk(ce[c + 2]), as_pk(ce[c + 3]), as_pk(dn[c]) };
        mlv_pk16 t[1];
        mlv_median5(v, t);
        o[c] = t[0];
    }
}
__device__ __forceinline__ mlv_pk16 pk_med3(mlv_pk16 a, mlv_pk16 b, mlv_pk16 c) { return mlv_mx(mlv_mn(a, b), mlv_mn(mlv_mx(a, b), c)); }
__device__ __forceinline__ void pstrip_median9(const uint32_t (*pk)[PW], int jrow, int col0, mlv_pk16 (&o)[STRIP])
{
    mlv_pk16 lo[STRIP + 2], mi[STRIP + 2], hi[STRIP + 2];
    uint32_t v[3][8];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const uint4 a = *(const uint4 *)&pk[jrow + 1 + r][col0], b = *(const uint4 *)&pk[jrow + 1 + r][col0 + 4];
        v[r][0] = a.x; v[r][1] = a.y; v[r][2] = a.z; v[r][3] = a.w; v[r][4] = b.x; v[r][5] = b.y; v[r][6] = b.z; v[r][7] = b.w;
    }
#pragma unroll
    for (int c = 0; c < STRIP + 2; c++) {
        const mlv_pk16 a = as_pk(v[0][c + 1]), b = as_pk(v[1][c + 1]), d = as_pk(v[2][c + 1]);
        const mlv_pk16 mn = mlv_mn(a, b), mx = mlv_mx(a, b);
        lo[c] = mlv_mn(mn, d);
        hi[c] = mlv_mx(mx, d);
        mi[c] = mlv_mx(mn, mlv_mn(mx, d));
    }
#pragma unroll
    for (int c = 0; c < STRIP; c++)
        o[c] = pk_med3(mlv_mx(mlv_mx(lo[c], lo[c + 1]), lo[c + 2]), pk_med3(mi[c], mi[c + 1], mi[c + 2]),
                       mlv_mn(mlv_mn(hi[c], hi[c + 1]), hi[c + 2]));
}
// certain: every half of every median strictly inside (-32768, 32767).  t = v + 32769 (wraps) is 1 for -32768, 0 for 32767 and at
// least 2 otherwise: the smallest t of the strip's eight halves tells (four adds, three minima, one saturating subtract)
__device__ __forceinline__ bool pk_uncertain(const mlv_pk16 (&o)[STRIP])
{
    typedef unsigned short upk16 __attribute__((ext_vector_type(2)));
    const upk16 off = { 32769, 32769 }, two = { 2, 2 };
    upk16 m = __builtin_bit_cast(upk16, o[0]) + off;
#pragma unroll
    for (int c = 1; c < STRIP; c++) m = __builtin_elementwise_min(m, (upk16)(__builtin_bit_cast(upk16, o[c]) + off));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(m, two)) != 0x00020002u;
}

// EV triples of NC cells on the loader's common path with the colour differences packed against the tile's reference: k_frame_dev.h's
// cell_multi_ev_fast with the end rearranged -- R's (and B's) exponent part, table value and -(green + reference) meet in ONE
// three-input add (the biased EV of R is never formed): 9 operations per cell behind the look-ups where ev + cells + pack took 11
__device__ __forceinline__ uint32_t add3_u32(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <int NC, bool SPREAD>
__device__ __forceinline__ void cell_multi_pk_fast(const uint32_t *p0, const uint32_t *p1, int black, const uint16_t *t, int ref_r, int ref_b,
                                                   int (&ge)[NC], uint32_t (&pk)[NC])
{
    uint32_t fb[4 * NC], tv[4 * NC];
    const float fmagic = 8388608.0f + (float)black;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const uint32_t px[4] = { p0[2 * c], p0[2 * c + 1], p1[2 * c], p1[2 * c + 1] };
#pragma unroll
        for (int i = 0; i < 4; i++) fb[4 * c + i] = __float_as_uint(__uint_as_float(px[i] | 0x4B000000u) - fmagic);
    }
#pragma unroll
    for (int i = 0; i < 4 * NC; i++) tv[i] = *(const uint16_t *)((const char *)t + t16_offset<SPREAD>(fb[i]));
    // (exponent parts: R and B each, the two greens as ONE shift of their sum -- the low ten bits of a pixel's float are zero)
    uint32_t exr[NC], exb[NC], exg[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) { exr[c] = fb[4 * c] >> 8; exb[c] = fb[4 * c + 3] >> 8; exg[c] = (fb[4 * c + 1] + fb[4 * c + 2]) >> 8; }
#pragma unroll
    for (int c = 0; c < NC; c += 2) {                    // opaque uses: the reads stay unconditional and back to back
        asm volatile("" :: "v"(exr[c]), "v"(exb[c]), "v"(exg[c]), "v"(exr[c + 1]), "v"(exb[c + 1]), "v"(exg[c + 1]));
    }
#pragma unroll
    for (int i = 0; i < 4 * NC; i += 8) {
        asm volatile("" :: "v"(tv[i]), "v"(tv[i + 1]), "v"(tv[i + 2]), "v"(tv[i + 3]), "v"(tv[i + 4]), "v"(tv[i + 5]), "v"(tv[i + 6]), "v"(tv[i + 7]));
    }
    uint32_t nref_r = 0u - (uint32_t)ref_r, nref_b = 0u - (uint32_t)ref_b;
    asm("" : "+v"(nref_r), "+v"(nref_b));       // (opaque: else -(ref + green) is formed, two instructions where one subtract does)
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const uint32_t gb = add3_u32(exg[c], tv[4 * c + 1], tv[4 * c + 2]) >> 1;        // biased EV of the cell's green (both EVs >= 0: trunc == floor)
        ge[c] = (int)(gb - (127u << 15));
        uint32_t ngr = nref_r - gb, ngb = nref_b - gb;
        // |dr|, |db|, |ref| < 2^20 here: the plain differences do not wrap.  (The three-input adds are written out: left to itself the
        // compiler forms every pixel's EV and (reference + green) first -- six more instructions per cell)
        pk[c] = as_u(__builtin_amdgcn_cvt_pk_i16((int)add3_u32(exr[c], tv[4 * c + 0], ngr), (int)add3_u32(exb[c], tv[4 * c + 3], ngb)));
    }
}

template <int METHOD, class SM>
__device__ __forceinline__ void emit_item_p(SM &sm, int black, bool dark, bool slow, int ref_r, int ref_b, int p, int lk, bool edge,
                                            const uint32_t (&p0)[8], const uint32_t (&p1)[8])
{
    const int jj = p - HC;
    const bool keep = !edge && jj >= 0;
    int ge[4];
    uint32_t pk[4];
    if (!dark) cell_multi_pk_fast<4, SM::SPREAD>(p0, p1, black, sm.t16, ref_r, ref_b, ge, pk);
    else {
        int dr[4], db[4];
#pragma unroll
        for (int c = 0; c < 4; c += 2) {
            int g2[2], r2[2], b2[2];
            if (!slow) cell_multi_ev_dark<2, SM::SPREAD>(p0 + 2 * c, p1 + 2 * c, black, sm.t16, g2, r2, b2);
            else cell_pair_ev<SM::SPREAD>(p0 + 2 * c, p1 + 2 * c, black, sm.t16, true, g2, r2, b2);
            ge[c] = g2[0]; ge[c + 1] = g2[1]; dr[c] = r2[0]; dr[c + 1] = r2[1]; db[c] = b2[0]; db[c + 1] = b2[1];
        }
#pragma unroll
        for (int c = 0; c < 4; c++)    // (differences of wrapped values: saturating)
            pk[c] = as_u(__builtin_amdgcn_cvt_pk_i16(__builtin_elementwise_sub_sat(dr[c], ref_r), __builtin_elementwise_sub_sat(db[c], ref_b)));
    }
    const int ca = edge ? PW - HC : HC + 4 * lk, cb = edge ? 0 : ca + 2;
    const uint32_t prow = __umul24((uint32_t)p, (uint32_t)(PW * 4));
    char *ppk = (char *)&sm.pk[0][0] + prow;
    *(uint2 *)(ppk + 4 * ca) = make_uint2(pk[0], pk[1]);
    *(uint2 *)(ppk + 4 * cb) = make_uint2(pk[2], pk[3]);
    if (keep) {
        *(int4 *)&sm.ge[jj][4 * lk] = make_int4(ge[0], ge[1], ge[2], ge[3]);
        *(uint4 *)&sm.raw[2 * jj][8 * lk] = make_uint4(p0[0] | (p0[1] << 16), p0[2] | (p0[3] << 16), p0[4] | (p0[5] << 16), p0[6] | (p0[7] << 16));
        *(uint4 *)&sm.raw[2 * jj + 1][8 * lk] = make_uint4(p1[0] | (p1[1] << 16), p1[2] | (p1[3] << 16), p1[4] | (p1[5] << 16), p1[6] | (p1[7] << 16));
    }
}

template <int METHOD, bool PACKED, int VEC, bool SPREAD>
__global__ __launch_bounds__(256, 4) void k_frame_p(const FrameArgs a)
{
    static_assert(METHOD == 2 || METHOD == 3 || METHOD == 5, "chroma smoothing only");
    static_assert(VEC != 0, "rows of whole 8-pixel groups");
    constexpr bool CHAIN = METHOD == 5;
    using Smem = SmemP<SPREAD, CHAIN>;
    __shared__ Smem sm;
    constexpr int BPP = bpp_of(PACKED, VEC);
    constexpr int NEW0 = 2 * HC;
    const int tid = threadIdx.x;

    load_t16_rel<SPREAD>(sm.t16, cold_args()->t16, tid);
    {
        // rows handed down to the tile below: threads 0..67 the packed rows TCH.. -> 0..3, 68..131 the pixel rows, 132..163 the green EVs
        constexpr int C_PL = 2 * HC * PW * 4 / 16, C_RAW = 2 * HC * 2 * TCW * 2 / 16, C_GE = HC * TCW * 4 / 16, C_ALL = C_PL + C_RAW + C_GE;
        static_assert(C_ALL <= 256, "one 16-byte piece per thread");
        int to = (int)offsetof(Smem, carry_spare), delta = 0;
        if (tid < C_PL) { to = (int)offsetof(Smem, pk) + 16 * tid; delta = TCH * PW * 4; }
        else if (tid < C_PL + C_RAW) { to = (int)offsetof(Smem, raw) + 16 * (tid - C_PL); delta = 2 * TCH * 2 * TCW * 2; }
        else if (tid < C_ALL) { to = (int)offsetof(Smem, ge) + 16 * (tid - C_PL - C_RAW); delta = TCH * TCW * 4; }
        sm.carry_at[tid] = make_int2(to + delta, to);
    }
    const int tiles_per_frame = a.tiles_x * a.tiles_y;
    const bool pmap_ok = tiles_per_frame <= PMAP_WORDS * 32;
    int band_end;
    {
        KArgs ka = cold_args();
        if (ka->patch && pmap_ok) {
            for (int i = tid; i < PMAP_WORDS; i += 256) sm.has_patch[i] = 0;
            __syncthreads();
            const int *toff = ka->tile_off;
            for (int i = tid; i < tiles_per_frame; i += 256)
                if (toff[i + 1] != toff[i]) atomicOr(&sm.has_patch[i >> 5], 1u << (i & 31));
        }
        // the tile walk of k_frame (groups, runs, singles); what only thread 0 needs, when it draws, waits in LDS
        const int total = tiles_per_frame * ka->nframes;
        const int groups = ka->groups, nx = 8;
        const int grp = blockIdx.x % groups;
        const int gpx = (groups + nx - 1) / nx;
        const int grank = (groups % nx == 0) ? (grp % nx) * gpx + grp / nx : grp;
        const int gq = total / groups, grem = total - gq * groups;
        const int band_start = grank * gq + min(grank, grem);
        band_end = band_start + gq + (grank < grem ? 1 : 0);
        const int run = max(ka->run, 1);
        const int runs_len = max(band_end - band_start - ka->singles, 0) / run * run;
        if (tid == 0) { sm.walk[0] = band_start; sm.walk[1] = runs_len; sm.walk[2] = run; sm.walk[3] = grp; sm.walk[4] = 0; }
    }
    // thread 0: the next run of the group's range, or its next single tile
    auto draw = [&](int &nt, int &ne) {
        KArgs ka = cold_args();
        int *tickets = ka->tickets;
        const int band_start = sm.walk[0], runs_len = sm.walk[1], run = sm.walk[2], grp = sm.walk[3];
        if (!sm.walk[4]) {
            const int p = atomicAdd(&tickets[2 * grp], run);
            if (p < runs_len) { nt = band_start + p; ne = nt + run; return; }
            sm.walk[4] = 1;                                 // the runs of this group's range are all handed out
        }
        const int q = atomicAdd(&tickets[2 * grp + 1], 1);
        nt = min(band_start + runs_len + q, band_end);
        ne = nt + 1;
    };
    if (tid == 0) {
        int nt, ne;
        draw(nt, ne);
        sm.next_tile = nt; sm.next_end = ne;
        sm.unc[0] = 0; sm.unc[1] = 0; sm.low[0] = 0; sm.low[1] = 0; sm.dim[0] = 0; sm.dim[1] = 0;
    }
    __syncthreads();
    int t = __builtin_amdgcn_readfirstlane(sm.next_tile), t_end = __builtin_amdgcn_readfirstlane(sm.next_end);
    struct Pos { int f, tcol, trow; };
    auto pos_of = [&](int tt) {
        Pos p;
        p.f = tt / tiles_per_frame;
        const int r = tt - p.f * tiles_per_frame;
        p.tcol = r / a.tiles_y;
        p.trow = r - p.tcol * a.tiles_y;
        return p;
    };
    auto pos_below = [&](Pos p) {
        if (++p.trow == a.tiles_y) {
            p.trow = 0;
            if (++p.tcol == a.tiles_x) { p.tcol = 0; p.f++; }
        }
        return p;
    };
    uint32_t r0[4], r1[4];                               // prefetch registers of this thread's item
    // of the lane's item the prefetch of an interior tile needs two byte offsets: kept in registers; the item's full description is
    // derived from the lane's number where the loader (and the prefetch of a tile at the frame's border) needs it
    uint32_t pre_a, pre_b;
    {
        const bool e0 = tid >= N_MAIN;
        item_tile_offsets<BPP>(item_lane<PACKED, VEC>(tid & 15, e0), e0 ? tid - N_MAIN : tid >> 4, a.w, pre_a, pre_b);
    }
    // loader: threads 0..239 own the main item (row t / 16, group t % 16) of the tile's new rows, threads 240..254 the edge items;
    // what a lane's item is, is derived from its number where it is needed (a handful of operations per tile: kept in registers
    // through the median phase the item descriptions were what the allocator spilt)
    struct Src { const uint8_t *p; size_t stride; unsigned bytes; };
    auto src_of = [](KArgs ka) { Src s; s.p = ka->src; s.stride = ka->src_stride; s.bytes = ka->src_bytes; return s; };
    auto issue_tile = [&](const Src &sa, const Pos &p, int tl) {
        issue_tile_rows_pre<BPP, VEC>(r0, r1, sa.p + (size_t)p.f * sa.stride, sa.bytes, pre_a, pre_b,
                                      [&](int &row) { const bool l_edge = tl >= N_MAIN; row = l_edge ? tl - N_MAIN : tl >> 4; return item_lane<PACKED, VEC>(tl & 15, l_edge); },
                                      a.w, a.h, p.tcol * 2 * TCW, p.trow * 2 * TCH, NEW0);
    };
    // The reference of a tile that does not continue the one above: pixel (lane & 7, lane >> 3 & 1) of the 8 x 2 block at the tile's
    // centre (inside the frame, on even coordinates: R G1 / G2 B), one per lane; the loads go out with the tile's prefetch
    uint32_t smp = 0, smp_sh = 0;          // (the load only: what depends on it waits until the tile starts)
    // pixel (lane & 15, lane >> 4) of the 16 x 4 block at the tile's centre (inside the frame, on even coordinates: R G1 / G2 B)
    auto issue_sample = [&](const Src &sa, const Pos &p, int tl) {
        const uint8_t *frame = sa.p + (size_t)p.f * sa.stride;
        const int x0 = min(p.tcol * 2 * TCW + TCW, (a.w - 16) & ~1), y0 = min(p.trow * 2 * TCH + TCH - 1, (a.h - 4) & ~1);
        const uint32_t i = (uint32_t)(y0 + ((tl >> 4) & 3)) * (uint32_t)a.w + (uint32_t)(x0 + (tl & 15));
        if (BPP != 16) {
            const uint32_t bit = i * (uint32_t)BPP;          // < 2^28 pixels per frame (launcher): fits
            __builtin_memcpy(&smp, frame + 2 * (bit >> 4), 4);          // two 16-bit words of the stream (an address that is a multiple of 2)
            smp_sh = bit & 15u;
        } else smp = ((const uint16_t *)frame)[i];
    };
    // The reference: the median of five cells of the block, no two of them neighbours -- the pixel maps of real clips hold pairs of
    // defects two pixels apart (and the benchmark's frames do: a twin next to its defect made two of THREE sampled cells outliers, the
    // reference garbage and the whole tile uncertain)
    int ref_r = 0, ref_b = 0;
    auto take_sample = [&]() {
        uint32_t px = smp;
        if (BPP != 16) px = (__builtin_amdgcn_alignbit(smp, smp, 16) >> (32 - BPP - smp_sh)) & ((1u << BPP) - 1u);          // MSB-first: first word high
        const int l = min(max((int)px - a.black, 1), 16383);
        const int ix = ev_index(l);
        const int ev = ev_value(l, (int)sm.t16[SPREAD ? ix + (ix >> 7) : ix]);
        constexpr int CX[5] = { 0, 3, 6, 1, 5 }, CY[5] = { 0, 0, 0, 1, 1 };       // cell (cx, cy): R in lane 2 cx + 32 cy, G1 + 1, G2 + 16, B + 17
        int dr[5], db[5];
#pragma unroll
        for (int c = 0; c < 5; c++) {
            constexpr int dummy = 0; (void)dummy;
            const int l0 = 2 * CX[c] + 32 * CY[c];
            const int r = __builtin_amdgcn_readlane(ev, l0), g1 = __builtin_amdgcn_readlane(ev, l0 + 1);
            const int g2 = __builtin_amdgcn_readlane(ev, l0 + 16), b = __builtin_amdgcn_readlane(ev, l0 + 17);
            const int ge = (g1 + g2) >> 1;
            dr[c] = r - ge; db[c] = b - ge;
        }
        auto med5 = [](const int (&v)[5]) { return med3i(v[4], max(min(v[0], v[1]), min(v[2], v[3])), min(max(v[0], v[1]), max(v[2], v[3]))); };
        ref_r = med5(dr);
        ref_b = med5(db);
    };
    Pos cur = pos_of(min(t, max(band_end - 1, 0)));
    {
        const Src sa = src_of(cold_args());
        issue_tile(sa, cur, tid);
        issue_sample(sa, cur, tid);
    }
    __syncthreads();                           // T16 copy complete

    int fb_skip = 0, fb_wait = KF_P_WAIT_MIN;  // tiles still to go to the list unseen; how many after the next uncertain tile
    bool cont = false;
    int par = 0;                               // tile parity: which of the two `unc` slots this tile uses
    int low_prev = 3;                          // (the tile before this one: conservative until there is one)
    bool have_smp = true;                      // the sample of this tile is on its way (else: fetched when the tile starts)
    auto push = [&](int first, int n) {        // thread 0
        KArgs ka = cold_args();
        int *ctl = ka->wl_ctl;
        const int i = atomicAdd(&ctl[0], 1);
        ka->wl[i] = make_int2(first, n);
        atomicAdd(&ctl[3], n);                 // (statistics: tiles listed since the stream's state was created)
    };
    while (t < band_end) {
        if (fb_skip > 0) {
            // ---- tiles that go to the list unseen: the rest of this run, or as many of it as are still to be skipped
            const int n = min(fb_skip, t_end - t);
            fb_skip -= n;
            if (tid == 0) {
                push(t, n);
                int nt, ne;
                if (t + n < t_end) { nt = t + n; ne = t_end; }
                else draw(nt, ne);
                sm.next_tile = nt; sm.next_end = ne;
            }
            lds_barrier();
            t = __builtin_amdgcn_readfirstlane(sm.next_tile); t_end = __builtin_amdgcn_readfirstlane(sm.next_end);
            lds_barrier();                     // (all have read: thread 0 may write the slots again)
            cont = false;
            if (t < band_end) {
                cur = pos_of(t);
                const Src sa = src_of(cold_args());
                issue_tile(sa, cur, tid);
                issue_sample(sa, cur, tid);
                have_smp = true;
            }
            continue;
        }
        int nt = 0, ne = 0;
        if (tid == 0) {                        // the tile after this one: known, or drawn now and back long before it is needed
            if (t + 1 < t_end) { nt = t + 1; ne = t_end; }
            else draw(nt, ne);
        }
        const int trow = cur.trow, tx0 = cur.tcol * 2 * TCW, ty0 = cur.trow * 2 * TCH;
        const int tr = cur.trow * a.tiles_x + cur.tcol;
        const Src sa = src_of(cold_args());      // (asked for here, needed when the next tile is prefetched: no wait there)
        // ---- pixel-map entries of this tile (few tiles have any)
        bool tile_patched = false;
        int pbeg = 0, pend = 0;
        int4 my_rec = make_int4(-1, 0, 0, 0);
        if (a.patch) {
            tile_patched = !pmap_ok || ((sm.has_patch[tr >> 5] >> (tr & 31)) & 1u);
            if (tile_patched) {
                KArgs ka = cold_args();
                const int *toff = ka->tile_off;
                pbeg = toff[tr];
                pend = toff[tr + 1];
                if (pbeg + tid < pend) my_rec = (ka->cells + (size_t)cur.f * ka->n_rec)[pbeg + tid];
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (!cont) {
            if (!have_smp) issue_sample(sa, cur, tid);
            take_sample();
        }
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
        auto do_item = [&](const ItemLane &L, const uint32_t (&d0)[4], const uint32_t (&d1)[4], int p, int lk) {
            uint32_t p0[8], p1[8];
            unpack8<BPP>(d0, L.s0, L.s1, L.s23, p0);
            unpack8<BPP>(d1, L.s0 ^ L.flip, L.s1 ^ L.flip, L.s23 ^ L.flip, p1);
            uint32_t lo = min(p0[0], p1[0]), hi = max(p0[0], p1[0]);
#pragma unroll
            for (int i = 1; i < 8; i++) {
                lo = min(min(lo, p0[i]), p1[i]);
                if (!PACKED) hi = max(max(hi, p0[i]), p1[i]);
            }
            const bool odd = (int)lo <= a.black;
            const bool beyond = (!PACKED && (int)hi - a.black > 16383) || (PACKED && a.black < 0);
            const bool slow = (!PACKED || a.black < 0) && __any(beyond);
            const bool dark = slow || __any(odd);
            if (__any((int)lo <= a.black + 255) || slow) {             // (strip_output: what a tile of pixels >= 256 above black may skip)
                sm.dim[par] = 1;
                if (__any((int)lo <= a.black + 64)) sm.low[par] = 1;   // the stripes epilogue's "more than 64 above black" holds for no mask then
            }
            emit_item_p<METHOD, Smem>(sm, a.black, dark, slow, ref_r, ref_b, p, lk, L.edge, p0, p1);
        };
        if (!cont) {
            // the first tile of a run (or of a column): the four plane rows above the tile's own, straight from memory -- threads
            // 0..63 the main items of rows 0..3, 64..67 their edge items -- while the other waves convert the prefetched rows
            if (tid_o < N_TOP) {
                const bool te = tid_o >= N_TOP_MAIN;
                const int trw = te ? tid_o - N_TOP_MAIN : tid_o >> 4;
                const ItemLane TL = item_lane<PACKED, VEC>(tid_o & 15, te);
                uint32_t q0[4] = { 0, 0, 0, 0 }, q1[4] = { 0, 0, 0, 0 };
                issue_item<BPP, true>(q0, q1, frame_rsrc(sa.p + (size_t)cur.f * sa.stride, sa.bytes), TL, a.w, a.h, tx0, ty0, trw);
                do_item(TL, q0, q1, trw, tid_o & 15);
            }
        }
#ifndef KFP_EXP_NOLOAD
        if (tid_o < N_ITEMS) {
            const bool l_edge = tid_o >= N_MAIN;
            do_item(item_lane<PACKED, VEC>(tid_o & 15, l_edge), r0, r1, NEW0 + (l_edge ? tid_o - N_MAIN : tid_o >> 4), tid_o & 15);
        }
#else
        asm volatile("" :: "v"(r0[0]), "v"(r0[1]), "v"(r0[2]), "v"(r0[3]), "v"(r1[0]), "v"(r1[1]), "v"(r1[2]), "v"(r1[3]));
#endif
        if (tid == 0) { sm.next_tile = nt; sm.next_end = ne; }
        lds_barrier();
        if (tid == 0) { sm.unc[par ^ 1] = 0; sm.low[par ^ 1] = 0; sm.dim[par ^ 1] = 0; }          // (read by all before this barrier, written again behind the next tile's)
        // pixels at most 64 above black among the rows this tile loaded (or patched); with those of the tile before: among its output rows
        // (one scalar for both: bit 0 = low, bit 1 = dim)
        const int low_cur = tile_patched ? 3 : (__builtin_amdgcn_readfirstlane(sm.low[par]) | __builtin_amdgcn_readfirstlane(sm.dim[par]) << 1);
        const int t_next = __builtin_amdgcn_readfirstlane(sm.next_tile), t_end_next = __builtin_amdgcn_readfirstlane(sm.next_end);
        // the tile after this one continues it when it is the next of the list and not the top of a column
        const bool cont_next = t_next == t + 1 && trow + 1 < a.tiles_y && t_next < band_end;
        if (tile_patched) {
            KArgs ka = cold_args();
            const int4 *cells = ka->cells + (size_t)cur.f * ka->n_rec;
            auto patch_one = [&](int4 rec) {
                PatchCell c = patch_cell<METHOD, PACKED, Smem>(sm, a.black, rec, tx0, ty0);
                if (c.i < 0) return;
                sm.pk[c.j][c.i] = as_u(__builtin_amdgcn_cvt_pk_i16(__builtin_elementwise_sub_sat(c.dr, ref_r), __builtin_elementwise_sub_sat(c.db, ref_b)));
                const int ii = c.i - HC, jj = c.j - HC;
                if (ii >= 0 && ii < TCW && jj >= 0 && jj < RH) {
                    sm.ge[jj][ii] = c.ge;
                    *(uint32_t *)&sm.raw[2 * jj][2 * ii] = c.top;
                    *(uint32_t *)&sm.raw[2 * jj + 1][2 * ii] = c.bot;
                }
            };
            patch_one(my_rec);
            for (int base = pbeg + 256; base < pend; base += 256)         // a dense map (focus pixels): the rest
                patch_one(base + tid < pend ? cells[base + tid] : make_int4(-1, 0, 0, 0));
            lds_barrier();
        }
        __builtin_amdgcn_s_setprio(1);
        // ---- prefetch the next tile (and, where it starts anew, its sample) while the medians run
        Pos nxt = pos_below(cur);
        if (t_next != t + 1 || t_next >= band_end) nxt = pos_of(min(t_next, band_end - 1));
#ifndef KFP_EXP_NOPREF
        {
            int tid_p = tid;
            asm volatile("" : "+v"(tid_p));
            issue_tile(sa, nxt, tid_p);
            if (!cont_next) issue_sample(sa, nxt, tid_p);
        }
#endif
        // what the output stage needs of the cold arguments, in one go and early: the loads go out together, long before their first use
        const OutArgs oa = out_args(cold_args());

        // ---- medians + output: one thread = 4 cells = 8 px on two rows
        int tid_m = tid;
        asm volatile("" : "+v"(tid_m));
        // lane -> (row j, strip k).  5x5: 17 consecutive lanes per tile row (16 strips and the halo group); other methods: 16
        const int j_ = CHAIN ? (int)(__umul24((uint32_t)tid_m, 241u) >> 12) : tid_m >> 4;       // tid / 17 for tid < 256 (24-bit multiply: v_mul_lo_u32 takes twice the issue slots)
        const int k = CHAIN ? tid_m - (int)__umul24((uint32_t)j_, 17u) : tid_m & 15;
        const int j = min(j_, TCH - 1);
        const bool is_strip = CHAIN ? (k < 16 && tid_m < 17 * TCH) : tid_m < N_MAIN;
        const int y = ty0 + 2 * j, x = tx0 + 2 * STRIP * k;
        const bool smooth_row = y >= 4 && y < a.h - 5;                                    // chroma_smooth.c:25
        mlv_pk16 o[STRIP] = { { 0, 0 }, { 0, 0 }, { 0, 0 }, { 0, 0 } };
#ifndef KFP_EXP_NOMED
        if (CHAIN) {
            const int lane = tid_m & 63;
            const bool publishes = lane == 0 && tid_m != 0, collects = lane == 63 && tid_m < 192;
            const int wv = tid_m >> 6;
            PGroup g;
            pchain_group(sm.pk, j, STRIP * k, g);
            // (every wave makes its rank window before the hand-over: k_frame lets wave 0, which publishes nothing, make it behind the
            // barrier -- two copies of the network and the register moves that merge them)
            mlv_quad_mid6(g.p0, g.p1, g.q);
            uint32_t *const xo = (uint32_t *)((char *)&sm.xchg[0][0] + __umul24((uint32_t)wv, (uint32_t)(XCHG_WORDS * 4)));      // record of this wave's last lane
            if (publishes) pchain_publish(g, xo - XCHG_WORDS);
            lds_barrier();
            PNext n;
            pchain_fetch_lists(g, n);
            pchain_fetch_window(g, n);
            if (collects) { pchain_collect_lists(xo, n); pchain_collect_window(xo, n); }
            pchain_finish(g, n, o);
        } else if (smooth_row) {
            if (METHOD == 3) pstrip_median9(sm.pk, j, STRIP * k, o);
            else pstrip_median5(sm.pk, j, STRIP * k, o);
        }
        if (is_strip && smooth_row && pk_uncertain(o)) sm.unc[par] = 1;
#else
        { const uint4 v = *(const uint4 *)&sm.pk[j + 2][STRIP * k]; o[0] = as_pk(v.x); o[1] = as_pk(v.y); o[2] = as_pk(v.z); o[3] = as_pk(v.w); }
#endif
        // ---- R / B replacement, stripes, store: the medians come relative to the reference
#ifdef KFP_EXP_NOOUT
        if (is_strip && o[0].x == 12345 && o[1].y == 321) {
#else
        if (is_strip) {
#endif
            const int4 g4 = *(const int4 *)&sm.ge[j][STRIP * k];
            const int gev[STRIP] = { g4.x, g4.y, g4.z, g4.w };
            int er[STRIP], eb[STRIP];
#pragma unroll
            for (int c = 0; c < STRIP; c++) {
                er[c] = wadd(wadd(gev[c], ref_r), (int)o[c].x);
                eb[c] = wadd(wadd(gev[c], ref_b), (int)o[c].y);
            }
            strip_output<METHOD, PACKED, true, Smem>(sm, oa, a.w, a.h, a.black, cur.f, tx0, ty0, j, k, lanes_ge(y, 4) & lanes_lt(y, a.h - 5), gev, 0, er, eb, ((low_cur | low_prev) & 1) != 0, (low_cur | low_prev) == 0, true);
        }
        // ---- the rows the tile below shares with this one: read before the barrier that ends the tile, stored behind it -- one
        // 16-byte piece per thread, from and to where the thread's entry of the table says (threads without a piece copy a spare
        // 16 bytes onto themselves: no predicate, no address arithmetic; deriving the piece from the thread's number cost 19 vector
        // instructions per tile and wave)
        int4 carry = make_int4(0, 0, 0, 0);
        int2 c_at = make_int2(0, 0);
        if (cont_next) {
            c_at = sm.carry_at[tid];
            carry = *(const int4 *)((const char *)&sm + c_at.x);
        }
        lds_barrier();
#if defined(KFP_EXP_NOLOAD) || defined(KFP_EXP_NOPREF) || defined(KFP_EXP_NOMED) || defined(KFP_EXP_NOOUT)
        const int unc = 0 * __builtin_amdgcn_readfirstlane(sm.unc[par]);       // (the experiments' planes hold garbage)
#else
        const int unc = __builtin_amdgcn_readfirstlane(sm.unc[par]);
#endif
        if (cont_next) *(int4 *)((char *)&sm + c_at.y) = carry;
        have_smp = !cont_next;
        if (unc) {
            // a tile with uncertain strips: to the list (k_frame does it again, whole), the next fb_wait tiles with it unseen; where
            // none is skipped the next tile starts anew, with a reference of its own
            if (tid == 0) push(t, 1);
            fb_skip = fb_wait;
            fb_wait = min(2 * fb_wait + 1, KF_P_WAIT_MAX);
            cont = false;
        } else {
            fb_wait = KF_P_WAIT_MIN;
            cont = cont_next;
        }
        low_prev = low_cur;
        t = t_next;
        t_end = t_end_next;
        cur = nxt;
        par ^= 1;
    }
    if (tid == 0) {
        KArgs ka = cold_args();
        const int groups = ka->groups;
        int *tickets = ka->tickets;
        if (atomicAdd(&tickets[2 * groups], 1) == (int)gridDim.x - 1)
            for (int i = 0; i <= 2 * groups; i++) tickets[i] = 0;   // last workgroup out: ready for the next launch on this stream
    }
}

// (2x2 and 3x3: k_frame's int32 networks compile to three-input min / max / med3 and the packed two-input ones save them nothing;
// MLVFS_AMD_KF_P_ALL=1 sends them through k_frame_p all the same: A/B)
bool frame_p_exists(int method, int vec)
{
    static const bool all = [] { const char *e = getenv("MLVFS_AMD_KF_P_ALL"); return e && e[0] == '1'; }();
    return (method == 5 || (all && method != 0)) && vec != 0;
}

// ================================================================ k_frame_p5: the packed-once pass as a streaming kernel (round 5, end)
// k_frame_s's form (k_frame_s.hip) for cs5x5: a WAVE owns a column of the frame 62 items wide (lanes 0 / 63: the halo items), walks down it a
// cell row per step, keeps the five packed rows of the window in registers, gets the right-hand group's sorted columns and rank window
// from the next lane (v_mov_b32_dpp wave_shl, as k_frame_p's lanes do inside a row) and finishes row r - 2 when row r is in.  No
// s_barrier after the table load, no planes in LDS (only the pixels of the three rows between load and output wait there, in the
// wave's own 6 KiB): 113 VGPRs, 40 KiB of LDS, four workgroups per CU -- and 6.0 instead of 6.8 us per frame (profiles/r05/ab_p5.log).
//   * Reference: one per task (a column of KF_P5_SEG rows), the median of five cells at its centre like k_frame_p's.
//   * A strip whose packed median is not provably exact marks its lane and row; when the task ends, the tiles of k_frame's geometry
//     that those strips lie in go on the launch's work list and the list-mode k_frame does them again (what this kernel wrote there
//     is overwritten).
//   * Pixel-map cells: the records of the tiles that overlap the task's region are collected, one per lane (at most 64: a denser map
//     sends the whole region to the list), and replace the cell's pixels in the step that loads their row.
//   * Rows with pixels at or below black take the loader's second form, whole waves at a time; the output stage's variants are
//     chosen per row from the five rows of its window.
// Takes long launches of 14-bit streams (rows of whole 8-pixel groups, even heights, black >= 0, stripes packed or none); everything
// else, and every short launch, stays with k_frame_p / k_frame.
#ifndef KF_P5_SEG
#define KF_P5_SEG 60
#endif
__device__ __forceinline__ int dpp_prev_ii(int v) { return __builtin_amdgcn_mov_dpp(v, 0x138, 0xf, 0xf, true); }
template <bool SPREAD, int VEC>
__global__ __launch_bounds__(256, 4) void k_frame_p5(const FrameArgs a, int cols, int segs, int seg_rows, int fold, int S_OUT)
{
    constexpr int BPP = 14;                              // (S_OUT: items a wave writes per row, k_frame_dev.h: frame_stream_colw)
    __shared__ __align__(16) uint16_t t16[MLV_T16_N + (SPREAD ? 64 : 0)];
    // (the dark-clip table layout is 128 bytes longer: there only the 62 lanes that write a row park it, which keeps the workgroup at 40 KiB)
    constexpr int PARK_LANES = SPREAD ? 62 : 64;
    __shared__ uint4 park[4][3][2][PARK_LANES];
    load_t16_rel<SPREAD>(t16, cold_args()->t16, (int)threadIdx.x);
    __syncthreads();                                     // the only barrier
    const int lane = (int)threadIdx.x & 63;
    const int w = a.w, h = a.h, black = a.black;
    const int rows = h >> 1, gmax = (w >> 3) - 1;
    // fold > 1: the frame's last column is at most 64 / fold - 2 items wide and a wave takes `fold` of its segments at once, one per
    // group of 64 / fold lanes (each with its own two halo lanes): 3584 px = 7 columns of 62 items and one of 14 -- 7.25 columns' worth
    // of steps instead of 8
    const int ncols_full = fold > 1 ? cols - 1 : cols, nfolded = fold > 1 ? (segs + fold - 1) / fold : 0;
    const int per_frame = ncols_full * segs + nfolded, ntasks = cold_args()->nframes * per_frame;
    const uint32_t pitch = (uint32_t)(w >> 3) * 14u;
    int *tickets = cold_args()->tickets;
    uint4 (*const mypark)[2][PARK_LANES] = park[threadIdx.x >> 6];
    const int plane = SPREAD ? min(max(lane - 1, 0), 61) : lane;       // the lane's slot (SPREAD: lanes 0 and 63 share their neighbours', unused)
    const bool parks = !SPREAD || (lane >= 1 && lane <= 62);
    for (;;) {
        int task = 0;
        if (lane == 0) task = atomicAdd(&tickets[0], 1);
        task = __builtin_amdgcn_readfirstlane(task);
        if (task >= ntasks) break;
        const int f = task / per_frame, rem = task - f * per_frame;
        const bool folded = rem >= ncols_full * segs;
        const int c = folded ? ncols_full : rem / segs, sg = folded ? (rem - ncols_full * segs) * fold : rem - c * segs;
        const int j0 = sg * seg_rows, j1 = min(j0 + seg_rows, rows);              // (of the first group of lanes; the others lie roff rows further down)
        const int nparts = folded ? fold : 1, P = folded ? 64 / fold : 64;
        const int pl = lane & (P - 1), part = folded ? lane / P : 0;
        const int roff = part * seg_rows;
        const int g_true = c * S_OUT + pl - 1;
        const int g = min(max(g_true, 0), gmax);
        const bool writes = pl >= 1 && pl <= min(S_OUT, P - 2) && g_true <= gmax && j0 + roff < rows;
        const uint32_t gbyte = (uint32_t)g * 14u;
        const uint32_t sel = (g & 1) ? SEL_MIS : SEL_SWAP;
        const uint32_t sel1 = VEC == 2 ? sel ^ (SEL_SWAP ^ SEL_MIS) : sel;
        KArgs kt = cold_args();                          // (what a task needs once: read here, not held in scalar registers through the launch)
        const uint8_t *const frame = kt->src + (size_t)f * kt->src_stride;
        const mlv_i32x4 rs_in = frame_rsrc(frame, kt->src_bytes);
        const mlv_i32x4 rs_out = frame_rsrc(kt->dst + (size_t)f * kt->dst_stride, (uint32_t)w * (uint32_t)h * 2u);
        const int tx0 = 8 * (c * S_OUT - 1);
        const bool xm = c == 0 || 8 * (c * S_OUT + S_OUT) > w - 4;
        uint32_t dA0[4], dA1[4], dB0[4], dB1[4];
        auto issue = [&](int r, uint32_t (&d0)[4], uint32_t (&d1)[4]) {
            const int rr = min(max(r + roff, 0), rows - 1);
            const uint32_t o0u = __umul24((uint32_t)(2 * rr), pitch) + gbyte, o0 = o0u & ~3u, o1 = (o0u + pitch) & ~3u;
            const mlv_u32x2 a0 = mlv_rbl_x2(rs_in, (int)o0, 0, KF_SRC_AUX), b0 = mlv_rbl_x2(rs_in, (int)o0 + 8, 0, KF_SRC_AUX);
            const mlv_u32x2 a1 = mlv_rbl_x2(rs_in, (int)o1, 0, KF_SRC_AUX), b1 = mlv_rbl_x2(rs_in, (int)o1 + 8, 0, KF_SRC_AUX);
            d0[0] = a0.x; d0[1] = a0.y; d0[2] = b0.x; d0[3] = b0.y;
            d1[0] = a1.x; d1[1] = a1.y; d1[2] = b1.x; d1[3] = b1.y;
        };
        issue(j0 - 2, dA0, dA1);
        issue(j0 - 1, dB0, dB1);
        // ---- the task's reference: median of five cells (no two of them neighbours) of the 16 x 4 block at its centre
        int ref_r, ref_b;
        {
            const int x0 = min(max(8 * (c * S_OUT + S_OUT / 2), 0), (w - 16) & ~1), y0 = min(j0 + j1, (h - 4) & ~1) & ~1;
            const uint32_t px = fetch_clamped<BPP>(frame, w, h, x0 + (lane & 15), y0 + (lane >> 4));
            const int l = min(max((int)px - black, 1), 16383);
            const int ix = ev_index(l);
            const int ev = ev_value(l, (int)t16[SPREAD ? ix + (ix >> 7) : ix]);
            constexpr int CX[5] = { 0, 3, 6, 1, 5 }, CY[5] = { 0, 0, 0, 1, 1 };
            int sdr[5], sdb[5];
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const int l0 = 2 * CX[k] + 32 * CY[k];
                const int rr = __builtin_amdgcn_readlane(ev, l0), g1 = __builtin_amdgcn_readlane(ev, l0 + 1);
                const int g2 = __builtin_amdgcn_readlane(ev, l0 + 16), bb = __builtin_amdgcn_readlane(ev, l0 + 17);
                const int gg = (g1 + g2) >> 1;
                sdr[k] = rr - gg; sdb[k] = bb - gg;
            }
            auto med5 = [](const int (&v)[5]) { return med3i(v[4], max(min(v[0], v[1]), min(v[2], v[3])), min(max(v[0], v[1]), max(v[2], v[3]))); };
            ref_r = med5(sdr);
            ref_b = med5(sdb);
        }
        // ---- pixel-map cells of the region (rows j0 - 2 .. j1 + 1, this wave's 256 cells): one record per lane
        int n_pm = 0, pm_cell = -1;                      // (lane k < n_pm holds record k: cell = cy << 16 | cx)
        uint32_t pm_top = 0, pm_bot = 0;
        bool list_all = false;                           // more records than lanes: the whole region goes to k_frame
        if (kt->patch) {
            // The host lists a cell in every tile whose plane holds it (k_frame's tiles, halo included), tile by tile in row-major order:
            // one contiguous range of records per tile row of the region.  All ranges are fetched at once (at most six tile rows for 64
            // rows of cells, 64 records each: a denser map sends the region to the list); a cell counts where its own tile row is read
            // (its copies in the tiles left and right of its own remain: the same pixels twice).
            KArgs ka = kt;
            const int4 *cells = ka->cells + (size_t)f * ka->n_rec;
            const int *toff = ka->tile_off;
            const int tx_n = ka->tiles_x, ty_n = ka->tiles_y;
            const int cx_lo = 4 * (c * S_OUT - 1), cx_hi = cx_lo + 4 * P;      // cells of a group's lanes
            const int tc0 = max(cx_lo - HC, 0) / TCW, tc1 = min((cx_hi - 1 + HC) / TCW, tx_n - 1);
            for (int q = 0; q < nparts; q++) {
                const int j0q = j0 + q * seg_rows, j1q = min(j0q + seg_rows, rows);
                if (j0q >= rows) break;
                const int cy_lo = j0q - 2, cy_hi = j1q + 2;
                const int tr0 = max(cy_lo, 0) / TCH, tr1 = min((cy_hi - 1) / TCH, ty_n - 1), ntr = tr1 - tr0 + 1;
                constexpr int MAXTR = 6;
                int my_lo = 0, my_hi = 0;
                if (lane < ntr && lane < MAXTR) { my_lo = toff[(tr0 + lane) * tx_n + tc0]; my_hi = toff[(tr0 + lane) * tx_n + tc1 + 1]; }
                if (ntr > MAXTR) list_all = true;
                int4 rec[MAXTR];
#pragma unroll
                for (int i = 0; i < MAXTR; i++) {
                    const int rb = __builtin_amdgcn_readlane(my_lo, i), re = __builtin_amdgcn_readlane(my_hi, i);
                    rec[i] = (i < ntr && rb + lane < re) ? cells[rb + lane] : make_int4(-1, 0, 0, 0);
                    if (i < ntr && re - rb > 64) list_all = true;
                }
#pragma unroll
                for (int i = 0; i < MAXTR; i++) {
                    const int rcx = rec[i].x & 0xFFFF, rcy = rec[i].x >> 16;
                    const bool in = rec[i].x >= 0 && rcx >= cx_lo && rcx < cx_hi && rcy >= cy_lo && rcy < cy_hi && rcy / TCH == tr0 + i;
                    unsigned long long m = __ballot(in);
                    while (m) {
                        const int src = __builtin_ctzll(m);
                        m &= m - 1;
                        if (n_pm < 64) {
                            const int vx = __builtin_amdgcn_readlane(rec[i].x, src), vy = __builtin_amdgcn_readlane(rec[i].y, src), vz = __builtin_amdgcn_readlane(rec[i].z, src);
                            // kept as (the step that loads the cell's row) << 16 | (lane << 2 | cell of the lane's four)
                            const int at = (((vx >> 16) - q * seg_rows) << 16) | (4 * P * q + (vx & 0xFFFF) - cx_lo);
                            if (lane == n_pm) { pm_cell = at; pm_top = (uint32_t)vy; pm_bot = (uint32_t)vz; }
                            n_pm++;
                        } else list_all = true;
                    }
                }
            }
        }
        uint32_t pkr[5][4] = {};                          // packed rows r - 4 .. r
        int ge2[4] = { 0, 0, 0, 0 }, ge1[4] = { 0, 0, 0, 0 };
        int fl1 = 3, fl2 = 3, fl3 = 3, fl4 = 3;          // low / dim flags of rows r - 1 .. r - 4 (k_frame_s.hip)
        unsigned long long unc_lanes = 0;                // lanes with an uncertain strip in some row of the task
        int unc_r0 = 1 << 30, unc_r1 = -1;               // ... and the rows
        auto step = [&](int r, uint32_t (&d0)[4], uint32_t (&d1)[4]) {
            uint32_t p0[8], p1[8];
            unpack8<BPP>(d0, sel, sel, sel, p0);
            unpack8<BPP>(d1, sel1, sel1, sel1, p1);
            if (r + 2 <= j1 + 1) issue(r + 2, d0, d1);
            if (n_pm) {                                   // pixel-map cells of this row: the owning lane takes the record's pixels
                unsigned long long m = __ballot(lane < n_pm && (pm_cell >> 16) == r);
                while (m) {
                    const int src = __builtin_ctzll(m);
                    m &= m - 1;
                    const int cell = __builtin_amdgcn_readlane(pm_cell, src);
                    const uint32_t t_ = (uint32_t)__builtin_amdgcn_readlane((int)pm_top, src), b_ = (uint32_t)__builtin_amdgcn_readlane((int)pm_bot, src);
                    const int cxr = cell & 0xFFFF;                                         // cell within the wave's 256
                    const bool mine = (cxr >> 2) == lane;
                    const int cc = cxr & 3;
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (mine && cc == k) { p0[2 * k] = t_ & 0xFFFFu; p0[2 * k + 1] = t_ >> 16; p1[2 * k] = b_ & 0xFFFFu; p1[2 * k + 1] = b_ >> 16; }
                }
            }
            uint32_t lo = min(p0[0], p1[0]);
#pragma unroll
            for (int i = 1; i < 8; i++) lo = min(min(lo, p0[i]), p1[i]);
            const bool dark = __any((int)lo <= black);
            int fl0 = 0;
            if (__any((int)lo <= black + 255)) fl0 = __any((int)lo <= black + 64) ? 3 : 2;
            int ge[4];
            uint32_t pk[4];
#ifdef KFP5_EXP_NOCELL
            if (true) {
#pragma unroll
                for (int cc = 0; cc < 4; cc++) { ge[cc] = (int)((p0[2 * cc + 1] + p1[2 * cc]) << 12); pk[cc] = (p0[2 * cc] & 255u) | ((p1[2 * cc + 1] & 255u) << 16); }
            } else
#endif
            if (!dark) cell_multi_pk_fast<4, SPREAD>(p0, p1, black, t16, ref_r, ref_b, ge, pk);
            else {
#pragma unroll
                for (int cc = 0; cc < 4; cc += 2) {
                    int g2[2], r2[2], b2[2];
                    cell_multi_ev_dark<2, SPREAD>(p0 + 2 * cc, p1 + 2 * cc, black, t16, g2, r2, b2);
                    ge[cc] = g2[0]; ge[cc + 1] = g2[1];
                    pk[cc] = as_u(__builtin_amdgcn_cvt_pk_i16(__builtin_elementwise_sub_sat(r2[0], ref_r), __builtin_elementwise_sub_sat(b2[0], ref_b)));
                    pk[cc + 1] = as_u(__builtin_amdgcn_cvt_pk_i16(__builtin_elementwise_sub_sat(r2[1], ref_r), __builtin_elementwise_sub_sat(b2[1], ref_b)));
                }
            }
            {
                uint4 (&slot)[2][PARK_LANES] = mypark[(unsigned)(r + 6) % 3u];
                if (parks) {
                    slot[0][plane] = make_uint4(p0[0] | (p0[1] << 16), p0[2] | (p0[3] << 16), p0[4] | (p0[5] << 16), p0[6] | (p0[7] << 16));
                    slot[1][plane] = make_uint4(p1[0] | (p1[1] << 16), p1[2] | (p1[3] << 16), p1[4] | (p1[5] << 16), p1[6] | (p1[7] << 16));
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) { pkr[0][i] = pkr[1][i]; pkr[1][i] = pkr[2][i]; pkr[2][i] = pkr[3][i]; pkr[3][i] = pkr[4][i]; pkr[4][i] = pk[i]; }
            const int jr = r - 2, y = 2 * jr, yl = y + 2 * roff;
            if (jr >= j0 && jr < j1) {
                const unsigned long long msmooth = lanes_ge(yl, 4) & lanes_lt(yl, h - 5);  // chroma_smooth.c:25
                const bool smooth_row = y + 2 * (nparts - 1) * seg_rows >= 4 && y < h - 5; // (some group's row)
                mlv_pk16 o[STRIP] = { { 0, 0 }, { 0, 0 }, { 0, 0 }, { 0, 0 } };
#ifdef KFP5_EXP_NOMED
                if (smooth_row) {
#pragma unroll
                    for (int cc = 0; cc < 4; cc++) o[cc] = as_pk(pkr[2][cc]);
                    o[0] = as_pk((uint32_t)dpp_prev_ii((int)as_u(o[2])));
                } else
#endif
                if (smooth_row) {
                    PGroup gq;
                    {
                        mlv_pk16 col[4][5];
#pragma unroll
                        for (int rr = 0; rr < 5; rr++)
#pragma unroll
                            for (int cc = 0; cc < 4; cc++) col[cc][rr] = as_pk(pkr[rr][cc]);
#pragma unroll
                        for (int cc = 0; cc < 4; cc++) mlv_sort5(col[cc], gq.s[cc]);
                        mlv_merge55(gq.s[0], gq.s[1], gq.p0);
                        mlv_merge55(gq.s[2], gq.s[3], gq.p1);
                    }
                    mlv_quad_mid6(gq.p0, gq.p1, gq.q);
                    PNext n;
                    pchain_fetch_lists(gq, n);
                    pchain_fetch_window(gq, n);
                    mlv_pk16 oc[STRIP];
                    pchain_finish(gq, n, oc);
                    // the chain's four medians are those of the lane's cells 2, 3 and of the NEXT lane's cells 0, 1 (its window is the lane's
                    // group and the next one): the lane's own cells 0, 1 come from the lane before
                    o[0] = as_pk((uint32_t)dpp_prev_ii((int)as_u(oc[2]))); o[1] = as_pk((uint32_t)dpp_prev_ii((int)as_u(oc[3])));
                    o[2] = oc[0]; o[3] = oc[1];
                    const unsigned long long um = __ballot(writes && pk_uncertain(o)) & msmooth;
                    if (um) { unc_lanes |= um; unc_r0 = min(unc_r0, jr); unc_r1 = max(unc_r1, jr); }
                }
                int er[STRIP], eb[STRIP];
#pragma unroll
                for (int cc = 0; cc < STRIP; cc++) {
                    er[cc] = wadd(wadd(ge2[cc], ref_r), (int)o[cc].x);
                    eb[cc] = wadd(wadd(ge2[cc], ref_b), (int)o[cc].y);
                }
                uint32_t top[STRIP], bot[STRIP];
                {
                    const uint4 (&slot)[2][PARK_LANES] = mypark[(unsigned)(jr + 6) % 3u];
                    const uint4 t4 = slot[0][plane], b4 = slot[1][plane];
                    top[0] = t4.x; top[1] = t4.y; top[2] = t4.z; top[3] = t4.w;
                    bot[0] = b4.x; bot[1] = b4.y; bot[2] = b4.z; bot[3] = b4.w;
                }
                const OutArgs oa = out_args(cold_args());          // (read here, not held through the step: 39 -> 17 spilled scalars)
                const int fl = fl0 | fl1 | fl2 | fl3 | fl4;                                 // the five rows of the window
#define KFP5_OUT(CLAMP, XM, BRIGHT) strip_output_t<5, true, true, CLAMP, XM, false, BRIGHT, NoSmem, true>(NoSmem(), oa, w, h, black, f, tx0, 0, jr, pl, msmooth, \
                                                                                                          ge2, 0, er, eb, false, top, bot)
#ifdef KFP5_EXP_NOOUT
                if (er[0] == 0x12345 && eb[1] == 0x54321) KFP5_OUT(false, false, true);
                else if (er[0] != 0x12345 || eb[1] != 0x54321) { top[0] ^= er[0] ^ er[1] ^ er[2] ^ er[3]; bot[0] ^= eb[0] ^ eb[1] ^ eb[2] ^ eb[3]; }
                else
#endif
                if (fl & 1) { if (xm) KFP5_OUT(true, true, false); else KFP5_OUT(true, false, false); }
                else if (xm) KFP5_OUT(false, true, false);
                else if (fl == 0) KFP5_OUT(false, false, true);
                else KFP5_OUT(false, false, false);
#undef KFP5_OUT
                if (writes) {
                    const uint32_t vo = (__umul24((uint32_t)yl, (uint32_t)w) + (uint32_t)(8 * g)) * 2u;  // (rows below the frame: beyond the buffer's range)
                    const mlv_u32x4 vt = { top[0], top[1], top[2], top[3] }, vb_ = { bot[0], bot[1], bot[2], bot[3] };
                    mlv_rbs_x4(vt, rs_out, (int)vo, 0, 2);
                    mlv_rbs_x4(vb_, rs_out, (int)vo, w * 2, 2);
                }
            }
#pragma unroll
            for (int cc = 0; cc < 4; cc++) { ge2[cc] = ge1[cc]; ge1[cc] = ge[cc]; }
            fl4 = fl3; fl3 = fl2; fl2 = fl1; fl1 = fl0;
        };
        for (int r = j0 - 2; r <= j1 + 1; r += 2) {
            step(r, dA0, dA1);
            if (r + 1 <= j1 + 1) step(r + 1, dB0, dB1);
        }
        // ---- what this task could not settle goes to k_frame: the tiles (64 x 15 cells) its uncertain strips lie in
        if (list_all) { unc_lanes = ~0ull; unc_r0 = j0; unc_r1 = j1 - 1; }
        if (unc_lanes && lane == 0) {
            KArgs kl = cold_args();
            const int tnx = kl->tiles_x, tny = kl->tiles_y;
            int *ctl = kl->wl_ctl;
            for (int q = 0; q < nparts; q++) {
                const unsigned long long mq = P == 64 ? unc_lanes : (unc_lanes >> (q * P)) & ((1ull << P) - 1);
                const int r0 = unc_r0 + q * seg_rows, r1 = min(unc_r1 + q * seg_rows, rows - 1);
                if (!mq || r0 >= rows) continue;
                const int l0 = __builtin_ctzll(mq), l1 = 63 - __builtin_clzll(mq);
                const int cx0 = max(4 * (c * S_OUT + l0 - 1), 0), cx1 = min(4 * (c * S_OUT + l1 - 1) + 3, w / 2 - 1);
                const int tc0 = cx0 / TCW, tc1 = min(cx1 / TCW, tnx - 1), tr0 = r0 / TCH, tr1 = min(r1 / TCH, tny - 1);
                for (int tc = tc0; tc <= tc1; tc++) {
                    const int i = atomicAdd(&ctl[0], 1);
                    kl->wl[i] = make_int2(f * tnx * tny + tc * tny + tr0, tr1 - tr0 + 1);
                    atomicAdd(&ctl[3], tr1 - tr0 + 1);
                }
            }
        }
    }
    if (lane == 0) {
        const int nwaves = (int)gridDim.x * 4;
        if (atomicAdd(&tickets[1], 1) == nwaves - 1) { tickets[0] = 0; tickets[1] = 0; }
    }
}

// does the streaming form take this two-kernel launch, and in tasks of how many rows?  0: no  (launch_frame_p_kernel)
static int frame_p5_takes(int method, bool packed, int vec, int num_cu, const FrameArgs &a)
{
    if (method != 5 || !packed || (vec != 1 && vec != 2)) return 0;
    const char *e = getenv("MLVFS_AMD_KF_P5");                                  // 0 never, 1 (default) long launches, 2 whenever it can
    const int policy = e ? atoi(e) : 1;
    if (policy == 0) return 0;
    if ((a.stripes && !a.coef_pk) || a.black < 0) return 0;
    if (!(a.w >= 16 && a.w % 8 == 0 && a.h >= 2 && a.h % 2 == 0)) return 0;
    static const int env_seg = [] { const char *e = getenv("MLVFS_AMD_KF_P5_SEG"); return e ? atoi(e) : 0; }();      // (experiments)
    if (env_seg > 0) return env_seg;
    // at least 3.5 tasks per wave (k_frame_s.hip: why), in tasks of 60 rows or, for launches half as long, of 30 (two warm-up rows per
    // task: 3584x1320, us per frame at 50 / 100 / 200 / 400 frames per launch: k_frame_p 7.7 / 7.4 / 7.1 / 6.9, tasks of 60 rows 8.3 /
    // 7.4 / 6.6 / 6.1, of 30 rows 8.6 / 6.9 / 6.7 / 6.2; profiles/r05/ab_p5.log)
    const long long cols = frame_stream_cols(a.w), rows = a.h / 2;
    const long long waves = (long long)(num_cu > 0 ? num_cu : 256) * 16;
    for (int seg : { KF_P5_SEG, KF_P5_SEG / 2 })
        if ((long long)a.nframes * cols * ((rows + seg - 1) / seg) * 2 >= waves * 7) return seg;
    return policy == 2 ? KF_P5_SEG / 2 : 0;
}
// prefer_tiles: the stream's last launches listed more than a few per cent of their tiles (k_frame.hip: stream_state) -- footage with
// regions of uncertain strips, where k_frame_p's skipping of the tiles behind an uncertain one saves what k_frame_p5 would do in vain
// (low light, 400 frames per launch: 8.2 against 8.8 us per frame)
void launch_frame_p_kernel(int method, bool packed, int vec, bool spread, int grid, hipStream_t stream, const FrameArgs &a, bool prefer_tiles)
{
#ifndef KFP_ONLY
    const char *e5 = getenv("MLVFS_AMD_KF_P5");
    if (prefer_tiles && !(e5 && atoi(e5) == 2)) {}
    else if (const int seg_rows = frame_p5_takes(method, packed, vec, grid / 4, a)) {
        const int cols = frame_stream_cols(a.w), segs = (a.h / 2 + seg_rows - 1) / seg_rows;
        // a narrow last column: several of its segments side by side in one wave (k_frame_p5: fold)
        const int fold = frame_stream_fold(a.w, cols, segs);
#define KFP5_GO(S, V) hipLaunchKernelGGL((k_frame_p5<S, V>), dim3(grid), dim3(256), 0, stream, a, cols, segs, seg_rows, fold, frame_stream_colw())
        if (vec == 2) { if (spread) KFP5_GO(true, 2); else KFP5_GO(false, 2); }
        else { if (spread) KFP5_GO(true, 1); else KFP5_GO(false, 1); }
#undef KFP5_GO
        return;
    }
#endif
#ifdef KFP_ONLY          // (tools: one or two instantiations, seconds to compile)
    if (method == 5) hipLaunchKernelGGL((k_frame_p<5, true, 1, false>), dim3(grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((k_frame_p<2, true, 1, false>), dim3(grid), dim3(256), 0, stream, a);
#else
#define MLV_P(M, P, V, S) hipLaunchKernelGGL((k_frame_p<M, P, V, S>), dim3(grid), dim3(256), 0, stream, a)
#define MLV_P_S(M, S)                                                                                             \
    do {                                                                                                          \
        if (packed) { if (vec == 1) MLV_P(M, true, 1, S); else if (vec == 2) MLV_P(M, true, 2, S);                \
                      else if (vec == 3) MLV_P(M, true, 3, S); else if (vec == 4) MLV_P(M, true, 4, S); }         \
        else { if (vec == 1) MLV_P(M, false, 1, S); else if (vec == 2) MLV_P(M, false, 2, S); }                   \
    } while (0)
#define MLV_P_M(M) do { if (spread) MLV_P_S(M, true); else MLV_P_S(M, false); } while (0)
    if (method == 2) MLV_P_M(2);
    else if (method == 3) MLV_P_M(3);
    else if (method == 5) MLV_P_M(5);
#undef MLV_P_M
#undef MLV_P_S
#undef MLV_P
#endif
}

// the code object of this file loaded when the device context is created, like the other files of the frame path (runtime.cpp:
// get_device): without it the first clip of a process paid 1.7 ms for it at its first fused launch
void preload_k_frame_p() { hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, (const void *)k_frame_p<5, true, 1, false>); (void)hipGetLastError(); }

}  // namespace mlv
