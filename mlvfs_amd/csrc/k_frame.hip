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
#include "k_frame_dev.h"

namespace mlv {

template <int METHOD, bool PACKED, int VEC, bool SPREAD>
__global__ __launch_bounds__(256, 4) void k_frame(const FrameArgs a)
{
    constexpr bool CHAIN = METHOD == 5;
    using Smem = SmemT<SPREAD, CHAIN>;
    __shared__ Smem sm;                                  // static: a compile-time LDS base (a dynamic one costs an add per access)

    // (the arguments the common path touches once per tile or less are read where they are used: cold_args, k_frame_dev.h)
    // list mode: the runs of tiles that k_frame_p (same stream, before this launch) left to this kernel.  Every workgroup draws a
    // run first thing, so those whose number is beyond the list's length find none and end at once -- without touching a counter
    // (an empty list, the common case: 1 024 workgroups adding to one address made the empty launch 14 us long); the last of the
    // others to end zeroes the counters and leaves the stream's statistic in the host's word.
    const int list_mode = a.list_mode;
    const int wl_n = list_mode ? cold_args()->wl_ctl[0] : 0;     // (written by the launch before this one: final)
    auto list_done = [&]() {                             // thread 0 of a list-mode workgroup that had work and ends
        KArgs ka = cold_args();
        int *ctl = ka->wl_ctl, *stat = ka->wl_stat;
        if (atomicAdd(&ctl[2], 1) == min((int)gridDim.x, wl_n) - 1) {
            ctl[0] = 0; ctl[1] = 0; ctl[2] = 0;
            if (stat) { __atomic_store_n(stat, ctl[3], __ATOMIC_RELAXED); __threadfence_system(); }
        }
    };
    if (list_mode && (int)blockIdx.x >= wl_n) {
        if (wl_n == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
            KArgs ka = cold_args();
            int *stat = ka->wl_stat;
            if (stat) { __atomic_store_n(stat, ka->wl_ctl[3], __ATOMIC_RELAXED); __threadfence_system(); }
        }
        return;
    }
    if (METHOD != 0) load_t16_rel<SPREAD>(sm.t16, cold_args()->t16, threadIdx.x);       // (relative form: k_frame_dev.h)
    // Which tiles of a frame have pixel-map entries: one bit per tile in LDS.  Reading the tile's list bounds from HBM in
    // every iteration made each wave wait for ALL its outstanding loads (the prefetch of the next tile included) before the
    // median phase; now only the few tiles that are touched fetch their bounds.
    const bool pmap_ok = a.tiles_x * a.tiles_y <= PMAP_WORDS * 32;
    if (a.patch && pmap_ok) {
        for (int i = threadIdx.x; i < PMAP_WORDS; i += blockDim.x) sm.has_patch[i] = 0;
        __syncthreads();
        const int *toff = cold_args()->tile_off;
        for (int i = threadIdx.x; i < a.tiles_x * a.tiles_y; i += blockDim.x)
            if (toff[i + 1] != toff[i]) atomicOr(&sm.has_patch[i >> 5], 1u << (i & 31));
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
    int band_end;
    {
        KArgs ka = cold_args();
        const int total = tiles_per_frame * ka->nframes;          // < 2^30 (checked by the launcher)
        const int groups = ka->groups, nx = 8;
        const int grp = blockIdx.x % groups;
        const int gpx = (groups + nx - 1) / nx;                                    // groups per XCD
        const int grank = (groups % nx == 0) ? (grp % nx) * gpx + grp / nx : grp;  // position of the group's range in the tile list
        const int gq = total / groups, grem = total - gq * groups;
        const int band_start = list_mode ? 0 : grank * gq + min(grank, grem);
        band_end = list_mode ? total : band_start + gq + (grank < grem ? 1 : 0);
        const int run = max(ka->run, 1);
        const int runs_len = max(band_end - band_start - ka->singles, 0) / run * run;  // tiles of the range that go out in runs
        // (what only thread 0 needs, when it draws, waits in LDS)
        if (threadIdx.x == 0) { sm.walk[0] = band_start; sm.walk[1] = runs_len; sm.walk[2] = run; sm.walk[3] = grp; sm.walk[4] = 0; }
    }
    constexpr bool vec = VEC != 0;                       // w % 8 == 0, aligned buffers: vector loads and stores
    constexpr int BPP = bpp_of(PACKED, VEC);             // bits per pixel of the input
    constexpr int NEW0 = METHOD == 0 ? HC : 2 * HC;      // first plane row a tile loads itself (without chroma smoothing: no halo at all)

    uint32_t r0[4], r1[4];                               // prefetch registers of this thread's item
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // loader: threads 0..239 own the main item (row t / 16, group t % 16) of the tile's new rows, threads 240..254 the edge items
    const bool l_edge = tid >= N_MAIN;
    const int l_row = l_edge ? tid - N_MAIN : tid >> 4;
    const ItemLane IL = item_lane<PACKED, VEC>(tid & 15, l_edge);
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

    auto draw = [&](int &nt, int &ne) {                  // thread 0: the next run of the group's range, or its next single tile
        KArgs ka = cold_args();
        if (list_mode) {                                 // (or the next run of the list)
            const int i = atomicAdd(&ka->wl_ctl[1], 1);
            nt = band_end; ne = band_end + 1;
            if (i < wl_n) { const int2 e = ka->wl[i]; nt = e.x; ne = e.x + e.y; }
            return;
        }
        int *tickets = ka->tickets;
        const int band_start = sm.walk[0], runs_len = sm.walk[1], run = sm.walk[2], grp = sm.walk[3];
        if (!sm.walk[4]) {
            const int p = atomicAdd(&tickets[2 * grp], run);
            if (p < runs_len) { nt = band_start + p; ne = nt + run; return; }
            sm.walk[4] = 1;                              // the runs of this group's range are all handed out
        }
        const int q = atomicAdd(&tickets[2 * grp + 1], 1);
        nt = min(band_start + runs_len + q, band_end);
        ne = nt + 1;
    };
    if (threadIdx.x == 0) {
        int nt, ne;
        draw(nt, ne);
        sm.next_tile = nt; sm.next_end = ne;
        sm.dark_items[0] = 0; sm.dark_items[1] = 0; sm.fb_count = 0; sm.low[0] = 0; sm.low[1] = 0;
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
    struct Src { const uint8_t *src; size_t src_stride; unsigned src_bytes; };
    auto src_of = [](KArgs ka) { Src s; s.src = ka->src; s.src_stride = ka->src_stride; s.src_bytes = ka->src_bytes; return s; };
    auto issue_tile = [&](const Src *ka, const Pos &p) {
#ifndef KF_W_OLD_PREFETCH
        issue_tile_rows<BPP, VEC>(r0, r1, ka->src + (size_t)p.f * ka->src_stride, ka->src_bytes, IL, l_row, a.w, a.h, p.tcol * 2 * TCW, p.trow * 2 * TCH, NEW0);
#else
        const mlv_i32x4 rs = frame_rsrc(ka->src + (size_t)p.f * ka->src_stride, ka->src_bytes);
        issue_item<BPP>(r0, r1, rs, IL, a.w, a.h, p.tcol * 2 * TCW, p.trow * 2 * TCH, NEW0 + l_row);
#endif
    };
    Pos cur = pos_of(min(t, max(band_end - 1, 0)));
    { const Src s0 = src_of(cold_args()); if (vec) issue_tile(&s0, cur); }
    __syncthreads();                           // T16 copy complete

#ifdef KF_DIAG_TIMES
    const uint64_t rt0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long diag_n[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#endif
    int par = 0;                               // tile parity: which of the two dark_items counters this tile uses
#ifdef KF_W_LOW
    int lpar = 0, low_prev = 3;                // which of the two low-pixel records; the record of the tile before (bit 0: low, bit 1: dim)
#endif
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
#ifdef KF_W_HOIST_SRC
        const Src sa = src_of(cold_args());      // (asked for here, needed when the next tile is prefetched: no wait there)
#endif
        const int tr = cur.trow * a.tiles_x + cur.tcol;          // the tile's number in the (row-major) pixel-map lists
        // ---- pixel-map entries of this tile (few tiles have any): list bounds now -- the wait that the uniform load implies is
        // for data the loader needs anyway --, the first 256 records themselves in flight while the loader phase runs
        const bool tile_patched = a.patch && (!pmap_ok || ((sm.has_patch[tr >> 5] >> (tr & 31)) & 1u));      // wave-uniform
        int pbeg = 0, pend = 0;
        int4 my_rec = make_int4(-1, 0, 0, 0);
        if (tile_patched) {
            KArgs ka = cold_args();
            const int *toff = ka->tile_off;
            pbeg = toff[tr];
            pend = toff[tr + 1];
            if (pbeg + tid < pend) my_rec = (ka->cells + (size_t)f * ka->n_rec)[pbeg + tid];
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
            } else { KArgs ka = cold_args(); fetch_rows<BPP>(ka->src + (size_t)f * ka->src_stride, a.w, a.h, tx0, lk, ty0 - 2 * HC + 2 * p, L.edge, p0, p1); }
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
#ifdef KF_W_LOW
                // what strip_output may skip: bit 1 = some pixel less than 256 above black (or 16-bit input), bit 0 = at most 64 above
                if (!PACKED || a.black < 0 || __any((int)lo <= a.black + 255)) {
                    const int v = (PACKED && a.black >= 0 && !__any((int)lo <= a.black + 64)) ? 2 : 3;
                    if (lane == 0) atomicOr(&sm.low[lpar], v);        // (several waves: OR, not store)
                }
#endif
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
                const ItemLane TL = item_lane<PACKED, VEC>(tid_o & 15, te);
                uint32_t q0[4] = { 0, 0, 0, 0 }, q1[4] = { 0, 0, 0, 0 };
                if (vec) { KArgs ka = cold_args(); issue_item<BPP, true>(q0, q1, frame_rsrc(ka->src + (size_t)f * ka->src_stride, ka->src_bytes), TL, a.w, a.h, tx0, ty0, trw); }
                do_item(TL, q0, q1, trw, tid_o & 15);
            }
        }
        const bool has_item = METHOD == 0 ? tid_o < N_MAIN : tid_o < N_ITEMS;
        if (has_item) do_item(IL, r0, r1, NEW0 + l_row, tid_o & 15);
        if (threadIdx.x == 0) { sm.next_tile = nt; sm.next_end = ne; }
        lds_barrier();
        const int t_next = __builtin_amdgcn_readfirstlane(sm.next_tile), t_end_next = __builtin_amdgcn_readfirstlane(sm.next_end);
#ifdef KF_W_LOW
        if (tid == 0) sm.low[lpar ^ 1] = 0;      // (read by all before this barrier, written again behind the next tile's)
        const int low_cur = (METHOD == 0 || tile_patched) ? 3 : __builtin_amdgcn_readfirstlane(sm.low[lpar]);
#endif
        // the tile after this one continues it when it is the next of the list and not the top of a column
        const bool cont_next = METHOD != 0 && t_next == t + 1 && trow + 1 < a.tiles_y && t_next < band_end;       // scalar
        if (tile_patched) {
            PatchCell c = patch_cell<METHOD, PACKED, Smem>(sm, a.black, my_rec, tx0, ty0);
            patch_store<METHOD, Smem>(sm, c);
            KArgs kc = cold_args();
            const int4 *cells = kc->cells + (size_t)f * kc->n_rec;
            for (int base = pbeg + 256; base < pend; base += 256) {         // a dense map (focus pixels): the rest
                const int4 rec = base + tid < pend ? cells[base + tid] : make_int4(-1, 0, 0, 0);
                c = patch_cell<METHOD, PACKED, Smem>(sm, a.black, rec, tx0, ty0);
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
#ifdef KF_W_HOIST_SRC
        if (vec) issue_tile(&sa, nxt);
#else
        { const Src s1 = src_of(cold_args()); if (vec) issue_tile(&s1, nxt); }
#endif

        // ---- medians + output: one thread = 4 cells = 8 px on two rows
        // the rest of a strip once its medians are known: R / B replacement, stripes, store
        // the rest of a strip once its medians are known: R / B replacement, stripes, store (strip_output, k_frame_dev.h; this kernel
        // keeps no record of low pixels: always the clamped look-up and the masked stripes epilogue)
#ifdef KF_W_LOW
        const bool w_low_any = ((low_cur | low_prev) & 1) != 0, w_bright = (low_cur | low_prev) == 0;
#else
        const bool w_low_any = true, w_bright = false;
#endif
        const OutArgs oa = out_args(cold_args());
        auto finish_strip = [&](int jj, int kk, unsigned long long msmooth, const int (&mr)[STRIP], const int (&mb)[STRIP], bool store) {
            int gev[STRIP] = { 0, 0, 0, 0 }, er[STRIP] = { 0, 0, 0, 0 }, eb[STRIP] = { 0, 0, 0, 0 };
            if (METHOD != 0) {
                const int4 g4 = *(const int4 *)&sm.ge[jj][STRIP * kk];
                gev[0] = g4.x; gev[1] = g4.y; gev[2] = g4.z; gev[3] = g4.w;
#pragma unroll
                for (int c = 0; c < STRIP; c++) { er[c] = wadd(gev[c], mr[c]); eb[c] = wadd(gev[c], mb[c]); }
            }
            strip_output<METHOD, PACKED, vec, Smem>(sm, oa, a.w, a.h, a.black, f, tx0, ty0, jj, kk, msmooth, gev, 0, er, eb, w_low_any, w_bright, store);
        };
        const int y = ty0 + 2 * j;
        const bool smooth_row = METHOD != 0 && y >= 4 && y < a.h - 5;                       // chroma_smooth.c:25
        const unsigned long long msmooth_row = METHOD != 0 ? (lanes_ge(y, 4) & lanes_lt(y, a.h - 5)) : 0ull;
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
            if (is_strip && !unknown) finish_strip(j, k, msmooth_row, mr, mb, true);
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
            if (is_strip) finish_strip(j, k, msmooth_row, mr, mb, true);
        }
        // ---- the rows the tile below shares with this one: read before the barrier that ends the tile, stored behind it
        int4 carry = make_int4(0, 0, 0, 0);
        int tid_c = tid;
        asm volatile("" : "+v"(tid_c));
        const bool do_carry = cont_next && tid_c < C_ALL;
        int c_dst = 0;
        if (do_carry) {                              // (which piece is a lane's: derived here, not kept in registers through the tile)
            int c_delta;
            if (tid_c < C_PL) { c_dst = (int)offsetof(Smem, dr) + 16 * tid_c; c_delta = TCH * PW * 4; }
            else if (tid_c < 2 * C_PL) { c_dst = (int)offsetof(Smem, db) + 16 * (tid_c - C_PL); c_delta = TCH * PW * 4; }
            else if (tid_c < 2 * C_PL + C_RAW) { c_dst = (int)offsetof(Smem, raw) + 16 * (tid_c - 2 * C_PL); c_delta = 2 * TCH * 2 * TCW * 2; }
            else { c_dst = (int)offsetof(Smem, ge) + 16 * (tid_c - 2 * C_PL - C_RAW); c_delta = TCH * TCW * 4; }
            carry = *(const int4 *)((const char *)&sm + c_dst + c_delta);
        }
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
                    finish_strip(j2, k2, ~0ull, mr, mb, tid < nfb);
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
#ifdef KF_W_LOW
        lpar ^= 1; low_prev = low_cur;
#endif
    }
    if (list_mode) {
        if (threadIdx.x == 0) list_done();
    } else if (threadIdx.x == 0) {
        KArgs ka = cold_args();
        const int groups = ka->groups;
        int *tickets = ka->tickets;
        if (atomicAdd(&tickets[2 * groups], 1) == (int)gridDim.x - 1)
            for (int i = 0; i <= 2 * groups; i++) tickets[i] = 0;   // last workgroup out: ready for the next launch on this stream
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
#ifndef KF_P_BUSY_PERCENT
#define KF_P_BUSY_PERCENT 30     // of a launch's tiles listed for k_frame: the next launches skip the packed-once kernel
#endif
#ifndef KF_P_HOLD_MIN
#define KF_P_HOLD_MIN 8
#endif
#ifndef KF_P_HOLD_MAX
#define KF_P_HOLD_MAX 64
#endif
#ifndef KF_RUN_MAX
#define KF_RUN_MAX 22
#endif
namespace {
struct StreamState {
    int *tickets = nullptr;      // group tickets of the range-mode launches, then the three work-list counters (FrameArgs::wl_ctl)
    int2 *wl = nullptr;          // work list between k_frame_p and the list-mode k_frame: one entry per run of tiles
    long long wl_cap = 0;
    // Which kernels a launch on this stream uses (round 5).  The list-mode launch leaves the stream's count of listed tiles in a
    // page-locked word when it ends; the NEXT launches read it without waiting for anything: footage the packed-once kernel lists
    // most tiles of (hard colour edges, deep shadows) goes to k_frame alone for `hold` launches, then one launch probes again.
    int *h_stat = nullptr;       // (host pointer, mapped)
    int *d_stat = nullptr;
    int stat_seen = 0;           // h_stat when it was last looked at
    long long pending_tiles = 0; // tiles of the two-kernel launches whose list counts have not been seen yet
    int wide_left = 0, hold = 0; // launches still to go to k_frame alone; how many after the next busy probe
    bool some_listed = false;    // the last verdict found more than KF_P5_CALM_PERCENT of the tiles listed: the tile kernel k_frame_p, which
                                 // skips the tiles behind an uncertain one unseen, does better there than the streaming k_frame_p5
    // the same for the streaming cs2x2 kernel (k_frame_s.hip): its steps that took the loader's form for pixels at or below black,
    // out of all steps (word 1 of h_stat; the cumulative count lives in tickets[S_STAT_AT])
    int s_seen = 0;
    long long s_pending = 0;
    int s_wide_left = 0, s_hold = 0;
};
std::mutex g_ticket_mu;
std::map<std::pair<int, hipStream_t>, StreamState> g_tickets;
constexpr int TICKET_INTS = 2 * MAX_GROUPS + 1 + 4 + 2;
#ifndef KF_P5_CALM_PERCENT
#define KF_P5_CALM_PERCENT 5
#endif
constexpr int S_STAT_AT = 2 * MAX_GROUPS + 1 + 4;      // k_frame_s: cumulative dark steps
#ifndef KF_S_BUSY_PERCENT
#define KF_S_BUSY_PERCENT 10
#endif
}
// wl_tiles > 0: the launch may use both kernels and then needs a work list that holds that many entries; *two says whether it does
static int stream_state(hipStream_t stream, long long wl_tiles, int policy, StreamState *out, bool *two)
{
    int dev = 0;
    *two = false;
    if (hipGetDevice(&dev) != hipSuccess) return MLVFS_AMD_ERR_HIP;
    std::lock_guard<std::mutex> lk(g_ticket_mu);
    StreamState &st = g_tickets[{ dev, stream }];
    if (!st.tickets) {
        // zeroed ON THE LAUNCHING STREAM: the streams are non-blocking, a null-stream hipMemset is not ordered before their
        // kernels (it once landed in the middle of the first launch, the "done" count never completed and the next launch
        // on that stream started from stale tickets)
        if (hipMalloc(&st.tickets, TICKET_INTS * sizeof(int)) != hipSuccess ||
            hipMemsetAsync(st.tickets, 0, TICKET_INTS * sizeof(int), stream) != hipSuccess) {
            set_error("ticket counters: allocation failed");
            if (st.tickets) (void)hipFree(st.tickets);
            g_tickets.erase({ dev, stream });
            return MLVFS_AMD_ERR_HIP;
        }
    }
    if (wl_tiles > 0 && policy != 0) {
        if (!st.h_stat) {                                 // (without the word every launch uses both kernels)
            if (hipHostMalloc((void **)&st.h_stat, 64, hipHostMallocMapped) == hipSuccess) {
                st.h_stat[0] = 0; st.h_stat[1] = 0;
                if (hipHostGetDevicePointer((void **)&st.d_stat, st.h_stat, 0) != hipSuccess) { (void)hipHostFree(st.h_stat); st.h_stat = nullptr; st.d_stat = nullptr; }
            } else st.h_stat = nullptr;
            (void)hipGetLastError();
        }
        bool use_p = true;
        if (policy == 1 && st.h_stat) {
            const int seen = __atomic_load_n(st.h_stat, __ATOMIC_RELAXED);
            if (seen != st.stat_seen || st.pending_tiles > 0) {
                // what the launches that have ended since listed, against the tiles of all launches that were under way (a
                // launch that has not ended yet counts with its tiles and no listed ones: the verdict errs towards "calm")
                const long long listed = (long long)(unsigned)(seen - st.stat_seen);
                if (seen != st.stat_seen) {
                    const bool busy = listed * 100 > st.pending_tiles * KF_P_BUSY_PERCENT;
                    st.some_listed = listed * 100 > st.pending_tiles * KF_P5_CALM_PERCENT;
                    if (busy) { st.hold = st.hold ? std::min(2 * st.hold, KF_P_HOLD_MAX) : KF_P_HOLD_MIN; st.wide_left = st.hold; }
                    else st.hold = 0;
                    st.stat_seen = seen;
                    st.pending_tiles = 0;
                }
            }
            if (st.wide_left > 0) { st.wide_left--; use_p = false; }
        }
        if (use_p) {
            if (wl_tiles > st.wl_cap) {
                // (hipFree waits for the device: no launch on this stream still reads the old list)
                if (st.wl) (void)hipFree(st.wl);
                st.wl = nullptr; st.wl_cap = 0;
                const long long cap = std::max(wl_tiles, 4096ll);
                if (hipMalloc(&st.wl, (size_t)cap * sizeof(int2)) != hipSuccess) { set_error("work list: allocation of %lld entries failed", cap); return MLVFS_AMD_ERR_HIP; }
                st.wl_cap = cap;
            }
            st.pending_tiles += wl_tiles;
            *two = true;
        }
    }
    *out = st;
    return MLVFS_AMD_OK;
}
// k_frame_s's launches: `steps` wave-steps are about to be launched; *use_s says whether the streaming kernel takes them.  Footage
// with pixels at or below black in most rows (deep shadows, colour patches) runs the loader's slower form in whole waves there, and
// k_frame -- which decides per item group and hides it behind its other workgroups -- is faster (cs2x2, 3584x1320: low light 5.4
// against 6.7 us per frame, colour patches 6.0 against 8.4; the benchmark's frames 4.8 against 4.7): such streams go back to
// k_frame for a while, like the packed-once kernel's (stream_state).
static int stream_state_s(hipStream_t stream, long long steps, StreamState *out, bool *use_s)
{
    bool two = false;
    int rc = stream_state(stream, 0, 0, out, &two);
    if (rc) return rc;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_ticket_mu);
    StreamState &st = g_tickets[{ dev, stream }];
    *use_s = true;
    const char *e = getenv("MLVFS_AMD_KF_S");
    const int policy = e ? atoi(e) : 1;                      // 2: always (frame_s_takes)
    if (policy == 1) {
        if (!st.h_stat) {
            if (hipHostMalloc((void **)&st.h_stat, 64, hipHostMallocMapped) == hipSuccess) {
                st.h_stat[0] = 0; st.h_stat[1] = 0;
                if (hipHostGetDevicePointer((void **)&st.d_stat, st.h_stat, 0) != hipSuccess) { (void)hipHostFree(st.h_stat); st.h_stat = nullptr; st.d_stat = nullptr; }
            } else st.h_stat = nullptr;
            (void)hipGetLastError();
        }
        if (st.h_stat) {
            const int seen = __atomic_load_n(&st.h_stat[1], __ATOMIC_RELAXED);
            if (seen != st.s_seen) {
                const long long dark = (long long)(unsigned)(seen - st.s_seen);
                const bool busy = dark * 100 > st.s_pending * KF_S_BUSY_PERCENT;
                if (busy) { st.s_hold = st.s_hold ? std::min(2 * st.s_hold, KF_P_HOLD_MAX) : KF_P_HOLD_MIN; st.s_wide_left = st.s_hold; }
                else st.s_hold = 0;
                st.s_seen = seen;
                st.s_pending = 0;
            }
            if (st.s_wide_left > 0) { st.s_wide_left--; *use_s = false; }
        }
    }
    if (*use_s) st.s_pending += steps;
    *out = st;
    return MLVFS_AMD_OK;
}
// The streams this library creates (per host thread, per host pipeline) give their counters back when they are destroyed: a
// recycled stream handle then starts from freshly zeroed counters instead of whatever an aborted launch left behind, and
// retired worker threads leak nothing.  Streams the caller owns keep their 8 KiB (and their list) until the process ends.
void release_stream_state(int device, hipStream_t stream)
{
    std::lock_guard<std::mutex> lk(g_ticket_mu);
    auto it = g_tickets.find({ device, stream });
    if (it == g_tickets.end()) return;
    if (it->second.tickets) (void)hipFree(it->second.tickets);
    if (it->second.wl) (void)hipFree(it->second.wl);
    if (it->second.h_stat) (void)hipHostFree(it->second.h_stat);
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
    if (b >= E2D_RECORDS_EXT) return;
    auto entry = [&](int i) { return (uint16_t)((((int)u16[i & 32767]) >> (13 - (i >> 15))) + black); };
    if (b >= E2D_RECORDS) { e2d[b] = make_uint2(entry(E2R_ENTRIES - 1), 0u); return; }       // beyond the table: its last value
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
    MLV_HIP(hipMalloc(&t, sizeof(uint2) * (E2D_RECORDS_EXT + 1)));          // the records, then the counter of the check
    int *d_bad = (int *)(t + E2D_RECORDS_EXT);
    int bad = -1;
    (void)hipMemsetAsync(d_bad, 0, sizeof(int), stream);
    hipLaunchKernelGGL(k_build_e2d, dim3((E2D_RECORDS_EXT + 255) / 256), dim3(256), 0, stream, dev->luts.u16, black, t, d_bad);
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

// The packed-once kernel (k_frame_p.hip) exists for the chroma-smoothing methods on the vector layouts; everything else is
// k_frame's alone.  MLVFS_AMD_KF_P=0 sends every launch to k_frame as rounds 1-4 did (A/B).
bool frame_p_exists(int method, int vec);
void launch_frame_p_kernel(int method, bool packed, int vec, bool spread, int grid, hipStream_t stream, const FrameArgs &a, bool prefer_tiles);
// k_frame_s.hip: cs2x2 as a streaming kernel without barriers (what it takes: frame_s_takes)
bool frame_s_takes(int method, bool packed, int vec, int num_cu, const FrameArgs &a);
void launch_frame_s_kernel(int method, bool spread, int vec, int num_cu, hipStream_t stream, const FrameArgs &a);
long long frame_s_steps(const FrameArgs &a);

template <int METHOD, bool PACKED, int VEC, bool SPREAD>
static int launch_frame_t(const FrameArgs &a_in, int num_cu, hipStream_t stream)
{
    const long long total = (long long)a_in.tiles_x * a_in.tiles_y * a_in.nframes;
    static const int env_wgs = [] { const char *e = getenv("MLVFS_AMD_KF_WGS_PER_CU"); return e ? atoi(e) : 0; }();      // (occupancy experiments)
    int grid = (num_cu > 0 ? num_cu : 256) * (env_wgs > 0 ? env_wgs : 4);          // 4 workgroups per CU (39 KiB LDS, <= 128 VGPRs)
    grid = (grid + 7) / 8 * 8;
    if (grid > total) grid = (int)((total + 7) / 8 * 8);
    if (frame_s_takes(METHOD, PACKED, VEC, num_cu, a_in)) {
        FrameArgs as = a_in;
        StreamState sst;
        bool use_s = true;
        const int rcs = stream_state_s(stream, frame_s_steps(a_in), &sst, &use_s);
        if (rcs) return rcs;
        if (use_s) {
        as.tickets = sst.tickets;
        as.wl_ctl = sst.tickets + S_STAT_AT;                  // (k_frame_s: [0] its cumulative count of dark steps)
        as.wl_stat = sst.d_stat ? sst.d_stat + 1 : nullptr;
        KernelTimer &tms = kernel_timer();
        const bool timed_s = tms.on && tms.used + 2 <= (int)tms.ev.size();
        if (timed_s) MLV_HIP(hipEventRecord(tms.ev[tms.used], stream));
        launch_frame_s_kernel(METHOD, SPREAD, VEC, num_cu, stream, as);
        if (timed_s) { MLV_HIP(hipEventRecord(tms.ev[tms.used + 1], stream)); tms.used += 2; }
        MLV_HIP(hipGetLastError());
        return MLVFS_AMD_OK;
        }
    }
    auto kern = k_frame<METHOD, PACKED, VEC, SPREAD>;
    // MLVFS_AMD_KF_P: 0 = k_frame alone (rounds 1-4), 1 = both kernels, k_frame alone while the footage is busy (default), 2 = always both
    const char *e_p = getenv("MLVFS_AMD_KF_P");                 // (read at every launch: the tests switch it)
    const int env_p = e_p ? atoi(e_p) : 1;
    FrameArgs a = a_in;
    StreamState st;
    bool two = false;
    int rc = stream_state(stream, frame_p_exists(METHOD, VEC) ? total : 0, env_p, &st, &two);
    if (rc) return rc;
    a.wl_stat = st.d_stat;
    a.tickets = st.tickets;
    a.wl = st.wl;
    a.wl_ctl = st.tickets + 2 * MAX_GROUPS + 1;
    a.list_mode = 0;
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
    if (two) {
        // k_frame_p does every tile whose packed medians are certain and lists the rest; k_frame in list mode does those again
        // (nothing listed: its workgroups end at once).  Both inside the timer's bracket: the pair is the pass.
        launch_frame_p_kernel(METHOD, PACKED, VEC, SPREAD, grid, stream, a, st.some_listed);
        a.list_mode = 1;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, stream, a);
    } else
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, stream, a);
    if (timed) { MLV_HIP(hipEventRecord(tm.ev[tm.used + 1], stream)); tm.used += 2; }
    MLV_HIP(hipGetLastError());
    static const bool env_dbg = [] { const char *e = getenv("MLVFS_AMD_KF_P_DEBUG"); return e && atoi(e) != 0; }();
    if (two && env_dbg) {                                // (A/B aid: how many tiles the packed-once kernel left to k_frame)
        int listed = 0;
        static int before = 0;
        if (hipStreamSynchronize(stream) == hipSuccess && hipMemcpy(&listed, a.wl_ctl + 3, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "k_frame_p<%d>: %lld tiles, %d listed for k_frame (%.2f %%)\n", METHOD, total, listed - before, 100.0 * (listed - before) / (double)total);
            if (listed - before > 0 && listed - before < 64) {       // a few: which ones (the list's first entries as the last launch left them)
                int2 e[8];
                if (hipMemcpy(e, a.wl, sizeof(e), hipMemcpyDeviceToHost) == hipSuccess)
                    for (int i = 0; i < 8; i++) {
                        const int tpf = a.tiles_x * a.tiles_y, r = e[i].x % tpf;
                        fprintf(stderr, "    entry %d: frame %d tile column %d row %d, %d tile(s)\n", i, e[i].x / tpf, r / a.tiles_y, r % a.tiles_y, e[i].y);
                    }
            }
            before = listed;
        }
    }
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
