// k_frame_s.hip -- the fused pass for cs2x2 and cs3x3 (chroma_smooth.c:22-71 with CHROMA_SMOOTH_2X2 / _3X3: the plus-shaped five, the
// nine) as a STREAMING kernel without barriers and without planes in LDS (round 5, end).
//
// k_frame's cs2x2 instantiation executes 5 % fewer instructions than it did and takes the same time: its time is the
// dependent chain of a tile -- prefetched words -> unpack -> table gathers -> barrier -> medians -> look-ups -> stores -> barrier --
// at four workgroups per CU.  A plus-shaped window needs one cell to the left and right and one row above and below, nothing a
// workgroup has to share through LDS:
//   * a WAVE owns a column of the frame 62 items wide (an item = 4 cells = 8 x 2 pixels, one per lane; lanes 0 and 63 hold the
//     halo items whose neighbouring cell the outermost output items need) and walks down it one cell row per step;
//   * the colour differences of the row above and the row being loaded stay in registers (the stencil's vertical taps), the
//     horizontal taps of a lane's outer cells come from the neighbouring lanes (v_mov_b32_dpp wave_shr / wave_shl);
//   * a step unpacks and converts row r, then finishes row r - 1: medians, look-ups, R / B replacement, stripes, two 16-byte stores;
//     the words of rows r + 1 and r + 2 are under way meanwhile (one row ahead: 6.7 us per frame, two: 4.75);
//   * no s_barrier after the table is in LDS, 16 KiB of LDS per workgroup (the raw2ev table), 126 VGPRs: four workgroups per CU;
//   * waves draw their tasks (frame, column, 60 rows) from one counter; the last wave out zeroes it for the next launch.
// Same arithmetic as k_frame: the loader's cell functions, mlv_median5, strip_output_t (k_frame_dev.h) -- results identical.
// Measured and not kept (profiles/r05/ab_kframe_s.log): compiled for five workgroups per CU (<= 96 VGPRs: 14-43 spilled, 6.7-10.5 us per
// frame against 4.75); a third row of prefetch with the previous row's pixels parked in LDS (+-0); a form that finishes row r - 2 while
// row r's table look-ups are under way, so that neither wait is exposed (186 VGPRs: the in-flight conversion, three rows of colour
// differences, the medians' operands and two sets of look-ups do not fit four waves per SIMD).
// What it takes: 14-bit streams whose rows are whole 8-pixel groups (on the buffers the vector path wants), even heights, no pixel
// map, stripes in the packed 16-bit form (or none), black >= 0.  Everything else stays with k_frame (k_frame.hip: launch_frame_t).
#include "k_frame_dev.h"

namespace mlv {

#ifndef KF_S_SEG
#define KF_S_SEG 60
#endif
#ifndef KF_S_PARK
#define KF_S_PARK 0                   // 1: a row's pixels wait for their medians in LDS instead of in registers (measured: no gain)
#endif
#ifndef KF_S_DEPTH
#define KF_S_DEPTH 2                  // rows of prefetch under way (2 or 3; 3 needs KF_S_PARK to stay within 128 registers)
#endif
#ifndef KF_S_NC
#define KF_S_NC 4                     // cells converted at once (their table look-ups in flight together)
#endif
#ifndef KF_S_WGS
#define KF_S_WGS 4                    // workgroups per CU the kernel is compiled for (<= 128 VGPRs; five would need <= 96: it spills)
#endif

__device__ __forceinline__ int dpp_prev_i(int v) { return __builtin_amdgcn_mov_dpp(v, 0x138, 0xf, 0xf, true); }        // wave_shr:1: lane l gets lane l - 1's

// VEC = 1: rows of whole 16-pixel groups (every row starts dword-aligned); VEC = 2: w % 16 == 8 (odd pixel rows start two bytes into a
// dword: their groups' alignment is the other way round -- the selectors flip, as in k_frame's loader)
// METHOD = 2: the plus-shaped five; METHOD = 3: the 3x3 nine (chroma_smooth.c with CHROMA_SMOOTH_3X3) -- the same rows in registers,
// sorted columns of three (k_frame_dev.h: strip_median9's scheme), the neighbouring lanes' edge columns by DPP
template <bool SPREAD, int VEC, int METHOD>
__global__ __launch_bounds__(256, KF_S_WGS) void k_frame_s(const FrameArgs a, int cols, int segs, int seg_rows, int fold, int S_OUT)
{
    constexpr int BPP = 14;
    __shared__ __align__(16) uint16_t t16[MLV_T16_N + (SPREAD ? 64 : 0)];
#if KF_S_PARK                                            // (experiment: a row's pixels wait one step for their medians in LDS -- the wave's own
                                                         // 4 KiB, two slots by row parity, no barrier involved -- instead of in registers)
    __shared__ uint4 park[4][2][2][64];
#endif
    load_t16_rel<SPREAD>(t16, a.t16, (int)threadIdx.x);
    __syncthreads();                                     // the only barrier: from here on the waves are on their own
    const int lane = (int)threadIdx.x & 63;
    const int w = a.w, h = a.h, black = a.black;
    const int rows = h >> 1, gmax = (w >> 3) - 1;
    // fold > 1: a last column of at most 64 / fold - 2 items, `fold` of its segments side by side in one wave (k_frame_p.hip: k_frame_p5)
    const int ncols_full = fold > 1 ? cols - 1 : cols, nfolded = fold > 1 ? (segs + fold - 1) / fold : 0;
    const int per_frame = ncols_full * segs + nfolded, ntasks = a.nframes * per_frame;
    const uint32_t pitch = (uint32_t)(w >> 3) * 14u;     // bytes per pixel row (VEC 1: a multiple of 28, rows start dword-aligned)
    const OutArgs oa = out_args(cold_args());
    int *tickets = a.tickets;
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
        const int pl = lane & (P - 1), roff = folded ? lane / P * seg_rows : 0;
        // the lane's 8-pixel group; lanes outside the frame (the halo lanes of the first and last column, the lanes behind a narrow
        // last column) take a group inside it: their values are never used, and never "dark"
        const int g_true = c * S_OUT + pl - 1;
        const int g = min(max(g_true, 0), gmax);
        const bool writes = pl >= 1 && pl <= min(S_OUT, P - 2) && g_true <= gmax && j0 + roff < rows;
        const uint32_t gbyte = (uint32_t)g * 14u;
        const bool mis = (g & 1) != 0;                   // the group starts in the upper half of a dword
        const uint32_t sel = mis ? SEL_MIS : SEL_SWAP;
        const uint32_t sel1 = VEC == 2 ? sel ^ (SEL_SWAP ^ SEL_MIS) : sel;      // the odd pixel row of the pair
        const mlv_i32x4 rs_in = frame_rsrc(a.src + (size_t)f * a.src_stride, a.src_bytes);
        const mlv_i32x4 rs_out = frame_rsrc(oa.dst + (size_t)f * oa.dst_stride, (uint32_t)w * (uint32_t)h * 2u);
        const int tx0 = 8 * (c * S_OUT - 1);             // x of lane 0's item
        const bool xm = c == 0 || 8 * (c * S_OUT + S_OUT) > w - 4;      // the column touches the frame's left or right margin

        uint32_t dA0[4], dA1[4], dB0[4], dB1[4], dC0[4], dC1[4];       // three rows under way: HBM's latency is two to three steps long
#if KF_S_PARK
        uint4 (*const mypark)[2][64] = park[threadIdx.x >> 6];
#endif
        auto issue = [&](int r, uint32_t (&d0)[4], uint32_t (&d1)[4]) {
            const int rr = min(max(r + roff, 0), rows - 1);     // (rows above / below the frame: never used either)
            const uint32_t o0u = __umul24((uint32_t)(2 * rr), pitch) + gbyte, o0 = o0u & ~3u, o1 = (o0u + pitch) & ~3u;
            const mlv_u32x2 a0 = mlv_rbl_x2(rs_in, (int)o0, 0, KF_SRC_AUX), b0 = mlv_rbl_x2(rs_in, (int)o0 + 8, 0, KF_SRC_AUX);
            const mlv_u32x2 a1 = mlv_rbl_x2(rs_in, (int)o1, 0, KF_SRC_AUX), b1 = mlv_rbl_x2(rs_in, (int)o1 + 8, 0, KF_SRC_AUX);
            d0[0] = a0.x; d0[1] = a0.y; d0[2] = b0.x; d0[3] = b0.y;
            d1[0] = a1.x; d1[1] = a1.y; d1[2] = b1.x; d1[3] = b1.y;
        };
        // rows r - 2 (colour differences only) and r - 1 (everything: it is finished when row r is in)
        int dr2[STRIP] = { 0, 0, 0, 0 }, db2[STRIP] = { 0, 0, 0, 0 };
        int dr1[STRIP] = { 0, 0, 0, 0 }, db1[STRIP] = { 0, 0, 0, 0 }, ge1[STRIP] = { 0, 0, 0, 0 };
#if !KF_S_PARK
        uint32_t top1[STRIP] = { 0, 0, 0, 0 }, bot1[STRIP] = { 0, 0, 0, 0 };
#endif
        int flags1 = 3, flags2 = 3;                      // bit 0: a pixel at most 64 above black, bit 1: less than 256 above (rows r - 1, r - 2)
        int dark_steps = 0;
        issue(j0 - 1, dA0, dA1);
        issue(j0, dB0, dB1);
        if (KF_S_DEPTH == 3) issue(j0 + 1, dC0, dC1);
        auto step = [&](int r, uint32_t (&d0)[4], uint32_t (&d1)[4]) {
            uint32_t p0[8], p1[8];
            unpack8<BPP>(d0, sel, sel, sel, p0);
            unpack8<BPP>(d1, sel1, sel1, sel1, p1);
            if (r + KF_S_DEPTH <= j1) issue(r + KF_S_DEPTH, d0, d1);       // the row this set is needed for next goes out while this one is converted
            uint32_t lo = min(p0[0], p1[0]);
#pragma unroll
            for (int i = 1; i < 8; i++) lo = min(min(lo, p0[i]), p1[i]);
            const bool dark = __any((int)lo <= black);
            dark_steps += dark ? 1 : 0;
            int flags0 = 0;
            if (__any((int)lo <= black + 255)) flags0 = __any((int)lo <= black + 64) ? 3 : 2;
            int ge[STRIP], dr[STRIP], db[STRIP];
            if (!dark) {
#if KF_S_NC == 4
                cell_multi_ev_fast<4, SPREAD>(p0, p1, black, t16, ge, dr, db);
#else
#pragma unroll
                for (int cc = 0; cc < 4; cc += 2) {
                    int g2[2], r2[2], b2[2];
                    cell_multi_ev_fast<2, SPREAD>(p0 + 2 * cc, p1 + 2 * cc, black, t16, g2, r2, b2);
                    ge[cc] = g2[0]; ge[cc + 1] = g2[1]; dr[cc] = r2[0]; dr[cc + 1] = r2[1]; db[cc] = b2[0]; db[cc + 1] = b2[1];
                }
#endif
            } else {
#pragma unroll
                for (int cc = 0; cc < 4; cc += 2) {
                    int g2[2], r2[2], b2[2];
                    cell_multi_ev_dark<2, SPREAD>(p0 + 2 * cc, p1 + 2 * cc, black, t16, g2, r2, b2);
                    ge[cc] = g2[0]; ge[cc + 1] = g2[1]; dr[cc] = r2[0]; dr[cc + 1] = r2[1]; db[cc] = b2[0]; db[cc + 1] = b2[1];
                }
            }
            uint32_t top[STRIP], bot[STRIP];
#pragma unroll
            for (int cc = 0; cc < STRIP; cc++) { top[cc] = p0[2 * cc] | (p0[2 * cc + 1] << 16); bot[cc] = p1[2 * cc] | (p1[2 * cc + 1] << 16); }
#if KF_S_PARK
            {
                uint4 (&slot)[2][64] = mypark[r & 1];
                slot[0][lane] = make_uint4(top[0], top[1], top[2], top[3]);
                slot[1][lane] = make_uint4(bot[0], bot[1], bot[2], bot[3]);
            }
#endif
            if (r - 1 >= j0) {
                // ---- row r - 1: medians of the plus-shaped five, then k_frame's output stage on registers
                const int jr = r - 1, y = 2 * jr, yl = y + 2 * roff;
                const bool smooth_row = y + 2 * (nparts - 1) * seg_rows >= 4 && y < h - 5;     // chroma_smooth.c:25 (scalar: some group's row)
                int er[STRIP] = { 0, 0, 0, 0 }, eb[STRIP] = { 0, 0, 0, 0 };
#if KF_S_PARK
                uint32_t top1[STRIP], bot1[STRIP];
                {
                    const uint4 (&slot)[2][64] = mypark[(r - 1) & 1];
                    const uint4 t4 = slot[0][lane], b4 = slot[1][lane];
                    top1[0] = t4.x; top1[1] = t4.y; top1[2] = t4.z; top1[3] = t4.w;
                    bot1[0] = b4.x; bot1[1] = b4.y; bot1[2] = b4.z; bot1[3] = b4.w;
                }
#endif
                if (smooth_row && METHOD == 2) {
                    const int lr = dpp_prev_i(dr1[3]), lb = dpp_prev_i(db1[3]);                // the cell left of cell 0: the lane before's last
                    const int rr_ = dpp_next_i(dr1[0]), rb = dpp_next_i(db1[0]);               // the cell right of cell 3: the next lane's first
#pragma unroll
                    for (int cc = 0; cc < STRIP; cc++) {
                        const int vr[5] = { dr2[cc], cc ? dr1[cc - 1] : lr, dr1[cc], cc < 3 ? dr1[cc + 1] : rr_, dr[cc] };
                        const int vb[5] = { db2[cc], cc ? db1[cc - 1] : lb, db1[cc], cc < 3 ? db1[cc + 1] : rb, db[cc] };
                        int o[1];
                        mlv_median5(vr, o); er[cc] = wadd(ge1[cc], o[0]);
                        mlv_median5(vb, o); eb[cc] = wadd(ge1[cc], o[0]);
                    }
                }
                if (smooth_row && METHOD == 3) {
                    // columns of three (rows r - 2, r - 1, r) sorted once; the lane before's last column and the next lane's first by DPP
                    auto med9 = [&](const int (&up)[STRIP], const int (&mid)[STRIP], const int (&dn)[STRIP], int (&e)[STRIP]) {
                        int lo[STRIP + 2], mi[STRIP + 2], hi[STRIP + 2];
#pragma unroll
                        for (int cc = 0; cc < STRIP; cc++) {
                            lo[cc + 1] = min(min(up[cc], mid[cc]), dn[cc]);
                            hi[cc + 1] = max(max(up[cc], mid[cc]), dn[cc]);
                            mi[cc + 1] = med3i(up[cc], mid[cc], dn[cc]);
                        }
                        lo[0] = dpp_prev_i(lo[STRIP]); mi[0] = dpp_prev_i(mi[STRIP]); hi[0] = dpp_prev_i(hi[STRIP]);
                        lo[STRIP + 1] = dpp_next_i(lo[1]); mi[STRIP + 1] = dpp_next_i(mi[1]); hi[STRIP + 1] = dpp_next_i(hi[1]);
#pragma unroll
                        for (int cc = 0; cc < STRIP; cc++)
                            e[cc] = wadd(ge1[cc], med3i(max(max(lo[cc], lo[cc + 1]), lo[cc + 2]), med3i(mi[cc], mi[cc + 1], mi[cc + 2]),
                                                         min(min(hi[cc], hi[cc + 1]), hi[cc + 2])));
                    };
                    med9(dr2, dr1, dr, er);
                    med9(db2, db1, db, eb);
                }
                const unsigned long long msmooth = lanes_ge(yl, 4) & lanes_lt(yl, h - 5);       // (a lane mask in a register pair: put_rb moves it to VCC)
                const int fl = flags0 | flags1 | flags2;
                // (the variants of strip_output, chosen by scalars: margins, low pixels, bright rows)
#define KFS_OUT(CLAMP, XM, BRIGHT) strip_output_t<METHOD, true, true, CLAMP, XM, false, BRIGHT, NoSmem, true>(NoSmem(), oa, w, h, black, f, tx0, 0, jr, pl, msmooth, \
                                                                                                         ge1, 0, er, eb, false, top1, bot1)
                if (fl & 1) { if (xm) KFS_OUT(true, true, false); else KFS_OUT(true, false, false); }
                else if (xm) KFS_OUT(false, true, false);
                else if (fl == 0) KFS_OUT(false, false, true);
                else KFS_OUT(false, false, false);
#undef KFS_OUT
                if (writes) {
                    const uint32_t vo = (__umul24((uint32_t)yl, (uint32_t)w) + (uint32_t)(8 * g)) * 2u;     // (rows below the frame: beyond the buffer's range)
                    const mlv_u32x4 vt = { top1[0], top1[1], top1[2], top1[3] }, vb_ = { bot1[0], bot1[1], bot1[2], bot1[3] };
                    mlv_rbs_x4(vt, rs_out, (int)vo, 0, 2);                                     // (2: non-temporal)
                    mlv_rbs_x4(vb_, rs_out, (int)vo, w * 2, 2);
                }
            }
#pragma unroll
            for (int cc = 0; cc < STRIP; cc++) {
                dr2[cc] = dr1[cc]; db2[cc] = db1[cc];
                dr1[cc] = dr[cc]; db1[cc] = db[cc]; ge1[cc] = ge[cc];
#if !KF_S_PARK
                top1[cc] = top[cc]; bot1[cc] = bot[cc];
#endif
            }
            flags2 = flags1; flags1 = flags0;
        };
        for (int r = j0 - 1; r <= j1; r += KF_S_DEPTH) {
            step(r, dA0, dA1);
            if (r + 1 <= j1) step(r + 1, dB0, dB1);
            if (KF_S_DEPTH == 3 && r + 2 <= j1) step(r + 2, dC0, dC1);
        }
        if (lane == 0 && dark_steps) atomicAdd(&a.wl_ctl[0], dark_steps);       // (what the host's choice of kernel for this stream looks at)
    }
    // the last wave out leaves the two counters as it found them (the next launch on this stream starts from zero)
    if (lane == 0) {
        const int nwaves = (int)gridDim.x * 4;
        if (atomicAdd(&tickets[1], 1) == nwaves - 1) {
            tickets[0] = 0; tickets[1] = 0;
            if (a.wl_stat) { __atomic_store_n(a.wl_stat, a.wl_ctl[0], __ATOMIC_RELAXED); __threadfence_system(); }
        }
    }
}

// does the streaming kernel take this launch?  (k_frame.hip: launch_frame_t asks before it sets up its own)
bool frame_s_takes(int method, bool packed, int vec, int num_cu, const FrameArgs &a)
{
    // MLVFS_AMD_KF_S: 0 never, 1 (default) long launches of footage without many pixels at or below black, 2 whenever the kernel can
    // (read at every launch: the tests switch it)
    if ((method != 2 && method != 3) || !packed || (vec != 1 && vec != 2)) return false;
    const char *e = getenv("MLVFS_AMD_KF_S");
    const int policy = e ? atoi(e) : 1;
    if (policy == 0) return false;
    if (a.patch || (a.stripes && !a.coef_pk) || a.black < 0) return false;
    if (!(a.w >= 16 && a.w % 8 == 0 && a.h >= 2 && a.h % 2 == 0)) return false;
    // Long launches only: a task is a column of 60 rows (~50 us of one wave), and a wave needs a handful of them for the chip to end
    // together -- 3584x1320, us per frame at 8 / 25 / 50 / 100 / 200 / 400 frames per launch: k_frame 10.7 / 5.7 / 5.7 / 5.5 / 5.3 / 4.9,
    // this kernel 13.1 / 7.3 / 7.1 / 5.5 / 5.0 / 4.8 (shorter tasks do not help: two rows of warm-up each; profiles/r05/ab_kframe_s.log)
    const long long cols = frame_stream_cols(a.w), segs = (a.h / 2 + KF_S_SEG - 1) / KF_S_SEG;
    const long long waves = (long long)(num_cu > 0 ? num_cu : 256) * KF_S_WGS * 4;
    return policy == 2 || (long long)a.nframes * cols * segs * 2 >= waves * 7;             // >= 3.5 tasks per wave
}

// wave-steps of a launch (what the dark steps it reports are a share of)
long long frame_s_steps(const FrameArgs &a)
{
    const long long cols = frame_stream_cols(a.w);
    const int fold = frame_stream_fold(a.w, (int)cols, (a.h / 2 + KF_S_SEG - 1) / KF_S_SEG);
    return (long long)a.nframes * ((cols - 1) * fold + 1) * (a.h / 2) / fold;
}

void launch_frame_s_kernel(int method, bool spread, int vec, int num_cu, hipStream_t stream, const FrameArgs &a)
{
    const int cols = frame_stream_cols(a.w), rows = a.h / 2;
    static const int env_seg = [] { const char *e = getenv("MLVFS_AMD_KF_S_SEG"); return e ? atoi(e) : 0; }();
    const int seg_rows = std::max(env_seg > 0 ? env_seg : KF_S_SEG, 1), segs = (rows + seg_rows - 1) / seg_rows;
    const int fold = frame_stream_fold(a.w, cols, segs);
    const long long tasks = (long long)a.nframes * (fold > 1 ? (cols - 1) * segs + (segs + fold - 1) / fold : cols * segs);
    int grid = (num_cu > 0 ? num_cu : 256) * KF_S_WGS;   // five workgroups per CU: 20 waves, 80 KiB of LDS (five copies of the table)
    if ((long long)grid * 4 > tasks) grid = (int)((tasks + 3) / 4);
#define KFS_GO(S, V, M) hipLaunchKernelGGL((k_frame_s<S, V, M>), dim3(grid), dim3(256), 0, stream, a, cols, segs, seg_rows, fold, frame_stream_colw())
#define KFS_M(S, V) do { if (method == 3) KFS_GO(S, V, 3); else KFS_GO(S, V, 2); } while (0)
    if (vec == 2) { if (spread) KFS_M(true, 2); else KFS_M(false, 2); }
    else { if (spread) KFS_M(true, 1); else KFS_M(false, 1); }
#undef KFS_M
#undef KFS_GO
}

void preload_k_frame_s() { hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, (const void *)k_frame_s<false, 1, 2>); (void)hipGetLastError(); }

}  // namespace mlv
