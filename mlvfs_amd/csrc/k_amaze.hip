// k_amaze.hip -- the AMaZE demosaic of the dual-ISO path (mlvfs/amaze_demosaic_RT.c:113-1487,
// the SSE2 variant that x86-64 builds of MLVFS run) for gfx950.
//
// One workgroup of 1024 threads per 160x160 tile (128x128 output pixels + 16 px apron), the same tiling as the
// reference because several passes are sequential INSIDE a tile and therefore depend on where the tile starts:
//   * the vertical colour differences are refined in place top-down (a row reads the refined row two above);
//   * the horizontal ones see the refined left neighbour in lanes 0,1 of each 4-lane group and the unrefined
//     one in lanes 2,3 -- lanes 2,3 only depend on unrefined data, so the row is computed in parallel with
//     lanes 0,1 re-deriving their left neighbour;
//   * the hvwt and pmwt "ask the neighbours" passes read the already updated row above: one LDS-resident row
//     per barrier;
//   * the Nyquist majority vote runs in raster order: executed as anti-diagonal wavefronts (2*row + col) in
//     LDS, which visit every dependency in raster order; skipped when no site of the tile is flagged.
// Every other pass is a parallel sweep over the tile.  The per-tile planes (2 MB) live in an HBM scratch block
// that is zeroed once at allocation: like the reference's calloc'ed block, the rows/columns that no pass ever
// writes (above row 4, left of column 4) read as zero.
//
// Arithmetic is IEEE binary32 in the reference's operation order (built with -ffp-contract=off; HIP's float
// division is correctly rounded), the vector loops' overrun past the scalar bounds included, so the three
// planes come out bit-identical to the reference's.  No MFMA: this is stencil/select work.
#include "amaze_math.h"
#include <cstdlib>
#include <map>

namespace mlv {

using namespace amz;

namespace {

struct Tile {                       // pointers into this tile's scratch block
    float *cfa, *green, *delsq, *dw0, *dw1, *vcd, *hcd, *vcdalt, *hcdalt, *cdsq, *dgv, *dgh, *hcd2;
    float *hvwt, *dgrb0, *dgrb1, *delp, *delm, *rbint, *curv_h, *curv_v, *sqm, *sqp, *pmwt, *rbm, *rbp;
};
static_assert(AMAZE_TILE_FLOATS == 13 * TT + 13 * HALF, "scratch layout");

}  // namespace

// Stale planes.  The reference keeps ONE block of tile planes for the whole image and never clears it between tiles
// (amaze_demosaic_RT.c:244).  A tile whose bottom/right apron does not reach row/column 160 (the image ends less than
// 144 rows/columns after the tile's origin) reads, past what it wrote itself, what earlier tiles left in the block:
// cfa columns of the tile before it, hcd/hcdalt/vcd/vcdalt feeding the Nyquist test, pmwt feeding the "ask the
// neighbours" sweep -- and that reaches its output pixels.  Tiles whose aprons are complete only ever read zeros
// there and are independent of the block's history.  The launch plan (amaze_launch) reproduces the history that matters:
//   * the incomplete tiles at the right end of a tile row (the last one, and the one before it when the last is
//     narrower than 32) are processed by the workgroup of the last complete tile, one after the other in its block
//     (`chain_len`);
//   * an incomplete tile ROW (the last, and the one before it when the last is shorter than 32) runs as its own launch
//     whose workgroups first copy the block in which the previous row ended (`copy_from`);
//   * when a bottom apron ran off its plane into pmwt (see the loader) the rows after it carry that garbage from tile
//     to tile, and small images (fewer than 3 tile columns) have no complete tile to start a chain from: both are
//     walked by ONE workgroup in the reference's order (`rows_per_wg`, chain_len = tiles_x).
// Not reproduced: what equal-size neighbours leave INSIDE each other's written range; the only reads of that kind
// are the overrun lanes of the pmwt sweep, which are 0 and provably stay 0 unless the garbage case above applies.
#ifdef AMAZE_DIAG
__device__ unsigned long long g_amaze_stamps[16];
#define AMZ_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && tk == 0) g_amaze_stamps[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AMZ_STAMP(k) do { } while (0)
#endif
__global__ __launch_bounds__(1024) void k_amaze(const float *__restrict__ raw, int w, int h, float *__restrict__ red,
                                                 float *__restrict__ green_out, float *__restrict__ blue, float *__restrict__ scratch,
                                                 int tiles_x, int row0, int wgs_per_row, int chain_len, int rows_per_wg, int copy_from,
                                                 size_t plane_stride, size_t scratch_stride, const int *__restrict__ h_of, int h_stride,
                                                 int nfx, int nfy, int dead_rows, const int *__restrict__ r2e, int ev_black, int *__restrict__ gray)
{
    __shared__ unsigned char s_nyq[HALF];
    __shared__ float s_w[HALF];
    const int tid = threadIdx.x, nt = blockDim.x;
    {   // frame of a batch (blockIdx.y): its planes, its scratch blocks; only the frames of THIS launch plan's height take part
        const size_t f = blockIdx.y;
        if (h_of && h_of[f * (size_t)h_stride] != h) return;
        raw += f * plane_stride; red += f * plane_stride; green_out += f * plane_stride; blue += f * plane_stride;
        if (r2e) gray += f * plane_stride;
        scratch += f * scratch_stride;
    }
    const int ty0 = row0 + (int)(blockIdx.x / wgs_per_row), tx0 = (int)(blockIdx.x % wgs_per_row);
    const int ntiles = (tx0 == wgs_per_row - 1) ? chain_len : 1;
    if (tx0 < nfx && ty0 < nfy && ntiles == 1 && rows_per_wg == 1 && copy_from < 0) return;       // a complete tile: k_amaze_rows has it
    if (tx0 == wgs_per_row - 1 && ty0 < dead_rows && rows_per_wg == 1 && copy_from < 0) return;    // a chain without output whose head k_amaze_rows has (amaze_rows_extra)
    float *const block = scratch + (size_t)(ty0 * tiles_x + tx0) * AMAZE_TILE_FLOATS;
    if (copy_from >= 0) {
        const float4 *src = (const float4 *)(scratch + (size_t)copy_from * AMAZE_TILE_FLOATS);
        float4 *dst = (float4 *)block;
        for (int n = tid; n < AMAZE_TILE_FLOATS / 4; n += nt) dst[n] = src[n];
        __syncthreads();
    }
    for (int tk = 0; tk < rows_per_wg * ntiles; tk++) {
    const int ty = ty0 + tk / ntiles, tx = tx0 + tk % ntiles;
    const int top = -16 + ty * (T - 32), left = -16 + tx * (T - 32);
    const int bottom = min(top + T, h + 16), right = min(left + T, w + 16);
    const int rr1 = bottom - top, cc1 = right - left;
    const int rrmin = top < 0 ? 16 : 0, ccmin = left < 0 ? 16 : 0;
    const int rrmax = bottom > h ? h - top : rr1, ccmax = right > w ? w - left : cc1;

    Tile t;
    {
        float *p = block;
        float **full[13] = { &t.cfa, &t.green, &t.delsq, &t.dw0, &t.dw1, &t.vcd, &t.hcd, &t.vcdalt, &t.hcdalt, &t.cdsq, &t.dgv, &t.dgh, &t.hcd2 };
        for (int k = 0; k < 13; k++) { *full[k] = p; p += TT; }
        float **half[13] = { &t.hvwt, &t.dgrb0, &t.dgrb1, &t.delp, &t.delm, &t.rbint, &t.curv_h, &t.curv_v, &t.sqm, &t.sqp, &t.pmwt, &t.rbm, &t.rbp };
        for (int k = 0; k < 13; k++) { *half[k] = p; p += HALF; }
    }
    const float *c = t.cfa;
    auto RAW = [&](int y, int x) -> float { return x < w ? raw[(size_t)y * w + x] : 0.0f; };   // the reference's rows are zero padded

    AMZ_STAMP(0);
    // ---- tile load + mirrored apron (:361-469; w % 4 == 0, so the vector groups of the loader never straddle a region).
    // The fills address the tile by FLAT index and run in the reference's order, because they overlap: a right-edge
    // fill that starts less than 16 columns before the end of a tile row (ccmax + 16 > 160) runs on into the first
    // columns of the next row, and a bottom fill that starts less than 16 rows before row 160 runs off the plane into
    // whatever follows it in the reference's block (cfa -> pmwt, rgbgreen -> delhvsqsum, 64 bytes further on).
    {
        auto put = [&](int idx, float v, bool into_green) {
            v = v / 65535.0f;
            if (idx < TT) { t.cfa[idx] = v; if (into_green) t.green[idx] = v; }
            else if (idx >= TT + 16) { t.pmwt[idx - TT - 16] = v; if (into_green) t.delsq[idx - TT - 16] = v; }
        };
        const int rin = rrmax - rrmin, cin = ccmax - ccmin;
        const bool has_top = rrmin > 0, has_bot = rrmax < rr1, has_left = ccmin > 0, has_right = ccmax < cc1;
        // phase 1: interior, top, bottom, left (disjoint)
        for (int n = tid; n < rin * cin; n += nt) {
            const int rr = rrmin + n / cin, cc = ccmin + n % cin;
            put(rr * T + cc, RAW(rr + top, cc + left), true);
        }
        if (has_top)
            for (int n = tid; n < 16 * cin; n += nt) {
                const int rr = n / cin, cc = ccmin + n % cin;
                put(rr * T + cc, RAW(32 - rr + top, cc + left), fc(rr, cc) == 1);
            }
        if (has_bot)
            for (int n = tid; n < 16 * cin; n += nt) {
                const int r2 = n / cin, cc = ccmin + n % cin;
                put((rrmax + r2) * T + cc, RAW(h - r2 - 2, left + cc), true);
            }
        if (has_left)
            for (int n = tid; n < rin * 16; n += nt) {
                const int rr = rrmin + n / 16, cc = n % 16;
                put(rr * T + cc, RAW(rr + top, 32 - cc + left), fc(rr, cc) == 1);
            }
        __syncthreads();
        // phase 2: right, top-left, bottom-right (the right fill may overwrite phase-1 pixels of the next row)
        if (has_right)
            for (int n = tid; n < rin * 16; n += nt) {
                const int rr = rrmin + n / 16, c2 = n % 16;
                put(rr * T + ccmax + c2, RAW(top + rr, w - c2 - 2), fc(rr, c2) == 1);
            }
        if (has_top && has_left)
            for (int n = tid; n < 256; n += nt) {
                const int rr = n / 16, cc = n % 16;
                put(rr * T + cc, RAW(32 - rr, 32 - (cc & ~3) + (cc & 3)), true);          // 4 ascending pixels per group (:423-430)
            }
        if (has_bot && has_right)
            for (int n = tid; n < 256; n += nt) {
                const int r2 = n / 16, c2 = n % 16;
                put((rrmax + r2) * T + ccmax + c2, RAW(h - r2 - 2, w - (c2 & ~3) - 2 + (c2 & 3)), true);
            }
        __syncthreads();
        // phase 3: top-right, bottom-left
        if (has_top && has_right)
            for (int n = tid; n < 256; n += nt) {
                const int rr = n / 16, c2 = n % 16;
                put(rr * T + ccmax + c2, RAW(32 - rr, w - c2 - 2), fc(rr, c2) == 1);
            }
        if (has_bot && has_left)
            for (int n = tid; n < 256; n += nt) {
                const int r2 = n / 16, cc = n % 16;
                put((rrmax + r2) * T + cc, RAW(h - r2 - 2, 32 - cc), fc(r2, cc) == 1);
            }
    }
    for (int n = tid; n < HALF; n += nt) { s_nyq[n] = 0; t.rbint[n] = 0.0f; }
    __syncthreads();

    AMZ_STAMP(1);
    // ---- gradients (:537-613)
    {
        const int cw = (cc1 + 3) & ~3, rows = rr1 - 4;
        for (int n = tid; n < rows * cw; n += nt) {
            const int i = (2 + n / cw) * T + n % cw;
            const float dh = fabsf(c[i + 1] - c[i - 1]), dv = fabsf(c[i + V1] - c[i - V1]);
            t.dw1[i] = EPS + fabsf(c[i + 2] - c[i]) + fabsf(c[i] - c[i - 2]) + dh;
            t.dw0[i] = EPS + fabsf(c[i + V2] - c[i]) + fabsf(c[i] - c[i - V2]) + dv;
            t.delsq[i] = dh * dh + dv * dv;
        }
        const int lanes = 4 * iters(6, cc1 - 6, 8), rows2 = rr1 - 12;
        for (int n = tid; n < rows2 * lanes; n += nt) {
            const int rr = 6 + n / lanes, m = n % lanes;
            const int g = (rr & 1) ? 0 : 1;
            const int i = rr * T + 6 + 2 * m, ig = i + g, ic = i + 1 - g, j = (rr * T + 6) / 2 + m;
            t.delp[j] = fabsf(c[ic + P1] - c[ic - P1]);
            t.delm[j] = fabsf(c[ic + M1] - c[ic - M1]);
            t.sqp[j] = sq(c[ig] - c[ig - P1]) + sq(c[ig] - c[ig + P1]);
            t.sqm[j] = sq(c[ig] - c[ig - M1]) + sq(c[ig] - c[ig + M1]);
        }
    }
    __syncthreads();

    AMZ_STAMP(2);
    // ---- directional colour differences (:622-675)
    {
        const float *d0 = t.dw0, *d1 = t.dw1;
        const int lanes = 4 * iters(4, cc1 - 7, 4), rows = rr1 - 8;
        for (int n = tid; n < rows * lanes; n += nt) {
            const int rr = 4 + n / lanes, cc = 4 + n % lanes, i = rr * T + cc;
            const float sgn = ((rr + cc) & 1) ? -1.0f : 1.0f;
            const float cru = c[i - V1] * (d0[i - V2] + d0[i]) / (d0[i - V2] * (EPS + c[i]) + d0[i] * (EPS + c[i - V2]));
            const float crd = c[i + V1] * (d0[i + V2] + d0[i]) / (d0[i + V2] * (EPS + c[i]) + d0[i] * (EPS + c[i + V2]));
            const float crl = c[i - 1] * (d1[i - 2] + d1[i]) / (d1[i - 2] * (EPS + c[i]) + d1[i] * (EPS + c[i - 2]));
            const float crr = c[i + 1] * (d1[i + 2] + d1[i]) / (d1[i + 2] * (EPS + c[i]) + d1[i] * (EPS + c[i + 2]));
            const float guha = c[i - V1] + 0.5f * (c[i] - c[i - V2]), gdha = c[i + V1] + 0.5f * (c[i] - c[i + V2]);
            const float glha = c[i - 1] + 0.5f * (c[i] - c[i - 2]), grha = c[i + 1] + 0.5f * (c[i] - c[i + 2]);
            float guar = fabsf(1.0f - cru) < ARTHRESH ? c[i] * cru : guha, gdar = fabsf(1.0f - crd) < ARTHRESH ? c[i] * crd : gdha;
            float glar = fabsf(1.0f - crl) < ARTHRESH ? c[i] * crl : glha, grar = fabsf(1.0f - crr) < ARTHRESH ? c[i] * crr : grha;
            const float hwt = d1[i - 1] / (d1[i - 1] + d1[i + 1]), vwt = d0[i - V1] / (d0[i + V1] + d0[i - V1]);
            const float ginth = hwt * grha + (1.0f - hwt) * glha, gintv = vwt * gdha + (1.0f - vwt) * guha;
            const float halt = sgn * (ginth - c[i]), valt = sgn * (gintv - c[i]);
            t.hcdalt[i] = halt;
            t.vcdalt[i] = valt;
            const bool clip = c[i] > CLIP_PT8 || gintv > CLIP_PT8 || ginth > CLIP_PT8;
            if (clip) { guar = guha; gdar = gdha; glar = glha; grar = grha; }
            t.vcd[i] = clip ? valt : sgn * ((vwt * gdar + (1.0f - vwt) * guar) - c[i]);
            t.hcd[i] = clip ? halt : sgn * ((hwt * grar + (1.0f - hwt) * glar) - c[i]);
            t.dgv[i] = fminv(sq(guha - gdha), sq(guar - gdar));
            t.dgh[i] = fminv(sq(glha - grha), sq(glar - grar));
        }
    }
    __syncthreads();

    AMZ_STAMP(3);
    // ---- refinement of the colour differences (:766-801)
    {
        const int lanes = 4 * iters(4, cc1 - 4, 4), rows = rr1 - 8;
        // horizontal: lanes 2,3 of a group read unrefined neighbours only; lanes 0,1 need the refined value of
        // the previous group's lanes 2,3, which they derive themselves.  Output goes to hcd2.
        // The reference refines hcd in place, so what it reads past the columns this tile wrote ([4, 4 + lanes)) are the
        // REFINED leftovers of an earlier tile: those live in hcd2 here.
        for (int n = tid; n < rows * lanes; n += nt) {
            const int rr = 4 + n / lanes, cc = 4 + n % lanes, i = rr * T + cc;
            const float sgn = ((rr + cc) & 1) ? -1.0f : 1.0f;
            auto H = [&](int q) { const int qc = q - rr * T; return (qc >= 4 && qc < 4 + lanes) ? t.hcd[q] : t.hcd2[q]; };
            auto refined = [&](int q, float leftval) {
                const float hv = var3(leftval, H(q), H(q + 2)), hav = var3(t.hcdalt[q - 2], t.hcdalt[q], t.hcdalt[q + 2]);
                return bound_difference(hav < hv ? t.hcdalt[q] : H(q), sgn, c[q], c[q - 1], c[q + 1]);   // sgn(q-2) == sgn(q)
            };
            float leftval = H(i - 2);
            if (((cc - 4) & 3) < 2 && cc >= 8) leftval = refined(i - 2, H(i - 4));
            t.hcd2[i] = refined(i, leftval);
        }
        __syncthreads();
        // vertical: one thread per (column, row parity) walks down; also the squared difference of the two
        // The walk carries ONE value from row to row (the refined difference two rows up); everything else it reads is unrefined
        // data of rows it has not reached.  The operands of the rows ahead are in flight while a row is computed and stored: a
        // ring of VCD_AHEAD rows (the in-place store made the compiler wait for each row's loads in turn: 1 200 cycles per row, a
        // quarter of the tile's time).
        constexpr int VCD_AHEAD = 4;
        for (int n = tid; n < 2 * lanes; n += nt) {
            const int cc = 4 + n / 2;
            const int r0 = 4 + (n & 1), r_end = rr1 - 4;
            if (r0 >= r_end) continue;
            const int i0 = r0 * T + cc;
            const int last = i0 + ((r_end - 1 - r0) / 2) * V2;           // index of the walk's last row (loads beyond it are clamped to it)
            float up = t.vcd[i0 - V2];                       // rows 2,3 are never written: zero like the reference's block
            // per row: vcd two rows below, vcdalt two rows below, cfa above / at / below, hcd2
            float q_v[VCD_AHEAD], q_a[VCD_AHEAD], q_cm[VCD_AHEAD], q_c0[VCD_AHEAD], q_cp[VCD_AHEAD], q_h[VCD_AHEAD];
#pragma unroll
            for (int d = 0; d < VCD_AHEAD; d++) {
                const int j = min(i0 + d * V2, last);
                q_v[d] = t.vcd[j + V2]; q_a[d] = t.vcdalt[j + V2]; q_cm[d] = c[j - V1]; q_c0[d] = c[j]; q_cp[d] = c[j + V1]; q_h[d] = t.hcd2[j];
            }
            float v0 = t.vcd[i0], a_1 = t.vcdalt[i0 - V2], a0 = t.vcdalt[i0];
            int i = i0;
            for (int rr = r0; rr < r_end; rr += 2 * VCD_AHEAD) {
#pragma unroll
                for (int d = 0; d < VCD_AHEAD; d++) {
                    if (rr + 2 * d < r_end) {
                        const float sgn = ((rr + cc) & 1) ? -1.0f : 1.0f;            // (rows of one walk have one parity)
                        const float v1 = q_v[d], a1 = q_a[d], cm = q_cm[d], c0 = q_c0[d], cp = q_cp[d], h2 = q_h[d];
                        // refill this slot with the row VCD_AHEAD steps ahead
                        const int j = min(i + VCD_AHEAD * V2, last);
                        q_v[d] = t.vcd[j + V2]; q_a[d] = t.vcdalt[j + V2]; q_cm[d] = c[j - V1]; q_c0[d] = c[j]; q_cp[d] = c[j + V1]; q_h[d] = t.hcd2[j];
                        const float vv = var3(up, v0, v1), vav = var3(a_1, a0, a1);
                        const float v = bound_difference(vav < vv ? a0 : v0, sgn, c0, cm, cp);
                        t.vcd[i] = v;
                        t.cdsq[i] = sq(v - h2);
                        up = v;
                        v0 = v1; a_1 = a0; a0 = a1;
                        i += V2;
                    }
                }
            }
        }
    }
    __syncthreads();
    const float *hc = t.hcd2, *vc = t.vcd;

    AMZ_STAMP(4);
    // ---- horizontal/vertical weight (:881-925) and the Nyquist texture test (:969-996)
    int flagged = 0;
    {
        const int rows = rr1 - 12;
        const int lanes_max = 4 * iters(6, cc1 - 6, 8);
        for (int n = tid; n < rows * lanes_max; n += nt) {
            const int rr = 6 + n / lanes_max, m = n % lanes_max, cc0 = 6 + (rr & 1);
            if (m >= 4 * iters(cc0, cc1 - 6, 8)) continue;
            const int i = rr * T + cc0 + 2 * m;
            const float uave = vc[i] + vc[i - V1] + vc[i - V2] + vc[i - V3], dave = vc[i] + vc[i + V1] + vc[i + V2] + vc[i + V3];
            float vu = sq(vc[i] - uave) + sq(vc[i - V1] - uave) + sq(vc[i - V2] - uave) + sq(vc[i - V3] - uave);
            float vd = sq(vc[i] - dave) + sq(vc[i + V1] - dave) + sq(vc[i + V2] - dave) + sq(vc[i + V3] - dave);
            const float hwt = t.dw1[i - 1] / (t.dw1[i - 1] + t.dw1[i + 1]), vwt = t.dw0[i - V1] / (t.dw0[i + V1] + t.dw0[i - V1]);
            const float lave = hc[i] + hc[i - 1] + hc[i - 2] + hc[i - 3], rave = hc[i] + hc[i + 1] + hc[i + 2] + hc[i + 3];
            float hl = sq(hc[i] - lave) + sq(hc[i - 1] - lave) + sq(hc[i - 2] - lave) + sq(hc[i - 3] - lave);
            float hr = sq(hc[i] - rave) + sq(hc[i + 1] - rave) + sq(hc[i + 2] - rave) + sq(hc[i + 3] - rave);
            const float vcdvar = EPSSQ + vwt * vd + (1.0f - vwt) * vu, hcdvar = EPSSQ + hwt * hr + (1.0f - hwt) * hl;
            vu = t.dgv[i] + t.dgv[i - V1] + t.dgv[i - V2];
            vd = t.dgv[i] + t.dgv[i + V1] + t.dgv[i + V2];
            hl = t.dgh[i] + t.dgh[i - 1] + t.dgh[i - 2];
            hr = t.dgh[i] + t.dgh[i + 1] + t.dgh[i + 2];
            const float vcdvar1 = EPSSQ + vwt * vd + (1.0f - vwt) * vu, hcdvar1 = EPSSQ + hwt * hr + (1.0f - hwt) * hl;
            const float varwt = hcdvar / (vcdvar + hcdvar), diffwt = hcdvar1 / (vcdvar1 + hcdvar1);
            const bool agree = (0.5f - varwt) * (0.5f - diffwt) > 0.0f && fabsf(0.5f - diffwt) < fabsf(0.5f - varwt);
            t.hvwt[i / 2] = agree ? varwt : diffwt;
        }
        const float G_ODD[4] = { 0.14659727707323927f, 0.103592713382435f, 0.0732036125103057f, 0.0365543548389495f };
        const float G_GRAD[6] = { 0.07384411893421103f, 0.06207511968171489f, 0.0521818194747806f,
                                  0.03687419286733595f, 0.03099732204057846f, 0.018413194161458882f };
        const float *q = t.cdsq, *d = t.delsq;
        const int sites_max = iters(6, cc1 - 6, 2);
        for (int n = tid; n < rows * sites_max; n += nt) {
            const int rr = 6 + n / sites_max, cc = 6 + (rr & 1) + 2 * (n % sites_max);
            if (cc >= cc1 - 6) continue;
            const int i = rr * T + cc;
            float test = (G_ODD[0] * q[i] + G_ODD[1] * (q[i - M1] + q[i + P1] + q[i - P1] + q[i + M1]) +
                          G_ODD[2] * (q[i - V2] + q[i - 2] + q[i + 2] + q[i + V2]) + G_ODD[3] * (q[i - M2] + q[i + P2] + q[i - P2] + q[i + M2]));
            test -= NYQTHRESH * (G_GRAD[0] * d[i] + G_GRAD[1] * (d[i - V1] + d[i + 1] + d[i - 1] + d[i + V1]) +
                                 G_GRAD[2] * (d[i - M1] + d[i + P1] + d[i - P1] + d[i + M1]) + G_GRAD[3] * (d[i - V2] + d[i - 2] + d[i + 2] + d[i + V2]) +
                                 G_GRAD[4] * (d[i - 2 * T - 1] + d[i - 2 * T + 1] + d[i - T - 2] + d[i - T + 2] + d[i + T - 2] + d[i + T + 2] +
                                              d[i + 2 * T - 1] + d[i + 2 * T + 1]) +
                                 G_GRAD[5] * (d[i - M2] + d[i + P2] + d[i - P2] + d[i + M2]));
            if (test > 0) { s_nyq[i / 2] = 1; flagged = 1; }
        }
    }
    flagged = __syncthreads_or(flagged);

    if (flagged) {
        AMZ_STAMP(5);
        // ---- majority vote in raster order (:998-1010) as wavefronts of constant 2*row + col
        const int tau_lo = 2 * 8 + 8, tau_hi = 2 * (rr1 - 9) + (cc1 - 9);
        for (int tau = tau_lo; tau <= tau_hi; tau++) {
            const int rr = 8 + ((tau ^ 8) & 1) + 2 * tid;       // rows with the parity of tau
            if (rr < rr1 - 8) {
                const int cc = tau - 2 * rr;
                if (cc >= 8 + (rr & 1) && cc < cc1 - 8) {
                    const int i = rr * T + cc;
                    const unsigned nsum = s_nyq[(i - V2) / 2] + s_nyq[(i - M1) / 2] + s_nyq[(i + P1) / 2] + s_nyq[(i - 2) / 2] + s_nyq[i / 2] +
                                          s_nyq[(i + 2) / 2] + s_nyq[(i - P1) / 2] + s_nyq[(i + M1) / 2] + s_nyq[(i + V2) / 2];
                    if (nsum > 4) s_nyq[i / 2] = 1;
                    if (nsum < 4) s_nyq[i / 2] = 0;
                }
            }
            __syncthreads();
        }
        AMZ_STAMP(6);
        // ---- area interpolation in Nyquist regions (:1016-1044)
        const int rows = rr1 - 16, sites_max = iters(8, cc1 - 8, 2);
        for (int n = tid; n < rows * sites_max; n += nt) {
            const int rr = 8 + n / sites_max, cc = 8 + (rr & 1) + 2 * (n % sites_max);
            if (cc >= cc1 - 8) continue;
            const int i = rr * T + cc;
            if (!s_nyq[i / 2]) continue;
            float sumh = 0, sumv = 0, sumsqh = 0, sumsqv = 0, area = 0;
            for (int a = -6; a < 7; a += 2)
                for (int b = -6; b < 7; b += 2) {
                    const int j = (rr + a) * T + cc + b;
                    if (!s_nyq[j / 2]) continue;
                    sumh += c[j] - half_exp(c[j - 1] + c[j + 1]);
                    sumv += c[j] - half_exp(c[j - V1] + c[j + V1]);
                    sumsqh += half_exp(sq(c[j] - c[j - 1]) + sq(c[j] - c[j + 1]));
                    sumsqv += half_exp(sq(c[j] - c[j - V1]) + sq(c[j] - c[j + V1]));
                    area += 1;
                }
            const float hvar = EPSSQ + fabsf(area * sumsqh - sumh * sumh), vvar = EPSSQ + fabsf(area * sumsqv - sumv * sumv);
            t.hvwt[i / 2] = hvar / (vvar + hvar);
        }
        __syncthreads();
    }

    AMZ_STAMP(7);
    // ---- G at R/B sites (:1046-1073): the weight update reads the updated row above -> one row per barrier in LDS
    for (int n = tid; n < HALF; n += nt) s_w[n] = t.hvwt[n];
    __syncthreads();
    for (int rr = 8; rr < rr1 - 8; rr++) {
        const int cc = 8 + (rr & 1) + 2 * tid;
        if (cc < cc1 - 8) {
            const int i = rr * T + cc, j = i / 2;
            const float alt = quarter_exp(s_w[(i - M1) / 2] + s_w[(i + P1) / 2] + s_w[(i - P1) / 2] + s_w[(i + M1) / 2]);
            if (fabsf(0.5f - s_w[j]) < fabsf(0.5f - alt)) s_w[j] = alt;
        }
        __syncthreads();
    }
    for (int n = tid; n < HALF; n += nt) t.hvwt[n] = s_w[n];
    {
        float *g = t.green;
        const int rows = rr1 - 16, sites_max = iters(8, cc1 - 8, 2);
        for (int n = tid; n < rows * sites_max; n += nt) {
            const int rr = 8 + n / sites_max, cc = 8 + (rr & 1) + 2 * (n % sites_max);
            if (cc >= cc1 - 8) continue;
            const int i = rr * T + cc, j = i / 2;
            const float hw = s_w[j];
            const float dg = hc[i] * (1.0f - hw) + vc[i] * hw;
            t.dgrb0[j] = dg;
            const float gi = c[i] + dg;
            g[i] = gi;
            if (s_nyq[j]) {
                t.curv_h[j] = sq(gi - half_exp(g[i - 1] + g[i + 1]));
                t.curv_v[j] = sq(gi - half_exp(g[i - V1] + g[i + V1]));
            } else {
                t.curv_h[j] = 0.0f;
                t.curv_v[j] = 0.0f;
            }
        }
        __syncthreads();
        if (flagged) {                                         // Nyquist refinement (:1081-1101)
            const float G_QUINC[4] = { 0.169917f, 0.108947f, 0.069855f, 0.0287182f };
            for (int n = tid; n < rows * sites_max; n += nt) {
                const int rr = 8 + n / sites_max, cc = 8 + (rr & 1) + 2 * (n % sites_max);
                if (cc >= cc1 - 8) continue;
                const int i = rr * T + cc, j = i / 2;
                if (!s_nyq[j]) continue;
                auto ring = [&](const float *a) {
                    return G_QUINC[0] * a[j] + G_QUINC[1] * (a[(i - M1) / 2] + a[(i + P1) / 2] + a[(i - P1) / 2] + a[(i + M1) / 2]) +
                           G_QUINC[2] * (a[(i - V2) / 2] + a[(i - 2) / 2] + a[(i + 2) / 2] + a[(i + V2) / 2]) +
                           G_QUINC[3] * (a[(i - M2) / 2] + a[(i + P2) / 2] + a[(i - P2) / 2] + a[(i + M2) / 2]);
                };
                const float gvarh = EPSSQ + ring(t.curv_h), gvarv = EPSSQ + ring(t.curv_v);
                const float dg = (hc[i] * gvarv + vc[i] * gvarh) / (gvarv + gvarh);
                t.dgrb0[j] = dg;
                g[i] = c[i] + dg;
            }
            __syncthreads();
        }
    }

    AMZ_STAMP(8);
    // ---- diagonal interpolation (:1112-1276)
    {
        const float G_EVEN[2] = { 0.13719494435797422f, 0.05640252782101291f };
        const int rows = rr1 - 16, lanes_max = 4 * iters(8, cc1 - 8, 8);
        for (int n = tid; n < rows * lanes_max; n += nt) {
            const int rr = 8 + n / lanes_max, m = n % lanes_max, cc0 = 8 + (rr & 1);
            if (m >= 4 * iters(cc0, cc1 - 8, 8)) continue;
            const int b0 = rr * T + cc0, i = b0 + 2 * m, j = b0 / 2 + m;
            const float se = diag_estimate(c[i], c[i + M1], c[i + M2]), nw = diag_estimate(c[i], c[i - M1], c[i - M2]);
            const float base_m = EPS + t.delm[j];
            const float wse = base_m + t.delm[(b0 + M1) / 2 + m] + t.delm[(b0 + M2) / 2 + m];
            const float wnw = base_m + t.delm[(b0 - M1) / 2 + m] + t.delm[(b0 - M2) / 2 + m];
            t.rbm[j] = diag_bound((wse * nw + wnw * se) / (wse + wnw), c[i], c[i - M1], c[i + M1]);
            const float ne = diag_estimate(c[i], c[i + P1], c[i + P2]), sw = diag_estimate(c[i], c[i - P1], c[i - P2]);
            const float base_p = EPS + t.delp[j];
            const float wne = base_p + t.delp[(b0 + P1) / 2 + m] + t.delp[(b0 + P2) / 2 + m];
            const float wsw = base_p + t.delp[(b0 - P1) / 2 + m] + t.delp[(b0 - P2) / 2 + m];
            t.rbp[j] = diag_bound((wne * sw + wsw * ne) / (wne + wsw), c[i], c[i - P1], c[i + P1]);
            auto even_ring = [&](const float *a) {
                return EPSSQ + (G_EVEN[0] * (a[(b0 - V1) / 2 + m] + a[(b0 - 1) / 2 + m] + a[(b0 + 1) / 2 + m] + a[(b0 + V1) / 2 + m]) +
                                G_EVEN[1] * (a[(b0 - V2 - 1) / 2 + m] + a[(b0 - V2 + 1) / 2 + m] + a[(b0 - 2 - V1) / 2 + m] + a[(b0 + 2 - V1) / 2 + m] +
                                             a[(b0 - 2 + V1) / 2 + m] + a[(b0 + 2 + V1) / 2 + m] + a[(b0 + V2 - 1) / 2 + m] + a[(b0 + V2 + 1) / 2 + m]));
            };
            const float varm = even_ring(t.sqm);
            t.pmwt[j] = varm / (even_ring(t.sqp) + varm);
        }
        __syncthreads();
        for (int n = tid; n < HALF; n += nt) s_w[n] = t.pmwt[n];
        __syncthreads();
        for (int rr = 10; rr < rr1 - 10; rr++) {               // reads the updated row above (:1266-1276)
            const int cc0 = 10 + (rr & 1), lanes = 4 * iters(cc0, cc1 - 10, 8);
            if (tid < lanes) {
                const int b0 = rr * T + cc0, j = b0 / 2 + tid;
                const float alt = 0.25f * (s_w[(b0 - M1) / 2 + tid] + s_w[(b0 + P1) / 2 + tid] + s_w[(b0 - P1) / 2 + tid] + s_w[(b0 + M1) / 2 + tid]);
                const float cur = s_w[j];
                s_w[j] = fabsf(0.5f - cur) < fabsf(0.5f - alt) ? alt : cur;
            }
            __syncthreads();
        }
        // rbint from the row's final weight (a site's weight is written once, in its own row's step): all rows at once, instead of
        // three global loads inside every step of the walk above (that was a quarter of the tile's time)
        {
            const int rows_w = rr1 - 20, lanes_w = 4 * iters(10, cc1 - 10, 8);
            for (int n = tid; n < rows_w * lanes_w; n += nt) {
                const int rr = 10 + n / lanes_w, m = n % lanes_w, cc0 = 10 + (rr & 1);
                if (m >= 4 * iters(cc0, cc1 - 10, 8)) continue;
                const int b0 = rr * T + cc0, j = b0 / 2 + m;
                const float nw = s_w[j];
                t.rbint[j] = 0.5f * (c[b0 + 2 * m] + t.rbm[j] * (1.0f - nw) + t.rbp[j] * nw);
            }
        }
        for (int n = tid; n < HALF; n += nt) t.pmwt[n] = s_w[n];
        // (s_w keeps pmwt for the next sweep; rbint was written through to HBM: make it visible)
        __syncthreads();
        const int rows3 = rr1 - 24, sites_max = iters(12, cc1 - 12, 2);
        for (int n = tid; n < rows3 * sites_max; n += nt) {
            const int rr = 12 + n / sites_max, cc = 12 + (rr & 1) + 2 * (n % sites_max);
            if (cc >= cc1 - 12) continue;
            const int i = rr * T + cc, j = i / 2;
            const float hw = t.hvwt[j];
            if (fabsf(0.5f - s_w[j]) < fabsf(0.5f - hw)) continue;
            // sic: the half-width rbint plane is offset by a FULL row in the reference (indx1 - v1), :1289-1290
            const float rb = t.rbint[j], rbu = t.rbint[j - V1], rbd = t.rbint[j + V1], rbl = t.rbint[j - 1], rbr = t.rbint[j + 1];
            // the reference divides in double (a 2.0 literal) and stores a float: for float operands that IS the correctly rounded
            // float quotient (53 >= 2 * 24 + 2 bits: the second rounding is innocuous), so binary32 division gives the same bits
            const float cru = (c[i - V1] * 2.0f) / (EPS + rb + rbu), crd = (c[i + V1] * 2.0f) / (EPS + rb + rbd);
            const float crl = (c[i - 1] * 2.0f) / (EPS + rb + rbl), crr = (c[i + 1] * 2.0f) / (EPS + rb + rbr);
            const float gu = fabsf(1.0f - cru) < ARTHRESH ? rb * cru : c[i - V1] + half_exp(rb - rbu);
            const float gd = fabsf(1.0f - crd) < ARTHRESH ? rb * crd : c[i + V1] + half_exp(rb - rbd);
            const float gl = fabsf(1.0f - crl) < ARTHRESH ? rb * crl : c[i - 1] + half_exp(rb - rbl);
            const float gr = fabsf(1.0f - crr) < ARTHRESH ? rb * crr : c[i + 1] + half_exp(rb - rbr);
            float gv = (t.dw0[i - V1] * gd + t.dw0[i + V1] * gu) / (t.dw0[i + V1] + t.dw0[i - V1]);
            float gh = (t.dw1[i - 1] * gr + t.dw1[i + 1] * gl) / (t.dw1[i - 1] + t.dw1[i + 1]);
            if (gv < rb) {
                if (2.0f * gv < rb) gv = ulim(gv, c[i - V1], c[i + V1]);
                else { const float wt = (2.0f * (rb - gv)) / (EPS + gv + rb); gv = wt * gv + (1.0f - wt) * ulim(gv, c[i - V1], c[i + V1]); }
            }
            if (gh < rb) {
                if (2.0f * gh < rb) gh = ulim(gh, c[i - 1], c[i + 1]);
                else { const float wt = (2.0f * (rb - gh)) / (EPS + gh + rb); gh = wt * gh + (1.0f - wt) * ulim(gh, c[i - 1], c[i + 1]); }
            }
            if (gh > CLIP_PT) gh = ulim(gh, c[i - 1], c[i + 1]);
            if (gv > CLIP_PT) gv = ulim(gv, c[i - V1], c[i + V1]);
            const float gi = gh * (1.0f - hw) + gv * hw;
            t.green[i] = gi;
            t.dgrb0[j] = gi - c[i];
        }
    }
    __syncthreads();

    AMZ_STAMP(9);
    // ---- chrominance (:1345-1395)
    {
        const int rows = iters(13, rr1 - 12, 2), sites = iters(13, cc1 - 12, 2);
        for (int n = tid; n < rows * sites; n += nt) {           // B sites: G-B moves to its own plane
            const int rr = 13 + 2 * (n / sites), j = (rr * T + 13) / 2 + n % sites;
            t.dgrb1[j] = t.dgrb0[j];
            t.dgrb0[j] = 0.0f;
        }
        __syncthreads();
        const int rows2 = rr1 - 28, lanes_max = 4 * iters(14, cc1 - 14, 8);
        for (int n = tid; n < rows2 * lanes_max; n += nt) {
            const int rr = 14 + n / lanes_max, m = n % lanes_max, cc0 = 14 + (rr & 1);
            if (m >= 4 * iters(cc0, cc1 - 14, 8)) continue;
            float *D = (1 - fc(rr, cc0) / 2) ? t.dgrb1 : t.dgrb0;
            const int b = rr * T + cc0;
            auto DD = [&](int o) { return D[(b + o) / 2 + m]; };
            const float wnw = 1.0f / (EPS + fabsf(DD(-M1) - DD(M1)) + fabsf(DD(-M1) - DD(-M3)) + fabsf(DD(M1) - DD(-M3)));
            const float wne = 1.0f / (EPS + fabsf(DD(P1) - DD(-P1)) + fabsf(DD(P1) - DD(P3)) + fabsf(DD(-P1) - DD(P3)));
            const float wsw = 1.0f / (EPS + fabsf(DD(-P1) - DD(P1)) + fabsf(DD(-P1) - DD(M3)) + fabsf(DD(P1) - DD(-P3)));
            const float wse = 1.0f / (EPS + fabsf(DD(M1) - DD(-M1)) + fabsf(DD(M1) - DD(-P3)) + fabsf(DD(-M1) - DD(M3)));
            D[b / 2 + m] = (wnw * (1.325f * DD(-M1) - 0.175f * DD(-M3) - 0.075f * DD(-M1 - 2) - 0.075f * DD(-M1 - V2)) +
                            wne * (1.325f * DD(P1) - 0.175f * DD(P3) - 0.075f * DD(P1 + 2) - 0.075f * DD(P1 + V2)) +
                            wsw * (1.325f * DD(-P1) - 0.175f * DD(-P3) - 0.075f * DD(-P1 - 2) - 0.075f * DD(-P1 - V2)) +
                            wse * (1.325f * DD(M1) - 0.175f * DD(M3) - 0.075f * DD(M1 + 2) - 0.075f * DD(M1 + V2))) /
                           (wnw + wne + wsw + wse);
        }
    }
    __syncthreads();

    AMZ_STAMP(10);
    // ---- the three planes of the tile interior (:1397-1470)
    {
        const float *hw = t.hvwt, *g = t.green;
        const int rows = rr1 - 32, cols = cc1 - 32, gcols = 4 * iters(16, cc1 - 19, 4);
        for (int n = tid; n < rows * cols; n += nt) {
            const int rr = 16 + n / cols, cc = 16 + n % cols, i = rr * T + cc;
            if (cc + left >= w) continue;                          // the reference writes these into its 16-float row padding
            const size_t o = (size_t)(rr + top) * w + (cc + left);
            float r, b;
            if (fc(rr, cc) == 1) {
                const float wu = hw[(i - V1) / 2], wr = 1.0f - hw[(i + 1) / 2], wl = 1.0f - hw[(i - 1) / 2], wd = hw[(i + V1) / 2];
                const float inv = 1.0f / (wu + wr + wl + wd);
                r = 65535.0f * (g[i] - (wu * t.dgrb0[(i - V1) / 2] + wr * t.dgrb0[(i + 1) / 2] + wl * t.dgrb0[(i - 1) / 2] + wd * t.dgrb0[(i + V1) / 2]) * inv);
                b = 65535.0f * (g[i] - (wu * t.dgrb1[(i - V1) / 2] + wr * t.dgrb1[(i + 1) / 2] + wl * t.dgrb1[(i - 1) / 2] + wd * t.dgrb1[(i + V1) / 2]) * inv);
            } else {
                r = 65535.0f * (g[i] - t.dgrb0[i / 2]);
                b = 65535.0f * (g[i] - t.dgrb1[i / 2]);
            }
            if (r2e) {
                // the conversion's look-ups instead of the planes (amaze_math.h: ev_of_planes; the three pointers are int planes then;
                // widths the conversion accepts are multiples of 4: green is written wherever red and blue are)
                int er, eg, eb, ey;
                amz::ev_of_planes(r2e, ev_black, r, g[i] * 65535.0f, b, er, eg, eb, ey);
                ((int *)red)[o] = er; ((int *)green_out)[o] = eg; ((int *)blue)[o] = eb; gray[o] = ey;
                continue;
            }
            red[o] = r;
            blue[o] = b;
            if (cc - 16 < gcols) green_out[o] = g[i] * 65535.0f;
        }
    }
    __syncthreads();
    AMZ_STAMP(11);
    }   // tiles of this workgroup
}

int amaze_launch(const float *d_raw, int w, int h, float *d_red, float *d_green, float *d_blue, float *d_scratch, hipStream_t s,
                 int nframes, size_t plane_stride, size_t scratch_stride, const int *h_of, int h_stride, float *d_rows_dbg,
                 const int *d_r2e, int ev_black, int *d_gray)
{
    if (d_r2e && (w % 4 != 0 || d_rows_dbg || !d_gray)) { set_error("amaze_launch: EV planes need a width that is a multiple of 4 and no debug planes"); return MLVFS_AMD_ERR_ARG; }
    const int step = AMAZE_TS - 32;
    const int tiles_x = (w + 16 + step - 1) / step, tiles_y = (h + 16 + step - 1) / step;
    const int cc1_last = w + 16 - (-16 + (tiles_x - 1) * step), rr1_last = h + 16 - (-16 + (tiles_y - 1) * step);
    static const int threads = [] { const char *e = getenv("MLVFS_AMD_AMAZE_THREADS"); const int v = e ? atoi(e) : 1024; return v >= 64 && v <= 1024 ? v / 64 * 64 : 1024; }();
    int nfx = 0, nfy = 0;
    amaze_rows_extent(w, h, &nfx, &nfy);                                         // the complete tiles go through LDS (k_amaze_rows.hip)
    const int dead_rows = d_rows_dbg ? 0 : amaze_rows_extra(w, h, nframes);               // and so do the heads of chains that have no output
    auto launch = [&](int row0, int nrows, int wgs_per_row, int chain_len, int rows_per_wg, int copy_from) {
        hipLaunchKernelGGL(k_amaze, dim3(nrows * wgs_per_row, nframes), dim3(threads), 0, s, d_raw, w, h, d_red, d_green, d_blue, d_scratch, tiles_x,
                           row0, wgs_per_row, chain_len, rows_per_wg, copy_from, plane_stride, scratch_stride, h_of, h_stride, nfx, nfy, dead_rows, d_r2e, ev_black, d_gray);
    };
    // The complete tiles run on a side stream, next to this stream's launches for the incomplete ones: those are few workgroups in
    // two dependent launches (1.5 ms of a mostly idle chip per batch of 8 at 3584x1320); k_amaze_rows draws its tiles from a counter,
    // so its workgroups that start late (behind the first launch below) just take fewer.
    struct Side {
        hipStream_t st = nullptr; hipEvent_t go = nullptr, done = nullptr; int *ctr = nullptr; int cap = 0;
        ~Side() { if (ctr) (void)hipFree(ctr); if (go) (void)hipEventDestroy(go); if (done) (void)hipEventDestroy(done); if (st) (void)hipStreamDestroy(st); }
    };
    // per thread and device (until round 4: per caller stream as well -- an entry per stream a caller ever passed, never pruned, and a
    // recycled stream handle inherited the old one).  One side stream serves all of a thread's caller streams: everything that touches
    // `ctr` is queued on it, in order, and an event's wait is bound to the record that precedes it, so re-recording go / done for the
    // next call does not disturb a wait that is already queued
    static thread_local std::map<int, Side> t_side;
    Side *side = nullptr;
    bool rows_pending = false;
    if (nfx) {
        int dev = 0;
        MLV_HIP(hipGetDevice(&dev));
        side = &t_side[dev];
        if (!side->st) {
            MLV_HIP(hipStreamCreateWithFlags(&side->st, hipStreamNonBlocking));
            MLV_HIP(hipEventCreateWithFlags(&side->go, hipEventDisableTiming));
            MLV_HIP(hipEventCreateWithFlags(&side->done, hipEventDisableTiming));
        }
        if (side->cap < nframes) {
            if (side->ctr) { MLV_HIP(hipStreamSynchronize(side->st)); MLV_HIP(hipFree(side->ctr)); side->ctr = nullptr; }
            side->cap = nframes > 64 ? nframes : 64;
            MLV_HIP(hipMalloc(&side->ctr, sizeof(int) * side->cap));
        }
        MLV_HIP(hipEventRecord(side->go, s));
        MLV_HIP(hipStreamWaitEvent(side->st, side->go, 0));
        MLV_HIP(hipMemsetAsync(side->ctr, 0, sizeof(int) * nframes, side->st));
        rows_pending = true;
    }
    auto rows_now = [&]() -> int {                                              // after this stream's first launch has been submitted
        if (!rows_pending) return MLVFS_AMD_OK;
        rows_pending = false;
        const int rc = amaze_rows_launch(d_raw, w, h, d_red, d_green, d_blue, side->st, nframes, plane_stride, h_of, h_stride, d_rows_dbg, side->ctr, d_r2e, ev_black, d_gray);
        if (rc) return rc;
        MLV_HIP(hipEventRecord(side->done, side->st));
        return MLVFS_AMD_OK;
    };
    static const bool rows_only = [] { const char *e = getenv("MLVFS_AMD_AMAZE_ROWS_ONLY"); return e && atoi(e); }();      // timing experiments
    if (rows_only && nfx) {
        const int rc = rows_now();
        if (rc) return rc;
        MLV_HIP(hipStreamWaitEvent(s, side->done, 0));
        return MLVFS_AMD_OK;
    }
    // incomplete tiles at the right end of a row, chained behind the last complete one
    const int incomplete_x = cc1_last >= AMAZE_TS ? 0 : (cc1_last < 32 ? 2 : 1);
    if (tiles_x < incomplete_x + 1 || tiles_x < 3) {
        launch(0, 1, 1, tiles_x, tiles_y, -1);                                 // small image: the reference's order, one workgroup
    } else {
        const int chain_len = incomplete_x + 1, wgs_per_row = tiles_x - incomplete_x;
        const int incomplete_y = rr1_last >= AMAZE_TS ? 0 : (rr1_last < 32 ? 2 : 1);
        const int rows_a = tiles_y - incomplete_y > 0 ? tiles_y - incomplete_y : 0;
        const bool garbage = rr1_last > 32 && rr1_last < 48;                   // the row above the last ran its bottom apron into pmwt
        if (rows_a > 0) launch(0, rows_a, wgs_per_row, chain_len, 1, -1);
        { const int rc = rows_now(); if (rc) return rc; }
        for (int ty = rows_a; ty < tiles_y; ty++) {
            const int src = ty > 0 ? (ty - 1) * tiles_x + (wgs_per_row - 1) : -1;   // block the previous row ended in
            if (garbage && ty == tiles_y - 1) {
                // the chained copy must land in this row's first block, then the whole row runs in it
                launch(ty, 1, 1, tiles_x, 1, src);
            } else {
                launch(ty, 1, wgs_per_row, chain_len, 1, src);
            }
        }
    }
    { const int rc = rows_now(); if (rc) return rc; }
    if (side && nfx) MLV_HIP(hipStreamWaitEvent(s, side->done, 0));
    MLV_HIP(hipGetLastError());
#ifdef AMAZE_DIAG
    {
        static int shown = 0;
        if (shown++ == 3) {
            unsigned long long st[16];
            hipStreamSynchronize(s);
            hipMemcpyFromSymbol(st, HIP_SYMBOL(g_amaze_stamps), sizeof st);
            const char *names[12] = { "load + apron", "gradients", "directional differences", "refinement (vcd walk)", "hv weight + nyquist test",
                                      "majority vote", "area interpolation", "G at R/B (row by row)", "diagonal interpolation (row by row)",
                                      "chrominance", "output planes", "" };
            fprintf(stderr, "AMAZE_DIAG cycles of workgroup 0, first tile (s_memtime):");
            for (int k = 0; k + 1 < 12; k++) fprintf(stderr, "\n  %-36s %8lld", names[k], (long long)(st[k + 1] - st[k]));
            fprintf(stderr, "\n  total %lld\n", (long long)(st[11] - st[0]));
        }
    }
#endif
    return MLVFS_AMD_OK;
}

size_t amaze_scratch_bytes(int w, int h)
{
    const size_t tiles_x = (w + 16 + (AMAZE_TS - 32) - 1) / (AMAZE_TS - 32), tiles_y = (h + 16 + (AMAZE_TS - 32) - 1) / (AMAZE_TS - 32);
    return tiles_x * tiles_y * (size_t)AMAZE_TILE_FLOATS * sizeof(float);
}

}  // namespace mlv
