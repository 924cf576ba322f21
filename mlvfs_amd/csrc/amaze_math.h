// amaze_math.h -- the scalar pieces of the AMaZE demosaic (mlvfs/amaze_demosaic_RT.c, SSE2 variant) shared by the two kernels that
// implement it: k_amaze.hip (a tile's planes in an HBM block: incomplete tiles and the chains that depend on stale planes) and
// k_amaze_rows.hip (complete tiles, row-streamed through LDS).  IEEE binary32 in the reference's operation order.
#pragma once
#include "clip.h"
#include "dualiso.h"

namespace mlv {
namespace amz {

constexpr int T = AMAZE_TS, TT = T * T, HALF = TT / 2;
constexpr int V1 = T, V2 = 2 * T, V3 = 3 * T, P1 = -T + 1, P2 = -2 * T + 2, P3 = -3 * T + 3, M1 = T + 1, M2 = 2 * T + 2, M3 = 3 * T + 3;
constexpr float EPS = 1e-5f, EPSSQ = 1e-10f, ARTHRESH = 0.75f, NYQTHRESH = 0.5f, CLIP_PT = 1.0f, CLIP_PT8 = 0.8f;


__device__ __forceinline__ int fc(int r, int c) { return (r & 1) == (c & 1) ? ((r & 1) ? 2 : 0) : 1; }
__device__ __forceinline__ float sq(float a) { return a * a; }
__device__ __forceinline__ float fminv(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float fmaxv(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ float lim(float a, float lo, float hi) { return fmaxv(lo, fminv(a, hi)); }
__device__ __forceinline__ float ulim(float a, float b, float c) { return b < c ? lim(a, b, c) : lim(a, c, b); }
__device__ __forceinline__ float half_exp(float d)          // xdiv2f: exponent - 1 unless zero
{
    int i = __float_as_int(d);
    if (i & 0x7FFFFFFF) i -= 1 << 23;
    return __int_as_float(i);
}
__device__ __forceinline__ float quarter_exp(float d)       // xdivf(d, 2)
{
    int i = __float_as_int(d);
    if (i & 0x7FFFFFFF) i -= 2 << 23;
    return __int_as_float(i);
}
__device__ __forceinline__ int iters(int start, int end, int step) { return end > start ? (end - start + step - 1) / step : 0; }
__device__ __forceinline__ float var3(float a, float b, float c) { return 3.0f * (sq(a) + sq(b) + sq(c)) - sq(a + b + c); }

// amaze_demosaic_RT.c:777-799
__device__ __forceinline__ float bound_difference(float cd, float sgn, float centre, float lo, float hi)
{
    const float nsgn = -sgn, sgn3 = 3.0f * sgn;
    const float gint = sgn * cd + centre, t2 = sgn3 * cd;
    const float wt = 1.0f + t2 / (EPS + gint + centre);
    const float alt = nsgn * (centre - ulim(gint, lo, hi));
    float r = (t2 < -(centre + gint)) ? alt : wt * cd + (1.0f - wt) * alt;
    r = (nsgn * cd > 0.0f) ? r : cd;
    return gint > CLIP_PT ? alt : r;
}
// :1120-1124
__device__ __forceinline__ float diag_estimate(float centre, float near, float far)
{
    const float ratio = (near + near) / (EPS + centre + far);
    return fabsf(1.0f - ratio) < ARTHRESH ? centre * ratio : near + 0.5f * (centre - far);
}
// :1133-1139
__device__ __forceinline__ float diag_bound(float rb, float centre, float lo, float hi)
{
    const float lim1 = ulim(rb, lo, hi);
    const float wt = 2.0f * (centre - rb) / (EPS + rb + centre);
    float r = wt * rb + (1.0f - wt) * lim1;
    r = (rb + rb < centre) ? lim1 : r;
    r = (rb < centre) ? r : rb;
    return r > CLIP_PT ? ulim(r, lo, hi) : r;
}

// What the dual-ISO conversion looks up of the three planes (hdr.c:1041-1062; k_dualiso.hip: k_di_amaze_ev did this in a pass of its own
// until round 5): green's scaling undone, the three clamped to 20 bits, interp_raw2ev of each and of the gray value.  r, g, b: the
// planes' values as the kernels would store them (65535 x the tile's value).
__device__ __forceinline__ void ev_of_planes(const int *__restrict__ r2e, int black, float r, float g_stored, float b, int &ev_r, int &ev_g, int &ev_b, int &ev_gray)
{
    const float fb = (float)black, hi = 1048575.0f;
    const float g = (g_stored - fb) * 2.0f + fb;
    const float gc = g < hi ? (g > 0.0f ? g : 0.0f) : hi, rc = r < hi ? (r > 0.0f ? r : 0.0f) : hi, bc = b < hi ? (b > 0.0f ? b : 0.0f) : hi;
    ev_g = r2e[(int)gc]; ev_r = r2e[(int)rc]; ev_b = r2e[(int)bc];
    ev_gray = r2e[(unsigned)(gc / 2 + rc / 4 + bc / 4)];
}

}  // namespace amz

}  // namespace mlv
