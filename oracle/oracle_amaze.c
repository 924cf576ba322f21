/*
 * oracle_amaze.c -- CPU restatement of the AMaZE demosaic that the dual-ISO
 * conversion uses as a temporary interpolator (mlvfs/amaze_demosaic_RT.c:113-1487,
 * called from mlvfs/hdr.c:1034).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Own code and structure; every pass cites
 * the reference lines it follows.
 *
 * WHICH reference: on x86-64 gcc defines __SSE2__, so the reference that MLVFS ships is
 * the SSE2 variant of every pass that has one.  It differs from the scalar variant in
 * ways that matter for bit-exactness, all reproduced here with explicit 4-lane groups:
 *   - vector loops run a few sites past the scalar loop bounds (stride 4 / stride 8);
 *   - the tile loader stores G and non-G sites alike into the green plane, and two of
 *     the corner fills copy 4 ascending pixels per group instead of mirroring them;
 *   - the in-place refinement of the horizontal colour differences (hcd) sees the NEW
 *     value of the left neighbour in lanes 0,1 and the OLD one in lanes 2,3
 *     (amaze_demosaic_RT.c:766-801);
 *   - halving is a multiply by 0.5f in the vector code and an exponent decrement
 *     (xdiv2f) in the scalar passes; 0.5f / 0.25f comparisons are float in the vector
 *     code and double in the scalar code.
 * Sequential (in-place) semantics that both variants share are kept as well: vcd
 * refinement reads the refined value two rows up; the hvwt and pmwt passes read the
 * already updated row above; the Nyquist majority vote runs in raster order.
 * The work arrays are allocated (zeroed) once per call and NOT cleared between tiles,
 * like the reference's calloc'ed block (amaze_demosaic_RT.c:244).
 *
 * Arithmetic: IEEE binary32 with the reference's operation order; compile without FMA
 * contraction (oracle/Makefile: -ffp-contract=off, no -march=native).
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define T 160                      /* tile side, amaze_demosaic_RT.c:136 */
#define TT (T * T)
#define HALF (TT / 2)
enum { V1 = T, V2 = 2 * T, V3 = 3 * T, P1 = -T + 1, P2 = -2 * T + 2, P3 = -3 * T + 3, M1 = T + 1, M2 = 2 * T + 2, M3 = 3 * T + 3 };

static const float EPS = 1e-5f, EPSSQ = 1e-10f, ARTHRESH = 0.75f, NYQTHRESH = 0.5f, CLIP_PT = 1.0f, CLIP_PT8 = 0.8f;
static const float G_ODD[4] = { 0.14659727707323927f, 0.103592713382435f, 0.0732036125103057f, 0.0365543548389495f };
static const float G_GRAD[6] = { 0.07384411893421103f, 0.06207511968171489f, 0.0521818194747806f,
                                 0.03687419286733595f, 0.03099732204057846f, 0.018413194161458882f };
static const float G_EVEN[2] = { 0.13719494435797422f, 0.05640252782101291f };
static const float G_QUINC[4] = { 0.169917f, 0.108947f, 0.069855f, 0.0287182f };

/* RGGB colour of a site: 0 R, 1 G, 2 B (amaze_demosaic_RT.c:41-49) */
static inline int fc(int r, int c) { return (r & 1) == (c & 1) ? ((r & 1) ? 2 : 0) : 1; }

static inline float sq(float a) { return a * a; }
static inline float fminv(float a, float b) { return a < b ? a : b; }          /* _mm_min_ps / MIN */
static inline float fmaxv(float a, float b) { return a > b ? a : b; }          /* _mm_max_ps / MAX */
static inline float lim(float a, float lo, float hi) { return fmaxv(lo, fminv(a, hi)); }
static inline float ulim(float a, float b, float c) { return b < c ? lim(a, b, c) : lim(a, c, b); }

/* exponent tricks of the scalar passes (amaze_demosaic_RT.c:87-99) */
static inline float half_exp(float d)
{
    union { float f; int32_t i; } u = { d };
    if (u.i & 0x7FFFFFFF) u.i -= 1 << 23;
    return u.f;
}
static inline float quarter_exp(float d)
{
    union { float f; int32_t i; } u = { d };
    if (u.i & 0x7FFFFFFF) u.i -= 2 << 23;
    return u.f;
}

/* The members follow the order of the reference's block (amaze_demosaic_RT.c:248-273) where it matters: its planes
 * are 64 bytes apart, and two fills of the tile loader run past their plane -- a bottom fill that starts less than
 * 16 rows before row 160 continues from cfa into pmwt and from rgbgreen into delhvsqsum. */
typedef struct {
    float green[TT], gap0[16], delsq[TT], dw0[TT], dw1[TT], vcd[TT], hcd[TT], vcdalt[TT], hcdalt[TT], cdsq[TT], dgv[TT], dgh[TT];
    float hvwt[HALF], dgrb0[HALF], dgrb1[HALF], delp[HALF], delm[HALF], rbint[HALF], curv_h[HALF], curv_v[HALF], sqm[HALF], sqp[HALF];
    float cfa[TT], gap1[16], pmwt[HALF], rbm[HALF], rbp[HALF];
    unsigned char nyq[HALF];
    float slack[64];               /* vector loops may touch a few floats past the last array */
} tile_t;

typedef struct { const float *raw; int w, h, pitch; float *red, *green, *blue; } image_t;
#define RAW(img, y, x) ((img)->raw[(size_t)(y) * (img)->pitch + (x)])

/* ------------------------------------------------------------------ tile load + mirrored borders, :361-469 */
static void load_tile(tile_t *t, const image_t *im, int top, int left, int rr1, int cc1, int rrmin, int rrmax, int ccmin, int ccmax)
{
    const int w = im->w, h = im->h;
#define PUT_ALL(i, v) do { float v_ = (v) / 65535.0f; t->cfa[i] = v_; t->green[i] = v_; } while (0)
#define PUT_G(i, r, c, v) do { float v_ = (v) / 65535.0f; t->cfa[i] = v_; if (fc(r, c) == 1) t->green[i] = v_; } while (0)
    for (int rr = rrmin; rr < rrmax; rr++) {
        int cc = ccmin;
        for (; cc < ccmax - 3; cc += 4)
            for (int k = 0; k < 4; k++) PUT_ALL(rr * T + cc + k, RAW(im, rr + top, cc + left + k));
        for (; cc < ccmax; cc++) PUT_G(rr * T + cc, rr, cc, RAW(im, rr + top, cc + left));
    }
    if (rrmin > 0)
        for (int rr = 0; rr < 16; rr++)
            for (int cc = ccmin; cc < ccmax; cc++) PUT_G(rr * T + cc, rr, cc, RAW(im, 32 - rr + top, cc + left));
    if (rrmax < rr1)
        for (int rr = 0; rr < 16; rr++)
            for (int cc = ccmin; cc < ccmax; cc += 4)
                for (int k = 0; k < 4; k++) PUT_ALL((rrmax + rr) * T + cc + k, RAW(im, h - rr - 2, left + cc + k));
    if (ccmin > 0)
        for (int rr = rrmin; rr < rrmax; rr++)
            for (int cc = 0; cc < 16; cc++) PUT_G(rr * T + cc, rr, cc, RAW(im, rr + top, 32 - cc + left));
    if (ccmax < cc1)
        for (int rr = rrmin; rr < rrmax; rr++)
            for (int cc = 0; cc < 16; cc++) PUT_G(rr * T + ccmax + cc, rr, cc, RAW(im, top + rr, w - cc - 2));
    if (rrmin > 0 && ccmin > 0)                        /* 4 ascending pixels per group (vector load), :423-430 */
        for (int rr = 0; rr < 16; rr++)
            for (int cc = 0; cc < 16; cc += 4)
                for (int k = 0; k < 4; k++) PUT_ALL(rr * T + cc + k, RAW(im, 32 - rr, 32 - cc + k));
    if (rrmax < rr1 && ccmax < cc1)
        for (int rr = 0; rr < 16; rr++)
            for (int cc = 0; cc < 16; cc += 4)
                for (int k = 0; k < 4; k++) PUT_ALL((rrmax + rr) * T + ccmax + cc + k, RAW(im, h - rr - 2, w - cc - 2 + k));
    if (rrmin > 0 && ccmax < cc1)
        for (int rr = 0; rr < 16; rr++)
            for (int cc = 0; cc < 16; cc++) PUT_G(rr * T + ccmax + cc, rr, cc, RAW(im, 32 - rr, w - cc - 2));
    if (rrmax < rr1 && ccmin > 0)
        for (int rr = 0; rr < 16; rr++)
            for (int cc = 0; cc < 16; cc++) PUT_G((rrmax + rr) * T + cc, rr, cc, RAW(im, h - rr - 2, 32 - cc));
#undef PUT_ALL
#undef PUT_G
}

/* ------------------------------------------------------------------ gradients, :537-613 */
static void gradients(tile_t *t, int rr1, int cc1)
{
    const float *c = t->cfa;
    for (int rr = 2; rr < rr1 - 2; rr++)
        for (int cc = 0; cc < cc1; cc += 4)
            for (int k = 0; k < 4; k++) {
                const int i = rr * T + cc + k;
                const float dh = fabsf(c[i + 1] - c[i - 1]), dv = fabsf(c[i + V1] - c[i - V1]);
                t->dw1[i] = EPS + fabsf(c[i + 2] - c[i]) + fabsf(c[i] - c[i - 2]) + dh;
                t->dw0[i] = EPS + fabsf(c[i + V2] - c[i]) + fabsf(c[i] - c[i - V2]) + dv;
                t->delsq[i] = dh * dh + dv * dv;
            }
    for (int rr = 6; rr < rr1 - 6; rr++) {
        const int g = (fc(rr, 2) & 1) ? 0 : 1;          /* offset of the G site inside the pair starting at an even column */
        for (int cc = 6; cc < cc1 - 6; cc += 8)
            for (int k = 0; k < 4; k++) {
                const int i = rr * T + cc + 2 * k, ig = i + g, ic = i + 1 - g;     /* ig: G site, ic: R/B site of the pair */
                t->delp[(rr * T + cc) / 2 + k] = fabsf(c[ic + P1] - c[ic - P1]);
                t->delm[(rr * T + cc) / 2 + k] = fabsf(c[ic + M1] - c[ic - M1]);
                t->sqp[(rr * T + cc) / 2 + k] = sq(c[ig] - c[ig - P1]) + sq(c[ig] - c[ig + P1]);
                t->sqm[(rr * T + cc) / 2 + k] = sq(c[ig] - c[ig - M1]) + sq(c[ig] - c[ig + M1]);
            }
    }
}

/* ------------------------------------------------------------------ directional colour differences, :622-675 */
static void colour_differences(tile_t *t, int rr1, int cc1)
{
    const float *c = t->cfa, *d0 = t->dw0, *d1 = t->dw1;
    for (int rr = 4; rr < rr1 - 4; rr++)
        for (int cc = 4; cc < cc1 - 7; cc += 4)
            for (int k = 0; k < 4; k++) {
                const int i = rr * T + cc + k;
                const float sgn = ((rr + cc + k) & 1) ? -1.0f : 1.0f;            /* + at R/B sites, - at G sites */
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
                t->hcdalt[i] = sgn * (ginth - c[i]);
                t->vcdalt[i] = sgn * (gintv - c[i]);
                const int clip = c[i] > CLIP_PT8 || gintv > CLIP_PT8 || ginth > CLIP_PT8;
                if (clip) { guar = guha; gdar = gdha; glar = glha; grar = grha; }
                t->vcd[i] = clip ? t->vcdalt[i] : sgn * ((vwt * gdar + (1.0f - vwt) * guar) - c[i]);
                t->hcd[i] = clip ? t->hcdalt[i] : sgn * ((hwt * grar + (1.0f - hwt) * glar) - c[i]);
                t->dgv[i] = fminv(sq(guha - gdha), sq(guar - gdar));
                t->dgh[i] = fminv(sq(glha - grha), sq(glar - grar));
            }
}

static inline float var3(float a, float b, float c) { return 3.0f * (sq(a) + sq(b) + sq(c)) - sq(a + b + c); }

/* one refined colour difference, :777-799 (same formula for hcd with lo/hi = left/right and vcd with up/down) */
static inline float bound_difference(float cd, float sgn, float centre, float lo, float hi)
{
    const float nsgn = -sgn, sgn3 = 3.0f * sgn;
    const float gint = sgn * cd + centre, t2 = sgn3 * cd;
    const float wt = 1.0f + t2 / (EPS + gint + centre);
    const float alt = nsgn * (centre - ulim(gint, lo, hi));
    float r = (t2 < -(centre + gint)) ? alt : wt * cd + (1.0f - wt) * alt;
    r = (nsgn * cd > 0.0f) ? r : cd;
    return gint > CLIP_PT ? alt : r;
}

/* ------------------------------------------------------------------ in-place refinement, :766-801 */
static void refine_differences(tile_t *t, int rr1, int cc1)
{
    const float *c = t->cfa;
    for (int rr = 4; rr < rr1 - 4; rr++)
        for (int cc = 4; cc < cc1 - 4; cc += 4) {
            float h[4], v[4];
            for (int k = 0; k < 4; k++) {                /* loads of the whole group happen before its stores */
                const int i = rr * T + cc + k;
                const float hv = var3(t->hcd[i - 2], t->hcd[i], t->hcd[i + 2]), hav = var3(t->hcdalt[i - 2], t->hcdalt[i], t->hcdalt[i + 2]);
                const float vv = var3(t->vcd[i - V2], t->vcd[i], t->vcd[i + V2]), vav = var3(t->vcdalt[i - V2], t->vcdalt[i], t->vcdalt[i + V2]);
                const float sgn = ((rr + cc + k) & 1) ? -1.0f : 1.0f;
                h[k] = bound_difference(hav < hv ? t->hcdalt[i] : t->hcd[i], sgn, c[i], c[i - 1], c[i + 1]);
                v[k] = bound_difference(vav < vv ? t->vcdalt[i] : t->vcd[i], sgn, c[i], c[i - V1], c[i + V1]);
            }
            for (int k = 0; k < 4; k++) {
                const int i = rr * T + cc + k;
                t->hcd[i] = h[k];
                t->vcd[i] = v[k];
                t->cdsq[i] = sq(v[k] - h[k]);
            }
        }
}

/* ------------------------------------------------------------------ horizontal/vertical weight, :881-925 */
static void direction_weights(tile_t *t, int rr1, int cc1)
{
    const float *vc = t->vcd, *hc = t->hcd;
    for (int rr = 6; rr < rr1 - 6; rr++)
        for (int cc = 6 + (fc(rr, 2) & 1); cc < cc1 - 6; cc += 8)
            for (int k = 0; k < 4; k++) {
                const int i = rr * T + cc + 2 * k;
                const float uave = vc[i] + vc[i - V1] + vc[i - V2] + vc[i - V3], dave = vc[i] + vc[i + V1] + vc[i + V2] + vc[i + V3];
                float vu = sq(vc[i] - uave) + sq(vc[i - V1] - uave) + sq(vc[i - V2] - uave) + sq(vc[i - V3] - uave);
                float vd = sq(vc[i] - dave) + sq(vc[i + V1] - dave) + sq(vc[i + V2] - dave) + sq(vc[i + V3] - dave);
                const float hwt = t->dw1[i - 1] / (t->dw1[i - 1] + t->dw1[i + 1]), vwt = t->dw0[i - V1] / (t->dw0[i + V1] + t->dw0[i - V1]);
                const float lave = hc[i] + hc[i - 1] + hc[i - 2] + hc[i - 3], rave = hc[i] + hc[i + 1] + hc[i + 2] + hc[i + 3];
                float hl = sq(hc[i] - lave) + sq(hc[i - 1] - lave) + sq(hc[i - 2] - lave) + sq(hc[i - 3] - lave);
                float hr = sq(hc[i] - rave) + sq(hc[i + 1] - rave) + sq(hc[i + 2] - rave) + sq(hc[i + 3] - rave);
                const float vcdvar = EPSSQ + vwt * vd + (1.0f - vwt) * vu, hcdvar = EPSSQ + hwt * hr + (1.0f - hwt) * hl;
                vu = t->dgv[i] + t->dgv[i - V1] + t->dgv[i - V2];
                vd = t->dgv[i] + t->dgv[i + V1] + t->dgv[i + V2];
                hl = t->dgh[i] + t->dgh[i - 1] + t->dgh[i - 2];
                hr = t->dgh[i] + t->dgh[i + 1] + t->dgh[i + 2];
                const float vcdvar1 = EPSSQ + vwt * vd + (1.0f - vwt) * vu, hcdvar1 = EPSSQ + hwt * hr + (1.0f - hwt) * hl;
                const float varwt = hcdvar / (vcdvar + hcdvar), diffwt = hcdvar1 / (vcdvar1 + hcdvar1);
                const int agree = (0.5f - varwt) * (0.5f - diffwt) > 0.0f && fabsf(0.5f - diffwt) < fabsf(0.5f - varwt);
                t->hvwt[i / 2] = agree ? varwt : diffwt;
            }
}

/* ------------------------------------------------------------------ Nyquist texture test + vote + area interpolation, :969-1044 */
static void nyquist_pass(tile_t *t, int rr1, int cc1)
{
    const float *q = t->cdsq, *d = t->delsq, *c = t->cfa;
    unsigned char *ny = t->nyq;
    for (int rr = 6; rr < rr1 - 6; rr++)
        for (int cc = 6 + (fc(rr, 2) & 1); cc < cc1 - 6; cc += 2) {
            const int i = rr * T + cc;
            float test = (G_ODD[0] * q[i] + G_ODD[1] * (q[i - M1] + q[i + P1] + q[i - P1] + q[i + M1]) +
                          G_ODD[2] * (q[i - V2] + q[i - 2] + q[i + 2] + q[i + V2]) + G_ODD[3] * (q[i - M2] + q[i + P2] + q[i - P2] + q[i + M2]));
            test -= NYQTHRESH * (G_GRAD[0] * d[i] + G_GRAD[1] * (d[i - V1] + d[i + 1] + d[i - 1] + d[i + V1]) +
                                 G_GRAD[2] * (d[i - M1] + d[i + P1] + d[i - P1] + d[i + M1]) + G_GRAD[3] * (d[i - V2] + d[i - 2] + d[i + 2] + d[i + V2]) +
                                 G_GRAD[4] * (d[i - 2 * T - 1] + d[i - 2 * T + 1] + d[i - T - 2] + d[i - T + 2] + d[i + T - 2] + d[i + T + 2] +
                                              d[i + 2 * T - 1] + d[i + 2 * T + 1]) +
                                 G_GRAD[5] * (d[i - M2] + d[i + P2] + d[i - P2] + d[i + M2]));
            if (test > 0) ny[i / 2] = 1;
        }
    for (int rr = 8; rr < rr1 - 8; rr++)                 /* majority vote, in place and in raster order */
        for (int cc = 8 + (fc(rr, 2) & 1); cc < cc1 - 8; cc += 2) {
            const int i = rr * T + cc;
            const unsigned n = ny[(i - V2) / 2] + ny[(i - M1) / 2] + ny[(i + P1) / 2] + ny[(i - 2) / 2] + ny[i / 2] + ny[(i + 2) / 2] +
                               ny[(i - P1) / 2] + ny[(i + M1) / 2] + ny[(i + V2) / 2];
            if (n > 4) ny[i / 2] = 1;
            if (n < 4) ny[i / 2] = 0;
        }
    for (int rr = 8; rr < rr1 - 8; rr++)
        for (int cc = 8 + (fc(rr, 2) & 1); cc < cc1 - 8; cc += 2) {
            const int i = rr * T + cc;
            if (!ny[i / 2]) continue;
            float sumh = 0, sumv = 0, sumsqh = 0, sumsqv = 0, area = 0;
            for (int a = -6; a < 7; a += 2)
                for (int b = -6; b < 7; b += 2) {
                    const int j = (rr + a) * T + cc + b;
                    if (!ny[j / 2]) continue;
                    sumh += c[j] - half_exp(c[j - 1] + c[j + 1]);
                    sumv += c[j] - half_exp(c[j - V1] + c[j + V1]);
                    sumsqh += half_exp(sq(c[j] - c[j - 1]) + sq(c[j] - c[j + 1]));
                    sumsqv += half_exp(sq(c[j] - c[j - V1]) + sq(c[j] - c[j + V1]));
                    area += 1;
                }
            const float hvar = EPSSQ + fabsf(area * sumsqh - sumh * sumh), vvar = EPSSQ + fabsf(area * sumsqv - sumv * sumv);
            t->hvwt[i / 2] = hvar / (vvar + hvar);
        }
}

/* ------------------------------------------------------------------ G at R/B sites + Nyquist refinement, :1046-1101 */
static void populate_green(tile_t *t, int rr1, int cc1)
{
    float *hw = t->hvwt, *g = t->green;
    for (int rr = 8; rr < rr1 - 8; rr++)                 /* reads the updated row above: sequential over rows */
        for (int cc = 8 + (fc(rr, 2) & 1); cc < cc1 - 8; cc += 2) {
            const int i = rr * T + cc, j = i / 2;
            const float alt = quarter_exp(hw[(i - M1) / 2] + hw[(i + P1) / 2] + hw[(i - P1) / 2] + hw[(i + M1) / 2]);
            if (fabsf(0.5f - hw[j]) < fabsf(0.5f - alt)) hw[j] = alt;
            t->dgrb0[j] = t->hcd[i] * (1.0f - hw[j]) + t->vcd[i] * hw[j];
            g[i] = t->cfa[i] + t->dgrb0[j];
            if (t->nyq[j]) {
                t->curv_h[j] = sq(g[i] - half_exp(g[i - 1] + g[i + 1]));
                t->curv_v[j] = sq(g[i] - half_exp(g[i - V1] + g[i + V1]));
            } else
                t->curv_h[j] = t->curv_v[j] = 0;
        }
    for (int rr = 8; rr < rr1 - 8; rr++)
        for (int cc = 8 + (fc(rr, 2) & 1); cc < cc1 - 8; cc += 2) {
            const int i = rr * T + cc, j = i / 2;
            if (!t->nyq[j]) continue;
#define RING(a) (G_QUINC[0] * a[j] + G_QUINC[1] * (a[(i - M1) / 2] + a[(i + P1) / 2] + a[(i - P1) / 2] + a[(i + M1) / 2]) + \
                 G_QUINC[2] * (a[(i - V2) / 2] + a[(i - 2) / 2] + a[(i + 2) / 2] + a[(i + V2) / 2]) +                          \
                 G_QUINC[3] * (a[(i - M2) / 2] + a[(i + P2) / 2] + a[(i - P2) / 2] + a[(i + M2) / 2]))
            const float gvarh = EPSSQ + RING(t->curv_h), gvarv = EPSSQ + RING(t->curv_v);
#undef RING
            t->dgrb0[j] = (t->hcd[i] * gvarv + t->vcd[i] * gvarh) / (gvarv + gvarh);
            g[i] = t->cfa[i] + t->dgrb0[j];
        }
}

/* one diagonal estimate of the opposite colour, :1120-1124 */
static inline float diag_estimate(float centre, float near, float far)
{
    const float ratio = (near + near) / (EPS + centre + far);
    return fabsf(1.0f - ratio) < ARTHRESH ? centre * ratio : near + 0.5f * (centre - far);
}
/* saturation bound of a diagonal interpolation, :1133-1139 */
static inline float diag_bound(float rb, float centre, float lo, float hi)
{
    const float lim1 = ulim(rb, lo, hi);
    const float wt = 2.0f * (centre - rb) / (EPS + rb + centre);
    float r = wt * rb + (1.0f - wt) * lim1;
    r = (rb + rb < centre) ? lim1 : r;
    r = (rb < centre) ? r : rb;
    return r > CLIP_PT ? ulim(r, lo, hi) : r;
}

/* ------------------------------------------------------------------ diagonal interpolation, :1112-1276 */
static void diagonal_pass(tile_t *t, int rr1, int cc1)
{
    const float *c = t->cfa;
    for (int rr = 8; rr < rr1 - 8; rr++)
        for (int cc = 8 + (fc(rr, 2) & 1); cc < cc1 - 8; cc += 8)
            for (int k = 0; k < 4; k++) {
                const int i = rr * T + cc + 2 * k, j = (rr * T + cc) / 2 + k;
                const float se = diag_estimate(c[i], c[i + M1], c[i + M2]), nw = diag_estimate(c[i], c[i - M1], c[i - M2]);
                const float base_m = EPS + t->delm[j];
                const float wse = base_m + t->delm[(rr * T + cc + M1) / 2 + k] + t->delm[(rr * T + cc + M2) / 2 + k];
                const float wnw = base_m + t->delm[(rr * T + cc - M1) / 2 + k] + t->delm[(rr * T + cc - M2) / 2 + k];
                t->rbm[j] = diag_bound((wse * nw + wnw * se) / (wse + wnw), c[i], c[i - M1], c[i + M1]);
                const float ne = diag_estimate(c[i], c[i + P1], c[i + P2]), sw = diag_estimate(c[i], c[i - P1], c[i - P2]);
                const float base_p = EPS + t->delp[j];
                const float wne = base_p + t->delp[(rr * T + cc + P1) / 2 + k] + t->delp[(rr * T + cc + P2) / 2 + k];
                const float wsw = base_p + t->delp[(rr * T + cc - P1) / 2 + k] + t->delp[(rr * T + cc - P2) / 2 + k];
                t->rbp[j] = diag_bound((wne * sw + wsw * ne) / (wne + wsw), c[i], c[i - P1], c[i + P1]);
#define EVEN_RING(a, b) (EPSSQ + (G_EVEN[0] * (a[(b - V1) / 2 + k] + a[(b - 1) / 2 + k] + a[(b + 1) / 2 + k] + a[(b + V1) / 2 + k]) + \
                                  G_EVEN[1] * (a[(b - V2 - 1) / 2 + k] + a[(b - V2 + 1) / 2 + k] + a[(b - 2 - V1) / 2 + k] + a[(b + 2 - V1) / 2 + k] + \
                                               a[(b - 2 + V1) / 2 + k] + a[(b + 2 + V1) / 2 + k] + a[(b + V2 - 1) / 2 + k] + a[(b + V2 + 1) / 2 + k])))
                const int b0 = rr * T + cc;
                const float varm = EVEN_RING(t->sqm, b0);
                t->pmwt[j] = varm / (EVEN_RING(t->sqp, b0) + varm);
#undef EVEN_RING
            }
    for (int rr = 10; rr < rr1 - 10; rr++)               /* reads the updated row above: sequential over rows */
        for (int cc = 10 + (fc(rr, 2) & 1); cc < cc1 - 10; cc += 8) {
            float nw[4];
            const int b0 = rr * T + cc, j0 = b0 / 2;
            for (int k = 0; k < 4; k++) {
                const float alt = 0.25f * (t->pmwt[(b0 - M1) / 2 + k] + t->pmwt[(b0 + P1) / 2 + k] + t->pmwt[(b0 - P1) / 2 + k] + t->pmwt[(b0 + M1) / 2 + k]);
                const float cur = t->pmwt[j0 + k];
                nw[k] = fabsf(0.5f - cur) < fabsf(0.5f - alt) ? alt : cur;
            }
            for (int k = 0; k < 4; k++) {
                t->pmwt[j0 + k] = nw[k];
                t->rbint[j0 + k] = 0.5f * (c[b0 + 2 * k] + t->rbm[j0 + k] * (1.0f - nw[k]) + t->rbp[j0 + k] * nw[k]);
            }
        }
    for (int rr = 12; rr < rr1 - 12; rr++)
        for (int cc = 12 + (fc(rr, 2) & 1); cc < cc1 - 12; cc += 2) {
            const int i = rr * T + cc, j = i / 2;
            if (fabsf(0.5f - t->pmwt[j]) < fabsf(0.5f - t->hvwt[j])) continue;
            const float rb = t->rbint[j], rbu = t->rbint[j - V1], rbd = t->rbint[j + V1], rbl = t->rbint[j - 1], rbr = t->rbint[j + 1];
            /* sic: rbint is a half-width array but the reference offsets it by a FULL row (indx1-v1), :1289-1290 */
            const float cru = c[i - V1] * 2.0 / (EPS + rb + rbu), crd = c[i + V1] * 2.0 / (EPS + rb + rbd);
            const float crl = c[i - 1] * 2.0 / (EPS + rb + rbl), crr = c[i + 1] * 2.0 / (EPS + rb + rbr);
            const float gu = fabsf(1.0f - cru) < ARTHRESH ? rb * cru : c[i - V1] + half_exp(rb - rbu);
            const float gd = fabsf(1.0f - crd) < ARTHRESH ? rb * crd : c[i + V1] + half_exp(rb - rbd);
            const float gl = fabsf(1.0f - crl) < ARTHRESH ? rb * crl : c[i - 1] + half_exp(rb - rbl);
            const float gr = fabsf(1.0f - crr) < ARTHRESH ? rb * crr : c[i + 1] + half_exp(rb - rbr);
            float gv = (t->dw0[i - V1] * gd + t->dw0[i + V1] * gu) / (t->dw0[i + V1] + t->dw0[i - V1]);
            float gh = (t->dw1[i - 1] * gr + t->dw1[i + 1] * gl) / (t->dw1[i - 1] + t->dw1[i + 1]);
            if (gv < rb) {
                if (2 * gv < rb) gv = ulim(gv, c[i - V1], c[i + V1]);
                else { const float wt = 2.0 * (rb - gv) / (EPS + gv + rb); gv = wt * gv + (1.0f - wt) * ulim(gv, c[i - V1], c[i + V1]); }
            }
            if (gh < rb) {
                if (2 * gh < rb) gh = ulim(gh, c[i - 1], c[i + 1]);
                else { const float wt = 2.0 * (rb - gh) / (EPS + gh + rb); gh = wt * gh + (1.0f - wt) * ulim(gh, c[i - 1], c[i + 1]); }
            }
            if (gh > CLIP_PT) gh = ulim(gh, c[i - 1], c[i + 1]);
            if (gv > CLIP_PT) gv = ulim(gv, c[i - V1], c[i + V1]);
            t->green[i] = gh * (1.0f - t->hvwt[j]) + gv * t->hvwt[j];
            t->dgrb0[j] = t->green[i] - c[i];
        }
}

/* ------------------------------------------------------------------ chrominance, :1345-1450 */
static void chrominance_pass(tile_t *t, const image_t *im, int top, int left, int rr1, int cc1)
{
    for (int rr = 13; rr < rr1 - 12; rr += 2)            /* B sites: move G-B out of the G-R plane */
        for (int cc = 13, j = (rr * T + cc) / 2; cc < cc1 - 12; cc += 2, j++) { t->dgrb1[j] = t->dgrb0[j]; t->dgrb0[j] = 0; }
    for (int rr = 14; rr < rr1 - 14; rr++) {
        const int cc0 = 14 + (fc(rr, 2) & 1);
        float *D = (1 - fc(rr, cc0) / 2) ? t->dgrb1 : t->dgrb0;
        for (int cc = cc0; cc < cc1 - 14; cc += 8)
            for (int k = 0; k < 4; k++) {
                const int b = rr * T + cc;
#define DD(o) D[(b + (o)) / 2 + k]
                const float wnw = 1.0f / (EPS + fabsf(DD(-M1) - DD(M1)) + fabsf(DD(-M1) - DD(-M3)) + fabsf(DD(M1) - DD(-M3)));
                const float wne = 1.0f / (EPS + fabsf(DD(P1) - DD(-P1)) + fabsf(DD(P1) - DD(P3)) + fabsf(DD(-P1) - DD(P3)));
                const float wsw = 1.0f / (EPS + fabsf(DD(-P1) - DD(P1)) + fabsf(DD(-P1) - DD(M3)) + fabsf(DD(P1) - DD(-P3)));
                const float wse = 1.0f / (EPS + fabsf(DD(M1) - DD(-M1)) + fabsf(DD(M1) - DD(-P3)) + fabsf(DD(-M1) - DD(M3)));
                const float v = (wnw * (1.325f * DD(-M1) - 0.175f * DD(-M3) - 0.075f * DD(-M1 - 2) - 0.075f * DD(-M1 - V2)) +
                                 wne * (1.325f * DD(P1) - 0.175f * DD(P3) - 0.075f * DD(P1 + 2) - 0.075f * DD(P1 + V2)) +
                                 wsw * (1.325f * DD(-P1) - 0.175f * DD(-P3) - 0.075f * DD(-P1 - 2) - 0.075f * DD(-P1 - V2)) +
                                 wse * (1.325f * DD(M1) - 0.175f * DD(M3) - 0.075f * DD(M1 + 2) - 0.075f * DD(M1 + V2))) /
                                (wnw + wne + wsw + wse);
                D[b / 2 + k] = v;
#undef DD
            }
    }
    const float *hw = t->hvwt, *g = t->green;
    for (int rr = 16; rr < rr1 - 16; rr++) {
        const int row = rr + top;
        float *R = im->red + (size_t)row * im->pitch, *B = im->blue + (size_t)row * im->pitch;
        for (int cc = 16; cc < cc1 - 16; cc++) {         /* the reference walks pairs; per site it is this */
            const int i = rr * T + cc, col = cc + left;
            if (fc(rr, cc) == 1) {                        /* G site: both differences come from the 4 neighbours */
                const float wu = hw[(i - V1) / 2], wr = 1.0f - hw[(i + 1) / 2], wl = 1.0f - hw[(i - 1) / 2], wd = hw[(i + V1) / 2];
                const float inv = 1.0f / (wu + wr + wl + wd);
                R[col] = 65535.0f * (g[i] - (wu * t->dgrb0[(i - V1) / 2] + wr * t->dgrb0[(i + 1) / 2] + wl * t->dgrb0[(i - 1) / 2] + wd * t->dgrb0[(i + V1) / 2]) * inv);
                B[col] = 65535.0f * (g[i] - (wu * t->dgrb1[(i - V1) / 2] + wr * t->dgrb1[(i + 1) / 2] + wl * t->dgrb1[(i - 1) / 2] + wd * t->dgrb1[(i + V1) / 2]) * inv);
            } else {
                R[col] = 65535.0f * (g[i] - t->dgrb0[i / 2]);
                B[col] = 65535.0f * (g[i] - t->dgrb1[i / 2]);
            }
        }
        float *G = im->green + (size_t)row * im->pitch;
        for (int cc = 16; cc < cc1 - 19; cc += 4)
            for (int k = 0; k < 4; k++) G[cc + k + left] = g[rr * T + cc + k] * 65535.0f;
    }
}

/* raw/red/green/blue: h rows of `pitch` floats (pitch >= w + 16; the reference over-reads up to w+1, hdr.c:969-975).
 * Requires w % 4 == 0 like the SSE2 reference needs for a fully written green plane.  Returns 0, or -1 on bad arguments. */
int orc_amaze_demosaic(const float *raw, int w, int h, int pitch, float *red, float *green, float *blue)
{
    if (w <= 0 || h <= 0 || pitch < w + 16) return -1;
    tile_t *t = (tile_t *)calloc(1, sizeof(tile_t));
    if (!t) return -1;
    const image_t im = { raw, w, h, pitch, red, green, blue };
    for (int top = -16; top < h; top += T - 32)
        for (int left = -16; left < w; left += T - 32) {
            memset(t->nyq, 0, sizeof t->nyq);
            memset(t->rbint, 0, sizeof t->rbint);
            const int bottom = top + T < h + 16 ? top + T : h + 16, right = left + T < w + 16 ? left + T : w + 16;
            const int rr1 = bottom - top, cc1 = right - left;
            const int rrmin = top < 0 ? 16 : 0, ccmin = left < 0 ? 16 : 0;
            const int rrmax = bottom > h ? h - top : rr1, ccmax = right > w ? w - left : cc1;
            load_tile(t, &im, top, left, rr1, cc1, rrmin, rrmax, ccmin, ccmax);
            gradients(t, rr1, cc1);
            colour_differences(t, rr1, cc1);
            refine_differences(t, rr1, cc1);
            direction_weights(t, rr1, cc1);
            nyquist_pass(t, rr1, cc1);
            populate_green(t, rr1, cc1);
            diagonal_pass(t, rr1, cc1);
            chrominance_pass(t, &im, top, left, rr1, cc1);
        }
    free(t);
    return 0;
}
