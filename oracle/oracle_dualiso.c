/*
 * oracle_dualiso.c -- CPU restatement of the full dual-ISO conversion
 * (cr2hdr 20-bit), mlvfs/hdr.c:230-1957: interp_method = 1 ("mean23",
 * hdr.c:1231-1304) and interp_method = 0 (AMaZE + edge-directed interpolation,
 * hdr.c:914-1229; the demosaic itself is oracle_amaze.c).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Own code, own structure; every stage
 * cites the reference lines it follows.  Behaviour reproduced on purpose:
 *  - MLVFS passes active_area.x1 = 0 (hdr.c:1938), which empties the black-noise
 *    loops: every noise figure is the default 8.0 (hdr.c:329-333);
 *  - fast_randn_init() is never called (hdr.c:391), so the dither of the
 *    20 -> 16 bit step is 0 and the step is deterministic;
 *  - the EV tables are cached per black level in FOUR independent function-local
 *    caches and are NOT rebuilt when only the white level changes
 *    (hdr.c:1080,1240,1575,1672): the white level of the first frame processed
 *    with a given black level sticks.  orc_dualiso_reset() forgets the caches
 *    (a fresh process).
 */
#include "oracle.h"

#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define EVR 32768
#define N20 (1 << 20)
#define CLAMP(x, lo, hi) ((x) < (lo) ? (lo) : ((x) > (hi) ? (hi) : (x)))
#define IMAX(a, b) ((a) > (b) ? (a) : (b))
#define IMIN(a, b) ((a) < (b) ? (a) : (b))
#define IABS(a) ((a) > 0 ? (a) : -(a))

/* ---------------------------------------------------------------- k-th smallest (wirth.h:37-66: any exact selector) */
static int cmp_int(const void *a, const void *b) { int x = *(const int *)a, y = *(const int *)b; return (x > y) - (x < y); }
static int kth_smallest(int *v, int n, int k) { qsort(v, n, sizeof(int), cmp_int); return v[k]; }
static int median_lower(int *v, int n) { return kth_smallest(v, n, (n & 1) ? n / 2 : n / 2 - 1); }

/* ---------------------------------------------------------------- 20-bit EV tables, hdr.c:839-874 */
typedef struct { int black; int *raw2ev; int *ev2raw_base; } evlut_t;

static void lut_build(evlut_t *L, int black, int white)
{
    if (!L->raw2ev) { L->raw2ev = (int *)malloc(sizeof(int) * N20); L->ev2raw_base = (int *)malloc(sizeof(int) * 24 * EVR); }
    int *ev2raw = L->ev2raw_base + 10 * EVR;
    for (int i = 0; i < N20; i++) {
        double signal = i / 64.0 - black / 64.0;
        if (signal < -1023) signal = -1023;
        L->raw2ev[i] = signal > 0 ? (int)round(log2(1 + signal) * EVR) : -(int)round(log2(1 - signal) * EVR);
    }
    for (int i = -10 * EVR; i < 0; i++) {
        double v = black + 64 - round(64 * pow(2, (double)-i / EVR));
        ev2raw[i] = (int)CLAMP(v, 0, black);
    }
    for (int i = 0; i < 14 * EVR; i++) {
        double v = black - 64 + round(64 * pow(2, (double)i / EVR));
        ev2raw[i] = (int)CLAMP(v, black, N20 - 1);
        if (i >= L->raw2ev[white]) ev2raw[i] = IMAX(ev2raw[i], white);
    }
    ev2raw[L->raw2ev[0]] = 0;                      /* keep "bad" pixels */
    L->black = black;
}

static evlut_t g_lut_interp = { -1, 0, 0 }, g_lut_amaze = { -1, 0, 0 }, g_lut_mix = { -1, 0, 0 }, g_lut_blend = { -1, 0, 0 };
static double *g_fullres_curve;
static int g_fullres_black = -1;

void orc_dualiso_reset(void)
{
    g_lut_interp.black = g_lut_amaze.black = g_lut_mix.black = g_lut_blend.black = -1;
    g_fullres_black = -1;
}

static evlut_t *lut_get(evlut_t *L, int black, int white)
{
    if (L->black != black) lut_build(L, black, white);      /* white is NOT part of the key (reference quirk) */
    return L;
}

static const double *fullres_curve(int black)                /* hdr.c:890-913 */
{
    if (!g_fullres_curve) g_fullres_curve = (double *)malloc(sizeof(double) * N20);
    if (g_fullres_black == black) return g_fullres_curve;
    g_fullres_black = black;
    for (int i = 0; i < N20; i++) {
        double sig = i / 64.0 - black / 64.0;
        double ev2 = log2(sig > 1 ? sig : 1);
        double t = ev2 - 4;
        t = CLAMP(t, 0, 4);
        g_fullres_curve[i] = (-cos(t * M_PI / 4) + 1) / 2;
    }
    return g_fullres_curve;
}

/* ---------------------------------------------------------------- hdr_check, hdr.c:407-439 */
static int looks_like_dual_iso(const uint16_t *img, int w, int h, int black, int white)
{
    static double ev[16384];
    static int ev_ready;
    if (!ev_ready) { for (int i = 0; i < 16384; i++) ev[i] = log2((double)i) * EVR; ev_ready = 1; }   /* main.c:136-148 */
#define EVF(p) (((p) - black) >= 0 && ((p) - black) < 16384 ? ev[(p) - black] : 0.0)
    double acc = 0;
    int num = 0;
    for (int y = 2; y < h - 2; y++)
        for (int x = 2; x < w - 2; x++) {
            int p = img[x + y * w], p2 = img[x + (y + 2) * w];
            if ((p > black + 32 || p2 > black + 32) && p < white && p2 < white) {
                double d = EVF(p2) - EVF(p);
                acc += d > 0 ? d : -d;
                num++;
            }
        }
#undef EVF
    return acc / num > 0.5;
}

/* ---------------------------------------------------------------- pattern detection, hdr.c:441-636 */
static int is_rggb(const uint16_t *img, int w, int h)
{
    int *hist = (int *)calloc(4 * 16384, sizeof(int));
    for (int y = 0; y < h / 4 * 4; y++)
        for (int x = 0; x < w; x++) hist[((y % 2) * 2 + (x % 2)) * 16384 + (img[x + y * w] & 16383)]++;
    for (int k = 0; k < 4; k++) { int acc = 0; for (int i = 0; i < 16384; i++) { acc += hist[k * 16384 + i]; hist[k * 16384 + i] = acc; } }
    double d_rggb = 0, d_gbrg = 0;
    for (int i = 0; i < 16384; i++) {
        d_rggb += IABS(hist[16384 + i] - hist[2 * 16384 + i]);
        d_gbrg += IABS(hist[i] - hist[3 * 16384 + i]);
    }
    free(hist);
    return d_rggb < d_gbrg;
}

static int bright_dark_fields(const uint16_t *img, int w, int h, int black, int ay1, int is_bright[4])
{
    const int white = 10000;
    int *hist = (int *)calloc(4 * 16384, sizeof(int));
    for (int y = (ay1 + 3) & ~3; y < h / 4 * 4; y++)
        for (int x = 0; x < w; x++)
            if ((x % 2) != (y % 2)) hist[(y % 4) * 16384 + (img[x + y * w] & 16383)]++;
    int total = 0;
    for (int i = 0; i < 16384; i++) total += hist[i];
    int acc[4] = { 0 }, raw[4] = { 0 }, off[4] = { 0 };
    const int ref_max = (int)(total * 0.998), ref_off = (int)(total * 0.05);
    for (int ref = 0; ref < ref_max; ref++) {
        for (int i = 0; i < 4; i++)
            while (acc[i] < ref) { acc[i] += hist[i * 16384 + raw[i]]; raw[i]++; }
        if (ref < ref_off && IMAX(IMAX(raw[0], raw[1]), IMAX(raw[2], raw[3])) < black + (white - black) / 4)
            memcpy(off, raw, sizeof off);
        if (raw[0] >= white || raw[1] >= white || raw[2] >= white || raw[3] >= white) break;
    }
    free(hist);
    for (int i = 0; i < 4; i++) raw[i] -= off[i];
    int s[4];
    memcpy(s, raw, sizeof s);
    qsort(s, 4, sizeof(int), cmp_int);
    double med = (s[1] + s[2]) / 2;                          /* integer division, hdr.c:614 */
    for (int i = 0; i < 4; i++) is_bright[i] = raw[i] > med;
    if (is_bright[0] + is_bright[1] + is_bright[2] + is_bright[3] != 2) return 0;
    if (is_bright[0] == is_bright[2] || is_bright[1] == is_bright[3]) return 0;
    return 1;
}

static void white_levels(const uint16_t *img, int w, int h, int ay1, const int is_bright[4], int *white_dark, int *white_bright)
{                                                            /* hdr.c:250-300 */
    const int max_pix = w * h / 2 / 9;
    int *px[2] = { (int *)malloc(sizeof(int) * IMAX(max_pix, 1)), (int *)malloc(sizeof(int) * IMAX(max_pix, 1)) };
    int cnt[2] = { 0, 0 };
    for (int y = ay1; y < h; y += 3)
        for (int x = 0; x < w; x += 3) {
            int c = is_bright[y % 4];
            cnt[c] = IMIN(cnt[c], max_pix - 1);
            px[c][cnt[c]] = -(int)img[x + y * w];
            cnt[c]++;
        }
    int w0 = -kth_smallest(px[0], cnt[0], 10) - 100;
    int w1 = -kth_smallest(px[1], cnt[1], 50) - 1500;
    *white_dark = CLAMP(w0, 10000, 16383);
    *white_bright = CLAMP(w1, 5000, 16383);
    free(px[0]); free(px[1]);
}

/* ---------------------------------------------------------------- exposure matching, hdr.c:638-823 */
static int match_exposures(uint32_t *raw, int w, int h, int ay1, int black20, int white20_in, const int is_bright[4],
                           double *corr_ev, int *white_darkened, double *a_out, double *b_out)
{
    const int white20 = IMIN(white20_in, *white_darkened);
    const int black = black20 / 16, white = white20 / 16;
    const int clip0 = white - black, clip = (int)(clip0 * 0.95);
    const int y0 = ay1 + 2;
    int *dark = (int *)calloc((size_t)w * h, sizeof(int)), *bright = (int *)calloc((size_t)w * h, sizeof(int));
#define P16(x, y) ((int)((raw[(x) + (y) * w] >> 4) & 0xFFFF))
    for (int y = y0; y < h - 2; y += 3) {
        int *native = is_bright[y % 4] ? bright : dark, *interp = is_bright[y % 4] ? dark : bright;
        for (int x = 0; x < w; x += 3) {
            int pa = P16(x, y - 2) - black, pb = P16(x, y + 2) - black, pn = P16(x, y) - black;
            int pi = (pa + pb + 1) / 2;
            if (pa >= clip || pb >= clip) pi = clip0;
            if (pi >= clip) pn = clip0;
            interp[x + y * w] = pi;
            native[x + y * w] = pn;
        }
    }
#undef P16
    const int nmax = (w + 2) * (h + 2) / 9;
    int *tmp = (int *)malloc(sizeof(int) * IMAX(nmax, 1));
    int n = 0;
    for (int y = y0; y < h - 2; y += 3)
        for (int x = 0; x < w; x += 3) { int b = bright[x + y * w]; if (b < clip) tmp[n++] = b; }
    int bmed = median_lower(tmp, n);
    int b_lo = kth_smallest(tmp, n, n * 98 / 100), b_hi = kth_smallest(tmp, n, (int)(n * 99.9 / 100));
    n = 0;
    for (int y = y0; y < h - 2; y += 3)
        for (int x = 0; x < w; x += 3) { if (bright[x + y * w] < clip) tmp[n++] = dark[x + y * w]; }
    int dmed = median_lower(tmp, n);

    const int hi_nmax = nmax / 50;
    int hi_n = 0;
    int *hd = (int *)malloc(sizeof(int) * (hi_nmax + h + 8)), *hb = (int *)malloc(sizeof(int) * (hi_nmax + h + 8));
    for (int y = y0; y < h - 2; y += 3)
        for (int x = 0; x < w; x += 3) {
            int d = dark[x + y * w], b = bright[x + y * w];
            if (b >= b_hi || b <= b_lo) continue;
            hd[hi_n] = d; hb[hi_n] = b; hi_n++;
            if (hi_n >= hi_nmax) break;                      /* leaves only the x loop (reference quirk) */
        }
    double a = 0, b = 0;
    int best = 0;
    for (double ev = 0; ev < 6; ev += 0.002) {
        double ta = pow(2, -ev), tb = dmed - bmed * ta;
        int score = 0;
        for (int i = 0; i < hi_n; i++) { int e = hd[i] - (hb[i] * ta + tb); if (IABS(e) < 50) score++; }
        if (score > best) { best = score; a = ta; b = tb; }
    }
    free(hd); free(hb); free(tmp); free(dark); free(bright);

    const double b20 = b * 16;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int p = raw[x + y * w];
            if (p == 0) continue;
            if (is_bright[y % 4]) p = (p - black20) * a + black20 + b20 * a;
            else p = p - b20 + b20 * a;
            raw[x + y * w] = CLAMP(p, 0, 0xFFFFF);
        }
    *white_darkened = (white20 - black20 + b20) * a + black20;
    double factor = 1 / a;
    *a_out = a; *b_out = b;
    if (factor < 1.2 || !isfinite(factor)) return 0;
    *corr_ev = log2(factor);
    return 1;
}

/* ---------------------------------------------------------------- mean23 interpolation, hdr.c:341-368, 1231-1304 */
static int mean2(int a, int b, int white) { return (a >= white || b >= white) ? white : (a + b) / 2; }
static int mean3(int a, int b, int c, int white)
{
    int m = (a + b + c) / 3;
    return (a >= white || b >= white || c >= white) ? IMAX(m, white) : m;
}

static void interpolate_mean23(const uint32_t *raw, uint32_t *dark, uint32_t *bright, int w, int h, int black, int white_lvl,
                               int white_darkened, const int is_bright[4])
{
    evlut_t *L = lut_get(&g_lut_interp, black, white_lvl);
    const int *r2e = L->raw2ev, *e2r = L->ev2raw_base + 10 * EVR;
#define R(x, y) ((int)raw[(x) + (y) * w])
    for (int y = 2; y < h - 2; y++) {
        uint32_t *native = is_bright[y % 4] ? bright : dark, *interp = is_bright[y % 4] ? dark : bright;
        const int white = !is_bright[y % 4] ? white_darkened : white_lvl;
        const int s = (is_bright[y % 4] == is_bright[(y + 1) % 4]) ? -1 : 1;
        for (int x = 2; x < w - 3; x += 2) {
            if (y % 2 == 0) {
                int ri = mean2(r2e[R(x, y - 2)], r2e[R(x, y + 2)], r2e[white]);
                int gi = mean3(r2e[R(x + 2, y + s)], r2e[R(x, y + s)], r2e[R(x + 1, y - 2 * s)], r2e[white]);
                interp[x + y * w] = e2r[ri];
                interp[x + 1 + y * w] = e2r[gi];
            } else {
                int bi = mean2(r2e[R(x + 1, y - 2)], r2e[R(x + 1, y + 2)], r2e[white]);
                int gi = mean3(r2e[R(x + 1, y + s)], r2e[R(x - 1, y + s)], r2e[R(x, y - 2 * s)], r2e[white]);
                interp[x + y * w] = e2r[gi];
                interp[x + 1 + y * w] = e2r[bi];
            }
            native[x + y * w] = R(x, y);
            native[x + 1 + y * w] = R(x + 1, y);
        }
    }
#undef R
}

/* borders (after either interpolator), hdr.c:1306-1353 */
static void interpolate_borders(const uint32_t *raw, uint32_t *dark, uint32_t *bright, int w, int h, const int is_bright[4])
{
#define R(x, y) ((int)raw[(x) + (y) * w])
    for (int y = 0; y < 3; y++)
        for (int x = 0; x < w; x++) {
            uint32_t *native = is_bright[y % 4] ? bright : dark, *interp = is_bright[y % 4] ? dark : bright;
            interp[x + y * w] = R(x, y + 2); native[x + y * w] = R(x, y);
        }
    for (int y = h - 4; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint32_t *native = is_bright[y % 4] ? bright : dark, *interp = is_bright[y % 4] ? dark : bright;
            interp[x + y * w] = R(x, y - 2); native[x + y * w] = R(x, y);
        }
    for (int y = 2; y < h; y++) {
        uint32_t *native = is_bright[y % 4] ? bright : dark, *interp = is_bright[y % 4] ? dark : bright;
        for (int x = 0; x < 2; x++) { interp[x + y * w] = R(x, y - 2); native[x + y * w] = R(x, y); }
        for (int x = w - 3; x < w; x++) { interp[x + y * w] = R(x - 2, y - 2); native[x + y * w] = R(x - 2, y); }
    }
#undef R
}

/* ---------------------------------------------------------------- AMaZE + edge-directed interpolation, hdr.c:914-1229 */
static const struct { signed char ack[2], a[2], b[2], bck[2]; } k_edge_dirs[11] = {   /* y entries are multiplied by s */
    { {-4, 2}, {-2, 1}, { 4, -2}, { 6, -3} }, { {-3, 2}, {-1, 1}, { 3, -2}, { 4, -3} }, { {-2, 2}, {-1, 1}, { 2, -2}, { 3, -3} },
    { {-1, 2}, {-1, 1}, { 1, -2}, { 2, -3} }, { {-1, 2}, { 0, 1}, { 1, -2}, { 1, -3} }, { { 0, 2}, { 0, 1}, { 0, -2}, { 0, -3} },
    { { 1, 2}, { 0, 1}, {-1, -2}, {-1, -3} }, { { 1, 2}, { 1, 1}, {-1, -2}, {-2, -3} }, { { 2, 2}, { 1, 1}, {-2, -2}, {-3, -3} },
    { { 3, 2}, { 1, 1}, {-3, -2}, {-4, -3} }, { { 4, 2}, { 2, 1}, {-4, -2}, {-6, -3} } };

static void interpolate_amaze(const uint32_t *raw, uint32_t *dark, uint32_t *bright, int w, int h, int black, int white_lvl,
                              int white_darkened, const int is_bright[4], double stats_out[2])
{
    const int pitch = w + 16;                                /* hdr.c:969 */
    int *squeezed = (int *)calloc(h, sizeof(int));
    float *cfa = (float *)calloc((size_t)h * pitch, sizeof(float)), *red = (float *)calloc((size_t)h * pitch, sizeof(float)),
          *green = (float *)calloc((size_t)h * pitch, sizeof(float)), *blue = (float *)calloc((size_t)h * pitch, sizeof(float));
    /* squeeze: rows of one exposure become adjacent (dark from the top, bright from h/4*2), greens halved; hdr.c:977-1026 */
    for (int pass = 0; pass < 2; pass++) {
        int yh = -1;
        for (int y = 0; y < h; y++) {
            if (is_bright[y % 4] != pass) continue;
            if (yh < 0) yh = pass ? h / 4 * 2 + y : y;
            for (int x = 0; x < w; x++) {
                int p = raw[x + y * w];
                if (x % 2 != y % 2) p = (p - black) / 2 + black;
                cfa[(size_t)yh * pitch + x] = p;
            }
            squeezed[y] = yh++;
            if (pass && yh >= h) break;                      /* rows that do not fit keep squeezed[y] = 0 (reference quirk) */
        }
    }
    orc_amaze_demosaic(cfa, w, h, pitch, red, green, blue);
    uint32_t *gray = (uint32_t *)malloc((size_t)w * h * 4);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            size_t i = (size_t)y * pitch + x;
            float g = (green[i] - black) * 2 + black;
            green[i] = g < 0xFFFFF ? (g > 0 ? g : 0) : 0xFFFFF;              /* MAX(MIN(x,hi),lo), hdr.c:1046-1048 */
            red[i] = red[i] < 0xFFFFF ? (red[i] > 0 ? red[i] : 0) : 0xFFFFF;
            blue[i] = blue[i] < 0xFFFFF ? (blue[i] > 0 ? blue[i] : 0) : 0xFFFFF;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            size_t i = (size_t)squeezed[y] * pitch + x;
            gray[x + y * w] = green[i] / 2 + red[i] / 4 + blue[i] / 4;
        }
    unsigned char *dir = (unsigned char *)malloc((size_t)w * h);
    memset(dir, 5, (size_t)w * h);
    const double *fc = fullres_curve(black);
    evlut_t *L = lut_get(&g_lut_amaze, black, white_lvl);
    const int *r2e = L->raw2ev, *e2r = L->ev2raw_base + 10 * EVR;
    long semi = 0, notover = 0, deep = 0, notshadow = 0;
    for (int y = 5; y < h - 5; y++) {
        const int s = (is_bright[y % 4] == is_bright[(y + 1) % 4]) ? -1 : 1;
        for (int x = 5; x < w - 5; x++) {
            int search;
            if (!is_bright[y % 4]) { search = !(fc[raw[x + y * w]] > 0.8); if (search) deep++; else notshadow++; }
            else { search = !((int)raw[x + y * w] < white_darkened); if (search) semi++; else notover++; }
            if (!search) continue;
            int e_best = INT_MAX, d_best = 5;
            for (int d = 0; d < 11; d++) {
                int e = 0;
                for (int j = -5; j <= 5; j++) {
#define GEV(q) r2e[gray[x + k_edge_dirs[d].q[0] + j + (y + k_edge_dirs[d].q[1] * s) * w]]
                    const int p1 = GEV(ack), p2 = GEV(a), p3 = GEV(b), p4 = GEV(bck);
#undef GEV
                    e += IABS(p1 - p2) + IABS(p2 - p3) + IABS(p3 - p4);
                }
                e += IABS(d - 5) * EVR / 8;
                if (e < e_best) { e_best = e; d_best = d; }
            }
            dir[x + y * w] = (unsigned char)d_best;
        }
    }
    if (stats_out) { stats_out[0] = semi * 100.0 / (semi + notover); stats_out[1] = deep * 100.0 / (deep + notshadow); }
    for (int y = 2; y < h - 2; y++) {
        uint32_t *native = is_bright[y % 4] ? bright : dark, *interp = is_bright[y % 4] ? dark : bright;
        const int s = (is_bright[y % 4] == is_bright[(y + 1) % 4]) ? -1 : 1;
        for (int x = 2; x < w - 2; x++) {
            const float *plane = (y % 2 == 0) ? (x % 2 == 0 ? red : green) : (x % 2 == 0 ? green : blue);
            const int d = dir[x + y * w], dd[3] = { d, IMIN(d + 1, 10), IMAX(d - 1, 0) };
            int pi[3];
            for (int k = 0; k < 3; k++) {
                int pa = (int)plane[(size_t)squeezed[y + k_edge_dirs[dd[k]].a[1] * s] * pitch + x + k_edge_dirs[dd[k]].a[0]];
                int pb = (int)plane[(size_t)squeezed[y + k_edge_dirs[dd[k]].b[1] * s] * pitch + x + k_edge_dirs[dd[k]].b[0]];
                pa = CLAMP(pa, 0, 0xFFFFF); pb = CLAMP(pb, 0, 0xFFFFF);
                pi[k] = (r2e[pa] * 2 + r2e[pb]) / 3;
            }
            interp[x + y * w] = e2r[(2 * pi[0] + pi[1] + pi[2]) / 4];
            native[x + y * w] = raw[x + y * w];
        }
    }
    free(dir); free(gray); free(cfa); free(red); free(green); free(blue); free(squeezed);
}

/* ---------------------------------------------------------------- alias map, hdr.c:1382-1486 */
static void alias_map_build(uint16_t *amap, const uint32_t *fullres, const uint32_t *halfres, const uint32_t *bright,
                            int w, int h, int dark_noise, int black, const int *r2e)
{
    const double *fc = fullres_curve(black);
    uint16_t *aux = (uint16_t *)malloc((size_t)w * h * 2);
#define SKIP(i) (fc[bright[i]] > 0.8)
    for (int i = 0; i < w * h; i++) {
        if (SKIP(i)) continue;
        int f = fullres[i], hh = halfres[i];
        int e_lin = IABS(f - hh);
        e_lin = IMAX(e_lin - dark_noise * 3 / 2, 0);
        int e_log = IABS(r2e[f] - r2e[hh]);
        amap[i] = IMIN(IMIN(e_lin / 2, e_log / 16), 65530);
    }
    memcpy(aux, amap, (size_t)w * h * 2);
    static const int8_t nb[37][2] = {
        {-2,-6},{0,-6},{2,-6}, {-4,-4},{-2,-4},{0,-4},{2,-4},{4,-4},
        {-6,-2},{-4,-2},{-2,-2},{0,-2},{2,-2},{4,-2},{6,-2}, {-6,0},{-4,0},{-2,0},{0,0},{2,0},{4,0},{6,0},
        {-6,2},{-4,2},{-2,2},{0,2},{2,2},{4,2},{6,2}, {-4,4},{-2,4},{0,4},{2,4},{4,4}, {-2,6},{0,6},{2,6} };
    for (int y = 6; y < h - 6; y++)
        for (int x = 6; x < w - 6; x++) {
            if (SKIP(x + y * w)) continue;
            int v[37];
            for (int k = 0; k < 37; k++) v[k] = -(int)amap[(x + nb[k][0]) + (y + nb[k][1]) * w];
            aux[x + y * w] = (uint16_t)(-kth_smallest(v, 37, 5));
        }
#define A(dx, dy) ((int)aux[(x + (dx)) + (y + (dy)) * w])
    for (int y = 6; y < h - 6; y++)
        for (int x = 6; x < w - 6; x++) {
            if (SKIP(x + y * w)) continue;
            int plus2 = A(0, -2) + A(-2, 0) + A(2, 0) + A(0, 2);
            int diag2 = A(-2, -2) + A(2, -2) + A(-2, 2) + A(2, 2);
            int odd = A(-2, -2) + A(2, -2) + A(-2, -2) + A(2, -2) + A(-2, 2) + A(2, 2) + A(-2, 2) + A(2, 2);   /* sic, hdr.c:1450 */
            int plus6 = A(0, -6) + A(-6, 0) + A(6, 0) + A(0, 6);
            int ring = A(-2, -6) + A(2, -6) + A(-6, -2) + A(6, -2) + A(-6, 2) + A(6, 2) + A(-2, 6) + A(2, 6);
            int c = A(0, 0) + plus2 * 820 / 1024 + diag2 * 657 / 1024 + plus2 * 421 / 1024 + odd * 337 / 1024 +
                    diag2 * 173 / 1024 + plus6 * 139 / 1024 + ring * 111 / 1024 + ring * 57 / 1024;
            amap[x + y * w] = (uint16_t)c;
        }
#undef A
#undef SKIP
    for (int y = 2; y < h - 2; y += 2)
        for (int x = 2; x < w - 2; x += 2) {
            int c = IMAX(IMAX(amap[x + y * w], amap[x + 1 + y * w]), IMAX(amap[x + (y + 1) * w], amap[x + 1 + (y + 1) * w]));
            c = IMIN(c, 15000);
            amap[x + y * w] = amap[x + 1 + y * w] = amap[x + (y + 1) * w] = amap[x + 1 + (y + 1) * w] = (uint16_t)c;
        }
    free(aux);
}

/* ---------------------------------------------------------------- chroma smoothing of the 20-bit planes,
 * hdr.c:1488-1522 (chroma_smooth.c:22-71 instantiated for uint32_t, black = 0, 20-bit tables) */
static void chroma_smooth20(const uint32_t *inp, uint32_t *out, int w, int h, int method, const int *r2e, const int *e2r)
{
    const int reach = method == 5 ? 4 : 2;
    for (int y = 4; y < h - 5; y += 2)
        for (int x = 4; x < w - 4; x += 2) {
            const int ge = (r2e[inp[x + 1 + y * w]] + r2e[inp[x + (y + 1) * w]]) / 2;
            if (ge < 2 * EVR) continue;
            int mr[25], mb[25], k = 0;
            for (int i = -reach; i <= reach; i += 2)
                for (int j = -reach; j <= reach; j += 2) {
                    if (method == 2 && IABS(i) + IABS(j) == 4) continue;
                    const uint32_t *c = inp + (x + i) + (size_t)(y + j) * w;
                    const int g = (r2e[c[1]] + r2e[c[w]]) / 2;
                    mr[k] = r2e[c[0]] - g;
                    mb[k] = r2e[c[w + 1]] - g;
                    k++;
                }
            const int dr = kth_smallest(mr, k, k / 2), db = kth_smallest(mb, k, k / 2);
            if (ge + dr <= EVR || ge + db <= EVR) continue;
            out[x + y * w] = e2r[CLAMP(ge + dr, 0, 14 * EVR - 1)];
            out[x + 1 + (y + 1) * w] = e2r[CLAMP(ge + db, 0, 14 * EVR - 1)];
        }
}

/* ---------------------------------------------------------------- whole conversion */
int orc_cr2hdr20(uint16_t *image, int w_in, int h_in, int black14, int white14, int interp_method, int use_fullres,
                 int use_alias_map, int chroma_smooth_method, int levels_out[2], double scalars_out[8])
{
    levels_out[0] = black14; levels_out[1] = white14;
    if (interp_method != 0 && interp_method != 1) return -1;
    if (interp_method == 0 && (w_in % 4)) return -1;         /* the SSE2 reference leaves part of the green plane unwritten then */
    if (chroma_smooth_method != 2 && chroma_smooth_method != 3 && chroma_smooth_method != 5)
        chroma_smooth_method = 0;                    /* the reference only logs an error (hdr.c:1518) and keeps the unsmoothed copies */
    if (!looks_like_dual_iso(image, w_in, h_in, black14, white14)) return 0;
    int w = w_in, h = h_in;
    if (w <= 0 || h <= 0) return 0;
    uint16_t *img = image;
    const int rggb = is_rggb(img, w, h);
    const int ay1 = rggb ? 0 : 1;                            /* active_area.y1 after the GBRG row skip */
    if (!rggb) { img += w; h--; }                            /* hdr.c:1783-1790 */
    int is_bright[4];
    if (!bright_dark_fields(img, w, h, black14, ay1, is_bright)) return 0;

    const int black = black14 * 64;
    int white = white14 * 64, white_bright;
    { int wd, wb; white_levels(img, w, h, ay1, is_bright, &wd, &wb); white = wd * 64; white_bright = wb * 64; }
    const double dark_noise = 8.0 * 64, dark_noise_ev = 3.0 + 6;      /* hdr.c:329-333, 876-888, 1817-1821 */

    size_t n = (size_t)w * h;
    uint32_t *raw = (uint32_t *)malloc(n * 4);
    for (size_t i = 0; i < n; i++) raw[i] = ((uint32_t)img[i] << 6) & 0xFFFFF;      /* hdr.c:825-837 */
    uint32_t *dark = (uint32_t *)calloc(n, 4), *bright = (uint32_t *)calloc(n, 4), *fullres = (uint32_t *)calloc(n, 4),
             *halfres = (uint32_t *)calloc(n, 4), *fullres_s = fullres, *halfres_s = halfres;
    uint16_t *over = (uint16_t *)calloc(n, 2), *amap = use_alias_map ? (uint16_t *)calloc(n, 2) : NULL;

    double corr_ev = 0, ma = 0, mb = 0;
    int white_darkened = white_bright, ret = 0;
    if (match_exposures(raw, w, h, ay1, black, white, is_bright, &corr_ev, &white_darkened, &ma, &mb)) {
        const double lowiso_dr = log2(white - black) - dark_noise_ev;
        if (interp_method == 0) interpolate_amaze(raw, dark, bright, w, h, black, white, white_darkened, is_bright, NULL);
        else interpolate_mean23(raw, dark, bright, w, h, black, white, white_darkened, is_bright);
        interpolate_borders(raw, dark, bright, w, h, is_bright);
        if (use_fullres)                                     /* hdr.c:1355-1380 */
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++) {
                    size_t i = x + (size_t)y * w;
                    if (is_bright[y % 4]) { int f = bright[i]; fullres[i] = f < white_darkened ? f : IMAX(f, (int)dark[i]); }
                    else fullres[i] = dark[i];
                }
        /* mix_images, hdr.c:1524-1661 */
        double overlap = lowiso_dr - corr_ev;
        overlap -= fmin(3, overlap - 3);
        if (overlap >= 0.5) {
            const double max_ev = log2(white / 64 - black / 64);
            evlut_t *Lm = lut_get(&g_lut_mix, black, white);
            const int *r2e = Lm->raw2ev, *e2r = Lm->ev2raw_base + 10 * EVR;
            for (size_t i = 0; i < n; i++) {
                int b = bright[i], d = dark[i];
                double sig = (b & 0xFFFFF) / 64.0 - black / 64.0;
                double ev = log2(sig > 1 ? sig : 1) + corr_ev;
                double t = ev - (max_ev - overlap);
                t = t < overlap ? t : overlap;
                t = t > 0 ? t : 0;
                double k = (-cos(t * M_PI / overlap) + 1) / 2;
                k = CLAMP(k, 0, 1);
                int mixed = r2e[b] * (1 - k) + r2e[d] * k;
                halfres[i] = e2r[mixed];
            }
            if (chroma_smooth_method) {                      /* hdr.c:1612-1619; without fullres the "smooth" plane IS fullres (all zero) */
                halfres_s = (uint32_t *)malloc(n * 4);
                memcpy(halfres_s, halfres, n * 4);
                chroma_smooth20(halfres, halfres_s, w, h, chroma_smooth_method, r2e, e2r);
                if (use_fullres) {
                    fullres_s = (uint32_t *)malloc(n * 4);
                    memcpy(fullres_s, fullres, n * 4);
                    chroma_smooth20(fullres, fullres_s, w, h, chroma_smooth_method, r2e, e2r);
                }
            }
            if (amap) alias_map_build(amap, fullres_s, halfres_s, bright, w, h, (int)dark_noise, black, r2e);
            uint16_t *aux = (uint16_t *)malloc(n * 2);
            for (size_t i = 0; i < n; i++) over[i] = ((int)bright[i] >= white_darkened || (int)dark[i] >= white) ? 100 : 0;
            memcpy(aux, over, n * 2);
#define O(dx, dy) ((int)aux[(x + (dx)) + (y + (dy)) * w])
            for (int y = 3; y < h - 3; y++)
                for (int x = 3; x < w - 3; x++)
                    over[x + y * w] = (uint16_t)(O(0, 0) + (O(0, -1) + O(-1, 0) + O(1, 0) + O(0, 1)) * 820 / 1024 +
                                                 (O(-1, -1) + O(1, -1) + O(-1, 1) + O(1, 1)) * 657 / 1024);
#undef O
            free(aux);
            /* final_blend, hdr.c:1663-1758 */
            const double *fc = fullres_curve(black);
            evlut_t *Lb = lut_get(&g_lut_blend, black, white);
            const int *br2e = Lb->raw2ev, *be2r = Lb->ev2raw_base + 10 * EVR;
            for (size_t i = 0; i < n; i++) {
                int b = bright[i];
                int hrev = br2e[halfres_s[i]], frev = br2e[fullres[i]], frsev = br2e[fullres_s[i]];
                double f = fc[b & 0xFFFFF], c = 0;
                if (amap) { c = amap[i] / (double)15000; c = CLAMP(c, 0, 1); }
                double ovf = over[i] / 200.0;
                ovf = CLAMP(ovf, 0, 1);
                c = c > ovf ? c : ovf;
                double noisy = ovf > 1 - f ? ovf : 1 - f;
                f = f > c ? f : c;
                double fev = noisy * frsev + (1 - noisy) * frev;
                int sig = ((int)dark[i] + (int)bright[i]) / 2;
                double lim = (double)(sig - black) / (4 * dark_noise);
                double fm = f < lim ? f : lim;
                f = fm > 0 ? fm : 0;
                int out = hrev * (1 - f) + fev * f;
                out = CLAMP(out, -10 * EVR, 14 * EVR - 1);
                raw[i] = be2r[out];
            }
            for (size_t i = 0; i < n; i++) {                 /* hdr.c:1760-1772 (dither term is 0) */
                int v = (int)(raw[i] / 16.0 + 0.0f + 0.5);
                img[i] = (uint16_t)CLAMP(v, 0, 0xFFFF);
            }
            ret = 1;
        }
    }
    if (scalars_out) {
        scalars_out[0] = rggb; scalars_out[1] = is_bright[0] * 8 + is_bright[1] * 4 + is_bright[2] * 2 + is_bright[3];
        scalars_out[2] = white; scalars_out[3] = white_bright; scalars_out[4] = ma; scalars_out[5] = mb;
        scalars_out[6] = corr_ev; scalars_out[7] = white_darkened;
    }
    if (fullres_s != fullres) free(fullres_s);
    if (halfres_s != halfres) free(halfres_s);
    free(raw); free(dark); free(bright); free(fullres); free(halfres); free(over); free(amap);
    if (ret) { levels_out[0] = black14 * 4; levels_out[1] = white14 * 4; }
    return ret;
}
