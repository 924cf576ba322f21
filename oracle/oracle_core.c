/*
 * oracle_core.c -- CPU restatement: EV tables, bit unpack, chroma smoothing,
 * bad/focus pixel repair, vertical-stripe correction, histogram, glibc rand().
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Own code; the reference location of
 * each algorithm is cited per function.  Signed overflow in the reference is
 * de-facto two's-complement wrap: all EV arithmetic
 * here is done in uint32_t and reinterpreted, so the behaviour is defined.
 */
#include "oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* wrap-around helpers                                                       */
static inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static inline int32_t wshl8(int32_t a)           { return (int32_t)((uint32_t)a << 8); }
/* the reference's ABS macro: (a > 0 ? a : -a), -INT_MIN wraps to INT_MIN     */
static inline int32_t wabs(int32_t a)            { return a > 0 ? a : (int32_t)(0u - (uint32_t)a); }
static inline int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }
/* C division that cannot trap: INT_MIN / -1 is a hardware fault in the
 * reference; we define it as INT_MIN (never reached by any test input).      */
static inline int32_t sdiv(int32_t a, int32_t b) { return (a == INT_MIN && b == -1) ? INT_MIN : a / b; }

/* ------------------------------------------------------------------------ */
/* EV tables -- mlvfs/main.c:128-196                                          */

static int32_t ev_of_linear(int i)
{
    /* main.c:163-167: (int)(log2(i) * EV_RESOLUTION); log2(0) = -inf and the
     * x86 cvttsd2si of -inf yields the "integer indefinite" value INT_MIN.    */
    if (i <= 0) return i == 0 ? INT_MIN : 0;
    return (int32_t)(log2((double)i) * ORC_EV_RES);
}

void orc_build_raw2ev(int black, int32_t *out, int n)
{
    for (int p = 0; p < n; p++) {
        int lin = p - black;                     /* pointer pre-offset, main.c:175 */
        out[p] = (lin >= 0 && lin < 16384) ? ev_of_linear(lin) : 0;
    }
}

void orc_build_raw2evf(int black, double *out, int n)
{
    for (int p = 0; p < n; p++) {
        int lin = p - black;                     /* main.c:136-148 */
        out[p] = (lin >= 0 && lin < 16384) ? log2((double)lin) * ORC_EV_RES : 0.0;
    }
}

void orc_build_ev2raw(int32_t *out)
{
    /* main.c:181-196: ev2raw[i] = (int)pow(2, (float)i / EV_RESOLUTION)       */
    for (int i = ORC_EV2RAW_LO; i < ORC_EV2RAW_HI; i++)
        out[i - ORC_EV2RAW_LO] = (int32_t)pow(2.0, (double)((float)i / ORC_EV_RES));
}

/* cached tables for the stage functions below */
static int32_t g_ev2raw[24 * ORC_EV_RES];
static int     g_ev2raw_ready;
static int32_t g_raw2ev[65536];
static int     g_raw2ev_black = -1;

static const int32_t *ev2raw_table(void)
{
    if (!g_ev2raw_ready) { orc_build_ev2raw(g_ev2raw); g_ev2raw_ready = 1; }
    return g_ev2raw - ORC_EV2RAW_LO;             /* indexable by signed EV */
}

static const int32_t *raw2ev_table(int black)
{
    if (black > ORC_MAX_BLACK) return NULL;      /* main.c:170-174 */
    if (g_raw2ev_black != black) {
        /* the reference indexes a 32768-entry static array; anything past it is
         * out of bounds there.  We keep 65536 entries (0 past the table) so that
         * 16-bit inputs stay defined.                                         */
        orc_build_raw2ev(black, g_raw2ev, 65536);
        g_raw2ev_black = black;
    }
    return g_raw2ev;
}

static inline uint16_t ev_to_pixel(const int32_t *ev2raw, int32_t ev, int black)
{
    return (uint16_t)(ev2raw[clampi(ev, 0, 14 * ORC_EV_RES - 1)] + black);
}

/* ------------------------------------------------------------------------ */
/* bit unpack -- mlvfs/dng.c:813-872                                          */
/* Pixel i occupies bits [i*bpp, (i+1)*bpp) of an MSB-first bit stream that is
 * stored as little-endian 16-bit words (mlvfs/raw.h:41-79).                  */
size_t orc_unpack_bits(const uint16_t *packed, uint8_t *out, int64_t offset,
                       size_t max_size, int bpp)
{
    uint32_t first_px  = (uint32_t)(offset > 0 ? offset : 0) / 2;
    uint32_t first_wrd = first_px * (uint32_t)bpp / 16;        /* dng.c:816 */
    size_t   lead      = offset < 0 ? (size_t)(-offset) : 0;
    size_t   out_bytes = max_size - lead;
    uint32_t mask      = (1u << bpp) - 1u;
    uint16_t *dst      = (uint16_t *)(out + lead + offset % 2);
    uint32_t npx       = (uint32_t)(out_bytes / 2);

    for (uint32_t k = 0; k < npx; k++) {
        uint32_t bitpos = (first_px + k) * (uint32_t)bpp;
        uint32_t word   = bitpos / 16 - first_wrd;   /* caller hands us a buffer that starts at first_wrd */
        uint32_t sh     = bitpos % 16;
        uint32_t two    = ((uint32_t)packed[word] << 16) | packed[word + 1];
        /* the reference rotates right by 16 + (32 - bpp) - sh on the LE dword;
         * after the half-swap that is a rotate by (32 - bpp - sh), whose low bpp
         * bits are bits [32-bpp-sh, 32-sh) of `two`.                           */
        uint32_t rot    = (32u - (uint32_t)bpp - sh) & 31u;
        uint32_t v      = rot ? ((two >> rot) | (two << (32u - rot))) : two;
        dst[k] = (uint16_t)(v & mask);
    }
    return max_size;
}

/* ------------------------------------------------------------------------ */
/* chroma smoothing -- mlvfs/cs.c:49-84, mlvfs/chroma_smooth.c:22-71          */

static int32_t median_of(int32_t *v, int n)
{
    /* any exact selector equals opt_med5/9/25 (mlvfs/opt_med.h) on ints       */
    for (int i = 1; i < n; i++) {
        int32_t key = v[i]; int j = i - 1;
        while (j >= 0 && v[j] > key) { v[j + 1] = v[j]; j--; }
        v[j + 1] = key;
    }
    return v[n / 2];
}

int orc_chroma_smooth(uint16_t *img, int w, int h, int black, int method)
{
    if (method != 2 && method != 3 && method != 5) return -1;   /* cs.c:78-80 */
    const int32_t *raw2ev = raw2ev_table(black);
    if (!raw2ev) return 0;                                       /* cs.c:58 */
    const int32_t *ev2raw = ev2raw_table();

    size_t npix = (size_t)w * h;
    uint16_t *src = (uint16_t *)malloc(npix * sizeof(uint16_t)); /* pristine copy, cs.c:60-65 */
    if (!src) return 0;
    memcpy(src, img, npix * sizeof(uint16_t));

    const int reach = (method == 5) ? 4 : 2;                     /* CHROMA_SMOOTH_MAX_IJ */

    for (int y = 4; y < h - 5; y += 2) {
        for (int x = 4; x < w - 4; x += 2) {
            /* green EV of the centre cell; C division truncates toward zero   */
            int32_t ge = wadd(raw2ev[src[x + 1 + y * w]], raw2ev[src[x + (y + 1) * w]]) / 2;
            if (ge < 2 * ORC_EV_RES) continue;                   /* chroma_smooth.c:35 */

            int32_t dr[25], db[25]; int n = 0;
            for (int i = -reach; i <= reach; i += 2) {
                for (int j = -reach; j <= reach; j += 2) {
                    if (method == 2 && abs(i) + abs(j) == 4) continue;   /* plus-shaped 5 */
                    const uint16_t *c0 = src + (x + i) + (size_t)(y + j) * w;
                    const uint16_t *c1 = c0 + w;
                    int32_t gk = wadd(raw2ev[c0[1]], raw2ev[c1[0]]) / 2;
                    dr[n] = wsub(raw2ev[c0[0]], gk);
                    db[n] = wsub(raw2ev[c1[1]], gk);
                    n++;
                }
            }
            int32_t mr = median_of(dr, n);
            int32_t mb = median_of(db, n);
            if (wadd(ge, mr) <= ORC_EV_RES) continue;            /* chroma_smooth.c:64-65 */
            if (wadd(ge, mb) <= ORC_EV_RES) continue;
            img[x + y * w]           = ev_to_pixel(ev2raw, wadd(ge, mr), black);
            img[x + 1 + (y + 1) * w] = ev_to_pixel(ev2raw, wadd(ge, mb), black);
        }
    }
    free(src);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* pixel interpolators -- mlvfs/cs.c:87-168                                    */

typedef struct { uint16_t *img; int w; int black; const int32_t *r2e; const int32_t *e2r; } fixctx_t;

static inline int32_t evp(const fixctx_t *c, int i) { return c->r2e[c->img[i]]; }

/* gradient magnitude between the +-1 / +-3 neighbours along stride s          */
static inline void grads(const fixctx_t *c, int i, int s, int32_t *d_plus, int32_t *d_minus)
{
    *d_plus  = wabs(wsub(evp(c, i + 3 * s), evp(c, i + s)));
    *d_minus = wabs(wsub(evp(c, i - s), evp(c, i - 3 * s)));
}

static inline int32_t wterm(const fixctx_t *c, int i, int32_t coef) { return wmul(evp(c, i), coef) >> 8; }

static void fix_along(const fixctx_t *c, int i, int s)            /* cs.c:87-129 */
{
    int32_t dp, dm; grads(c, i, s, &dp, &dm);
    int32_t sum = wadd(dp, dm);
    if (sum == 0) { c->img[i] = c->img[i + 2 * s]; return; }
    int32_t cp = sdiv(wshl8(wsub(sum, dp)), sum);
    int32_t cm = sdiv(wshl8(wsub(sum, dm)), sum);
    int32_t ev = wadd(wterm(c, i + 2 * s, cp), wterm(c, i - 2 * s, cm));
    c->img[i] = ev_to_pixel(c->e2r, ev, c->black);
}

static void fix_cross(const fixctx_t *c, int i)                   /* cs.c:131-168 */
{
    int32_t vp, vm, hp, hm;
    grads(c, i, c->w, &vp, &vm);
    grads(c, i, 1, &hp, &hm);
    int32_t sum = wadd(wadd(hp, hm), wadd(vp, vm));
    if (sum == 0) { c->img[i] = c->img[i + 2]; return; }
    int32_t den = wmul(3, sum);
    int32_t cvp = sdiv(wshl8(wsub(sum, vp)), den);
    int32_t cvm = sdiv(wshl8(wsub(sum, vm)), den);
    int32_t chp = sdiv(wshl8(wsub(sum, hp)), den);
    int32_t chm = sdiv(wshl8(wsub(sum, hm)), den);
    int32_t ev = wadd(wadd(wterm(c, i + 2 * c->w, cvp), wterm(c, i - 2 * c->w, cvm)),
                      wadd(wterm(c, i + 2, chp), wterm(c, i - 2, chm)));
    c->img[i] = ev_to_pixel(c->e2r, ev, c->black);
}

/* ------------------------------------------------------------------------ */
/* bad pixels -- mlvfs/cs.c:220-331                                            */

size_t orc_detect_bad_pixels(const uint16_t *img, int w, int h, int black,
                             int aggressive, int crop_x, int crop_y,
                             orc_pixel_t *out, size_t cap)
{
    const int32_t *raw2ev = raw2ev_table(black);
    if (!raw2ev) return 0;
    const int dark_lo = black - 12 * 8, dark_hi = black + 12 * 8;    /* cs.c:255-257 */
    size_t count = 0;

    for (int y = 6; y < h - 6; y++) {
        for (int x = 6; x < w - 6; x++) {
            int p = img[x + y * w];
            /* three largest of the 8 same-colour neighbours at distance 2     */
            int top[3] = { -1, -1, -1 };
            for (int dy = -2; dy <= 2; dy += 2)
                for (int dx = -2; dx <= 2; dx += 2) {
                    if (!dx && !dy) continue;
                    int q = img[(x + dx) + (y + dy) * w];
                    if (q >= top[0])      { top[2] = top[1]; top[1] = top[0]; top[0] = q; }
                    else if (q >= top[1]) { top[2] = top[1]; top[1] = q; }
                    else if (q > top[2])  { top[2] = q; }
                }
            int bad = 0;
            if (p < dark_lo) bad = 1;                                              /* cold */
            else if (wsub(raw2ev[p], raw2ev[top[1]]) > 2 * ORC_EV_RES && p > dark_hi) bad = 1;   /* hot */
            else if (aggressive) {
                if ((wsub(raw2ev[p], raw2ev[top[1]]) > ORC_EV_RES ||
                     wsub(raw2ev[p], raw2ev[top[2]]) > ORC_EV_RES) && p > dark_hi) bad = 1;
            }
            if (bad) {
                if (count < cap) { out[count].x = x + crop_x; out[count].y = y + crop_y; }
                count++;
            }
        }
    }
    return count;
}

void orc_apply_bad_pixels(uint16_t *img, int w, int h, int black,
                          const orc_pixel_t *map, size_t count,
                          int crop_x, int crop_y, int dual_iso)
{
    fixctx_t c = { img, w, black, raw2ev_table(black), ev2raw_table() };
    if (!c.r2e) return;
    for (size_t m = 0; m < count; m++) {                       /* list order matters */
        int x = map[m].x - crop_x, y = map[m].y - crop_y;
        if (x > 2 && x < w - 3 && y > 2 && y < h - 3) {
            if (dual_iso) fix_along(&c, x + y * w, 1); else fix_cross(&c, x + y * w);
        }
    }
}

void orc_apply_focus_pixels(uint16_t *img, int w, int h, int black,
                            const orc_pixel_t *map, size_t count,
                            int crop_x, int crop_y, int dual_iso)
{
    fixctx_t c = { img, w, black, raw2ev_table(black), ev2raw_table() };
    if (!c.r2e) return;
    for (size_t m = 0; m < count; m++) {
        int x = map[m].x - crop_x, y = map[m].y - crop_y;
        int i = x + y * w;
        if (x > 2 && x < w - 3 && y > 2 && y < h - 3) {
            if (dual_iso) fix_along(&c, i, 1); else fix_cross(&c, i);
        } else if (i > 0 && i < w * h) {                       /* cs.c:479-500 */
            int h_edge = (x >= w - 3 && x < w) || (x >= 0 && x <= 3);
            int v_edge = (y >= h - 3 && y < h) || (y >= 0 && y <= 3);
            if (h_edge && !v_edge && !dual_iso) fix_along(&c, i, w);
            else if (v_edge && !h_edge)         fix_along(&c, i, 1);
            else if (x >= 0 && x <= 3)          img[i] = img[i + 2];
            else if (x >= w - 3 && x < w)       img[i] = img[i - 2];
        }
    }
}

/* ------------------------------------------------------------------------ */
/* vertical stripes -- mlvfs/stripes.c:108-266                                 */

/* The 24 add_pixel calls of one 8-pixel group (stripes.c:175-203), as
 * (histogram, index of reference pixel, index of corrected pixel) with pixel
 * indices 0..9 = pa..ph,pa2,pb2.                                             */
static const uint8_t k_pairs[24][3] = {
    {2,0,2},{2,0,2},{2,0,2},{2,8,2},  {3,1,3},{3,1,3},{3,1,3},{3,9,3},
    {4,0,4},{4,0,4},{4,8,4},{4,8,4},  {5,1,5},{5,1,5},{5,9,5},{5,9,5},
    {6,0,6},{6,8,6},{6,8,6},{6,8,6},  {7,1,7},{7,9,7},{7,9,7},{7,9,7},
};

int orc_stripes_compute(const uint16_t *img, int w, int h, int black, int white,
                        int frame_size, int (*rand_fn)(void), int32_t coeffs[8],
                        int32_t *hist_out, int32_t *num_out)
{
    if (!rand_fn) rand_fn = rand;
    int32_t *hist = (int32_t *)calloc(8 * 65536, sizeof(int32_t));
    int32_t num[8] = { 0 };
    const double too_bright = white / 1.5;                      /* stripes.c:116 */

    for (int y = 0; y < h; y++) {
        const uint16_t *row = img + (size_t)y * w;
        for (int x = 0; x < w - 10; x += 8) {
            int px[10];
            for (int k = 0; k < 10; k++) px[k] = row[x + k] - black;
            for (int c = 0; c < 24; c++) {
                int a = px[k_pairs[c][1]], b = px[k_pairs[c][2]];
                int lo = a < b ? a : b, hi = a < b ? b : a;
                if (lo < 32) continue;                          /* too noisy */
                if (hi > too_bright) continue;                  /* too bright */
                double af = a + (rand_fn() % 1024) / 1024.0 - 0.5;   /* dither, stripes.c:129-130 */
                double bf = b + (rand_fn() % 1024) / 1024.0 - 0.5;
                double ev = log2(af / bf);
                int bin = clampi((int)(65536 / 2 + ev * 65536 / 2), 0, 65535);   /* F2H */
                hist[k_pairs[c][0] * 65536 + bin]++;
                num[k_pairs[c][0]]++;
            }
        }
    }

    for (int j = 0; j < 8; j++) {                               /* stripes.c:218-234 */
        if (num[j] < frame_size / 128) continue;
        int t = 0;
        for (int k = 0; k < 65536; k++) {
            t += hist[j * 65536 + k];
            if (t >= num[j] / 2) {
                coeffs[j] = (int32_t)(pow(2.0, (double)(k - 32768) / 32768) * 65536);
                break;
            }
        }
    }
    coeffs[0] = coeffs[1] = 65536;

    int needed = 0;
    for (int j = 0; j < 8; j++) {
        double c = (double)coeffs[j] / 65536;
        if (c < 0.998 || c > 1.002) needed = 1;
    }
    if (hist_out) memcpy(hist_out, hist, 8 * 65536 * sizeof(int32_t));
    if (num_out)  memcpy(num_out, num, sizeof(num));
    free(hist);
    return needed;
}

void orc_stripes_apply(uint16_t *img, size_t npix, int w, int black, int white,
                       int needed, const int32_t coeffs[8], int64_t offset)
{
    if (!needed) return;                                        /* stripes.c:252 */
    if (w % 8 != 0) return;
    uint16_t blk = (uint16_t)black, wht = (uint16_t)white;
    size_t start = (size_t)(offset % 8);
    for (size_t i = 0; i < npix; i++) {
        double coef = coeffs[(i + start) % 8];
        if (coef != 0 && img[i] > blk + 64) {
            /* every operand is an integer below 2^53: the double expression of
             * stripes.c:263 is evaluated exactly as written.                   */
            double v = (img[i] - blk) * coef / 65536 + blk;
            img[i] = (uint16_t)(wht < v ? wht : v);
        }
    }
}

/* ------------------------------------------------------------------------ */
/* histogram -- mlvfs/histogram.c:33-84 (16-bit counters that wrap)            */

orc_hist_t *orc_hist_create(uint16_t white)
{
    orc_hist_t *h = (orc_hist_t *)malloc(sizeof *h);
    if (!h) return NULL;
    h->white = white; h->count = 0;
    h->bins = (uint16_t *)calloc((size_t)white + 1, sizeof(uint16_t));
    return h;
}

void orc_hist_add(orc_hist_t *h, const uint16_t *data, uint32_t size, uint16_t skip)
{
    uint32_t step = (uint32_t)skip + 1;
    for (uint32_t i = 0; i < size; i += step) {
        uint16_t v = data[i] < h->white ? data[i] : h->white;
        h->bins[v]++;
    }
    h->count += size / step;
}

uint16_t orc_hist_median(const orc_hist_t *h)
{
    uint32_t half = h->count / 2, acc = 0;
    for (uint32_t i = 0; i <= h->white; i++) {
        acc += h->bins[i];
        if (acc > half) return (uint16_t)i;
    }
    return 0;
}

void orc_hist_destroy(orc_hist_t *h) { if (h) { free(h->bins); free(h); } }

/* ------------------------------------------------------------------------ */
/* glibc rand(): TYPE_3 additive-feedback generator (public algorithm,
 * glibc stdlib/random_r.c; not vendored by the reference, system glibc).     */

void orc_rand_seed(orc_rand_t *st, unsigned seed)
{
    /* srandom_r: x[0] = seed, x[i] = 16807 * x[i-1] mod (2^31 - 1) (Schrage),
     * x[31..33] = x[0..2], then x[i] = x[i-31] + x[i-3] (mod 2^32); the first
     * 310 generated words are discarded.                                      */
    uint32_t x[34 + 310];
    int32_t word = (int32_t)(seed ? seed : 1);
    x[0] = (uint32_t)word;
    for (int i = 1; i < 31; i++) {
        int64_t hi = word / 127773, lo = word % 127773;
        int64_t nxt = 16807 * lo - 2836 * hi;
        if (nxt < 0) nxt += 2147483647;
        word = (int32_t)nxt;
        x[i] = (uint32_t)word;
    }
    for (int i = 31; i < 34; i++) x[i] = x[i - 31];
    for (int i = 34; i < 34 + 310; i++) x[i] = x[i - 31] + x[i - 3];
    for (int i = 0; i < 31; i++) st->ring[i] = (int32_t)x[34 + 310 - 31 + i];
    st->pos = 0;
}

int orc_rand_next(orc_rand_t *st)
{
    /* x[n] = x[n-31] + x[n-3]; ring[pos] is x[n-31], ring[(pos+28)%31] is x[n-3] */
    uint32_t v = (uint32_t)st->ring[st->pos] + (uint32_t)st->ring[(st->pos + 28) % 31];
    st->ring[st->pos] = (int32_t)v;
    st->pos = (st->pos + 1) % 31;
    return (int)(v >> 1);
}

/* ------------------------------------------------------------------------ */
/* Row-shard form of the stripes histogram (for the multi-GPU host logic tests):
 * rows [row0,row1) only, dither values taken from a caller-supplied stream of
 * rand()%1024 values (two per accepted call, in raster order).  rnd == NULL just
 * counts the accepted calls.  Same arithmetic as orc_stripes_compute above.      */
int64_t orc_stripes_hist_rows(const uint16_t *img, int w, int row0, int row1, int black, int white,
                              const uint16_t *rnd, int64_t n_rnd, int32_t *hist, int32_t *num)
{
    const double too_bright = white / 1.5;
    int64_t calls = 0;
    for (int y = row0; y < row1; y++) {
        const uint16_t *row = img + (size_t)y * w;
        for (int x = 0; x < w - 10; x += 8) {
            int px[10];
            for (int k = 0; k < 10; k++) px[k] = row[x + k] - black;
            for (int c = 0; c < 24; c++) {
                int a = px[k_pairs[c][1]], b = px[k_pairs[c][2]];
                int lo = a < b ? a : b, hi = a < b ? b : a;
                if (lo < 32 || hi > too_bright) continue;
                if (rnd && 2 * calls + 1 < n_rnd) {
                    double af = a + rnd[2 * calls] / 1024.0 - 0.5;
                    double bf = b + rnd[2 * calls + 1] / 1024.0 - 0.5;
                    int bin = clampi((int)(65536 / 2 + log2(af / bf) * 65536 / 2), 0, 65535);
                    hist[k_pairs[c][0] * 65536 + bin]++;
                    num[k_pairs[c][0]]++;
                }
                calls++;
            }
        }
    }
    return calls;
}
