/*
 * oracle_patternnoise.c -- CPU restatement of the row/column pattern-noise
 * correction, mlvfs/patternnoise.c:49-380 (fix_pattern_noise; debug views: orc_fix_pattern_noise_dbg).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Own code.
 *
 * Structure of the algorithm (per direction; the row pass is the column pass on
 * the transposed frame, patternnoise.c:371-378):
 *   1. split the Bayer frame into four half-resolution planes (r, g1, g2, b);
 *   2. per plane pixel, take the run of horizontally adjacent pixels whose
 *      average green stays within `thr` of the centre (at most 25 to the left,
 *      24 to the right) and replace g1, g2, r-avg_g, b-avg_g by the LOWER median
 *      over that run (patternnoise.c:88-180);
 *   3. noise = original - smoothed; per column the lower median of the noise
 *      over unmasked pixels (mask: |in[i-2]-in[i+2]| > 500 on the FLAT plane
 *      array, or pixel >= white) becomes the column offset (needs >= 10 samples);
 *   4. subtract the offsets (clamp to +-32767), then remove the lower median of
 *      all offsets and clamp to [0, 32760] (patternnoise.c:185-282).
 * All plane values are int16_t and wrap like the reference's int16 stores.
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

static int cmp_int(const void *a, const void *b)
{
    int x = *(const int *)a, y = *(const int *)b;
    return (x > y) - (x < y);
}

/* lower median: element (n-1)/2 of the sorted run (wirth.h:129-131) */
static int lower_median_i16(const int16_t *v, int n)
{
    int tmp[64];
    for (int i = 0; i < n; i++) tmp[i] = v[i];
    qsort(tmp, n, sizeof(int), cmp_int);
    return tmp[(n & 1) ? n / 2 : n / 2 - 1];
}

static int lower_median_int(int *v, int n)
{
    qsort(v, n, sizeof(int), cmp_int);
    return v[(n & 1) ? n / 2 : n / 2 - 1];
}

typedef struct { int16_t *p[4]; } planes_t;       /* r, g1, g2, b */

static void smooth_rows(const planes_t *in, const planes_t *out, int w, int h, int reach, int thr)
{
    size_t n = (size_t)w * h;
    int16_t *avg = (int16_t *)malloc(n * 2), *drg = (int16_t *)malloc(n * 2), *dbg = (int16_t *)malloc(n * 2);
    for (size_t i = 0; i < n; i++) {
        avg[i] = (int16_t)(((int)in->p[1][i] + (int)in->p[2][i]) / 2);
        drg[i] = (int16_t)(in->p[0][i] - avg[i]);
        dbg[i] = (int16_t)(in->p[3][i] - avg[i]);
    }
    for (int y = 0; y < h; y++) {
        const int16_t *ag = avg + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            int centre = ag[x];
            int hi_lim = x + reach < w ? x + reach : w;         /* exclusive */
            int lo_lim = x - reach > 0 ? x - reach : 0;         /* inclusive */
            int xr = x + 1, xl = x - 1;
            while (xr < hi_lim && abs(ag[xr] - centre) <= thr) xr++;
            while (xl >= lo_lim && abs(ag[xl] - centre) <= thr) xl--;
            int cnt = xr - xl - 1;
            size_t o = (size_t)y * w + xl + 1, at = (size_t)y * w + x;
            int mg1 = lower_median_i16(in->p[1] + o, cnt);
            int mg2 = lower_median_i16(in->p[2] + o, cnt);
            int mg  = (mg1 + mg2) / 2;
            out->p[1][at] = (int16_t)mg1;
            out->p[2][at] = (int16_t)mg2;
            out->p[0][at] = (int16_t)(lower_median_i16(drg + o, cnt) + mg);
            out->p[3][at] = (int16_t)(lower_median_i16(dbg + o, cnt) + mg);
        }
    }
    free(avg); free(drg); free(dbg);
}

static void remove_column_offsets(int16_t *orig, const int16_t *smooth, int w, int h, int white, int flags)
{
    size_t n = (size_t)w * h;
    int16_t *noise = (int16_t *)malloc(n * 2);
    uint8_t *masked = (uint8_t *)malloc(n);
    int *offs = (int *)malloc(sizeof(int) * w);
    int *col = (int *)malloc(sizeof(int) * (w > h ? w : h));

    for (size_t i = 0; i < n; i++) {
        noise[i] = (int16_t)(orig[i] - smooth[i]);
        int16_t grad = (i >= 2 && i + 2 < n) ? (int16_t)(orig[i - 2] - orig[i + 2]) : 0;
        masked[i] = (abs((int)grad) > 500) || (orig[i] >= white);
    }
    if (flags & (2 | 4 | 8)) {                  /* patternnoise.c:215-240: debug views instead of the correction */
        for (size_t i = 0; i < n; i++)
            orig[i] = (flags & 2) ? smooth[i] : (flags & 4) ? (int16_t)((masked[i] ? -100 : noise[i]) + 100) : (int16_t)(masked[i] * 1000);
        free(noise); free(masked); free(offs); free(col);
        return;
    }
    for (int x = 0; x < w; x++) {
        int k = 0;
        for (int y = 0; y < h; y++)
            if (!masked[x + (size_t)y * w]) col[k++] = noise[x + (size_t)y * w];
        offs[x] = (k < 10) ? 0 : -lower_median_int(col, k);
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int v = (int)orig[x + (size_t)y * w] + offs[x];
            orig[x + (size_t)y * w] = (int16_t)(v < -32767 ? -32767 : (v > 32767 ? 32767 : v));
        }
    int mc = lower_median_int(offs, w);
    for (size_t i = 0; i < n; i++) {
        int v = (int)orig[i] - mc;
        orig[i] = (int16_t)(v < 0 ? 0 : (v > 32760 ? 32760 : v));
    }
    free(noise); free(masked); free(offs); free(col);
}

static void column_pass(int16_t *raw, int w, int h, int white, int flags)
{
    int hw = w / 2, hh = h / 2;
    size_t n = (size_t)hw * hh;
    planes_t src, den;
    for (int c = 0; c < 4; c++) { src.p[c] = (int16_t *)malloc(n * 2); den.p[c] = (int16_t *)malloc(n * 2); }
    /* plane c sits at (dx, dy) = (c & 1, c >> 1): r(0,0) g1(1,0) g2(0,1) b(1,1) */
    for (int c = 0; c < 4; c++)
        for (int y = c >> 1; y < h; y += 2)
            for (int x = c & 1; x < w; x += 2)
                src.p[c][(x / 2) + (size_t)(y / 2) * hw] = raw[x + (size_t)y * w];

    smooth_rows(&src, &den, hw, hh, 50 / 2, 500);
    for (int c = 0; c < 4; c++) remove_column_offsets(src.p[c], den.p[c], hw, hh, white, flags);

    for (int c = 0; c < 4; c++)
        for (int y = c >> 1; y < h; y += 2)
            for (int x = c & 1; x < w; x += 2)
                raw[x + (size_t)y * w] = src.p[c][(x / 2) + (size_t)(y / 2) * hw];
    for (int c = 0; c < 4; c++) { free(src.p[c]); free(den.p[c]); }
}

/* flags: the reference's debug_flags (patternnoise.h:19-24; patternnoise.c:363-379: with any flag set only one direction runs --
 * bit 0 chooses the row direction --, bits 1..3 choose a view) */
void orc_fix_pattern_noise_dbg(int16_t *raw, int w, int h, int white, int flags)
{
    if (!flags || !(flags & 1)) column_pass(raw, w, h, white, flags);
    if (flags && !(flags & 1)) return;

    size_t n = (size_t)w * h;
    int16_t *t = (int16_t *)malloc(n * 2);
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) t[y + (size_t)x * h] = raw[x + (size_t)y * w];
    column_pass(t, h, w, white, flags);
    for (int y = 0; y < w; y++) for (int x = 0; x < h; x++) raw[y + (size_t)x * w] = t[x + (size_t)y * h];
    free(t);
}

void orc_fix_pattern_noise(int16_t *raw, int w, int h, int white) { orc_fix_pattern_noise_dbg(raw, w, h, white, 0); }
