/*
 * oracle_hdr_preview.c -- CPU restatement of the fast dual-ISO preview
 * conversion, mlvfs/hdr.c:40-227 (hdr_convert_data).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Own code.
 *
 * Notes on behaviour that is reproduced on purpose:
 *  - the four row-phase green histograms use 16-bit counters and the
 *    `count += size / (skip + 1)` bookkeeping of mlvfs/histogram.c:57-64;
 *  - the CDF-matching loop of the reference (hdr.c:118-141) walks raw_hi up to
 *    the sample COUNT, i.e. it can index the histograms past `white`; every such
 *    read happens after the cumulative sum has reached its total, where no data
 *    point can be recorded any more, so reads past the table are treated as 0
 *    here and the catch-up loop is bounded;
 *  - double -> uint16_t conversions follow x86-64 gcc (cvttsd2si to 32 bits,
 *    low 16 bits kept);
 *  - rows are processed top-down in place (hdr.c:178-213): a row sees the
 *    already rewritten row y-2 and the not yet rewritten row y+2.
 *  The focus-pixel repair the reference performs between detection and
 *  matching (hdr.c:104) is the caller's business here (orc_apply_focus_pixels).
 */
#include "oracle.h"

#include <stdlib.h>

static inline uint16_t d2u16(double v) { return (uint16_t)(int32_t)v; }

static inline uint32_t bin_at(const orc_hist_t *h, int i)
{
    return (i >= 0 && i <= (int)h->white) ? h->bins[i] : 0u;
}

int orc_hdr_preview(uint16_t *img, int w_in, int h_in, int black_in, int white_in,
                    size_t max_size, double *a_out, double *b_out, int *dark_row_start_out)
{
    const uint16_t width = (uint16_t)w_in, height = (uint16_t)h_in;
    const uint16_t black = (uint16_t)black_in, white = (uint16_t)white_in;

    /* median green of every 4th row phase, hdr.c:52-64 */
    orc_hist_t *hist[4];
    int med[4];
    for (int i = 0; i < 4; i++) hist[i] = orc_hist_create(white);
    for (uint16_t y = 4; y < height - 4; y += 5) {
        int first_green = (y + 1) % 2;
        orc_hist_add(hist[y % 4], img + (size_t)y * width + first_green,
                     (uint32_t)(width - first_green), 3);
    }
    for (int i = 0; i < 4; i++) med[i] = (int)orc_hist_median(hist[i]) - black;

    /* which two of four consecutive rows are the dark ones, hdr.c:66-102 */
    static const int8_t layouts[4][6] = {
        /* bright pair, dark pair, lo hist, hi hist */
        { 2, 3, 0, 1, 0, 2 },
        { 0, 3, 1, 2, 1, 0 },
        { 0, 1, 2, 3, 2, 0 },
        { 1, 2, 0, 3, 0, 2 },
    };
    int start = -1;
    for (int k = 0; k < 4 && start < 0; k++) {
        const int8_t *L = layouts[k];
        /* the reference tests bright > 2*dark in a fixed order of four comparisons */
        if (med[L[0]] > med[L[2]] * 2 && med[L[0]] > med[L[3]] * 2 &&
            med[L[1]] > med[L[2]] * 2 && med[L[1]] > med[L[3]] * 2)
            start = k;
    }
    if (start < 0) {
        for (int i = 0; i < 4; i++) orc_hist_destroy(hist[i]);
        return 0;
    }
    const orc_hist_t *lo = hist[layouts[start][4]], *hi = hist[layouts[start][5]];

    /* dark-as-a-function-of-bright curve from the two CDFs, hdr.c:106-141 */
    const int min_pix = 100;
    int cap = width * height / min_pix + 1;
    int *px = (int *)malloc(sizeof(int) * cap);
    int *py = (int *)malloc(sizeof(int) * cap);
    double *pw = (double *)malloc(sizeof(double) * cap);
    int n = 0, acc_lo = 0, acc_hi = 0, raw_lo = 0, prev_acc_hi = 0;
    const int total = (int)hist[0]->count;

    for (int raw_hi = 0; raw_hi < total; raw_hi++) {
        acc_hi += (int)bin_at(hi, raw_hi);
        while (acc_lo < acc_hi && raw_lo <= 65536) { acc_lo += (int)bin_at(lo, raw_lo); raw_lo++; }
        if (raw_lo >= white) break;
        if (acc_hi - prev_acc_hi > min_pix) {
            if (acc_hi > total * 1 / 100 && acc_hi < total * 99.99 / 100) {
                int xb = raw_hi - black;
                px[n] = xb; py[n] = raw_lo - black;
                pw[n] = (double)(xb + 100 > 0 ? xb + 100 : 0);
                n++;
                prev_acc_hi = acc_hi;
            }
        }
    }

    /* weighted least squares, summation order of hdr.c:150-166 */
    double mx = 0, my = 0, mxy = 0, mx2 = 0, wsum = 0;
    for (int i = 0; i < n; i++) {
        mx  += px[i] * pw[i];
        my  += py[i] * pw[i];
        mxy += (double)px[i] * py[i] * pw[i];
        mx2 += (double)px[i] * px[i] * pw[i];
        wsum += pw[i];
    }
    mx /= wsum; my /= wsum; mxy /= wsum; mx2 /= wsum;
    const double a = (mxy - mx * my) / (mx2 - mx * mx);
    const double b = my - a * mx;
    free(px); free(py); free(pw);
    for (int i = 0; i < 4; i++) orc_hist_destroy(hist[i]);

    const uint16_t shadow = d2u16(black + 1 / (a * a) + b);

    #define SCALED(p) ({ double _v = ((p) - black) * a + black + b; (double)white < _v ? (double)white : _v; })
    for (int y = 0; y < height; y++) {
        uint16_t *row = img + (size_t)y * width;
        const int up = -2 * (int)width, dn = 2 * (int)width;
        if (((y - start + 4) % 4) >= 2) {                       /* bright row */
            for (int x = 0; x < width; x++) {
                if (row[x] >= white)
                    row[x] = (uint16_t)(y > 2 ? (y < height - 2 ? (row[x + up] + row[x + dn]) / 2 : row[x + up])
                                              : row[x + dn]);
                else
                    row[x] = d2u16(SCALED(row[x]));
            }
        } else {                                                /* dark row */
            for (int x = 0; x < width; x++) {
                if (row[x] < shadow) {
                    double v = y > 2 ? (y < height - 2 ? (row[x + up] + SCALED(row[x + dn])) / 2
                                                       : (double)row[x + up])
                                     : SCALED(row[x + dn]);
                    row[x] = d2u16(v);
                }
            }
        }
    }
    #undef SCALED

    size_t count = max_size / 2;                                /* hdr.c:217-222 */
    for (size_t i = 0; i < count; i++) img[i] = (uint16_t)(img[i] << 2);

    if (a_out) *a_out = a;
    if (b_out) *b_out = b;
    if (dark_row_start_out) *dark_row_start_out = start;
    return 1;
}
