/*
 * oracle_lj92.c -- CPU restatement of the reference's lossless-JPEG (ITU T.81 process 14, "LJ92") frame decoder and of
 * the untiling MLVFS applies to it.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 *   orc_lj92_info     mlvfs/lj92.c:82-94 (find), 96-274 (parseHuff, direct-LUT build), 276-283 (parseSof3), 285-290
 *                     (parseBlock), 595-626 (parseImage, findSoI), 650-687 (lj92_open)
 *   orc_lj92_decode   mlvfs/lj92.c:344-406 (nextdiff), 408-510 (parsePred6), 512-593 (parseScan), 689-702 (lj92_decode with
 *                     writeLength = width*height, skipLength = 0, no linearisation: how main.c:639 calls it)
 *   orc_lj92_untile   mlvfs/main.c:646-667 (rows and columns of the decoded image are de-interleaved: even first)
 *
 * Pinned against the reference's own lj92.c (one source file, built into oracle/_ref): streams made by the reference's
 * lj92_encode and hand-made streams with every predictor must decode identically (tests/test_lj92.py).
 * Scope of the restatement: well-formed streams.  On damaged streams the reference reads uninitialised table entries and
 * past its buffers; here such streams end in ORC_LJ92_CORRUPT and the two are not compared.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_LJ92_OK 0
#define ORC_LJ92_CORRUPT (-1)

typedef struct {
    int width, height, bits, predictor;
    int huffbits;                  /* longest code */
    int scan_offset;               /* first byte of entropy-coded data */
    uint8_t codelen[17];           /* BITS: codes of each length 1..16 */
    uint8_t values[256];           /* HUFFVAL in code order */
    int nvalues;
} orc_lj92_hdr;

static int be16(const uint8_t *p) { return (p[0] << 8) | p[1]; }

/* the reference's marker search: next 0xFF byte from ix, marker = the byte after it, ix ends two past the 0xFF */
static int next_marker(const uint8_t *d, int len, int *ix)
{
    int i = *ix;
    while (d[i] != 0xFF && i < len - 1) i++;
    i += 2;
    if (i >= len) return -1;
    *ix = i;
    return d[i - 1];
}

int orc_lj92_info(const uint8_t *d, int len, orc_lj92_hdr *h)
{
    memset(h, 0, sizeof *h);
    int ix = 0, have_tab = 0;
    if (len < 4 || next_marker(d, len, &ix) != 0xD8) return ORC_LJ92_CORRUPT;
    for (;;) {
        const int m = next_marker(d, len, &ix);
        if (m == 0xC4) {                                   /* DHT: only the last table counts; ix is NOT advanced (lj92.c:96-274) */
            const uint8_t *t = d + ix;
            if (ix + be16(t) >= len) return ORC_LJ92_CORRUPT;
            int n = 0;
            h->codelen[0] = 0;
            for (int i = 1; i <= 16; i++) { h->codelen[i] = t[2 + i]; n += t[2 + i]; }
            if (n > 256) return ORC_LJ92_CORRUPT;
            memcpy(h->values, t + 19, (size_t)n);
            h->nvalues = n;
            h->huffbits = 16;
            while (h->huffbits > 0 && !h->codelen[h->huffbits]) h->huffbits--;
            have_tab = 1;
        } else if (m == 0xC3) {                            /* SOF3 */
            if (ix + 6 >= len) return ORC_LJ92_CORRUPT;
            h->bits = d[ix + 2];
            h->height = be16(d + ix + 3);
            h->width = be16(d + ix + 5);
            ix += be16(d + ix);
        } else if (m == 0xD9) {                            /* EOI before any scan: open succeeds, decode would not */
            break;
        } else if (m == 0xDA) {                            /* SOS */
            const int ncomp = d[ix + 2];
            h->predictor = d[ix + 3 + 2 * ncomp];
            h->scan_offset = ix + be16(d + ix);
            break;
        } else if (m == -1) {
            return ORC_LJ92_CORRUPT;
        } else {                                           /* comment and anything else: skipped by length */
            ix += be16(d + ix);
            if (ix >= len) return ORC_LJ92_CORRUPT;
        }
    }
    if (h->width <= 0 || !have_tab || !h->scan_offset || h->predictor < 0 || h->predictor > 7) return ORC_LJ92_CORRUPT;
    return ORC_LJ92_OK;
}

/* canonical code -> direct table indexed by the next huffbits bits: (ssss << 8) | code length */
static uint16_t *build_lut(const orc_lj92_hdr *h)
{
    const int n = 1 << h->huffbits;
    uint16_t *lut = (uint16_t *)calloc((size_t)n, sizeof *lut);     /* unfilled entries stay 0 = "no such code" */
    if (!lut) return NULL;
    int i = 0, v = 0;
    for (int len = 1; len <= h->huffbits; len++)
        for (int k = 0; k < h->codelen[len]; k++, v++) {
            const int span = 1 << (h->huffbits - len);
            for (int r = 0; r < span && i < n; r++) lut[i++] = (uint16_t)((h->values[v] << 8) | len);
        }
    return lut;
}

typedef struct { const uint8_t *d; int len, ix; uint64_t acc; int cnt; int overrun; } bitreader;

/* bytes of the scan with the stuffed zero after every 0xFF removed, MSB first */
static void refill(bitreader *b, int need)
{
    while (b->cnt < need) {
        int byte = 0;
        if (b->ix < b->len) {
            byte = b->d[b->ix++];
            if (byte == 0xFF && b->ix < b->len && b->d[b->ix] == 0) b->ix++;
        } else b->overrun = 1;
        b->acc = (b->acc << 8) | (uint64_t)byte;
        b->cnt += 8;
    }
}

static int next_diff(bitreader *b, const uint16_t *lut, int huffbits, int *bad)
{
    refill(b, huffbits);
    const uint16_t e = lut[(b->acc >> (b->cnt - huffbits)) & ((1u << huffbits) - 1)];
    const int used = e & 0xFF, t = e >> 8;
    if (!used || t > 16) { *bad = 1; return 0; }
    b->cnt -= used;
    if (!t) return 0;
    refill(b, t);
    b->cnt -= t;
    int diff = (int)((b->acc >> b->cnt) & ((1u << t) - 1));
    if (diff < (1 << (t - 1))) diff += (int)(0xFFFFFFFFu << t) + 1;       /* negative half: value - (2^t - 1) */
    return diff;
}

int orc_lj92_decode(const uint8_t *d, int len, uint16_t *out)
{
    orc_lj92_hdr h;
    if (orc_lj92_info(d, len, &h) != ORC_LJ92_OK) return ORC_LJ92_CORRUPT;
    uint16_t *lut = build_lut(&h);
    uint16_t *rows = (uint16_t *)calloc((size_t)h.width * 2, sizeof *rows);
    if (!lut || !rows) { free(lut); free(rows); return -2; }
    uint16_t *cur = rows, *prev = rows + h.width;
    bitreader b = { d, len, h.scan_offset, 0, 0, 0 };
    int bad = 0;
    const int W = h.width, H = h.height;
    for (int r = 0; r < H && !bad; r++) {
        int left = 0;
        for (int c = 0; c < W; c++) {
            int px;
            if (r == 0 && c == 0) px = 1 << (h.bits - 1);
            else if (r == 0) px = left;
            else if (c == 0) px = prev[0];
            else switch (h.predictor) {
                case 0: px = 0; break;
                case 1: px = left; break;
                case 2: px = prev[c]; break;
                case 3: px = prev[c - 1]; break;
                case 4: px = left + prev[c] - prev[c - 1]; break;
                case 5: px = left + ((prev[c] - prev[c - 1]) >> 1); break;
                case 6: px = prev[c] + ((left - prev[c - 1]) >> 1); break;
                default: px = (left + prev[c]) >> 1; break;
            }
            left = px + next_diff(&b, lut, h.huffbits, &bad);      /* carried as int along the row, stored as 16 bits */
            if (bad) break;
            cur[c] = (uint16_t)left;
            out[(size_t)r * W + c] = (uint16_t)left;
        }
        uint16_t *t = cur; cur = prev; prev = t;
    }
    free(lut);
    free(rows);
    return (bad || b.overrun) ? ORC_LJ92_CORRUPT : ORC_LJ92_OK;
}

/* main.c:646-667: the decoded image, read as yres rows of xres values, holds the even rows before the odd rows and in
 * every row the even columns before the odd ones */
void orc_lj92_untile(const uint16_t *src, uint16_t *dst, int xres, int yres)
{
    for (int y = 0; y < yres; y++) {
        const int dy = ((2 * y) % yres) + ((2 * y) / yres);
        for (int x = 0; x < xres; x++) {
            const int dx = ((2 * x) % xres) + ((2 * x) / xres);
            dst[(size_t)dy * xres + dx] = src[(size_t)y * xres + x];
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------------------
 * orc_lj92_encode -- the reference's ENCODER (mlvfs/lj92.c:711-1144: frequencyScan 733-786, createEncodeTable 788-937,
 * writeHeader 939-977, writeBody 986-1099, writePost 979-984, lj92_encode 1104-1144), restated.  The reference keeps two row
 * caches and walks the tile once per pass; here the tile is gathered first and every prediction reads its three neighbours
 * directly.  Pinned against the reference's own lj92_encode byte for byte (tests/test_lj92_encode.py).  Cases in which the
 * reference indexes behind its arrays (a 17-bit difference, 17 classes in use, a code longer than 16 bits) return
 * ORC_LJ92_CORRUPT and are not compared.
 */
typedef struct { int bits[17], values[17], nvalues, len[17], code[17]; } orc_lj92_enctable;

static int enc_class(int d) { int s = 0; unsigned a = (unsigned)(d < 0 ? -d : d); while (a) { s++; a >>= 1; } return s; }

int orc_lj92_encode_table(const int hist[17], int npix, orc_lj92_enctable *t)
{
    /* 18 leaves (the 17 classes and a reserved leaf of frequency 1.0); a live group = bitmask of its leaves, kept at the index of
     * the leaf it grew from.  Two rarest: the LAST index among the smallest, then the FIRST index among the smallest of the rest */
    float f[18];
    unsigned members[18];
    int depth[18] = { 0 }, used = 0;
    for (int i = 0; i < 17; i++) { f[i] = (float)hist[i] / (float)npix; members[i] = 1u << i; used += hist[i] != 0; }
    f[17] = 1.0f; members[17] = 1u << 17;
    if (used == 0 || used == 17) return ORC_LJ92_CORRUPT;
    for (;;) {
        int lo = -1, lo2 = -1;
        for (int i = 0; i < 18; i++) if (f[i] > 0.0f && (lo < 0 || f[i] <= f[lo])) lo = i;
        for (int i = 17; i >= 0; i--) if (i != lo && f[i] > 0.0f && (lo2 < 0 || f[i] <= f[lo2])) lo2 = i;
        if (lo2 < 0) break;
        f[lo] += f[lo2]; f[lo2] = 0.0f;
        members[lo] |= members[lo2];
        for (int i = 0; i < 18; i++) if (members[lo] >> i & 1) depth[i]++;
    }
    memset(t, 0, sizeof *t);
    for (int i = 0; i < 18; i++) { if (depth[i] > 16) return ORC_LJ92_CORRUPT; if (depth[i]) { t->bits[depth[i]]++; t->nvalues++; } }
    int k = 0;
    for (int l = 1; l <= 16; l++) for (int s = 0; s < 17; s++) if (depth[s] == l) t->values[k++] = s;
    /* values[k..nvalues-1] stay 0: the reserved leaf has a code but no value */
    int pos_of[17] = { 0 }, pos = 0, code = 0, plen[18], pcode[18];
    for (int l = 1; l <= 16; l++, code <<= 1) for (int j = 0; j < t->bits[l]; j++, pos++) { plen[pos] = l; pcode[pos] = code++; }
    for (int i = 0; i < pos && i < 17; i++) pos_of[t->values[i]] = i;
    for (int s = 0; s < 17; s++) { t->len[s] = plen[pos_of[s]]; t->code[s] = pcode[pos_of[s]]; }
    return ORC_LJ92_OK;
}

/* returns the stream's length (> 0), ORC_LJ92_CORRUPT, or -100 when out is too small */
int orc_lj92_encode(const uint16_t *image, int width, int height, int bitdepth, int read_len, int skip_len,
                    const uint16_t *delin, int delin_len, uint8_t *out, int cap)
{
    const long long n = (long long)width * height;
    if (n <= 0 || n >= (1 << 27)) return ORC_LJ92_CORRUPT;
    int32_t *v = malloc(sizeof(int32_t) * n);
    int32_t *d = malloc(sizeof(int32_t) * n);
    const uint16_t *p = image;
    int run = read_len, bad = 0;
    for (long long i = 0; i < n; i++) {
        int x = *p++;
        if (delin) { if (x >= delin_len) { bad = 1; break; } x = delin[x]; }
        v[i] = x;
        if (--run == 0) { p += skip_len; run = read_len; }
    }
    int hist[18] = { 0 };
    for (long long i = 0; i < n && !bad; i++) {
        const int r = (int)(i / width), c = (int)(i % width);
        int px;
        if (r == 0) px = c ? v[i - 1] : 1 << (bitdepth - 1);
        else if (c == 0) px = v[i - width];
        else px = v[i - width] + ((v[i - 1] - v[i - width - 1]) >> 1);
        d[i] = v[i] - px;
        const int s = enc_class(d[i]);
        hist[s > 17 ? 17 : s]++;
    }
    orc_lj92_enctable t;
    int w = -1;
    if (!bad && !hist[17] && orc_lj92_encode_table(hist, (int)n, &t) == ORC_LJ92_OK) {
        uint8_t head[64];
        int hl = 0;
        const int sof[] = { 0xFF, 0xD8, 0xFF, 0xC3, 0, 11, bitdepth, height >> 8, height, width >> 8, width, 1, 0, 0x11, 0, 0xFF, 0xC4, 0, 19 + t.nvalues, 0 };
        for (unsigned i = 0; i < sizeof sof / sizeof *sof; i++) head[hl++] = (uint8_t)sof[i];
        for (int l = 1; l <= 16; l++) head[hl++] = (uint8_t)t.bits[l];
        for (int i = 0; i < t.nvalues; i++) head[hl++] = (uint8_t)t.values[i];
        const int sos[] = { 0xFF, 0xDA, 0, 8, 1, 0, 0, 6, 0, 0 };
        for (unsigned i = 0; i < sizeof sos / sizeof *sos; i++) head[hl++] = (uint8_t)sos[i];
        w = 0;
        #define ORC_PUT(b) do { if (w < cap) out[w] = (uint8_t)(b); w++; } while (0)
        for (int i = 0; i < hl; i++) ORC_PUT(head[i]);
        uint64_t acc = 0;                       /* pending bits, right-aligned */
        int pending = 0;
        for (long long i = 0; i < n; i++) {
            const int s = enc_class(d[i]);
            const unsigned extra = (unsigned)(d[i] < 0 ? d[i] + (1 << s) - 1 : d[i]) & ((1u << s) - 1u);
            acc = (acc << t.len[s]) | (unsigned)t.code[s]; pending += t.len[s];
            acc = (acc << s) | extra; pending += s;
            while (pending >= 8) {
                const unsigned byte = (unsigned)(acc >> (pending - 8)) & 0xFF;
                ORC_PUT(byte);
                if (byte == 0xFF) ORC_PUT(0);
                pending -= 8;
            }
        }
        if (pending) {
            const unsigned byte = (unsigned)(acc << (8 - pending)) & 0xFF;
            ORC_PUT(byte);
            if (byte == 0xFF) ORC_PUT(0);
        }
        ORC_PUT(0xFF); ORC_PUT(0xD9);
        #undef ORC_PUT
    }
    free(v);
    free(d);
    if (w < 0) return ORC_LJ92_CORRUPT;
    return w <= cap ? w : -100;
}
