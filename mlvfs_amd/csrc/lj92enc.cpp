// lj92enc.cpp -- host side of lj92_encode (lj92.h:65-68): the Huffman table of the reference's encoder and the call itself.
//
//   table      mlvfs/lj92.c:788-937 (createEncodeTable): code sizes by merging the two rarest of 18 entries in FLOAT frequencies
//              (17 SSSS classes + one reserved entry of frequency 1.0), the value list in order of size, canonical codes
//   header     mlvfs/lj92.c:939-984 (SOI, SOF3, DHT, SOS with predictor 6, EOI)
//   kernels    csrc/k_lj92enc.hip (histogram, bit packing, byte stuffing)
//
// What the reference's table builder does and a textbook one does not -- all of it kept, the streams are compared byte for byte:
//  * the reserved entry has the LARGEST frequency, so it takes a short code instead of the longest one;
//  * it is counted in BITS but has no value, so the value list is one entry short and its last entry is the zero the list was
//    cleared to: class 0 appears twice in the DHT, and because symbols are assigned walking the list upwards, SSSS = 0 is
//    written with the LAST (longest, all-ones) code;
//  * ties: the first candidate is the last of the rarest entries, the second the first of the remaining rarest.
// Where the reference runs off its arrays the call fails with LJ92_ERROR_CORRUPT instead: a difference that needs 17 bits
// (hist[17], huffsym[17]: 16-bit material only), all 17 classes in use (an 18th code is written behind huffenc[] / huffbits[]),
// a code longer than 16 bits (bits[17+]), a value beyond the delinearisation table, an empty image.
#include <cstdlib>
#include <cstring>

#include "lj92.h"

namespace mlv {

int lje_classify(const uint16_t *d_img, const uint16_t *d_delin, int delin_len, int width, uint32_t npix, int bitdepth, uint32_t *d_code,
                 uint32_t *d_hist, hipStream_t s);
uint32_t lje_blocks(uint64_t n);
int lje_pack(const uint32_t *d_code, uint32_t npix, const uint8_t len[17], const uint16_t codes[17], uint32_t *d_blocks, uint32_t *d_sums,
             uint32_t *d_bits, uint8_t *d_out, hipStream_t s);

namespace {

struct EncTable {
    int bits[17];              // codes per length (index 0 unused), the reserved entry included
    int values[17];            // DHT value list
    int nvalues;               // = sum of bits[]: one more than the classes in use
    uint8_t len[17];           // per SSSS class: length and code it is written with
    uint16_t code[17];
};

// returns nullptr or why the reference's own procedure leaves its arrays
const char *build_table(const uint32_t hist[17], int width_times_height, EncTable *t)
{
    enum { N = 18, NONE = -1 };
    float f[N];
    int size[N], chain[N];
    const float total = (float)width_times_height;
    int used = 0;
    for (int i = 0; i < 17; i++) { f[i] = (float)(int)hist[i] / total; used += hist[i] != 0; }
    f[17] = 1.0f;
    for (int i = 0; i < N; i++) { size[i] = 0; chain[i] = NONE; }
    if (used == 0) return "no pixels";
    if (used == 17) return "all 17 difference classes in use: the reference writes an 18th code behind its tables";
    for (;;) {
        int a = NONE, b = NONE;
        float fa = 3.0f, fb = 3.0f;
        for (int i = 0; i < N; i++) if (f[i] > 0.0f && f[i] <= fa) { fa = f[i]; a = i; }            // last of the rarest
        for (int i = 0; i < N; i++) if (i != a && f[i] > 0.0f && f[i] < fb) { fb = f[i]; b = i; }    // first of the rest
        if (b == NONE) break;
        f[a] += f[b];
        f[b] = 0.0f;
        // every member of both groups moves one level down; b's group is appended to a's
        int e = a;
        for (;; e = chain[e]) { size[e]++; if (chain[e] == NONE) break; }
        chain[e] = b;
        for (e = b; e != NONE; e = chain[e]) size[e]++;
    }
    memset(t, 0, sizeof *t);
    for (int i = 0; i < N; i++) {
        if (size[i] > 16) return "a Huffman code longer than 16 bits";
        if (size[i]) { t->bits[size[i]]++; t->nvalues++; }
    }
    int k = 0;
    for (int l = 1; l <= 16; l++)
        for (int j = 0; j < 17; j++) if (size[j] == l) t->values[k++] = j;
    // canonical codes in list order; list position -> class; the walk upwards lets the list's cleared last entry (class 0) win
    int lens[18], codes[18], n = 0;
    unsigned next = 0;
    for (int l = 1; l <= 16; l++) {
        for (int j = 0; j < t->bits[l]; j++) { lens[n] = l; codes[n] = (int)next++; n++; }
        next <<= 1;
    }
    int at[17] = { 0 };
    for (int i = 0; i < n && i < 17; i++) at[t->values[i]] = i;
    for (int s = 0; s < 17; s++) { t->len[s] = (uint8_t)lens[at[s]]; t->code[s] = (uint16_t)codes[at[s]]; }
    return nullptr;
}

int write_header(uint8_t *e, int width, int height, int bitdepth, const EncTable &t)
{
    int w = 0;
    auto put = [&](int v) { e[w++] = (uint8_t)v; };
    put(0xFF); put(0xD8);
    put(0xFF); put(0xC3); put(0); put(11); put(bitdepth); put(height >> 8); put(height); put(width >> 8); put(width); put(1); put(0); put(0x11); put(0);
    put(0xFF); put(0xC4); put(0); put(17 + 2 + t.nvalues); put(0);
    for (int l = 1; l <= 16; l++) put(t.bits[l]);
    for (int i = 0; i < t.nvalues; i++) put(t.values[i]);
    put(0xFF); put(0xDA); put(0); put(8); put(1); put(0); put(0); put(6); put(0); put(0);
    return w;
}

enum { LJ92_OK = 0, LJ92_CORRUPT = -1, LJ92_NO_MEMORY = -2 };      // lj92.h:29-35
size_t up256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace
}  // namespace mlv

using namespace mlv;

extern "C" {

// test hook (host only; tests/test_lj92_encode.py, hostcheck): the table the encoder would write for a histogram.
// out = bits[1..16], nvalues, values[17], len[17], code[17] as ints (68 of them).  Returns 0, or -1 with the error string set.
int mlvfs_amd_lj92_encode_table(const uint32_t hist[17], int npix, int *out)
{
    EncTable t;
    const char *why = build_table(hist, npix, &t);
    if (why) { set_error("lj92_encode: %s", why); return -1; }
    int k = 0;
    for (int l = 1; l <= 16; l++) out[k++] = t.bits[l];
    out[k++] = t.nvalues;
    for (int i = 0; i < 17; i++) out[k++] = t.values[i];
    for (int i = 0; i < 17; i++) out[k++] = t.len[i];
    for (int i = 0; i < 17; i++) out[k++] = t.code[i];
    return 0;
}

int lj92_encode(uint16_t *image, int width, int height, int bitdepth, int readLength, int skipLength,
                uint16_t *delinearize, int delinearizeLength, uint8_t **encoded, int *encodedLength)               // lj92.h:65-68
{
    if (!image || !encoded || !encodedLength || width <= 0 || height <= 0 || bitdepth < 1 || bitdepth > 16) { set_error("lj92_encode: bad argument"); return LJ92_CORRUPT; }
    const uint64_t npix64 = (uint64_t)width * height;
    if (npix64 >= (1u << 27)) { set_error("lj92_encode: more than 2^27 pixels"); return LJ92_NO_MEMORY; }
    const uint32_t npix = (uint32_t)npix64;
    if (delinearize && delinearizeLength <= 0) { set_error("lj92_encode: empty delinearisation table"); return LJ92_CORRUPT; }
    LibcRandGuard rand_guard;                      // HIP code may run: keep the caller's rand() stream out of its reach
    ThreadCtx *c = thread_ctx();
    if (!c) return LJ92_NO_MEMORY;
    // ---- device layout: d_a = tile, table, codes, counters; d_b = bit stream, stuffed bytes
    const uint32_t nb = lje_blocks(npix), nbb = lje_blocks((uint64_t)npix * 4 + 8);
    size_t a = 0;
    const size_t img_at = a; a += up256((size_t)npix * 2);
    const size_t delin_at = a; a += up256(delinearize ? (size_t)delinearizeLength * 2 : 0);
    const size_t code_at = a; a += up256((size_t)npix * 4);
    const size_t blocks_at = a; a += up256(((size_t)nb + nbb) * 4);
    const size_t hist_at = a; a += 256;                               // 19 counters, then the two sums at +128
    size_t b = 0;
    const size_t bits_at = b; b += up256(((size_t)npix + 2) * 4);
    const size_t out_at = b; b += up256((size_t)npix * 8 + 16);
    if (c->ensure(a, b)) return LJ92_NO_MEMORY;
    uint8_t *A = (uint8_t *)c->d_a, *B = (uint8_t *)c->d_b;
    hipStream_t s = c->stream;
    auto hip_ok = [](hipError_t e, const char *what) { if (e != hipSuccess) { set_error("lj92_encode: %s: %s", what, hipGetErrorString(e)); return false; } return true; };
    // ---- the tile: readLength values, skipLength apart (lj92.c:766-769), contiguous on the device
    if (readLength <= 0 || skipLength == 0 || (uint32_t)readLength >= npix) {
        if (!hip_ok(hipMemcpyAsync(A + img_at, image, (size_t)npix * 2, hipMemcpyHostToDevice, s), "upload")) return LJ92_CORRUPT;
    } else if (skipLength > 0) {
        const uint32_t rows = npix / (uint32_t)readLength, rest = npix - rows * (uint32_t)readLength;
        const size_t pitch = ((size_t)readLength + skipLength) * 2;
        if (!hip_ok(hipMemcpy2DAsync(A + img_at, (size_t)readLength * 2, image, pitch, (size_t)readLength * 2, rows, hipMemcpyHostToDevice, s), "upload")) return LJ92_CORRUPT;
        if (rest && !hip_ok(hipMemcpyAsync(A + img_at + (size_t)rows * readLength * 2, (const uint8_t *)image + rows * pitch, (size_t)rest * 2, hipMemcpyHostToDevice, s), "upload")) return LJ92_CORRUPT;
    } else {                                                           // overlapping or backwards runs: gathered here
        uint16_t *tile = (uint16_t *)malloc((size_t)npix * 2);
        if (!tile) return LJ92_NO_MEMORY;
        const uint16_t *p = image;
        for (uint32_t i = 0, scan = (uint32_t)readLength; i < npix; i++) { tile[i] = *p++; if (--scan == 0) { p += skipLength; scan = (uint32_t)readLength; } }
        const bool ok = hip_ok(hipMemcpyAsync(A + img_at, tile, (size_t)npix * 2, hipMemcpyHostToDevice, s), "upload") && hip_ok(hipStreamSynchronize(s), "upload");
        free(tile);
        if (!ok) return LJ92_CORRUPT;
    }
    if (delinearize && !hip_ok(hipMemcpyAsync(A + delin_at, delinearize, (size_t)delinearizeLength * 2, hipMemcpyHostToDevice, s), "upload")) return LJ92_CORRUPT;
    if (!hip_ok(hipMemsetAsync(A + hist_at, 0, 256, s), "memset") || !hip_ok(hipMemsetAsync(B + bits_at, 0, ((size_t)npix + 2) * 4, s), "memset")) return LJ92_CORRUPT;
    if (lje_classify((const uint16_t *)(A + img_at), delinearize ? (const uint16_t *)(A + delin_at) : nullptr, delinearizeLength, width, npix, bitdepth,
                     (uint32_t *)(A + code_at), (uint32_t *)(A + hist_at), s)) return LJ92_CORRUPT;
    uint32_t hist[19];
    if (!hip_ok(hipMemcpyAsync(hist, A + hist_at, sizeof hist, hipMemcpyDeviceToHost, s), "download") || !hip_ok(hipStreamSynchronize(s), "histogram")) return LJ92_CORRUPT;
    if (hist[18]) { set_error("lj92_encode: a value beyond the delinearisation table"); return LJ92_CORRUPT; }
    if (hist[17]) { set_error("lj92_encode: a difference of 17 bits (the reference counts and looks it up behind its tables)"); return LJ92_CORRUPT; }
    EncTable t;
    if (const char *why = build_table(hist, (int)npix, &t)) { set_error("lj92_encode: %s", why); return LJ92_CORRUPT; }
    uint32_t *d_sums = (uint32_t *)(A + hist_at + 128);
    if (lje_pack((const uint32_t *)(A + code_at), npix, t.len, t.code, (uint32_t *)(A + blocks_at), d_sums, (uint32_t *)(B + bits_at), B + out_at, s)) return LJ92_CORRUPT;
    uint32_t sums[2];
    if (!hip_ok(hipMemcpyAsync(sums, d_sums, sizeof sums, hipMemcpyDeviceToHost, s), "download") || !hip_ok(hipStreamSynchronize(s), "packing")) return LJ92_CORRUPT;
    const size_t body = (size_t)((sums[0] + 7) >> 3) + sums[1];
    uint8_t head[64];
    const int hl = write_header(head, width, height, bitdepth, t);
    if (hl + body + 2 > 0x7FFFFFFFu) { set_error("lj92_encode: stream longer than an int can say"); return LJ92_NO_MEMORY; }
    uint8_t *e = (uint8_t *)malloc(hl + body + 2);                     // the caller frees it (lj92.c:1136-1139)
    if (!e) return LJ92_NO_MEMORY;
    memcpy(e, head, hl);
    if (body && (!hip_ok(hipMemcpyAsync(e + hl, B + out_at, body, hipMemcpyDeviceToHost, s), "download") || !hip_ok(hipStreamSynchronize(s), "download"))) { free(e); return LJ92_CORRUPT; }
    e[hl + body] = 0xFF;
    e[hl + body + 1] = 0xD9;
    *encoded = e;
    *encodedLength = (int)(hl + body + 2);
    return LJ92_OK;
}

}  // extern "C"
