// lj92.cpp -- host side of the LJ92 decoder: JPEG marker parsing, the Huffman look-up table, work buffers.
//
//   stream structure   mlvfs/lj92.c:82-94 (marker search), 276-290 (SOF3, skipped blocks), 595-626 (marker loop), 512-520 (SOS)
//   look-up table      mlvfs/lj92.c:222-270 (direct table indexed by the longest code's worth of bits)
//   kernels            csrc/k_lj92.hip
// Supported: what the reference's decoder supports: one component, one table, predictors 0..7.  Predictor 6 -- the only one
// the reference's own encoder writes (lj92.c:951) -- and 1 have the fast kernels; 7 is limited to 8192 rows.
#include <cstring>
#include <map>
#include <vector>

#include "lj92.h"

namespace mlv {
namespace {

struct Parsed {
    int width = 0, height = 0, bits = 0, pred = -1, huffbits = 0, scan = 0;
    uint8_t count[17] = { 0 };
    const uint8_t *vals = nullptr;
    int nvals = 0;
};

int be16(const uint8_t *p) { return (p[0] << 8) | p[1]; }

// The reference looks for the next 0xFF from wherever the previous segment left it (it does not skip over a DHT's body) and
// takes the byte behind it as the marker; same here, so that the same streams are accepted.
bool parse(const uint8_t *d, int len, Parsed *h, const char **why)
{
    int ix = 0;
    auto marker = [&]() -> int {
        int i = ix;
        while (i < len - 1 && d[i] != 0xFF) i++;
        i += 2;
        if (i + 1 >= len) return -1;                     // every segment that is looked at starts with a 2-byte length: d[ix], d[ix + 1]
        ix = i;
        return d[i - 1];
    };
    *why = "not a lossless JPEG stream";
    if (len < 8 || marker() != 0xD8) return false;
    bool table = false;
    for (;;) {
        const int m = marker();
        if (m < 0) return false;
        if (m == 0xC4) {
            if (ix + be16(d + ix) >= len || ix + 19 > len) return false;
            int n = 0;
            for (int i = 1; i <= 16; i++) { h->count[i] = d[ix + 2 + i]; n += h->count[i]; }
            if (n > 256 || ix + 19 + n > len) return false;
            h->vals = d + ix + 19;
            h->nvals = n;
            for (h->huffbits = 16; h->huffbits > 0 && !h->count[h->huffbits];) h->huffbits--;
            table = true;
        } else if (m == 0xC3) {
            if (ix + 6 >= len) return false;
            h->bits = d[ix + 2];
            h->height = be16(d + ix + 3);
            h->width = be16(d + ix + 5);
            ix += be16(d + ix);
        } else if (m == 0xDA) {
            if (ix + 3 >= len) return false;
            const int nc = d[ix + 2];
            if (ix + 3 + 2 * nc >= len) return false;
            h->pred = d[ix + 3 + 2 * nc];
            h->scan = ix + be16(d + ix);
            break;
        } else if (m == 0xD9) {
            *why = "no scan before the end-of-image marker";
            return false;
        } else {
            ix += be16(d + ix);
            if (ix >= len) return false;
        }
    }
    if (!table || h->huffbits < 1 || h->width <= 0 || h->height <= 0 || h->bits < 1 || h->bits > 16 || h->scan >= len) return false;
    if (h->pred > 7) { *why = "predictor does not exist"; return false; }                   // lj92.c:517
    return true;
}

// Canonical codes in order (lj92.c:222-270 fills one flat table of 2^huffbits entries the same way); here the first
// LJ_L1_BITS bits index the first level and longer codes go through a second-level table per distinct prefix.
// Returns the number of entries used, or -1 when the table needs more second-level tables than fit.
int build_lut(const Parsed &h, uint16_t *lut)
{
    const int b1 = std::min(h.huffbits, LJ_L1_BITS), sub_bits = h.huffbits - b1, sub_n = 1 << sub_bits;
    int used = 1 << b1;
    memset(lut, 0, sizeof(uint16_t) * LJ_LUT_MAX);
    uint32_t code = 0;                                          // next canonical code, left-aligned to huffbits
    int v = 0;
    for (int len = 1; len <= h.huffbits; len++)
        for (int k = 0; k < h.count[len] && v < h.nvals; k++, v++) {
            const uint32_t span = 1u << (h.huffbits - len);
            if (code + span > (1u << h.huffbits)) return used;  // over-subscribed table: the remaining codes do not exist
            const uint16_t e = h.vals[v] <= 16 ? (uint16_t)((h.vals[v] << 8) | len) : 0;      // ssss > 16 cannot be decoded
            if (len <= b1) {
                for (uint32_t i = code >> sub_bits; i < (code + span) >> sub_bits; i++) lut[i] = e;
            } else {
                const uint32_t prefix = code >> sub_bits;
                if (!(lut[prefix] & 0x8000u)) {
                    if (used + sub_n > LJ_LUT_MAX) return -1;
                    lut[prefix] = (uint16_t)(0x8000u | used);
                    used += sub_n;
                }
                const int base = lut[prefix] & 0x7FFF;
                for (uint32_t i = code & (sub_n - 1); i < (code & (sub_n - 1)) + span; i++) lut[base + i] = e;
            }
            code += span;
        }
    return used;
}

struct Work {                              // per host thread and device, grow-only
    uint8_t *d_arena = nullptr, *h_stage = nullptr;
    size_t cap_arena = 0, cap_stage = 0;
    hipStream_t sub[2] = { nullptr, nullptr };          // sub-batches alternate between two streams: uploads overlap kernels
    hipEvent_t ready = nullptr, done[2] = { nullptr, nullptr };
    hipStream_t up = nullptr;                            // every sub-batch's upload, in order, on ONE stream (see the call)
    std::vector<hipEvent_t> uploaded;                   // one per sub-batch
    std::vector<std::pair<const void *, size_t>> last_key;      // (MLVFS_AMD_LJ92_NOUPLOAD: the frames of the call before)
    size_t last_arena = 0;
    bool same_as_before = false;
    Work() = default;
    Work(const Work &) = delete;
    ~Work()                                             // with the host thread that owned it
    {
        for (int k = 0; k < 2; k++) {
            if (sub[k]) { (void)hipStreamSynchronize(sub[k]); (void)hipStreamDestroy(sub[k]); }
            if (done[k]) (void)hipEventDestroy(done[k]);
        }
        if (ready) (void)hipEventDestroy(ready);
        if (up) { (void)hipStreamSynchronize(up); (void)hipStreamDestroy(up); }
        for (hipEvent_t e : uploaded) (void)hipEventDestroy(e);
        if (d_arena) (void)hipFree(d_arena);
        if (h_stage) (void)hipHostFree(h_stage);
    }
    int ensure(size_t arena, size_t stage)
    {
        if (!ready) {
            for (int k = 0; k < 2; k++) {
                MLV_HIP(hipStreamCreateWithFlags(&sub[k], hipStreamNonBlocking));
                MLV_HIP(hipEventCreateWithFlags(&done[k], hipEventDisableTiming));
            }
            MLV_HIP(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
            MLV_HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
        }
        if (arena > cap_arena) {
            if (d_arena) (void)hipFree(d_arena);
            d_arena = nullptr; cap_arena = 0;
            MLV_HIP(hipMalloc(&d_arena, arena));
            cap_arena = arena;
        }
        if (stage > cap_stage) {
            if (h_stage) (void)hipHostFree(h_stage);
            h_stage = nullptr; cap_stage = 0;
            MLV_HIP(hipHostMalloc(&h_stage, stage, hipHostMallocDefault));
            cap_stage = stage;
        }
        return MLVFS_AMD_OK;
    }
};
thread_local std::map<int, Work> t_work;

size_t up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace
}  // namespace mlv

using namespace mlv;

extern "C" int mlvfs_amd_lj92_info(const void *stream, size_t size, int dims[4])
{
    Parsed h;
    const char *why = "";
    if (!stream || size > 0x7FFFFFFF || !parse((const uint8_t *)stream, (int)size, &h, &why)) { set_error("lj92: %s", why); return MLVFS_AMD_ERR_ARG; }
    dims[0] = h.width; dims[1] = h.height; dims[2] = h.bits; dims[3] = h.pred;
    return MLVFS_AMD_OK;
}

extern "C" int mlvfs_amd_lj92_decode_dev(const void *const *streams, const size_t *sizes, int nframes, int xres, int yres,
                                         void *d_out, size_t out_stride, void *stream)
{
    ThreadCtx *c = thread_ctx();
    if (!c) return MLVFS_AMD_ERR_HIP;
    if (nframes <= 0) return MLVFS_AMD_OK;
    // xres == 0 (and yres == 0): the values stay in the decoder's own order, W x H of the JPEG (what lj92_decode hands to main.c)
    const bool raw_order = xres == 0 && yres == 0;
    if (!streams || !sizes || !d_out || (!raw_order && (xres <= 0 || yres <= 0 || out_stride < (size_t)xres * yres * 2))) { set_error("lj92: bad argument"); return MLVFS_AMD_ERR_ARG; }
    hipStream_t s = pick_stream(stream, c);
    std::vector<Parsed> hdr(nframes);
    // ---- layout: one staging block (raw scans, tables, frame records) and one device arena
    size_t stage = 0, arena = 0;
    struct Off { size_t raw, lut, ust, ust_len, blk, cmap, smap, wmap, gmap, wstart, gstart, diff, colsum, err; uint32_t raw_len, nwg, ngrp; };
    std::vector<Off> off(nframes);
    uint32_t max_raw = 0, max_nwg = 0, max_ngrp = 0;
    int max_w = 0, max_h = 0;
    for (int i = 0; i < nframes; i++) {
        const char *why = "";
        Parsed &h = hdr[i];
        if (!streams[i] || sizes[i] > 0x7FFFFFFF || !parse((const uint8_t *)streams[i], (int)sizes[i], &h, &why)) { set_error("lj92: frame %d: %s", i, why); return MLVFS_AMD_ERR_ARG; }
        if (h.pred < 0 || h.pred > 7) { set_error("lj92: frame %d: predictor %d does not exist", i, h.pred); return MLVFS_AMD_ERR_ARG; }
        if (h.pred == 7 && h.height > LJ_WAVE_MAX_H) { set_error("lj92: frame %d: predictor 7 is limited to %d rows", i, LJ_WAVE_MAX_H); return MLVFS_AMD_ERR_ARG; }
        if (raw_order && out_stride < (size_t)h.width * h.height * 2) { set_error("lj92: frame %d: output stride smaller than the %dx%d values", i, h.width, h.height); return MLVFS_AMD_ERR_ARG; }
        if (!raw_order && (long long)h.width * h.height != (long long)xres * yres) {
            set_error("lj92: frame %d: %dx%d values decoded, the video frame is %dx%d", i, h.width, h.height, xres, yres);
            return MLVFS_AMD_ERR_ARG;
        }
        Off &o = off[i];
        o.raw_len = (uint32_t)(sizes[i] - (size_t)h.scan);
        o.nwg = (o.raw_len + 8191) / 8192;
        if (o.nwg == 0) o.nwg = 1;
        o.ngrp = (o.nwg + 31) / 32;
        o.lut = stage; stage += up(sizeof(uint16_t) * LJ_LUT_MAX, 256);
        max_raw = std::max(max_raw, o.raw_len); max_nwg = std::max(max_nwg, o.nwg); max_ngrp = std::max(max_ngrp, o.ngrp);
        max_w = std::max(max_w, h.width); max_h = std::max(max_h, h.height);
    }
    const size_t frames_at = stage;
    stage += up(sizeof(LjFrame) * nframes, 256);
    arena = stage;                                           // the staging block (tables, frame records) is mirrored at the start of the arena
    for (int i = 0; i < nframes; i++) {
        Off &o = off[i];
        const Parsed &h = hdr[i];
        o.raw = arena; arena += up(o.raw_len + 16, 256);
        o.ust = arena; arena += up((size_t)o.nwg * 8192 + 8192, 256);
        o.ust_len = arena; arena += 256;
        o.blk = arena; arena += up(((size_t)o.raw_len / 4096 + 2) * 4, 256);
        o.cmap = arena; arena += (size_t)o.nwg * 256 * 32 * 2;
        o.smap = arena; arena += (size_t)o.nwg * 16 * 32 * 8;
        o.wmap = arena; arena += up((size_t)o.nwg * 32 * 8, 256);
        o.gmap = arena; arena += up((size_t)o.ngrp * 32 * 8, 256);
        o.wstart = arena; arena += up((size_t)o.nwg * 8, 256);
        o.gstart = arena; arena += up((size_t)o.ngrp * 8, 256);
        o.diff = arena; arena += up((size_t)h.width * h.height * 4, 256);
        o.colsum = arena; arena += up((size_t)h.width * 16 * 4, 256);
        o.err = arena; arena += 256;
    }
    Work &w = t_work[c->dev->id];
    int rc = w.ensure(arena, stage + (size_t)nframes * sizeof(int));
    if (rc) return rc;
    {   // the same frames as the call before? (pointers, sizes, layout)
        std::vector<std::pair<const void *, size_t>> key(nframes);
        for (int i = 0; i < nframes; i++) key[i] = { streams[i], sizes[i] };
        w.same_as_before = key == w.last_key && arena == w.last_arena;
        w.last_key.swap(key); w.last_arena = arena;
    }
    LjFrame *fr = (LjFrame *)(w.h_stage + frames_at);
    for (int i = 0; i < nframes; i++) {
        const Off &o = off[i];
        const Parsed &h = hdr[i];
        const int lut_n = build_lut(h, (uint16_t *)(w.h_stage + o.lut));
        if (lut_n < 0) { set_error("lj92: frame %d: Huffman table with more than %d long-code prefixes", i, (LJ_LUT_MAX - (1 << LJ_L1_BITS)) / 32); return MLVFS_AMD_ERR_ARG; }
        LjFrame &f = fr[i];
        uint8_t *A = w.d_arena;
        f.raw = A + o.raw; f.raw_len = o.raw_len;
        f.ust = A + o.ust; f.ust_len = (uint32_t *)(A + o.ust_len); f.blk_drop = (uint32_t *)(A + o.blk);
        f.lut = (const uint16_t *)(A + o.lut); f.huffbits = h.huffbits; f.lut_entries = lut_n;
        f.cmap = (uint16_t *)(A + o.cmap); f.smap = (uint2 *)(A + o.smap); f.wmap = (uint2 *)(A + o.wmap); f.gmap = (uint2 *)(A + o.gmap);
        f.wstart = (uint2 *)(A + o.wstart); f.gstart = (uint2 *)(A + o.gstart);
        f.diff = (int32_t *)(A + o.diff);
        f.colsum = (int32_t *)(A + o.colsum);
        f.out = (uint16_t *)((uint8_t *)d_out + (size_t)i * out_stride);
        f.W = h.width; f.H = h.height; f.bits = h.bits; f.pred = h.pred; f.xres = xres; f.yres = yres;
        f.nwg = o.nwg; f.ngrp = o.ngrp;
        f.err = (int *)(A + o.err);
    }
    MLV_HIP(hipMemcpyAsync(w.d_arena, w.h_stage, stage, hipMemcpyHostToDevice, s));
    for (int i = 0; i < nframes; i++) MLV_HIP(hipMemsetAsync(w.d_arena + off[i].err, 0, sizeof(int), s));
    MLV_HIP(hipEventRecord(w.ready, s));
    // Sub-batches of frames alternate between two streams, so that the upload of one overlaps the kernels of the other.  The
    // entropy-coded bytes go straight from the caller's memory (page-locked when they come from the reader's staging).
    static const int sub_env = [] { const char *e = getenv("MLVFS_AMD_LJ92_SUB"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();   // experiments
    const int SUB = sub_env ? sub_env : (nframes >= 8 ? 4 : (nframes + 1) / 2);
    // The uploads of all sub-batches go out on one stream of their own, in order: two streams' copies ran on two copy engines at 27 GB/s
    // together where one stream's run at 53 (tools/h2d_chunks.py, profiles/r05/h2d_chunks.log).  MLVFS_AMD_LJ92_UPSTREAM=0: as before.
    static const bool one_up = [] { const char *e = getenv("MLVFS_AMD_LJ92_UPSTREAM"); return !e || atoi(e) != 0; }();
    if (one_up) MLV_HIP(hipStreamWaitEvent(w.up, w.ready, 0));
    for (int j0 = 0, j = 0; j0 < nframes; j0 += SUB, j++) {
        const int n = std::min(SUB, nframes - j0);
        hipStream_t sj = w.sub[j & 1];
        hipStream_t su = one_up ? w.up : sj;
        if (j < 2) MLV_HIP(hipStreamWaitEvent(sj, w.ready, 0));
        // (MLVFS_AMD_LJ92_NOUPLOAD=1, measurement only: a call with the same frames as the call before it finds their bytes where that
        // call put them -- what the kernels do when no link stands before them; tools/lj92_bench.py)
        static const bool no_upload = [] {
            const char *e = getenv("MLVFS_AMD_LJ92_NOUPLOAD");
            const bool on = e && atoi(e) != 0;
            if (on) fprintf(stderr, "mlvfs_amd: MLVFS_AMD_LJ92_NOUPLOAD=1 -- a call with the same stream pointers and sizes as the call before it skips "
                                    "their upload (measurement only: stale results if the bytes behind the pointers changed)\n");
            return on;
        }();
        if (!(no_upload && w.same_as_before))
        for (int i = j0; i < j0 + n; i++)
            MLV_HIP(hipMemcpyAsync(w.d_arena + off[i].raw, (const uint8_t *)streams[i] + hdr[i].scan, off[i].raw_len, hipMemcpyHostToDevice, su));
        if (one_up) {
            while ((int)w.uploaded.size() <= j) { hipEvent_t e; MLV_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); w.uploaded.push_back(e); }
            MLV_HIP(hipEventRecord(w.uploaded[j], w.up));
            MLV_HIP(hipStreamWaitEvent(sj, w.uploaded[j], 0));
        }
        unsigned preds = 0;
        for (int i = j0; i < j0 + n; i++) preds |= 1u << hdr[i].pred;
        rc = lj92_launch((const LjFrame *)(w.d_arena + frames_at) + j0, n, max_raw, max_nwg, max_ngrp, max_w, max_h, preds, sj);
        if (rc) return rc;
    }
    for (int k = 0; k < 2; k++) {
        MLV_HIP(hipEventRecord(w.done[k], w.sub[k]));
        MLV_HIP(hipStreamWaitEvent(s, w.done[k], 0));
    }
    int *errs = (int *)(w.h_stage + stage);
    for (int i = 0; i < nframes; i++) MLV_HIP(hipMemcpyAsync(&errs[i], w.d_arena + off[i].err, sizeof(int), hipMemcpyDeviceToHost, s));
    MLV_HIP(hipStreamSynchronize(s));                        // the staging block is reused by the next call
    for (int i = 0; i < nframes; i++)
        if (errs[i]) {
            set_error("lj92: frame %d is damaged (%s)", i, (errs[i] & LJ_ERR_SHORT) ? "fewer symbols than pixels" : "invalid code or data ends early");
            return MLVFS_AMD_ERR_ARG;
        }
    return MLVFS_AMD_OK;
}


// ---------------------------------------------------------------- lj92.h: the decoder's own three calls (main.c:626-647)
// MLVFS's get_image_data opens the frame's JPEG, decodes it into a temporary buffer and untiles that into the frame itself.  With
// these three symbols `lj92.o` can leave the link as well: the decode runs on the GPU (32 ms per 3584x1320 frame in the
// reference's decoder on one host core).  Only what MLVFS passes is supported: skiplen 0, no linearisation table.
namespace {
struct LjHandle { const uint8_t *data; int len; int width, height, bits; };
enum { LJ92_OK = 0, LJ92_CORRUPT = -1, LJ92_NO_MEMORY = -2, LJ92_BAD_HANDLE = -3 };      // lj92.h:29-35
}

extern "C" {

int lj92_open(lj92 *lj, uint8_t *data, int datalen, int *width, int *height, int *bitdepth)        // lj92.h:42-46
{
    if (!lj) return LJ92_BAD_HANDLE;
    *lj = nullptr;
    Parsed h;
    const char *why = "";
    if (!data || datalen <= 0 || !parse(data, datalen, &h, &why)) { set_error("lj92_open: %s", why); return LJ92_CORRUPT; }
    LjHandle *hd = (LjHandle *)malloc(sizeof *hd);
    if (!hd) return LJ92_NO_MEMORY;
    *hd = LjHandle{ data, datalen, h.width, h.height, h.bits };
    if (width) *width = h.width;
    if (height) *height = h.height;
    if (bitdepth) *bitdepth = h.bits;
    *lj = (lj92)hd;
    return LJ92_OK;
}

void lj92_close(lj92 lj) { free(lj); }                                                                   // lj92.h:47

int lj92_decode(lj92 lj, uint16_t *target, int tlen, int skiplen, uint16_t *linearize, int linlen)     // lj92.h:55-58
{
    LjHandle *hd = (LjHandle *)lj;
    if (!hd) return LJ92_BAD_HANDLE;
    const size_t npix = (size_t)hd->width * hd->height;
    const bool plain = skiplen == 0 && !linearize;      // what MLVFS passes (main.c:626-647): the whole frame, value by value
    if (!target || (plain && (size_t)(tlen < 0 ? 0 : tlen) < npix)) { set_error("lj92_decode: target too small for the frame"); return LJ92_CORRUPT; }
    LibcRandGuard rand_guard;                      // HIP code may run: keep the caller's rand() stream out of its reach
    ThreadCtx *c = thread_ctx();
    if (!c || c->ensure(npix * 2, 0)) return LJ92_NO_MEMORY;
    const void *streams[1] = { hd->data };
    const size_t sizes[1] = { (size_t)hd->len };
    if (mlvfs_amd_lj92_decode_dev(streams, sizes, 1, 0, 0, c->d_a, npix * 2, c->stream) != MLVFS_AMD_OK) return LJ92_CORRUPT;
    std::vector<uint16_t> vals;
    uint16_t *dl = target;
    if (!plain) { vals.resize(npix); dl = vals.data(); }
    if (hipMemcpyAsync(dl, c->d_a, npix * 2, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
        set_error("lj92_decode: download failed");
        return LJ92_CORRUPT;
    }
    if (plain) return LJ92_OK;
    // The arguments MLVFS never passes, with the reference's bookkeeping (lj92.c:436-493 for predictor 6, :517-585 for the others):
    // every value goes through the table (`left > linlen` ends the decode as corrupt -- not tested in the first row of the
    // predictor-6 loop), and after every `tlen` values written the output skips `skiplen` more (the predictor-6 loop counts its
    // first value without looking at the counter: a block length of one never skips there).  Done on the host behind the GPU's
    // decode: the values themselves do not depend on either argument.
    Parsed h;
    const char *why = "";
    if (!parse(hd->data, hd->len, &h, &why)) { set_error("lj92_decode: %s", why); return LJ92_CORRUPT; }
    const bool p6 = h.pred == 6;
    uint16_t *out = target;
    int write = tlen;
    size_t cidx = 0;
    for (size_t i = 0; i < npix; i++) {
        const int left = vals[i];
        int linear = left;
        if (linearize) {
            if (!(p6 && i < (size_t)hd->width) && left > linlen) return LJ92_CORRUPT;
            linear = linearize[left];
        }
        out[cidx++] = (uint16_t)linear;
        if (p6 && i == 0) { --write; continue; }
        if (--write == 0) { out += skiplen; write = tlen; }
    }
    return LJ92_OK;
}

// Optional, for a maintainer who changes three lines of main.c: lj92_decode and the untiling loop behind it (main.c:646-667, 8 ms
// per 3584x1320 frame on a host core at -O2, 22 ms as MLVFS's Makefile builds it) in one call -- the frame as get_image_data hands it
// on, xres x yres pixels in host memory, decoded and untiled on the GPU (the device entry point's own output layout).
int mlvfs_amd_lj92_decode_untiled(lj92 lj, uint16_t *dst, int xres, int yres)
{
    LjHandle *hd = (LjHandle *)lj;
    if (!hd) return LJ92_BAD_HANDLE;
    const size_t npix = (size_t)hd->width * hd->height;
    if (!dst || xres <= 0 || yres <= 0 || (size_t)xres * yres != npix) { set_error("lj92: %dx%d values decoded, the video frame is %dx%d", hd->width, hd->height, xres, yres); return LJ92_CORRUPT; }
    LibcRandGuard rand_guard;
    ThreadCtx *c = thread_ctx();
    if (!c || c->ensure(npix * 2, 0)) return LJ92_NO_MEMORY;
    const void *streams[1] = { hd->data };
    const size_t sizes[1] = { (size_t)hd->len };
    if (mlvfs_amd_lj92_decode_dev(streams, sizes, 1, xres, yres, c->d_a, npix * 2, c->stream) != MLVFS_AMD_OK) return LJ92_CORRUPT;
    if (hipMemcpyAsync(dst, c->d_a, npix * 2, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
        set_error("lj92: download failed");
        return LJ92_CORRUPT;
    }
    return LJ92_OK;
}

}  // extern "C"
