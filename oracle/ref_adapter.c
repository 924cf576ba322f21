/*
 * ref_adapter.c -- plain-argument entry points around the REFERENCE's own
 * hot-path functions, built only into oracle/_ref/libmlvfs_ref.so.
 *
 * TEST INFRASTRUCTURE ONLY.  This file contains no algorithm: it fills a
 * `struct frame_headers` (declared by the reference's headers, which are
 * included from /root/reference at build time and never copied into this repo)
 * and calls the reference function unchanged.  It exists so that numpy buffers
 * can reach the reference through ctypes, and so that the reference can be timed
 * as bench.py's cpu_baseline ("kind": "reference").
 *
 * The reference objects import three caller-side symbols (get_raw2ev,
 * get_raw2evf, get_ev2raw: mlvfs/mlvfs.h:90-92).  Their definitions live in the
 * reference's main.c (lines 128-196), which cannot be compiled as a whole here
 * because it needs <fuse.h> (absent from this image; no stand-in header is
 * written).  oracle/Makefile therefore slices those functions -- with
 * mlv_get_frame_headers and get_image_data -- out of main.c BY NAME at build
 * time into a translation unit of this library (_ref/main_slices.c, deleted
 * after the build): the table builders here are the reference's own text.
 * ref_luts.c, the repo's restatement of them, is a separate library
 * (_ref/libref_luts.so) that tests/test_oracle_vs_ref.py checks against that text.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mlvfs.h"      /* reference header, via -I/root/reference/mlvfs */
#include "dng.h"
#include "cs.h"
#include "stripes.h"
#include "hdr.h"
#include "histogram.h"
#include "patternnoise.h"

static void fill(struct frame_headers *fh, int w, int h, int bpp, int black, int white)
{
    memset(fh, 0, sizeof *fh);
    fh->rawi_hdr.xRes = (uint16_t)w;
    fh->rawi_hdr.yRes = (uint16_t)h;
    fh->rawi_hdr.raw_info.width = w;
    fh->rawi_hdr.raw_info.height = h;
    fh->rawi_hdr.raw_info.pitch = w * bpp / 8;
    fh->rawi_hdr.raw_info.frame_size = (int)((int64_t)w * h * bpp / 8);
    fh->rawi_hdr.raw_info.bits_per_pixel = bpp;
    fh->rawi_hdr.raw_info.black_level = black;
    fh->rawi_hdr.raw_info.white_level = white;
    fh->rawi_hdr.raw_info.cfa_pattern = 0x02010100;
}

size_t ref_sizeof_frame_headers(void) { return sizeof(struct frame_headers); }

size_t ref_unpack(const uint16_t *packed, uint8_t *out, int64_t offset, size_t max_size,
                  int w, int h, int bpp)
{
    struct frame_headers fh; fill(&fh, w, h, bpp, 0, 0);
    return dng_get_image_data(&fh, (uint16_t *)packed, out, (off_t)offset, max_size);
}

void ref_chroma_smooth(uint16_t *img, int w, int h, int black, int method)
{
    struct frame_headers fh; fill(&fh, w, h, 14, black, 0);
    chroma_smooth(&fh, img, method);
}

/* fileGuid 0 never matches a cached map (cs.c:235), so detection re-runs on
 * every call, which is what a stateless checker wants.                        */
void ref_fix_bad_pixels(uint16_t *img, int w, int h, int black, int aggressive, int dual_iso,
                        int pan_x, int pan_y, uint64_t guid)
{
    struct frame_headers fh; fill(&fh, w, h, 14, black, 0);
    fh.vidf_hdr.panPosX = (uint16_t)pan_x;
    fh.vidf_hdr.panPosY = (uint16_t)pan_y;
    fh.file_hdr.fileGuid = guid;
    fix_bad_pixels(&fh, img, aggressive, dual_iso);
}

/* focus-pixel maps are looked up as "<cameraModel hex>_<width>x<height>.fpm" in
 * the current directory (cs.c:369-370); tests chdir into a temp dir first.     */
void ref_fix_focus_pixels(uint16_t *img, int w, int h, int black, int dual_iso,
                          uint32_t camera, int raw_w, int raw_h, int pan_x, int pan_y)
{
    struct frame_headers fh; fill(&fh, w, h, 14, black, 0);
    fh.idnt_hdr.cameraModel = camera;
    fh.rawi_hdr.raw_info.width = raw_w;
    fh.rawi_hdr.raw_info.height = raw_h;
    fh.vidf_hdr.panPosX = (uint16_t)pan_x;
    fh.vidf_hdr.panPosY = (uint16_t)pan_y;
    fix_focus_pixels(&fh, img, dual_iso);
}

void ref_free_focus_pixel_maps(void) { free_focus_pixel_maps(); }

int ref_stripes_compute(const uint16_t *img, int w, int h, int black, int white, int32_t coeffs[8])
{
    struct frame_headers fh; fill(&fh, w, h, 14, black, white);
    struct stripes_correction c; memset(&c, 0, sizeof c);
    memcpy(c.coeffficients, coeffs, sizeof c.coeffficients);
    stripes_compute_correction(&fh, &c, (uint16_t *)img, 0, (size_t)w * h);
    memcpy(coeffs, c.coeffficients, sizeof c.coeffficients);
    return c.correction_needed;
}

void ref_stripes_apply(uint16_t *img, int w, int h, int black, int white, int needed,
                       const int32_t coeffs[8])
{
    struct frame_headers fh; fill(&fh, w, h, 14, black, white);
    struct stripes_correction c; memset(&c, 0, sizeof c);
    c.correction_needed = needed;
    memcpy(c.coeffficients, coeffs, sizeof c.coeffficients);
    stripes_apply_correction(&fh, &c, img, 0, (size_t)w * h);
}

int ref_hdr_preview(uint16_t *img, int w, int h, int black, int white, int levels_out[2])
{
    struct frame_headers fh; fill(&fh, w, h, 14, black, white);
    int r = hdr_convert_data(&fh, img, 0, (size_t)w * h * 2);
    levels_out[0] = fh.rawi_hdr.raw_info.black_level;
    levels_out[1] = fh.rawi_hdr.raw_info.white_level;
    return r;
}

int ref_cr2hdr20(uint16_t *img, int w, int h, int black, int white, int interp_method,
                 int fullres, int use_alias_map, int chroma_smooth_method, int fix_bad_pixels_mode,
                 int levels_out[2])
{
    struct frame_headers fh; fill(&fh, w, h, 14, black, white);
    int r = cr2hdr20_convert_data(&fh, img, interp_method, fullres, use_alias_map,
                                  chroma_smooth_method, fix_bad_pixels_mode);
    levels_out[0] = fh.rawi_hdr.raw_info.black_level;
    levels_out[1] = fh.rawi_hdr.raw_info.white_level;
    return r;
}

/* the reference's AMaZE on contiguous planes (amaze_demosaic_RT.c:113); rows are `pitch` floats apart */
void amaze_demosaic_RT(float **rawData, float **red, float **green, float **blue, int winx, int winy, int winw, int winh);
void ref_amaze_demosaic(float *raw, int w, int h, int pitch, float *red, float *green, float *blue)
{
    float **rows = (float **)malloc(sizeof(float *) * 4 * h);
    for (int y = 0; y < h; y++) {
        rows[y] = raw + (size_t)y * pitch;
        rows[h + y] = red + (size_t)y * pitch;
        rows[2 * h + y] = green + (size_t)y * pitch;
        rows[3 * h + y] = blue + (size_t)y * pitch;
    }
    amaze_demosaic_RT(rows, rows + h, rows + 2 * h, rows + 3 * h, 0, 0, w, h);
    free(rows);
}

void ref_fix_pattern_noise(int16_t *raw, int w, int h, int white)
{
    fix_pattern_noise(raw, w, h, white, 0);
}
void ref_fix_pattern_noise_dbg(int16_t *raw, int w, int h, int white, int flags)
{
    fix_pattern_noise(raw, w, h, white, flags);
}

/* histogram.c pass-through (16-bit counters)                                   */
uint16_t ref_hist_median_of(const uint16_t *data, uint32_t size, uint16_t skip, uint16_t white)
{
    struct histogram *hg = hist_create(white);
    hist_add(hg, (uint16_t *)data, size, skip);
    uint16_t m = hist_median(hg);
    hist_destroy(hg);
    return m;
}

/* whole frame in process_frame order (mlvfs/main.c:942-997) for one frame of a
 * fresh clip: unpack -> [bad pixels] -> [chroma smooth] -> [stripes].          */
int ref_process_frame(const uint16_t *packed, uint16_t *img, int w, int h, int bpp, int black,
                      int white, int cs_method, int bad_pix, int stripes, uint64_t guid,
                      int32_t coeffs[8], int *needed_io, int compute_stripes)
{
    struct frame_headers fh; fill(&fh, w, h, bpp, black, white);
    fh.file_hdr.fileGuid = guid;
    dng_get_image_data(&fh, (uint16_t *)packed, (uint8_t *)img, 0, (size_t)w * h * 2);
    fix_focus_pixels(&fh, img, 0);
    if (bad_pix) fix_bad_pixels(&fh, img, bad_pix == 2, 0);
    if (cs_method) chroma_smooth(&fh, img, cs_method);
    if (stripes) {
        struct stripes_correction c; memset(&c, 0, sizeof c);
        c.correction_needed = *needed_io;
        memcpy(c.coeffficients, coeffs, sizeof c.coeffficients);
        if (compute_stripes) stripes_compute_correction(&fh, &c, img, 0, (size_t)w * h);
        stripes_apply_correction(&fh, &c, img, 0, (size_t)w * h);
        memcpy(coeffs, c.coeffficients, sizeof c.coeffficients);
        *needed_io = c.correction_needed;
    }
    return 1;
}

/* dng_get_header_data on a caller-supplied frame_headers image (592 bytes, the layout
 * tests/test_abi.py pins); the reference may rewrite the active area inside it. */
size_t ref_header_data(void *frame_headers_blob, uint8_t *out, int64_t offset, size_t max_size, double fps_override,
                       const char *mlv_basename)
{
    return dng_get_header_data((struct frame_headers *)frame_headers_blob, out, (off_t)offset, max_size, fps_override,
                               (char *)mlv_basename);
}

/* The reference's MLV index (index.c builds from its own single source file): the XREF block without / with the
 * .IDX file beside the clip, and the frame count derived from it. */
#include "index.h"
static size_t copy_xref(mlv_xref_hdr_t *x, uint8_t *out, size_t cap)
{
    if (!x) return 0;
    size_t n = x->blockSize;
    if (out && n <= cap) memcpy(out, x, n);
    free(x);
    return n;
}
size_t ref_mlv_new_index(const char *path, uint8_t *out, size_t cap) { return copy_xref(get_new_index(path), out, cap); }
size_t ref_mlv_get_index(const char *path, uint8_t *out, size_t cap) { return copy_xref(get_index(path), out, cap); }
int ref_mlv_frame_count(const char *path) { return mlv_get_frame_count(path); }

/* The reference's LJ92 codec (lj92.c builds from its single source file). */
#include "lj92.h"
int ref_lj92_decode(uint8_t *data, int len, uint16_t *out, int cap_px, int dims[3])
{
    lj92 h;
    int ret = lj92_open(&h, data, len, &dims[0], &dims[1], &dims[2]);
    if (ret != LJ92_ERROR_NONE) return ret;
    if ((long long)dims[0] * dims[1] > cap_px) { lj92_close(h); return -100; }
    ret = lj92_decode(h, out, dims[0] * dims[1], 0, NULL, 0);
    lj92_close(h);
    return ret;
}
int ref_lj92_encode(uint16_t *img, int w, int h, int bits, uint8_t *out, int cap)
{
    uint8_t *enc = NULL;
    int n = 0;
    int ret = lj92_encode(img, w, h, bits, w * h, 0, NULL, 0, &enc, &n);
    if (ret != LJ92_ERROR_NONE) return ret;
    if (n <= cap) memcpy(out, enc, n);
    free(enc);
    return n <= cap ? n : -100;
}
/* the encoder with all of its arguments (tile runs, delinearisation table): lj92.h:65-68 */
int ref_lj92_encode_tile(uint16_t *img, int w, int h, int bits, int read_len, int skip_len, uint16_t *delin, int delin_len, uint8_t *out, int cap)
{
    uint8_t *enc = NULL;
    int n = 0;
    int ret = lj92_encode(img, w, h, bits, read_len, skip_len, delin, delin_len, &enc, &n);
    if (ret != LJ92_ERROR_NONE) return ret;
    if (n <= cap) memcpy(out, enc, n);
    free(enc);
    return n <= cap ? n : -100;
}

/* The caller's chunk handling as the sliced mlv_get_frame_headers (main.c:429-558) sees it: resource_manager.c:285-317 without
 * KEEP_FILES_OPEN (the default, resource_manager.h:25) forwards to index.c's load_chunks / close_chunks.  (resource_manager.c
 * itself needs <fuse.h>.) */
FILE **mlvfs_load_chunks(const char *path, uint32_t *chunk_count) { return load_chunks(path, chunk_count); }
void mlvfs_close_chunks(FILE **chunk_files, uint32_t chunk_count) { close_chunks(chunk_files, chunk_count); }

int mlv_get_frame_headers(const char *mlv_filename, int index, struct frame_headers *frame_headers);
/* = mlv_get_frame_headers, the reference's own text; out must hold ref_sizeof_frame_headers() bytes */
int ref_mlv_frame_headers(const char *path, int index, uint8_t *out)
{
    struct frame_headers fh;
    const int ok = mlv_get_frame_headers(path, index, &fh);
    memcpy(out, &fh, sizeof fh);
    return ok;
}

/* LZMA payloads: the reference's vendored decoder / encoder (LZMA/LzmaLib.c) behind the layout main.c:598-616 reads:
 * [u32 decoded size][5 property bytes][stream].  The encoder only makes test streams. */
#include "LZMA/LzmaLib.h"
long ref_lzma_make_payload(const uint8_t *src, size_t n, uint8_t *out, size_t cap, int level, unsigned dict, int lc, int lp, int pb)
{
    if (cap < 9) return -1;
    size_t dest_len = cap - 9, props_len = 5;
    const int r = LzmaCompress(out + 9, &dest_len, src, n, out + 4, &props_len, level, dict, lc, lp, pb, 32, 1);
    if (r != SZ_OK || props_len != 5) return -1 - r;
    out[0] = (uint8_t)n; out[1] = (uint8_t)(n >> 8); out[2] = (uint8_t)(n >> 16); out[3] = (uint8_t)(n >> 24);
    return (long)(dest_len + 9);
}
int ref_lzma_uncompress(const uint8_t *payload, size_t size, uint8_t *out, size_t *out_len)
{
    size_t lzma_out_size = *(const uint32_t *)payload, lzma_in_size = size - LZMA_PROPS_SIZE - 4;      /* main.c:600-602 */
    const int r = LzmaUncompress(out, &lzma_out_size, payload + 4 + LZMA_PROPS_SIZE, &lzma_in_size, payload + 4, LZMA_PROPS_SIZE);
    *out_len = lzma_out_size;
    return r;
}

/* the animated preview, gif.c:82-244 (its frame fetch is the sliced get_image_data) */
#include "gif.h"
size_t ref_gif(const char *path, uint8_t *out, size_t cap)
{
    struct frame_headers fh;
    if (!mlv_get_frame_headers(path, 0, &fh)) return 0;
    const size_t n = gif_get_size(&fh);
    if (!out || cap < n) return n;
    return gif_get_data(path, out, 0, n) ? n : 0;
}
