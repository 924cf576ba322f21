/*
 * oracle.h -- CPU restatement of MLVFS's per-frame raw-processing path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and only as the checker / reported CPU baseline.
 * The product (mlvfs_amd/csrc -> libmlvfs_amd.so) never links or calls it.
 *
 * Every function is a plain-C restatement (own code, own structure) of the
 * algorithm of one reference function; the reference location it follows is
 * cited above each prototype (paths relative to /root/reference/).
 * Interfaces use plain geometry/level arguments instead of
 * `struct frame_headers` so numpy buffers can be passed through ctypes.
 *
 * Parity pinning: the reference ships no tests or golden vectors for this
 * path.  The oracle is therefore pinned against the
 * reference itself: oracle/Makefile compiles the reference's own hot-path
 * sources in place into oracle/_ref/libmlvfs_ref.so and tests/test_oracle_vs_ref.py
 * requires byte-identical outputs on seeded synthetic, adversarial and
 * edge-case frames; tests/golden/ holds vectors generated from that reference
 * build (tests/golden/make_golden.py).
 */
#ifndef MLVFS_ORACLE_H
#define MLVFS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_EV_RES      32768          /* mlvfs/mlvfs.h:87  EV_RESOLUTION */
#define ORC_MAX_BLACK   16384          /* mlvfs/mlvfs.h:88  MAX_BLACK     */
#define ORC_EV2RAW_LO   (-10 * ORC_EV_RES)
#define ORC_EV2RAW_HI   (14 * ORC_EV_RES)

/* ---- EV look-up tables: mlvfs/main.c:128-196 ---------------------------- */
/* raw2ev for one black level as a directly indexable table: out[p] for
 * p in [0, 16384 + 16384 - black) ... we expose n entries starting at p = 0.
 * out[p] = (int)(log2(p - black) * 32768) for p > black, INT_MIN at p == black
 * (x86 float->int of -inf), 0 below black.                                   */
void orc_build_raw2ev(int black, int32_t *out, int n);
void orc_build_raw2evf(int black, double *out, int n);
/* ev2raw[i] for i in [-10*32768, 14*32768): out[0] is index -10*32768.       */
void orc_build_ev2raw(int32_t *out /* 24*32768 entries */);

/* ---- unpack: mlvfs/dng.c:813-872 ---------------------------------------- */
size_t orc_unpack_bits(const uint16_t *packed, uint8_t *out, int64_t offset,
                       size_t max_size, int bpp);

/* ---- chroma smooth: mlvfs/cs.c:49-84 + mlvfs/chroma_smooth.c:22-71 ------ */
/* method in {2,3,5}; returns 0 on success, -1 on unsupported method.        */
int orc_chroma_smooth(uint16_t *img, int w, int h, int black, int method);

/* ---- bad / focus pixels: mlvfs/cs.c:87-331, 336-503 --------------------- */
typedef struct { int32_t x, y; } orc_pixel_t;
/* detection pass of fix_bad_pixels (cs.c:255-306); returns count, writes up to
 * cap entries (raster order, crop offsets added).                            */
size_t orc_detect_bad_pixels(const uint16_t *img, int w, int h, int black,
                             int aggressive, int crop_x, int crop_y,
                             orc_pixel_t *out, size_t cap);
/* ordered in-place application (cs.c:314-330).                               */
void orc_apply_bad_pixels(uint16_t *img, int w, int h, int black,
                          const orc_pixel_t *map, size_t count,
                          int crop_x, int crop_y, int dual_iso);
/* focus-pixel application incl. edge rules (cs.c:444-503).                   */
void orc_apply_focus_pixels(uint16_t *img, int w, int h, int black,
                            const orc_pixel_t *map, size_t count,
                            int crop_x, int crop_y, int dual_iso);

/* ---- vertical stripes: mlvfs/stripes.c:108-266 --------------------------- */
/* rand_fn == NULL -> libc rand().  coeffs[8] entries for under-populated
 * histograms are left untouched (the reference leaves them uninitialised).
 * hist_out (optional, 8*65536 ints) and num_out (optional, 8) receive the raw
 * histograms.  Returns correction_needed.                                    */
int orc_stripes_compute(const uint16_t *img, int w, int h, int black, int white,
                        int frame_size, int (*rand_fn)(void), int32_t coeffs[8],
                        int32_t *hist_out, int32_t *num_out);
/* rows [row0,row1) with an explicit dither stream (multi-GPU host-logic tests)   */
int64_t orc_stripes_hist_rows(const uint16_t *img, int w, int row0, int row1, int black, int white,
                              const uint16_t *rnd, int64_t n_rnd, int32_t *hist, int32_t *num);
void orc_stripes_apply(uint16_t *img, size_t npix, int w, int black, int white,
                       int needed, const int32_t coeffs[8], int64_t offset);

/* ---- histogram helper: mlvfs/histogram.c:33-84 -------------------------- */
typedef struct { uint16_t white; uint32_t count; uint16_t *bins; } orc_hist_t;
orc_hist_t *orc_hist_create(uint16_t white);
void orc_hist_add(orc_hist_t *h, const uint16_t *data, uint32_t size, uint16_t skip);
uint16_t orc_hist_median(const orc_hist_t *h);
void orc_hist_destroy(orc_hist_t *h);

/* ---- dual-ISO preview: mlvfs/hdr.c:40-227 -------------------------------- */
/* returns 1 when converted (levels then have to be multiplied by 4 by the
 * caller, as hdr.c:223-224 does), 0 when no interlaced pattern was found.    */
int orc_hdr_preview(uint16_t *img, int w, int h, int black, int white,
                    size_t max_size, double *a_out, double *b_out, int *dark_row_start_out);

/* ---- AMaZE demosaic (SSE2 variant = what x86-64 builds of MLVFS run): mlvfs/amaze_demosaic_RT.c:113-1487 ----
 * raw/red/green/blue: h rows of `pitch` floats, pitch >= w + 16 (hdr.c:969-975); values are the caller's scale
 * (the demosaic divides by 65535 on load and multiplies back on store).  Returns 0, -1 on bad arguments.        */
int orc_amaze_demosaic(const float *raw, int w, int h, int pitch, float *red, float *green, float *blue);

/* ---- full dual-ISO (cr2hdr 20-bit): mlvfs/hdr.c:230-1957, interp_method 0 (AMaZE + edge-directed) and 1 (mean23) ----- */
/* returns 1 converted / 0 not dual-ISO or failed / -1 configuration not restated.
 * levels_out = {black, white} the caller's frame_headers end up with (x4 on success);
 * scalars_out (optional) = rggb, is_bright bits, white, white_bright, a, b, corr_ev, white_darkened */
int orc_cr2hdr20(uint16_t *image, int w, int h, int black14, int white14, int interp_method, int use_fullres,
                 int use_alias_map, int chroma_smooth_method, int levels_out[2], double scalars_out[8]);
void orc_dualiso_reset(void);

/* ---- pattern noise: mlvfs/patternnoise.c:49-380 -------------------------- */
void orc_fix_pattern_noise(int16_t *raw, int w, int h, int white);
void orc_fix_pattern_noise_dbg(int16_t *raw, int w, int h, int white, int flags);   /* debug_flags of patternnoise.h:19-24 */

/* ---- glibc rand() restatement (TYPE_3 additive feedback, seed 1) --------- */
typedef struct { int32_t ring[31]; int pos; } orc_rand_t;
void orc_rand_seed(orc_rand_t *st, unsigned seed);
int  orc_rand_next(orc_rand_t *st);

/* ---- whole-frame pipeline in process_frame order: mlvfs/main.c:908-1005 -- */
typedef struct {
    int w, h, bpp, black, white, frame_size;
    int chroma_smooth;     /* 0,2,3,5 */
    int fix_bad_pixels;    /* 0,1,2   */
    int fix_stripes;       /* 0,1     */
} orc_cfg_t;

#ifdef __cplusplus
}
#endif
#endif
