// dualiso.h -- shared declarations of the full dual-ISO path (k_dualiso.hip, dualiso.cpp)
#pragma once
#include "common.h"

namespace mlv {

// histogram block the host decisions read (unsigned words), derived from the device block below
constexpr size_t DI_H_BAYER = 0;                         // [4][16384]  (y%2)*2 + (x%2)
constexpr size_t DI_H_GREEN0 = DI_H_BAYER + 4 * 16384;   // [4][16384]  greens by y%4, frame as is (RGGB)
constexpr size_t DI_H_GREEN1 = DI_H_GREEN0 + 4 * 16384;  // [4][16384]  greens by y%4, frame minus its first row (GBRG)
constexpr size_t DI_H_WHITE0 = DI_H_GREEN1 + 4 * 16384;  // [4][32768]  every 3rd pixel by y%4 (RGGB)
constexpr size_t DI_H_WHITE1 = DI_H_WHITE0 + 4 * 32768;  // [4][32768]  (GBRG)
constexpr size_t DI_HIST_WORDS = DI_H_WHITE1 + 4 * 32768;
// what the device produces (k_di_analyse): class histograms [y % 4][x & 1][16384] over ALL rows + the two white blocks
constexpr size_t DI_D_CLASS = 0;
constexpr size_t DI_D_WHITE0 = DI_D_CLASS + 8 * 16384;
constexpr size_t DI_D_WHITE1 = DI_D_WHITE0 + 4 * 32768;
constexpr size_t DI_D_WORDS = DI_D_WHITE1 + 4 * 32768;

constexpr int DI_HIST_OFF = 65536, DI_HIST_N = 131072;   // match_exposures histograms: value + OFF

struct DiParams {
    int w, h;                  // frame the conversion works on (one row shorter for GBRG)
    int ay1;                   // active_area.y1 (0 for RGGB, 1 after the GBRG row skip)
    int is_bright_bits;        // bit k = is_bright[k]
    int black20, white20;      // 20-bit levels (white from white_detect)
    int match_white20;         // MIN(white, white_bright): the clip level of match_exposures
    int white_darkened;
    int dark_noise;            // 8 * 64 (hdr.c:329-333, 1817)
    int use_fullres, use_alias_map, chroma_smooth;   // chroma_smooth: 0, 2, 3, 5
    int mix_lo, mix_hi;        // bright values below mix_lo mix with the weight of t = 0, from mix_hi on with that of t = overlap (hdr.c:1567-1575;
                               // dualiso.cpp: mix_band): only the band between them needs log2 of the signal and the cosine
    double a, b20, corr_ev, max_ev, overlap;
};

// Batches (mlvfs_amd_cr2hdr20_batch_dev): every kernel of the conversion takes the frame index from its grid (blockIdx.y, or .z
// for the kernels whose grid is two-dimensional) and its DiParams from a device array; the planes of frame f start f * stride
// elements behind the batch's base pointers.  A single conversion is a batch of one whose parameters travel by value.
struct DiBatch {
    const DiParams *pp;        // device array [nframes], or null: use `p0`
    DiParams p0;
    size_t S;                  // plane stride between frames, in pixels (a multiple of 64)
    size_t img_stride;         // bytes between the 16-bit input / output frames
    int nframes;
    int f0;                    // first frame of this launch: the kernels' frame index is f0 + the grid's (a batch may be launched in parts)
    int heights[2], nheights;  // the distinct row counts among the frames that are converted (H, and H - 1 for GBRG frames): AMaZE's
                               // launch plan depends on the rows, one plan per height
};
// frames a batch leaves alone (not dual ISO, detection failed) carry h = 0 in their DiParams

// what the device-side decisions of a batch hand to the host, per frame (k_di_decide_*), in ONE copy
struct DiDecide {
    double check_sum, check_n;             // hdr_check (hdr.c:407-439)
    int rggb;                              // identify_rggb_or_gbrg (hdr.c:441-495)
    int is_bright[4];                      // identify_bright_and_dark_fields (hdr.c:497-636); raw differences in bd_raw
    int bd_raw[4];
    int white_dark, white_bright;          // white_detect (hdr.c:250-300), 14-bit
    long long n;                           // samples of match_exposures' histograms (hdr.c:700-722)
    int bmed, b_lo, b_hi, dmed;
    int hi_n;                              // highlight pairs (hdr.c:735-746)
    int best;                              // index of the winning candidate slope, -1 none (hdr.c:752-772)
    int best_score;
    int check_ok;                          // hdr_check passed (or was passed before the drop-in path's pixel repairs)
};

struct DiLuts {                // device pointers; *_ev2raw are indexable from -10*32768
    const int *interp_raw2ev, *interp_ev2raw;      // the interpolator's cache: mean23's (hdr.c:1240) or AMaZE's (hdr.c:1080)
    const int *mix_raw2ev, *mix_ev2raw;
    const int *blend_raw2ev, *blend_ev2raw;
    const double *fullres_curve;   // [2^20]
    const double *log2sig;         // [2^20] log2(max(i/64 - black/64, 1))
    // the same tables re-packed for k_di_interp, whose time is the number of its table gathers (1.2 GB of L2 requests per batch of 8):
    const int2 *mix_pair;              // indexable from -10*32768: { mix_ev2raw[e], mix_raw2ev[mix_ev2raw[e]] }
    int fullres_thr;                   // fullres_curve[i] > 0.8 <=> i >= fullres_thr (the curve is monotone; checked when it is built)
    int fr_lo, fr_hi;                  // fullres_curve[i] == fr_lo_val for every i < fr_lo, == fr_hi_val for every i >= fr_hi (found in the table
    double fr_lo_val, fr_hi_val;       // when it is built): k_di_blend reads the 8 MB of doubles only for the band between
    int blend_is_mix;                  // the blend's raw2ev table equals the mix's (always, in practice: both are rebuilt in the same call with
                                       // the same levels): without chroma smoothing the interpolation hands the blend EV values instead of raw ones
};

constexpr int DI_STAT_SLOTS = 64;     // k_di_edge_dir spreads its four counters per frame over this many slots (same-address atomics serialise)
struct DiPlanes {
    uint32_t *raw, *dark, *bright, *fullres, *halfres, *fullres_s, *halfres_s;
    uint16_t *over, *amap, *aux, *amap2;
    int *cells;                // [3][h/2][w/2] work planes of the chroma smoothing
    // AMaZE path only
    float *cfa, *red, *green, *blue;   // squeezed Bayer plane and its demosaic, [h][w]
    int *ev_red, *ev_green, *ev_blue;  // interp_raw2ev of the clamped demosaic (k_di_amaze_clamp): each is looked up by up to six pixels
    int *gray_ev;              // raw2ev of the gray image, squeezed like the planes it comes from, [h][w]
    uint8_t *dir;              // chosen edge direction, [h][w]
    const int *sq_dst, *sq_row;        // per image row: squeezed row it is written to (-1 none) / looked up at (0 if none)
    unsigned *stats;           // semi-overexposed, not overexposed, deep shadow, not shadow (4 per slot, DI_STAT_SLOTS slots per frame)
    float *amaze_scratch;
    size_t amaze_scratch_stride;   // floats between the frames' scratch blocks
    size_t cells_stride;           // ints between the frames' chroma-smoothing work planes
};

constexpr int AMAZE_TS = 160;                                              // tile side, amaze_demosaic_RT.c:136
constexpr int AMAZE_TILE_FLOATS = 13 * AMAZE_TS * AMAZE_TS + 13 * AMAZE_TS * AMAZE_TS / 2;
// nframes planes of `plane_stride` floats each; h_of (device, per frame, stride in ints `h_stride`; null: every frame has `h` rows)
// gives the rows of each frame, 0 = skip the frame
int amaze_launch(const float *d_raw, int w, int h, float *d_red, float *d_green, float *d_blue, float *d_scratch, hipStream_t s,
                 int nframes = 1, size_t plane_stride = 0, size_t scratch_stride = 0, const int *h_of = nullptr, int h_stride = 0,
                 float *d_rows_dbg = nullptr, const int *d_r2e = nullptr, int ev_black = 0, int *d_gray = nullptr);
// d_r2e (round 5): the three plane pointers are INT planes and take interp_raw2ev of the clamped plane values, d_gray that of the gray
// value (what k_di_amaze_ev made of the float planes in a pass of its own: amaze_math.h, ev_of_planes)
size_t amaze_scratch_bytes(int w, int h);
// k_amaze_rows.hip: the complete tiles (the first nfx x nfy of the tile grid), row-streamed through LDS
extern int g_amaze_rows_mode;
extern int g_amaze_rows_extra_mode;
void amaze_rows_extent(int w, int h, int *nfx, int *nfy);
int amaze_rows_extra(int w, int h, int nframes);            // heads of k_amaze.hip's chains that k_amaze_rows takes too (rows 0 .. n - 1 of column nfx)
int amaze_rows_launch(const float *d_raw, int w, int h, float *d_red, float *d_green, float *d_blue, hipStream_t s, int nframes,
                      size_t plane_stride, const int *h_of, int h_stride, float *d_dbg, int *d_ctr /* nframes zeroed ints: the tile counters */,
                      const int *d_r2e = nullptr, int ev_black = 0, int *d_gray = nullptr);
int di_launch_amaze_interp(const void *d_img, const DiBatch &b, int h_launch, const DiLuts &L, const DiPlanes &P, hipStream_t s,
                           hipEvent_t after_amaze = nullptr, hipStream_t tail = nullptr);
// incl. the exposure match; after_amaze: recorded on s when AMaZE is through; tail: the stream that takes over from there (waits for after_amaze)

int di_launch_analyse(const void *d_img, int w, int H, int black, int white, const double *d_evf, unsigned *d_hist,
                      double *d_check, hipStream_t s, int nframes = 1, size_t img_stride = 0, size_t hist_stride = 0, size_t check_stride = 0);
int di_launch_subsample(const void *d_img, const DiBatch &b, int nsx, int nsy_max, size_t ns_stride, int *d_dark_s, int *d_bright_s,
                        unsigned *d_hist_bd, hipStream_t s);
int di_launch_hi_count(const int *d_bs, int nsx, int nsy_max, size_t ns_stride, int b_lo, int b_hi, const DiDecide *dd, const DiBatch &b,
                       int *d_rows, hipStream_t s);
int di_launch_hi_compact(const int *d_ds, const int *d_bs, int nsx, int nsy_max, size_t ns_stride, int b_lo, int b_hi, const DiDecide *dd,
                         const DiBatch &b, const int *d_rows, int *d_hi, size_t hi_stride, hipStream_t s);
int di_launch_score(const int *d_hi, size_t hi_stride, int hi_n, const double *d_ta, int ncand, int dmed, int bmed, const DiDecide *dd,
                    int nframes, int *d_score, int score_stride, hipStream_t s);
int di_launch_match(const void *d_img, const DiBatch &b, int h_launch, const DiLuts &L, const DiPlanes &P, hipStream_t s);   // mean23: + raw2ev of the result (P.ev_red)
int di_launch_convert(const DiBatch &b, int h_launch, const DiLuts &L, const DiPlanes &P, bool interp_done, void *d_out, hipStream_t s);
// device-side decisions of a batch (k_dualiso.hip: k_di_decide_*)
struct DiDecideBuffers {
    const unsigned *hist; size_t hist_stride;          // k_di_analyse's block per frame
    const double *check; size_t check_stride;
    unsigned *derived; size_t derived_stride;          // work: Bayer / green histograms with their row ranges applied, prefix sums
    const unsigned *hist_bd;                           // match_exposures' histograms, hist_b | hist_d per frame
    int *rows; int nsy_max;                            // counts | take | offset per frame
    const int *score; int score_stride; int ncand;
    int check_passed;                                  // the caller has evaluated hdr_check already (drop-in path: before its pixel repairs)
    DiDecide *dd;                                      // [nframes]
    DiParams *pp;                                      // [nframes]: the geometry part is written by the pattern step
};
size_t di_derived_words();
int di_launch_decide_pattern(const void *d_frames, const DiBatch &b, int H, int black14, const DiDecideBuffers &D, hipStream_t s);
int di_launch_decide_quantiles(const DiBatch &b, const DiDecideBuffers &D, hipStream_t s);
int di_launch_decide_rows(const DiBatch &b, const DiDecideBuffers &D, hipStream_t s);
int di_launch_decide_fit(const DiBatch &b, const DiDecideBuffers &D, hipStream_t s);

}  // namespace mlv
