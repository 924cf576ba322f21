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
    double a, b20, corr_ev, max_ev, overlap;
};

struct DiLuts {                // device pointers; *_ev2raw are indexable from -10*32768
    const int *interp_raw2ev, *interp_ev2raw;      // the interpolator's cache: mean23's (hdr.c:1240) or AMaZE's (hdr.c:1080)
    const int *mix_raw2ev, *mix_ev2raw;
    const int *blend_raw2ev, *blend_ev2raw;
    const double *fullres_curve;   // [2^20]
    const double *log2sig;         // [2^20] log2(max(i/64 - black/64, 1))
};

struct DiPlanes {
    uint32_t *raw, *dark, *bright, *fullres, *halfres, *fullres_s, *halfres_s;
    uint16_t *over, *amap, *aux, *amap2;
    int *cells;                // [3][h/2][w/2] work planes of the chroma smoothing
    // AMaZE path only
    float *cfa, *red, *green, *blue;   // squeezed Bayer plane and its demosaic, [h][w]
    int *gray_ev;              // raw2ev of the de-squeezed gray image, [h][w]
    uint8_t *dir;              // chosen edge direction, [h][w]
    const int *sq_dst, *sq_row;        // per image row: squeezed row it is written to (-1 none) / looked up at (0 if none)
    unsigned *stats;           // semi-overexposed, not overexposed, deep shadow, not shadow
    float *amaze_scratch;
};

constexpr int AMAZE_TS = 160;                                              // tile side, amaze_demosaic_RT.c:136
constexpr int AMAZE_TILE_FLOATS = 13 * AMAZE_TS * AMAZE_TS + 13 * AMAZE_TS * AMAZE_TS / 2;
int amaze_launch(const float *d_raw, int w, int h, float *d_red, float *d_green, float *d_blue, float *d_scratch, hipStream_t s);
size_t amaze_scratch_bytes(int w, int h);
int di_launch_amaze_interp(const DiParams &p, const DiLuts &L, const DiPlanes &P, hipStream_t s);

int di_launch_analyse(const void *d_img, int w, int H, int black, int white, const double *d_evf, unsigned *d_hist,
                      double *d_check, hipStream_t s);
int di_launch_subsample(const void *d_img, const DiParams &p, int nsx, int nsy, int *d_dark_s, int *d_bright_s,
                        unsigned *d_hist_b, unsigned *d_hist_d, hipStream_t s);
int di_launch_hi_count(const int *d_bs, int nsx, int nsy, int b_lo, int b_hi, int *d_counts, hipStream_t s);
int di_launch_hi_compact(const int *d_ds, const int *d_bs, int nsx, int nsy, int b_lo, int b_hi, const int *d_take, const int *d_offset,
                         int *d_hd, int *d_hb, hipStream_t s);
int di_launch_score(const int *d_hd, const int *d_hb, int hi_n, const double *d_cand, int ncand, int *d_score, hipStream_t s);
int di_launch_match(const void *d_img, const DiParams &p, const DiPlanes &P, hipStream_t s);
int di_launch_convert(const DiParams &p, const DiLuts &L, const DiPlanes &P, bool interp_done, void *d_out, hipStream_t s);

}  // namespace mlv
