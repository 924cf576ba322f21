// clip.h -- per-clip state (stripe coefficients + ordered pixel map) and the
// launcher prototypes shared between the host files and the kernel files.
#pragma once

#include "common.h"

#include <algorithm>
#include <mutex>
#include <vector>

namespace mlv {

// tile geometry of the fused kernel (k_frame.hip); the per-tile patch lists are built
// on the host with the same numbers
// One tile height since round 4: 15 cell rows (15 rows x 17 lanes of the 5x5 medians, 15 x 17 loader items: k_frame.hip).  The
// two-geometry plumbing of rounds 2-3 (16 rows without 5x5) is kept: both entries are the same now.
constexpr int FRAME_TCW = 64, FRAME_TCH = 15, FRAME_TCH5 = 15, FRAME_HC = 2;
constexpr int FRAME_GEOS = 2;                                   // tile geometries: 0 = FRAME_TCH rows, 1 = FRAME_TCH5 rows
inline int frame_tile_rows(int geo) { return geo == 1 ? FRAME_TCH5 : FRAME_TCH; }
inline int frame_geo_of(int method) { return method == 5 ? 1 : 0; }
inline int frame_tiles_x(int w) { return (w + 2 * FRAME_TCW - 1) / (2 * FRAME_TCW); }
inline int frame_tiles_y(int h, int geo) { return (h + 2 * frame_tile_rows(geo) - 1) / (2 * frame_tile_rows(geo)); }

// A Bayer cell that holds repaired pixels, as the fused kernel's tile lists name it: the pixel-map entry of each of its four
// pixels (index into the clip's ordered entry list; -1: the pixel keeps the frame's value)
struct CellRec {
    int cell;       // cell column | cell row << 16
    int e[4];       // pixel (x & 1) + 2 * (y & 1)
};

struct PatchView {            // what the fused kernel needs to apply a clip's pixel map (one tile geometry)
    const void *cells;        // int4[nframes][n_rec] {cell, R | G1 << 16, G2 | B << 16, -}: k_pixfix_cells
    int n_rec;
    const int *tile_off;      // CSR over the tiles of one frame: records of the cells that lie in the tile + halo
};

struct PixEntry {
    int pos;        // y * w + x in frame coordinates (-1: not applied)
    int kind;       // 0 skip, 1 cross (interpolate_pixel), 2 along x, 3 along y, 4 copy x+2, 5 copy x-2
    int emit;       // 1: this entry's value is the final value of its position
    int dep[12];    // per tap: index of the entry whose repaired value must be read, or -1
};

struct ThreadCtx;
struct Clip {
    Geom g{};
    int pan_x = 0, pan_y = 0;
    int device = 0;
    std::mutex mu;
    // stripes (mlvfs/stripes.h:30-36)
    int needed = 0;
    int32_t coef[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    // pixel map
    std::vector<int32_t> xy;
    int rules = 0, dual_iso = 0;
    int n_entries = 0, n_levels = 0, n_level0 = 0;       // n_level0: entries without a dependency (they come first)
    PixEntry *d_entries = nullptr;
    int *d_level_off = nullptr;
    // per tile geometry: the cells with repaired pixels, listed tile by tile (a cell in the halo of a neighbouring tile is
    // listed there too)
    int *d_tile_off[FRAME_GEOS] = { nullptr, nullptr };
    CellRec *d_tile_rec[FRAME_GEOS] = { nullptr, nullptr };
    int n_rec[FRAME_GEOS] = { 0, 0 };
    // One buffer per call holds the patch list {position, value} of every frame and, behind it, the cell values of every
    // frame for the tile geometry in use
    size_t patch_buffer_bytes(int nframes) const
    {
        return (size_t)nframes * ((size_t)n_entries * 8 + (size_t)std::max(n_rec[0], n_rec[1]) * 16) + 16;
    }
    static void *cells_of(void *patches, int n_entries, int nframes)
    {
        return (uint8_t *)patches + ((size_t)nframes * (size_t)n_entries * 8 + 15) / 16 * 16;
    }
    PatchView patch_view(void *patches, int nframes, int geo) const
    {
        return PatchView{ cells_of(patches, n_entries, nframes), n_rec[geo], d_tile_off[geo] };
    }
    void *d_patches = nullptr;
    size_t patch_bytes = 0;
    void *d_scratch = nullptr;
    size_t scratch_bytes = 0;
    // T16 layout of the fused kernel: 0 = not decided yet (from the first frame the clip processes), 1 = plain, 2 = spread
    int t16_layout = 0;
    void *d_unpacked = nullptr;      // 10 / 12-bit clips: frames unpacked to 16 bits before the fused kernel
    size_t unpacked_bytes = 0;

    ~Clip();
    int set_pixel_map(const int32_t *xy, size_t count, int rules, int dual_iso);
    int ensure_patches(int nframes);
    int ensure_scratch(size_t bytes);
    int ensure_unpacked(size_t bytes);
    int detect_bad_pixels(const void *d_frame, int aggressive, int dual_iso, hipStream_t stream);
    int fix_pixels(void *d_frames, size_t stride, int nframes, hipStream_t stream);
    // one frame, for callers that SHARE this clip between host threads (drop-in path): nothing of the clip is written --
    // the patch list lives in the calling thread's context, the black level is the frame's
    int fix_pixels_shared(void *d_frame, size_t stride, int black, ThreadCtx *c) const;
    int stripes_compute(const void *d_frame, int frame_size, int rand_mode, hipStream_t stream);
};

// one shard (rows [row0,row1)) of the stripes histogram computation
struct StripesWork {
    static constexpr int RECHECK_CAP = 1 << 16;
    Clip *owner = nullptr;
    Geom g{};
    int row0 = 0, row1 = 0, gpr = 0, n_groups = 0, nblk = 0;
    size_t o_counts = 0, o_bsum = 0, o_boff = 0, o_total = 0, o_hist = 0, o_num = 0, o_nre = 0, o_re = 0, o_copies = 0, bytes = 0;
    uint8_t *base = nullptr;
    int init(Clip *owner, const Geom &g, int row0, int row1);
    int count(const void *d_frame, long long *accepted, hipStream_t stream);
    int hist_dev(const void *d_frame, const void *d_rand, long long n_rand, int *d_hist, int *d_num, hipStream_t stream);
    int recheck_into(int32_t *hist_host_or_null, int *d_hist, hipStream_t stream);
    int hist_to_host(const void *d_frame, const void *d_rand, long long n_rand, int32_t *hist, int32_t num[8],
                     hipStream_t stream);
};

// device-level pixel repair shared by the drop-in symbols and the dual-ISO path (dropin.cpp)
struct ThreadCtx;
bool focus_map_applies(struct frame_headers *fh, ThreadCtx *c, int dual_iso);      // a map file exists and has entries for this frame
// n_patched (optional): entries of the patch list the repair left in c->d_patch ({position or -1, value} each)
int focus_pixels_device(struct frame_headers *fh, ThreadCtx *c, void *d_frame, int dual_iso, bool *changed, int *n_patched = nullptr);
int bad_pixels_device(struct frame_headers *fh, ThreadCtx *c, void *d_frame, int aggressive, int dual_iso, bool *changed, int *n_patched = nullptr);

void glibc_rand_stream(uint16_t *out, size_t n, uint64_t skip, unsigned seed);
int stripes_solve(const int32_t *hist, const int32_t num[8], int frame_size, int32_t coeffs[8]);

// optional HIP-event timing of the dominant kernel (mlvfs_amd_timer_*)
struct KernelTimer {
    std::vector<hipEvent_t> ev;      // pairs: start, stop
    int used = 0;
    bool on = false;
};
KernelTimer &kernel_timer();

// kernel launchers (k_*.hip)
int launch_unpack(const void *d_packed, size_t packed_stride, void *d_out, size_t out_stride, uint32_t first_px,
                  uint32_t npix, int bpp, int nframes, hipStream_t stream);
int launch_frame(const Device *dev, const Geom &g, bool packed, const void *src, size_t src_stride, void *dst,
                 size_t dst_stride, int nframes, int method, const PatchView *pv, bool stripes,
                 const int32_t *coef, hipStream_t stream, bool spread = false);       // spread: T16 layout for dark clips (k_frame.hip)
// share of sampled pixels of one frame that lie 1 .. 511 above black, in 1/1024 (synchronises the stream)
int dark_share(int packed_bpp, const void *d_frame, int w, int h, int black, hipStream_t stream, int *share_1024);      // packed_bpp 0: 16-bit frames
// does the fused kernel read this packed stream itself (k_frame.hip), or does it take an unpack pass first?
bool frame_kernel_takes(const Geom &g, const void *src, size_t src_stride, const void *dst, size_t dst_stride, int nframes);
// packed: 0 = 16-bit frames, 1 (`true`) = 14-bit stream, else the stream's bits per pixel (12, 10)
int launch_pixfix(int packed, const void *frames, size_t stride, int w, int black, const void *entries,
                  const int *level_off, int n_levels, int n_level0, int n_entries, void *patches, void *scatter,
                  size_t scatter_stride, int nframes, const DeviceLuts &luts, hipStream_t stream);
// both of the above for the fused kernel (one launch when the map is small)
int launch_pixfix_for_frame_kernel(int packed, const void *frames, size_t stride, int w, int h, int black, const void *entries,
                                   const int *level_off, int n_levels, int n_level0, int n_entries, void *patches,
                                   const CellRec *recs, int n_rec, void *cells, int nframes, const DeviceLuts &luts, hipStream_t stream);
// the four pixels of every listed cell after the repair (for the fused kernel): after launch_pixfix, on the same stream
int launch_pixfix_cells(int packed, const void *frames, size_t stride, int w, int h, const CellRec *recs, int n_rec,
                        const void *patches, int n_entries, void *cells, int nframes, hipStream_t stream);
int launch_deflicker_hist(const void *d_frame, uint32_t samples, uint32_t white, unsigned *d_hist, hipStream_t s);
int launch_hist_add(const void *d_frame, uint32_t first, uint32_t step, uint32_t samples, uint32_t white, unsigned *d_hist, hipStream_t s);
int launch_badpix_detect(const void *d_frame, int w, int h, int black, int aggressive, int crop_x, int crop_y,
                         void *d_mask, int words_per_row, int *d_row_count, void *d_list, int cap,
                         const DeviceLuts &luts, hipStream_t stream);
int stripes_groups_per_row(int w);
int launch_stripes_count(const void *d_frame, int w, int row0, int row1, int black, int white, unsigned char *d_counts,
                         int *d_block_sum, long long *d_block_off, long long *d_total, hipStream_t stream);
constexpr int RAND_CHUNK = 31 * 32;      // values per thread of the device generator of the rand() % 1024 stream
int launch_rand_stream(const uint32_t *d_start, const uint32_t *d_pow2, int npow, uint32_t nchunks, uint32_t *d_states, uint16_t *d_out,
                       size_t n, hipStream_t stream);
int rand_stream_device(uint16_t *d_out, size_t n, uint64_t skip, unsigned seed, hipStream_t stream);
size_t stripes_hist_copies_bytes();
int launch_hist_bump(int *d_hist, const int *d_idx, int n, hipStream_t stream);
int launch_stripes_hist(const void *d_frame, int w, int row0, int row1, int black, int white, const unsigned char *d_counts,
                        const long long *d_block_off, const void *d_rand, long long n_rand, int *d_hist, int *d_num,
                        void *d_recheck, int recheck_cap, int *d_n_recheck, void *d_copies, hipStream_t stream);
int launch_stripes_apply(void *d_frames, size_t stride, size_t npix, int w, int black, int white, const int32_t *coef,
                         int nframes, hipStream_t stream);

}  // namespace mlv
