/*
 * mlvfs_amd.h -- C ABI of libmlvfs_amd.so, the MI355X (gfx950) implementation of
 * MLVFS's per-frame raw-processing path.  Plain pointers and sizes only.
 *
 * PART 1 -- drop-in symbols.  Same names, prototypes, ownership and error
 *   behaviour as the reference's dng.o / cs.o / stripes.o / hdr.o / histogram.o /
 *   patternnoise.o for the hot path, so MLVFS's unchanged main.c links against
 *   this library instead (the binding a maintainer adds is shown in
 *   INTEGRATION.md).  They work IN PLACE ON HOST MEMORY and are synchronous:
 *   each call stages the frame to the GPU, runs the HIP kernels and copies back.
 *   Any HIP failure is reported on stderr and returns the reference's failure
 *   value (0 / no-op) -- there is NO CPU fallback in this library.
 *
 * PART 2 -- device-resident API (mlvfs_amd_*).  The same stages on frames that
 *   already live in HBM, batched over many frames per launch, plus the fused
 *   steady-state pipeline.  This is what bench.py times and what a frame
 *   prefetcher would call.  All device calls are asynchronous
 *   on the given hipStream_t (passed as void*; NULL = HIP's default stream)
 *   unless stated otherwise.
 */
#ifndef MLVFS_AMD_H
#define MLVFS_AMD_H

#include <stddef.h>
#include <stdint.h>
#include <sys/types.h>

#include "mlvfs_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ======================================================================== */
/* PART 1: drop-in symbols                                                   */

/* replaces mlvfs/dng.h:31 (dng.c:854-872): unpack bpp-bit packed pixels to u16 */
size_t dng_get_image_data(struct frame_headers *frame_headers, uint16_t *packed_bits,
                          uint8_t *output_buffer, off_t offset, size_t max_size);
/* replaces mlvfs/dng.h:29 (dng.c:597-789): the 65536-byte CinemaDNG header (TIFF IFD0 + EXIF IFD) of one frame,
 * built on the host from the MLV block headers; copies min(max_size, 65536) bytes starting at `offset` and returns
 * that count.  May rewrite frame_headers->rawi_hdr.raw_info.active_area like the reference does.  Host only:
 * works without a HIP device.                                                       */
size_t dng_get_header_data(struct frame_headers *frame_headers, uint8_t *output_buffer, off_t offset, size_t max_size,
                           double fps_override, char *mlv_basename);
/* replaces mlvfs/dng.h:30,32,33 (dng.c:797-800, 879-891) */
size_t dng_get_header_size(void);
size_t dng_get_image_size(struct frame_headers *frame_headers);
size_t dng_get_size(struct frame_headers *frame_headers);

/* replaces mlvfs/cs.h:27-30 (cs.c:49-84, 220-331, 336-503) */
void chroma_smooth(struct frame_headers *frame_headers, uint16_t *image_data, int method);
void fix_bad_pixels(struct frame_headers *frame_headers, uint16_t *image_data, int aggressive, int dual_iso);
void fix_focus_pixels(struct frame_headers *frame_headers, uint16_t *image_data, int dual_iso);
void free_focus_pixel_maps(void);

/* replaces mlvfs/stripes.h:30-43 (stripes.c:29-266) */
struct stripes_correction {
    struct stripes_correction *next;
    char *mlv_filename;
    int correction_needed;
    int coeffficients[8];          /* sic: the reference's spelling is part of the ABI */
};
struct stripes_correction *stripes_get_correction(const char *mlv_filename);
struct stripes_correction *stripes_new_correction(const char *mlv_filename);
void stripes_free_corrections(void);
void stripes_compute_correction(struct frame_headers *frame_headers, struct stripes_correction *correction,
                                uint16_t *image_data, off_t offset, size_t size);
void stripes_apply_correction(struct frame_headers *frame_headers, struct stripes_correction *correction,
                              uint16_t *image_data, off_t offset, size_t size);

/* replaces mlvfs/hdr.h:27 (hdr.c:40-227): fast dual-ISO preview; 1 = converted */
int hdr_convert_data(struct frame_headers *frame_headers, uint16_t *image_data, off_t offset, size_t max_size);

/* replaces mlvfs/hdr.h:28 (hdr.c:1932-1957): full dual-ISO conversion (cr2hdr 20-bit).
 * Every configuration main.c can ask for is built: interp_method 0 (AMaZE + edge-directed interpolation, --amaze-edge) and
 * 1 (mean23), fullres / alias map on or off, chroma_smooth 0 / 2 / 3 / 5 (any other value only logs, like hdr.c:1915-1927),
 * fix_bad_pixels_mode 0 / 1 / 2 (focus + bad pixel repair in dual-ISO mode first, hdr.c:1784-1790).  Returns 1 when the
 * frame was converted (levels in frame_headers are then multiplied by 4, hdr.c:1949-1950), 0 when the frame is not dual ISO
 * or cannot be converted -- a width that is not a multiple of 4 with interp_method 0 has no defined result in the
 * reference's SSE2 AMaZE and is refused --; the frame is then left untouched.                          */
int cr2hdr20_convert_data(struct frame_headers *frame_headers, uint16_t *image_data, int interp_method, int fullres,
                          int use_alias_map, int chroma_smooth, int fix_bad_pixels_mode);

/* replaces mlvfs/patternnoise.h:16 (patternnoise.c:357-380) */
void fix_pattern_noise(int16_t *raw, int w, int h, int white, int debug_flags);

/* replaces mlvfs/histogram.h:26-36 (histogram.c:33-84); host-side helper used by
 * main.c's deflicker (main.c:895-906)                                         */
struct histogram { uint16_t white; uint32_t count; uint16_t *data; };
struct histogram *hist_create(uint16_t white);
void hist_add(struct histogram *hist, uint16_t *data, uint32_t size, uint16_t skip);
uint16_t hist_median(struct histogram *hist);
void hist_destroy(struct histogram *hist);

/* Imported from the caller when present (weak): mlvfs/mlvfs.h:90-92, defined in
 * main.c:128-196.  When the caller provides them the device tables are built
 * from the caller's arrays; otherwise from the same formulas with the host libm. */
int *get_raw2ev(int black);
int *get_ev2raw(void);

/* ======================================================================== */
/* PART 2: device-resident API                                               */

#define MLVFS_AMD_OK            0
#define MLVFS_AMD_ERR_HIP      -1     /* a HIP call failed (see mlvfs_amd_last_error) */
#define MLVFS_AMD_ERR_ARG      -2     /* bad geometry / argument                      */
#define MLVFS_AMD_ERR_LUT      -3     /* host libm tables fail the compression check  */
#define MLVFS_AMD_ERR_NOMEM    -4

typedef struct {
    int32_t width, height;        /* rawi_hdr.xRes / yRes                         */
    int32_t bpp;                  /* raw_info.bits_per_pixel                       */
    int32_t black, white;         /* raw_info.black_level / white_level            */
    int32_t pan_x, pan_y;         /* vidf_hdr.panPosX / panPosY (pixel-map crop)   */
} mlvfs_amd_geom_t;

/* context: one per (host thread, device); created lazily */
int         mlvfs_amd_device_count(void);
/* PCI bus id ("0000:c1:00.0") of visible device `device` into out (len >= 16): the physical card, whatever the enumeration order;
 * and the device the calling thread is bound to (-1: not yet).  Threads that do not choose a device (mlvfs_amd_init) are spread
 * round-robin over the visible GPUs in the order of their PCI bus ids; the mapping is printed once (MLVFS_AMD_QUIET=1: not). */
int mlvfs_amd_device_pci_bus_id(int device, char *out, int len);
int mlvfs_amd_thread_device(void);
int         mlvfs_amd_init(int device);              /* binds the calling thread to `device` */
const char *mlvfs_amd_last_error(void);
const char *mlvfs_amd_version(void);

/* per-clip artefacts: stripe coefficients + ordered pixel map.
 * A handle also owns scratch that its calls reuse (the per-frame patch lists of the pixel repair): calls on ONE handle must
 * be ordered -- issue them on one stream, or serialise them --; to process one clip on several streams at once create a
 * handle per stream and give each the same stripes / pixel map.  (The drop-in symbols of PART 1 do that themselves: the
 * maps they cache per clip are shared read-only between libfuse's worker threads, each thread repairs into its own buffers.) */
typedef struct mlvfs_amd_clip mlvfs_amd_clip_t;
mlvfs_amd_clip_t *mlvfs_amd_clip_create(const mlvfs_amd_geom_t *geom);
void   mlvfs_amd_clip_destroy(mlvfs_amd_clip_t *clip);
int    mlvfs_amd_clip_set_stripes(mlvfs_amd_clip_t *clip, int needed, const int32_t coeffs[8]);
int    mlvfs_amd_clip_get_stripes(const mlvfs_amd_clip_t *clip, int *needed, int32_t coeffs[8]);
/* Layout of the raw2ev table inside the fused kernel: 0 = decide from the first frame this clip processes with
 * chroma smoothing (default; one stream synchronisation), 1 = plain, 2 = spread (dark footage).  Results are identical.   */
int    mlvfs_amd_clip_set_t16_layout(mlvfs_amd_clip_t *clip, int layout);
int    mlvfs_amd_clip_get_t16_layout(const mlvfs_amd_clip_t *clip);
/* xy = count (x,y) pairs in sensor coordinates (crop offsets included), list order
 * is application order.  kind: 0 = bad-pixel rules (cs.c:314-330), 1 = focus-pixel
 * rules incl. edges (cs.c:463-500).                                            */
int    mlvfs_amd_clip_set_pixel_map(mlvfs_amd_clip_t *clip, const int32_t *xy, size_t count, int kind, int dual_iso);
size_t mlvfs_amd_clip_get_pixel_map(const mlvfs_amd_clip_t *clip, int32_t *xy, size_t cap);

/* -- single stages on device frames ---------------------------------------- */
/* frames are `nframes` buffers spaced by the given strides (bytes)           */
int mlvfs_amd_unpack_dev(const mlvfs_amd_geom_t *geom, const void *d_packed, size_t packed_stride,
                         void *d_out, size_t out_stride, int nframes, void *stream);
int mlvfs_amd_chroma_smooth_dev(const mlvfs_amd_geom_t *geom, const void *d_in, void *d_out, size_t stride,
                                int method, int nframes, void *stream);
/* detection on one frame; fills the clip's pixel map (synchronises the stream) */
int mlvfs_amd_detect_bad_pixels_dev(mlvfs_amd_clip_t *clip, const void *d_frame, int aggressive, void *stream);
/* ordered repair with the clip's pixel map, in place                         */
int mlvfs_amd_fix_pixels_dev(mlvfs_amd_clip_t *clip, void *d_frames, size_t stride, int nframes, void *stream);
/* stripes: histogram of rows [row0,row1) of one frame.
 *   count pass  -> number of accepted add_pixel calls (synchronises)
 *   hist  pass  -> adds into d_hist (int32[8][65536]) and d_num (int32[8]);
 *                  d_rand = device array of rand()%1024 values (uint16), entry
 *                  2*k and 2*k+1 belong to this shard's k-th accepted call     */
int mlvfs_amd_stripes_count_dev(const mlvfs_amd_geom_t *geom, const void *d_frame, int row0, int row1,
                                int64_t *accepted, void *stream);
int mlvfs_amd_stripes_hist_dev(const mlvfs_amd_geom_t *geom, const void *d_frame, int row0, int row1,
                               const void *d_rand, int64_t n_rand, void *d_hist, void *d_num, void *stream);
/* host: histograms -> coefficients (stripes.c:207-246); coeffs of under-filled
 * histograms are left as passed in                                           */
int mlvfs_amd_stripes_solve(const int32_t *hist, const int32_t num[8], int frame_size, int32_t coeffs[8]);
/* whole computation for one device frame into the clip.  rand_mode 0: consume
 * libc rand() like the reference; 1: private glibc-compatible stream, seed 1   */
int mlvfs_amd_stripes_compute_dev(mlvfs_amd_clip_t *clip, const void *d_frame, int frame_size, int rand_mode,
                                  void *stream);
int mlvfs_amd_stripes_apply_dev(const mlvfs_amd_clip_t *clip, void *d_frames, size_t stride, int nframes,
                                void *stream);
/* glibc TYPE_3 rand() restatement: out[i] = rand()%1024 for calls skip..skip+n-1
 * after srand(seed)                                                            */
void mlvfs_amd_rand_stream(uint16_t *out, size_t n, uint64_t skip, unsigned seed);
/* the same values generated on the device into d_out (16-byte aligned device memory); synchronises the stream */
int mlvfs_amd_rand_stream_dev(void *d_out, size_t n, uint64_t skip, unsigned seed, void *stream);

/* -- fused steady-state pipeline (process_frame order, main.c:942-997) ------ */
/* packed 14-bit stream -> [pixel map repair] -> [chroma smooth] -> [stripes
 * apply] -> 16-bit frames, one pass over HBM (3.75 B/px).  Stages are enabled
 * by cs_method (0,2,3,5), fix_pixels and apply_stripes (uses the clip state).
 * Clips of another bit depth (10, 12, ... bits: geom.bpp) are unpacked to 16 bits first and then take the kernel's
 * 16-bit input path (one more pass over HBM).                                */
int mlvfs_amd_process_frames_dev(mlvfs_amd_clip_t *clip, const void *d_packed, size_t packed_stride,
                                 void *d_out, size_t out_stride, int nframes,
                                 int cs_method, int fix_pixels, int apply_stripes, void *stream);

/* The same stages on frames that are already 16-bit pixels in HBM (e.g. decoded LJ92 payloads); d_out != d_frames. */
int mlvfs_amd_process_unpacked_dev(mlvfs_amd_clip_t *clip, const void *d_frames, size_t stride, void *d_out, size_t out_stride,
                                   int nframes, int cs_method, int fix_pixels, int apply_stripes, void *stream);

/* The same fused pipeline for frames in HOST memory (a reader that has MLV payloads in RAM): chunks
 * of chunk_frames frames (<= 0: 8) travel H2D -> kernels -> D2H on three streams so that both copy directions and the
 * kernels overlap; returns when h_out is complete.  Strides are bytes between frames.  Full PCIe speed needs page-locked
 * buffers (mlvfs_amd_host_alloc, or memory the caller registered with the HIP runtime); pageable memory works, slower. */
int mlvfs_amd_process_frames_host(mlvfs_amd_clip_t *clip, const void *h_packed, size_t packed_stride, void *h_out,
                                  size_t out_stride, int nframes, int cs_method, int fix_pixels, int apply_stripes,
                                  int chunk_frames);
/* Page-locked buffers, pooled: a freed buffer is kept (up to 2 GiB in all) and handed out again for the next request of its
 * size, so a host may allocate and free one per frame like process_frame does with malloc (main.c:931).  With frame buffers
 * from here the drop-in symbols of PART 1 copy at the link's speed instead of the pageable path's. */
void *mlvfs_amd_host_alloc(size_t bytes);
/* Gives the buffer back to the pool.  A pointer that did not come from mlvfs_amd_host_alloc, and a buffer freed a second time,
 * are reported on stderr and otherwise ignored (the library frees nothing it does not own). */
void mlvfs_amd_host_free(void *p);
/* 1 when [p, p + bytes) lies inside a live buffer of mlvfs_amd_host_alloc (page-locked and mapped: the GPU can address it), else 0 */
int mlvfs_amd_host_owns(const void *p, size_t bytes);
/* bytes from p to the end of the live mlvfs_amd_host_alloc buffer that holds it (its size rounded up to 64 KiB); 0: in none */
size_t mlvfs_amd_host_size(const void *p);
/* is p inside any buffer mlvfs_amd_host_alloc has handed out, live or already freed?  (the allocation shim's free(): a second free of a
 * pool buffer goes back to the pool, which reports it, never to the C library) */
int mlvfs_amd_host_knows(const void *p);
/* Returns the pool's cached (free) buffers to the runtime; result: bytes released. */
size_t mlvfs_amd_host_trim(void);

/* -- LJ92 payloads ---------------------------------------- */
/* Lossless-JPEG frames of a compressed clip (MLV_VIDEO_CLASS_FLAG_LJ92; main.c:617-681: lj92_open + lj92_decode + the
 * untiling loop) decoded on the GPU.  streams[i] / sizes[i]: frame i's JPEG stream in HOST memory (the VIDF payload
 * behind its 4-byte size word); output: xres x yres 16-bit pixels per frame at d_out + i * out_stride, in the layout
 * dng_get_image_data produces, ready for the stages above.  One component, one Huffman table, predictors 0..7, like the
 * reference's decoder (6, what MLV writers use, and 1 are the fast ones).  Synchronises `stream` before returning.  dims of mlvfs_amd_lj92_info:
 * {width, height, bits, predictor} of the JPEG itself.                                                             */
int mlvfs_amd_lj92_info(const void *stream, size_t size, int dims[4]);
int mlvfs_amd_lj92_decode_dev(const void *const *streams, const size_t *sizes, int nframes, int xres, int yres,
                              void *d_out, size_t out_stride, void *stream);

/* The reference decoder's own three calls (lj92.h:40-58), what get_image_data makes of an LJ92 frame (main.c:626-647: lj92_open,
 * lj92_decode into a temporary buffer, then its own untiling loop): with them `lj92.o` can leave MLVFS's link too and the decode
 * runs on the GPU.  Values in the decoder's own order, width x height of the JPEG.  Only what MLVFS passes is supported:
 * skiplen 0, linearize NULL (anything else: LJ92_ERROR_CORRUPT, -1).                                                 */
typedef struct _ljp *lj92;
int lj92_open(lj92 *lj, uint8_t *data, int datalen, int *width, int *height, int *bitdepth);
void lj92_close(lj92 lj);
int lj92_decode(lj92 lj, uint16_t *target, int tlen, int skiplen, uint16_t *linearize, int linlen);
/* The reference's encoder (lj92.h:65-68, lj92.c:711-1144; nothing in MLVFS calls it -- it completes the export table of lj92.o):
 * a width x height tile read from `image` in runs of readLength values skipLength apart, optionally through a delinearisation
 * table, written as one-component lossless JPEG with predictor 6 -- byte for byte the reference's stream, its peculiar Huffman
 * table included (csrc/lj92enc.cpp).  *encoded is malloc'd, the caller frees it.  Host memory in and out; histogram, bit packing
 * and byte stuffing run on the GPU (csrc/k_lj92enc.hip).  Returns 0, LJ92_ERROR_NO_MEMORY (-2), or LJ92_ERROR_CORRUPT (-1) where
 * the reference would leave its own arrays (17-bit differences, all 17 classes in use, values beyond the table) or HIP fails. */
int lj92_encode(uint16_t *image, int width, int height, int bitdepth, int readLength, int skipLength,
                uint16_t *delinearize, int delinearizeLength, uint8_t **encoded, int *encodedLength);
/* Optional (needs three changed lines in main.c): lj92_decode plus the untiling loop get_image_data runs behind it
 * (main.c:646-667: 8-22 ms per 3584x1320 frame on a host core) in one call: xres x yres pixels in host memory, decoded and untiled
 * on the GPU.  0, or the lj92.h error codes.                                                                             */
int mlvfs_amd_lj92_decode_untiled(lj92 lj, uint16_t *dst, int xres, int yres);
/* host-only test hook: the encoder's Huffman table for a histogram of the 17 classes; out[68] = bits[1..16], number of DHT
 * values, the 17 values, then length and code per class.  0, or -1 (error string set) where lj92_encode would refuse.          */
int mlvfs_amd_lj92_encode_table(const uint32_t hist[17], int npix, int *out);

/* -- LZMA payloads ----------------------------------------- */
/* One VIDF payload of an LZMA-compressed clip (MLV_VIDEO_CLASS_FLAG_LZMA; main.c:598-616): [u32 size of the packed frame][5 LZMA
 * property bytes][LZMA stream] -> the packed frame, exactly what LzmaUncompress leaves in lzma_out for dng_get_image_data.  Host
 * code (the entropy decoding of a frame is one serial chain; the reader decodes one frame per thread): no HIP device needed.
 * Returns 0 and the decoded size, or LzmaDecode's error code (1 data, 4 unsupported properties, 6 input ends early), or
 * MLVFS_AMD_ERR_ARG when the buffer is smaller than the size word says.                                             */
int mlvfs_amd_lzma_uncompress(const void *payload, size_t size, void *dst, size_t dst_cap, size_t *out_size);

/* -- MLV container reader and prefetcher (host code) -------- */
/* Opens <name>.MLV and its chunks .M00, .M01, ... (index.c:367-424) and builds, once, what MLVFS rebuilds for every
 * frame it serves: the XREF index (index.c:216-341; use_idx_file != 0: taken from <name>.IDX when that exists, written
 * there otherwise, like get_index) and the block headers that belong to each video frame (main.c:429-558).  NULL on
 * failure.  Needs no HIP device except for mlvfs_amd_mlv_process.                                                     */
void  *mlvfs_amd_mlv_open(const char *mlv_path, int use_idx_file);
void   mlvfs_amd_mlv_close(void *reader);
int    mlvfs_amd_mlv_frame_count(const void *reader);            /* = mlv_get_frame_count, index.c:488-527 */
int    mlvfs_amd_mlv_chunk_count(const void *reader);
/* the XREF block (header + entries) exactly as make_index builds it; returns its size, copies if cap suffices */
size_t mlvfs_amd_mlv_xref(const void *reader, void *dst, size_t cap);
/* = mlv_get_frame_headers(path, index, out): 1 = found and a RAWI block precedes it, 0 otherwise */
int    mlvfs_amd_mlv_frame_headers(const void *reader, int index, struct frame_headers *out);
/* payloads of `count` frames (uncompressed clips) into dst, `stride` bytes apart, read by io_threads threads (<= 0: 8) */
int    mlvfs_amd_mlv_read_frames(const void *reader, int first, int count, void *dst, size_t stride, int io_threads);
/* file -> fused pipeline -> h_out: batches of batch_frames frames (<= 0: 32) are read into page-locked staging by
 * io_threads threads while the previous batch goes through mlvfs_amd_process_frames_host (LJ92 clips: through the GPU
 * decoder and mlvfs_amd_process_unpacked_dev).  `clip` must have the frames' geometry; a reader's staging and device buffers
 * belong to the device of the thread that first streams from it.                                                     */
int    mlvfs_amd_mlv_process(const void *reader, mlvfs_amd_clip_t *clip, int first, int count, void *h_out, size_t out_stride,
                             int cs_method, int fix_pixels, int apply_stripes, int batch_frames, int io_threads);
/* A DUAL-ISO clip, file -> batched full conversion -> h_out (main.c:942 + 956-959 per frame, cr2hdr20 with the headers' levels):
 * batches of batch_frames frames (<= 0: 8) are read while the previous batch is unpacked and converted in ONE submission
 * (mlvfs_amd_cr2hdr20_batch_dev) and the batch before travels back.  results[i] = 1: frame first + i converted (black and white
 * level 4x the headers'), 0: no dual-ISO frame, h_out holds it unpacked.  Plain and LZMA clips. */
int    mlvfs_amd_mlv_process_dualiso(const void *reader, int first, int count, void *h_out, size_t out_stride, int interp_method,
                                     int fullres, int use_alias_map, int chroma_smooth, int batch_frames, int io_threads, int *results);

/* -- animated GIF preview (gif.c:82-244) ---------------------- */
/* = gif_get_size: size of the preview file of a clip with these frame headers                                        */
size_t mlvfs_amd_gif_size(const struct frame_headers *frame_headers);
/* The preview of 10 frames (host memory, `stride` bytes apart: packed payloads of geom->bpp bits with one word of slack behind
 * each when packed != 0, else 16-bit frames): the pixel picking and the gamma map run on the GPU, the file's framing on the host;
 * file receives mlvfs_amd_gif_size bytes, byte for byte what gif_get_data builds.  packed == 2: only the rows the preview reads
 * (it picks ONE pixel of every 4x4 block, gif.c:197): yres / 4 row pieces per frame, stride / (yres / 4) bytes each, piece y
 * starting at the 16-bit word of the payload that holds pixel y * 4 * (xres / 4 * 4) + 1 (mlvfs_amd_mlv_gif_data reads just those
 * from an uncompressed clip: a quarter of the file).                                                                  */
int mlvfs_amd_gif_render(const mlvfs_amd_geom_t *geom, const void *h_frames, size_t stride, int packed, int nframes, uint8_t *file);
/* = gif_get_data on an opened clip (frames k * count / 10, k = 0..9; uncompressed, LZMA and LJ92 clips): copies
 * min(max_size, size - offset) bytes from `offset` on, returns max_size (0 on failure).                              */
size_t mlvfs_amd_mlv_gif_data(const void *reader, uint8_t *output_buffer, off_t offset, size_t max_size);
/* gif.h:27-28, the two calls MLVFS makes for a clip's _PREVIEW.GIF (main.c:1018-1022, 1212): gif_get_data opens the clip at `path`
 * with the library's reader and is mlvfs_amd_mlv_gif_data on it; with them `gif.o` can leave the link too.             */
size_t gif_get_data(const char *path, uint8_t *output_buffer, off_t offset, size_t max_size);
size_t gif_get_size(struct frame_headers *frame_headers);

/* dual-ISO preview on one device frame (hdr.c:40-227); returns 1 / 0 / <0    */
int mlvfs_amd_hdr_preview_dev(const mlvfs_amd_geom_t *geom, void *d_frame, size_t max_size, void *stream);

/* deflicker of main.c:895-906 on one device frame: the median of every second pixel (the reference's
 * 16-bit histogram counters included) against `target` -> raw_info.exposure_bias = {(int)(log2(...) * 10000), 10000}, which
 * dng_get_header_data writes as BaselineExposure.  size_bytes: the frame's size in bytes, as main.c:943 passes it.
 * Synchronises the stream.                                                                                        */
int mlvfs_amd_deflicker_dev(const mlvfs_amd_geom_t *geom, const void *d_frame, size_t size_bytes, int target,
                            int32_t exposure_bias[2], void *stream);

/* full dual-ISO conversion of one device frame, in place (hdr.c:1774-1957 without the
 * pixel-map repairs); returns 1 converted / 0 not dual ISO or not convertible / <0 error.
 * mlvfs_amd_dualiso_reset forgets the per-black-level table caches (a fresh process).  */
int mlvfs_amd_cr2hdr20_dev(const mlvfs_amd_geom_t *geom, void *d_frame, int interp_method, int fullres, int use_alias_map,
                           int chroma_smooth, void *stream);
/* The same for `nframes` frames of one clip geometry, `stride` bytes apart, in ONE submission: every stage is launched once for the
 * whole batch (the frame index is a grid dimension; AMaZE runs nframes x 319 tiles at 3584x1320 instead of 319 with a two-tile
 * tail), the integer decisions of hdr.c:441-636, 250-300 and 638-772 (pattern, bright / dark fields, white levels, order
 * statistics, highlight rows, slope fit) are made by single-workgroup kernels, and ONE host round trip per batch remains -- the
 * libm scalars (log2 of the fitted slope and of the levels) and the reference's progress lines -- plus the final wait.
 * results[f] (host memory) = 1 converted / 0 left alone (not dual ISO, detection failed), like the reference's return value
 * frame by frame; the function returns 0, or < 0 on an error of the library.  Frames are processed as if in order: the table caches
 * take the white level of the first frame that converts (hdr.c:1080,1240,1575,1672).  Work memory: ~0.93 GB per 3584x1320 frame
 * of the batch with the AMaZE interpolation, kept by the calling thread. */
int mlvfs_amd_cr2hdr20_batch_dev(const mlvfs_amd_geom_t *geom, void *d_frames, size_t stride, int nframes, int interp_method,
                                 int fullres, int use_alias_map, int chroma_smooth, int *results, void *stream);
void mlvfs_amd_dualiso_reset(void);
/* gives back the dual-ISO work memory the calling thread holds (it grows with the largest batch and is otherwise kept until the
 * thread ends) */
void mlvfs_amd_dualiso_trim(void);
/* the global decisions of the calling thread's last conversion: {RGGB?, is_bright[0..3] as bits 3..0, white, white of the bright
 * rows (20 bit), a, b of the exposure fit (hdr.c:638-823), ISO difference in EV, darkened white} */
void mlvfs_amd_dualiso_last_scalars(double out[8]);

/* AMaZE demosaic of a float RGGB plane in HBM (amaze_demosaic_RT.c:113, as called from hdr.c:1034 with
 * winx = winy = 0): d_raw and the three outputs are height rows of width floats (no row padding), values in
 * the caller's scale.  width must be a multiple of 4 (the reference's SSE2 build needs that for a fully
 * written green plane) and the plane at least 36x36.  Bit-identical to the x86-64 reference build.        */
int mlvfs_amd_amaze_demosaic_dev(const float *d_raw, int width, int height, float *d_red, float *d_green, float *d_blue, void *stream);
/* The split of a width x height plane between the two AMaZE kernels: the first nfx x nfy tiles (128-pixel grid, origin -16) are complete
 * -- 160 rows and columns inside the image, not the head of a chain of incomplete tiles -- and are row-streamed through LDS.  No GPU needed. */
void mlvfs_amd_amaze_rows_extent(int width, int height, int *nfx, int *nfy);
/* Test hook: tiles that head a chain of incomplete tiles WITHOUT output (widths that are a multiple of 128) through the row-streamed
 * kernel as well.  mode -1: MLVFS_AMD_AMAZE_ROWS_EXTRA decides (default off: measured 1 % slower per batch), 0 / 1: forced; returns
 * the mode before; *count (may be NULL): the number of such tiles of a width x height plane.  Results are bit-identical either way. */
int mlvfs_amd_amaze_rows_extra_mode(int mode, int width, int height, int *count);
/* Debug: the same with the tile planes copied out (26 planes per 160x160 tile, the layout of k_amaze.hip's block).  mode 0: every
 * tile through k_amaze.hip, blocks in tile order; mode 1: the complete tiles through k_amaze_rows.hip (LDS row streaming), their
 * planes numbered ty * nfx + tx; nfx x nfy = the complete tiles of the plane.  Synchronous. */
int mlvfs_amd_amaze_debug(const float *d_raw, int width, int height, float *d_red, float *d_green, float *d_blue, int mode,
                          float *d_planes, size_t planes_floats, int *nfx, int *nfy);

/* HIP-event timer around the dominant kernel (k_frame) of the calling thread's
 * launches, recorded on the stream the kernel is launched on (bench.py's
 * roofline figure).  begin: arm for up to max_launches launches.  end: waits for
 * the events, writes one duration in milliseconds per launch, returns the count. */
int mlvfs_amd_timer_begin(int max_launches);
int mlvfs_amd_timer_end(float *ms, int cap);

/* FRAME BRACKET.  process_frame (mlvfs/main.c:908-1005) brackets its stages with mlvfs_load_chunks (main.c:923) and
 * mlvfs_close_chunks (main.c:998, resource_manager.c:285-317) on the calling thread and nothing of MLVFS reads image_buffer->data
 * in between (deflicker's hist_add is served, see below).  Between mlvfs_amd_frame_begin() and mlvfs_amd_frame_end() on one thread
 * the drop-in stages of PART 1 therefore do not write the host buffer: unpack, bad-pixel repair (once the clip's map is known),
 * chroma smoothing and stripes are RECORDED and run as one launch of the fused kernel on the packed payload, and the frame is
 * downloaded once, inside mlvfs_amd_frame_end().  integration/mlvfs_amd_wrap.c + `-Wl,--wrap=mlvfs_load_chunks
 * -Wl,--wrap=mlvfs_close_chunks` make those two calls of the UNCHANGED main.c the bracket.
 * Outside a bracket every call completes before it returns -- gif_get_data (gif.c:90-221) uses load_chunks / close_chunks
 * directly and reads the frame right after get_image_data.  Both calls are cheap no-ops for a thread that does no pixel work in
 * between (mlv_get_frame_headers, main.c:434-555, brackets a header walk the same way); MLVFS_AMD_DEFER=0 in the environment
 * disables the deferral.  0 = ok / the host buffer is current. */
int mlvfs_amd_frame_begin(void);
int mlvfs_amd_frame_end(void);
/* inside a bracket: fetch the frame at `image_data` NOW (a host that wants the pixels before the bracket ends).  Does nothing
 * outside a bracket and for a buffer nothing is pending for.  0 = the host buffer is current. */
int mlvfs_amd_frame_sync(void *image_data);
/* out[0]: frames of this process whose recorded stages ran as ONE fused launch at the fetch; out[1]: frames whose recorded stages
 * had to run earlier (a call outside process_frame's order, the first frame of a clip, a focus-pixel map, pattern noise, dual ISO,
 * hist_add on the frame). */
void mlvfs_amd_dropin_stats(long long out[2]);
/* What the drop-in stages moved between host and device since the process started: {frame uploads, frame downloads, bytes up, bytes
 * down} (pixel repairs that fetch their few patched pixels are not counted).  Inside a frame bracket a frame costs one upload of its
 * payload and one download, whatever stages run in between. */
void mlvfs_amd_dropin_transfers(long long out[4]);
/* MLVFS_AMD_DROPIN_PROFILE=1 in the environment: where the wall time of the bracketed frames went, in ms summed over all threads:
 * {inside dng_get_image_data, of it the upload call, of it the wait for the upload's end, inside the recorded stage calls, inside
 * mlvfs_amd_frame_end, of it the fused launch's calls, of it the download call + the wait for it, number of frames}. */
void mlvfs_amd_dropin_profile(double out[8]);
/* Test hook: the calling thread's next fused launch of a bracketed frame (what = 1) or next frame download (what = 2) reports a HIP
 * error without touching the device.  What the caller then finds -- a zeroed frame and one line on stderr, never the bytes its
 * malloc returned -- is the library's failure policy ("When the device fails"; tests/test_failure_policy.py). */
void mlvfs_amd_test_fail_next(int what);
/* Test hook, host only: can the dither of stripes_compute_correction be taken from the application's libc generator in bulk
 * (glibc's TYPE_3 state layout, checked once per process on a generator of the library's own; csrc/runtime.cpp)?  1 yes, 0 no (the
 * values are then drawn by calling rand()).  Leaves the application's stream where it was.                              */
int mlvfs_amd_test_rand_layout(void);
/* Test hook, host only: which device the k-th worker thread without a device of its own choosing is bound to on a node whose cards
 * report `bus_ids[0..n)` (HIP ordinal -> PCI bus id): round-robin over the cards IN THE ORDER OF THEIR BUS IDS (csrc/runtime.cpp:
 * thread_ctx; resource_manager.c:111-118 is the pool this replaces).  device_of[0..workers) receives HIP ordinals.            */
int mlvfs_amd_test_device_order(const char *const *bus_ids, int n, int workers, int *device_of);

/* Test hook, host only: how the streaming forms of the fused pass (csrc/k_frame_s.hip, k_frame_p5 in csrc/k_frame_p.hip) cut a frame
 * into tasks -- columns of 62 items (8 pixels each), segments of seg_rows cell rows, `fold` segments of a narrow last column side by
 * side in one wave (a last column of <= 14 items: 4, <= 30 items: 2), tasks per frame.                                              */
int mlvfs_amd_test_stream_plan(int width, int height, int seg_rows, int *cols, int *segs, int *fold, int *tasks_per_frame);

/* self tests that need no GPU (selection networks, LUT identities): 0 = pass */
int mlvfs_amd_selftest_host(void);
/* the library's host EV tables against raw2ev_lin[16384] (index = pixel - black) and ev2raw[24 * 32768] (index 0 = EV -10 * 32768):
 * 0 = identical (main.c:128-196; no GPU needed) */
int mlvfs_amd_selftest_tables(const int32_t *raw2ev_lin, const int32_t *ev2raw);

#ifdef __cplusplus
}
#endif
#endif
