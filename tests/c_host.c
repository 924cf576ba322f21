/* A plain C host -- no Python, no torch -- that links libmlvfs_amd.so the way MLVFS would
 * and runs process_frame's call sequence (mlvfs/main.c:923-998) on one frame read from a file -- including the
 * mlvfs_load_chunks / mlvfs_close_chunks pair around it (tests/c_host_chunks.c plays resource_manager.c:285-317).  Built twice
 * by tests/test_gpu_c_host.py: as it is, and with integration/mlvfs_amd_wrap.c + -Wl,--wrap=... (the frame bracket); this file
 * is the same in both and calls nothing of the library beyond the reference's symbols.
 *   c_host <in: packed 14-bit words> <out: u16 pixels> w h black white cs bad_pixels stripes [dual_iso: 1 preview, 2 full]
 * The caller's table accessors (get_raw2ev / get_ev2raw, mlvfs/main.c:128-196) are deliberately NOT provided: the
 * library then builds the tables itself, as it does under Python.  tests/test_gpu_c_host.py compares the output with
 * the oracle. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mlvfs_abi.h"
#include "mlvfs_amd.h"

FILE **mlvfs_load_chunks(const char *path, uint32_t *chunk_count);        /* resource_manager.h */
void mlvfs_close_chunks(FILE **chunk_files, uint32_t chunk_count);

int main(int argc, char **argv)
{
    if (argc != 10 && argc != 11) { fprintf(stderr, "usage\n"); return 2; }
    const int dual_iso = argc == 11 ? atoi(argv[10]) : 0;
    const int w = atoi(argv[3]), h = atoi(argv[4]), black = atoi(argv[5]), white = atoi(argv[6]);
    const int cs = atoi(argv[7]), bad = atoi(argv[8]), stripes = atoi(argv[9]);
    struct frame_headers fh;
    memset(&fh, 0, sizeof fh);
    fh.rawi_hdr.xRes = (uint16_t)w; fh.rawi_hdr.yRes = (uint16_t)h;
    fh.rawi_hdr.raw_info.width = w; fh.rawi_hdr.raw_info.height = h;
    fh.rawi_hdr.raw_info.bits_per_pixel = 14;
    fh.rawi_hdr.raw_info.pitch = w * 14 / 8;
    fh.rawi_hdr.raw_info.frame_size = w * h * 14 / 8;
    fh.rawi_hdr.raw_info.black_level = black; fh.rawi_hdr.raw_info.white_level = white;
    const size_t npix = (size_t)w * h, nwords = (npix * 14 + 15) / 16 + 4;
    uint16_t *packed = calloc(nwords, 2), *img = malloc(npix * 2);
    uint32_t chunk_count = 0;
    FILE **chunk_files = mlvfs_load_chunks(argv[1], &chunk_count);           /* main.c:923 */
    if (!chunk_files || !chunk_count || !packed || !img) return 3;
    const size_t got = fread(packed, 2, nwords, chunk_files[0]);             /* get_image_data, main.c:696 */
    (void)got;
    if (dng_get_image_data(&fh, packed, (uint8_t *)img, 0, npix * 2) != npix * 2) { fprintf(stderr, "unpack failed\n"); return 4; }
    int is_dual_iso = 0;                                       /* main.c:951-973 */
    if (dual_iso == 1) is_dual_iso = hdr_convert_data(&fh, img, 0, npix * 2);
    else if (dual_iso == 2) is_dual_iso = cr2hdr20_convert_data(&fh, img, 0, 1, 1, cs, bad);
    if (!is_dual_iso) {
        fix_focus_pixels(&fh, img, 0);
        if (bad) fix_bad_pixels(&fh, img, bad == 2, 0);
    }
    if (cs && dual_iso != 2) chroma_smooth(&fh, img, cs);
    if (stripes) {
        struct stripes_correction *c = stripes_get_correction("c_host.MLV");
        if (!c) {
            c = stripes_new_correction("c_host.MLV");
            if (c) stripes_compute_correction(&fh, c, img, 0, npix);
        }
        stripes_apply_correction(&fh, c, img, 0, npix);
    }
    mlvfs_close_chunks(chunk_files, chunk_count);                            /* main.c:998 */
    FILE *f = fopen(argv[2], "wb");
    if (!f) return 5;
    fwrite(img, 2, npix, f);
    fprintf(stderr, "levels %d %d dual_iso %d\n", fh.rawi_hdr.raw_info.black_level, fh.rawi_hdr.raw_info.white_level, is_dual_iso);
    fclose(f);
    stripes_free_corrections();
    free_focus_pixel_maps();
    free(packed); free(img);
    return 0;
}
