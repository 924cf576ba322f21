/* mlvfs_amd_wrap.c -- the ONE file a MLVFS maintainer adds to the link (next to -lmlvfs_amd) to let a frame cross PCIe once in
 * each direction with main.c, gif.c and resource_manager.c byte for byte unchanged:
 *
 *     OBJS    += /path/to/repo/integration/mlvfs_amd_wrap.o
 *     LDFLAGS += -Wl,--wrap=mlvfs_load_chunks -Wl,--wrap=mlvfs_close_chunks
 *
 * process_frame (mlvfs/main.c:908-1005) opens its clip with mlvfs_load_chunks (main.c:923) before the first pixel stage and
 * closes it with mlvfs_close_chunks (main.c:998) after the last one, on the calling thread (definitions:
 * resource_manager.c:285-317).  GNU ld's --wrap sends main.o's calls here: the load arms the library's frame bracket, the close
 * ends it -- which is where the frame's recorded stages run as one fused launch and image_buffer->data is written, once
 * (include/mlvfs_amd.h: mlvfs_amd_frame_begin / mlvfs_amd_frame_end).  mlv_get_frame_headers and mlv_read_debug_log use the same
 * pair around header walks (main.c:338-417, 434-555): an empty bracket costs two thread-local stores.  gif_get_data calls
 * load_chunks / close_chunks of index.c directly (gif.c:90,210), is NOT wrapped, and so gets every unpack at once.
 */
#include <stdint.h>
#include <stdio.h>

#include "mlvfs_amd.h"

FILE **__real_mlvfs_load_chunks(const char *path, uint32_t *chunk_count);
void __real_mlvfs_close_chunks(FILE **chunk_files, uint32_t chunk_count);

FILE **__wrap_mlvfs_load_chunks(const char *path, uint32_t *chunk_count)
{
    FILE **chunk_files = __real_mlvfs_load_chunks(path, chunk_count);
    if (chunk_files && chunk_count && *chunk_count) mlvfs_amd_frame_begin();   /* the failure branch (main.c:924-928) never closes */
    return chunk_files;
}

void __wrap_mlvfs_close_chunks(FILE **chunk_files, uint32_t chunk_count)
{
    /* image_buffer->data is current from here on.  Should the fused launch or the download fail, the library has zeroed the frame
     * (never process_frame's malloc'ed bytes) and said so on stderr; the line below names the clip's side of it, like the
     * reference's err_printf does for its own failures */
    if (mlvfs_amd_frame_end() != 0) fprintf(stderr, "MLVFS: frame served black: %s\n", mlvfs_amd_last_error());
    __real_mlvfs_close_chunks(chunk_files, chunk_count);
}
