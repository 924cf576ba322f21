/* The part resource_manager.c:285-317 plays for tests/c_host.c and tools/dropin_bench_c.c: mlvfs_load_chunks /
 * mlvfs_close_chunks in a translation unit of their own (GNU ld's --wrap only redirects references that are undefined in the
 * object that makes them, like main.o's are).  One "chunk": the file itself. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

FILE **mlvfs_load_chunks(const char *path, uint32_t *chunk_count)
{
    *chunk_count = 0;
    FILE **chunks = malloc(sizeof *chunks);
    if (!chunks) return NULL;
    chunks[0] = fopen(path, "rb");
    if (!chunks[0]) { free(chunks); return NULL; }
    *chunk_count = 1;
    return chunks;
}

void mlvfs_close_chunks(FILE **chunk_files, uint32_t chunk_count)
{
    if (!chunk_files) return;
    for (uint32_t i = 0; i < chunk_count; i++)
        if (chunk_files[i]) fclose(chunk_files[i]);
    free(chunk_files);
}
