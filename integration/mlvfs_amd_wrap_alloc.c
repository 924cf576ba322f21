/* mlvfs_amd_wrap_alloc.c -- OPTIONAL second shim (link flags only, like mlvfs_amd_wrap.c): process_frame's large buffers come from
 * the library's pool of page-locked, GPU-mapped memory, so that the fused kernel at the end of a frame bracket writes
 * image_buffer->data itself, over the link, instead of the runtime copying it there (no copy engine, no staging through the
 * runtime's bounce buffers, no page faults on a fresh malloc):
 *
 *     OBJS    += /path/to/repo/integration/mlvfs_amd_wrap_alloc.o
 *     LDFLAGS += -Wl,--wrap=malloc -Wl,--wrap=calloc -Wl,--wrap=realloc -Wl,--wrap=free
 *
 * What it changes: malloc / calloc of MLVFS's OWN objects (GNU ld's --wrap redirects the references of the objects in the link, not
 * libc's or any shared library's) for sizes of at least MLVFS_AMD_POOL_MIN bytes (1 MiB) -- the frame buffer of main.c:931, the
 * payload buffer of main.c:687 -- go to mlvfs_amd_host_alloc; free recognises such a pointer (mlvfs_amd_host_knows: a range check without a lock for everything else) and returns it to
 * the pool (resource_manager.c:144 frees the frame buffer); everything else goes to the C library unchanged.  A block that MLVFS
 * allocates and a shared library frees (or the reverse) must stay below the threshold; MLVFS has none of that size.
 * Measured (tools/dropin_bench_c.sh, 3584x1320, unpack + bad pixels + cs5x5 + stripes): 16 worker threads 2 420 -> 3 618 frames/s. */
#include <stddef.h>
#include <string.h>

#include "mlvfs_amd.h"

#ifndef MLVFS_AMD_POOL_MIN
#define MLVFS_AMD_POOL_MIN ((size_t)1 << 20)
#endif

void *__real_malloc(size_t n);
void *__real_calloc(size_t n, size_t m);
void *__real_realloc(void *p, size_t n);
void __real_free(void *p);

void *__wrap_malloc(size_t n)
{
    if (n >= MLVFS_AMD_POOL_MIN) {
        void *p = mlvfs_amd_host_alloc(n);
        if (p) return p;                                   /* (no device / no page-locked memory left: the C library's) */
    }
    return __real_malloc(n);
}

void *__wrap_calloc(size_t n, size_t m)
{
    if (m && n > (size_t)-1 / m) return NULL;
    const size_t bytes = n * m;
    if (bytes >= MLVFS_AMD_POOL_MIN) {
        void *p = mlvfs_amd_host_alloc(bytes);
        if (p) { memset(p, 0, bytes); return p; }          /* (a pooled buffer has been used before) */
    }
    return __real_calloc(n, m);
}

void __wrap_free(void *p)
{
    /* (a pool buffer freed a second time is the pool's to report: handed to the C library it would corrupt its heap) */
    if (p && mlvfs_amd_host_knows(p)) mlvfs_amd_host_free(p);
    else __real_free(p);
}

void *__wrap_realloc(void *p, size_t n)
{
    const size_t have = p ? mlvfs_amd_host_size(p) : 0;
    if (have && n == 0) { mlvfs_amd_host_free(p); return NULL; }      /* realloc(p, 0) frees (a malloc(0) that returned NULL leaked the block) */
    if (have) {                                            /* a pooled block grows or shrinks by copy */
        if (n <= have && n >= MLVFS_AMD_POOL_MIN) return p;
        void *q = __wrap_malloc(n);
        if (!q) return NULL;
        memcpy(q, p, have < n ? have : n);
        mlvfs_amd_host_free(p);
        return q;
    }
    if (n >= MLVFS_AMD_POOL_MIN && p) {                    /* a small block that grows past the threshold: into the pool */
        void *q = mlvfs_amd_host_alloc(n);
        if (q) {
            void *r = __real_realloc(p, n);                /* (the C library knows the old size: let it move the bytes, then copy) */
            if (!r) { mlvfs_amd_host_free(q); return NULL; }
            memcpy(q, r, n);
            __real_free(r);
            return q;
        }
    }
    return __real_realloc(p, n);
}
