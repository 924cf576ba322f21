/* tools/dropin_bench_c.c -- what tools/dropin_bench.py measures, from a C host with pthreads (libfuse's worker pool in MLVFS):
 * process_frame's call sequence (mlvfs/main.c:923-998: mlvfs_load_chunks, unpack, focus pixels, bad pixels, cs5x5, stripes,
 * mlvfs_close_chunks) on 3584x1320 frames, every frame in a freshly allocated buffer, T threads x N frames.  Prints one JSON line.
 *   dropin_bench_c <packed frame 0> <packed frame 1> T N pinned
 * pinned = 1: frame and input buffers from mlvfs_amd_host_alloc / _free (page-locked, pooled) instead of malloc / free.
 * The file calls NOTHING of the library beyond the reference's symbols (mlvfs_amd_host_alloc for pinned = 1 aside).  Modes:
 * MLVFS_AMD_RESIDENT=0/1 in the environment; and the same source linked with integration/mlvfs_amd_wrap.c and
 * -Wl,--wrap=mlvfs_load_chunks -Wl,--wrap=mlvfs_close_chunks (tools/dropin_bench_c.sh: "wrap"), which makes the chunk calls the
 * library's frame bracket: one upload, one fused launch, one download per frame. */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "mlvfs_abi.h"
#include "mlvfs_amd.h"

enum { W = 3584, H = 1320, BLACK = 2048, WHITE = 15000 };
FILE **mlvfs_load_chunks(const char *path, uint32_t *chunk_count);        /* resource_manager.h; here: tests/c_host_chunks.c */
void mlvfs_close_chunks(FILE **chunk_files, uint32_t chunk_count);

static const char *g_path[2];
static uint16_t *g_packed[2];
static size_t g_words;
static int g_pinned, g_frames;
static uint64_t g_hash[256];

static uint64_t fnv1a(const void *p, size_t n)
{
    const uint8_t *b = p; uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ull; }
    return h;
}

static void *buf_alloc(size_t n) { return g_pinned ? mlvfs_amd_host_alloc(n) : malloc(n); }
static void buf_free(void *p) { if (g_pinned) mlvfs_amd_host_free(p); else free(p); }

static void one_frame(int idx, int k, int keep_hash)
{
    struct frame_headers fh;
    memset(&fh, 0, sizeof fh);
    fh.file_hdr.fileGuid = 0x1234;
    fh.rawi_hdr.xRes = W; fh.rawi_hdr.yRes = H;
    fh.rawi_hdr.raw_info.width = W; fh.rawi_hdr.raw_info.height = H;
    fh.rawi_hdr.raw_info.bits_per_pixel = 14;
    fh.rawi_hdr.raw_info.pitch = W * 14 / 8;
    fh.rawi_hdr.raw_info.frame_size = W * H * 14 / 8;
    fh.rawi_hdr.raw_info.black_level = BLACK; fh.rawi_hdr.raw_info.white_level = WHITE;
    const size_t npix = (size_t)W * H;
    uint32_t chunk_count = 0;
    FILE **chunk_files = mlvfs_load_chunks(g_path[(idx + k) & 1], &chunk_count);      /* main.c:923: fopen of every chunk */
    if (!chunk_files || !chunk_count) return;
    /* get_image_data (main.c:684-703): a buffer for the payload, read from the file (here: copied from memory), unpacked, freed */
    uint16_t *packed = buf_alloc(g_words * 2);
    memcpy(packed, g_packed[(idx + k) & 1], g_words * 2);
    uint16_t *img = buf_alloc(npix * 2);                                   /* main.c:931 */
    dng_get_image_data(&fh, packed, (uint8_t *)img, 0, npix * 2);
    buf_free(packed);
    fix_focus_pixels(&fh, img, 0);
    fix_bad_pixels(&fh, img, 0, 0);
    chroma_smooth(&fh, img, 5);
    struct stripes_correction *c = stripes_get_correction("dropin_bench.MLV");
    if (!c) {
        c = stripes_new_correction("dropin_bench.MLV");
        if (c) stripes_compute_correction(&fh, c, img, 0, npix);
    }
    stripes_apply_correction(&fh, c, img, 0, npix);
    mlvfs_close_chunks(chunk_files, chunk_count);                                     /* main.c:998 */
    if (keep_hash && ((idx + k) & 1)) {
        g_hash[idx] = fnv1a(img, npix * 2);
        const char *dump = getenv("DROPIN_DUMP");                                     /* frame 1 as thread 0 got it, for the caller to compare */
        if (dump && idx == 0) { FILE *d = fopen(dump, "wb"); if (d) { fwrite(img, 2, npix, d); fclose(d); } }
    }
    buf_free(img);
}

/* A worker of a long-running host has its stream and buffers already: two frames of warm-up per thread (the library creates the
 * thread's stream and device buffers with its first call: 50-100 ms, serialised inside the runtime), a barrier, then the clock. */
static pthread_barrier_t g_start;
static double g_t0;
static double now(void);
/* which device the library bound each worker to (round-robin in PCI order: csrc/runtime.cpp) and when it finished: the multi-GPU
 * report of the drop-in path */
static int g_dev_of[256];
static double g_done[256];

static void *worker(void *arg)
{
    const int idx = (int)(intptr_t)arg;
    one_frame(idx, 0, 0);
    one_frame(idx, 1, 0);
    if (pthread_barrier_wait(&g_start) == PTHREAD_BARRIER_SERIAL_THREAD) g_t0 = now();
    pthread_barrier_wait(&g_start);
    for (int k = 0; k < g_frames; k++) one_frame(idx, k, k >= g_frames - 2);
    g_done[idx] = now();
    g_dev_of[idx] = mlvfs_amd_thread_device();
    return NULL;
}

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv)
{
    if (argc != 6) { fprintf(stderr, "usage: dropin_bench_c packed0 packed1 threads frames pinned\n"); return 2; }
    const int T = atoi(argv[3]);
    g_frames = atoi(argv[4]);
    g_pinned = atoi(argv[5]);
    if (T < 1 || T > 256) return 2;
    g_words = ((size_t)W * H * 14 + 15) / 16 + 4;
    for (int i = 0; i < 2; i++) {
        g_path[i] = argv[1 + i];
        g_packed[i] = calloc(g_words, 2);
        FILE *f = fopen(argv[1 + i], "rb");
        if (!f || fread(g_packed[i], 2, g_words - 4, f) == 0) { fprintf(stderr, "cannot read %s\n", argv[1 + i]); return 3; }
        fclose(f);
    }
    FILE *quiet = freopen("/dev/null", "w", stdout);       /* the bad-pixel list goes to stdout like the reference's (cs.c:307-311) */
    (void)quiet;
    srand(1);
    one_frame(0, 0, 0);                                    /* clip state (map, coefficients) from frame 0 */
    one_frame(0, 1, 0);
    double fps[2];
    const int counts[2] = { 1, T };
    for (int r = 0; r < 2; r++) {
        pthread_t th[256];
        pthread_barrier_init(&g_start, NULL, counts[r]);
        for (int i = 0; i < counts[r]; i++) pthread_create(&th[i], NULL, worker, (void *)(intptr_t)i);
        for (int i = 0; i < counts[r]; i++) pthread_join(th[i], NULL);
        fps[r] = counts[r] * g_frames / (now() - g_t0);
        pthread_barrier_destroy(&g_start);
    }
    int same = 1;
    uint64_t ref = 0;
    for (int i = 0; i < T; i++) if (g_hash[i]) { if (!ref) ref = g_hash[i]; same = same && g_hash[i] == ref; }
    const char *mode = getenv("MLVFS_AMD_RESIDENT");
    long long st[2];
    mlvfs_amd_dropin_stats(st);
    fprintf(stderr, "{\"host\": \"C, pthreads\", \"resident\": \"%s\", \"fused_at_fetch\": %lld, \"run_early\": %lld, \"pinned_frame_buffers\": %s, \"frames_per_thread\": %d, \"fps_1_threads\": %.1f, "
            "\"fps_%d_threads\": %.1f, \"frame1_hash\": \"%016llx\", \"identical_between_threads\": %s}\n", mode ? mode : "0", st[0], st[1],
            g_pinned ? "true" : "false", g_frames, fps[0], T, fps[1], (unsigned long long)ref, same ? "true" : "false");
    {                                                      /* per device: workers, frames per second until its last worker finished */
        const int ndev = mlvfs_amd_device_count();
        fprintf(stderr, "{\"devices_visible\": %d, \"device_of_worker\": [", ndev);
        for (int i = 0; i < T; i++) fprintf(stderr, "%s%d", i ? ", " : "", g_dev_of[i]);
        fprintf(stderr, "], \"per_device\": [");
        int first = 1;
        for (int d = 0; d < (ndev > 0 ? ndev : 1); d++) {
            int nw = 0; double last = 0;
            char bus[64] = "";
            for (int i = 0; i < T; i++) if (g_dev_of[i] == d) { nw++; if (g_done[i] > last) last = g_done[i]; }
            if (!nw) continue;
            (void)mlvfs_amd_device_pci_bus_id(d, bus, (int)sizeof bus);
            fprintf(stderr, "%s{\"device\": %d, \"pci_bus_id\": \"%s\", \"workers\": %d, \"fps\": %.1f}", first ? "" : ", ", d, bus, nw, nw * g_frames / (last - g_t0));
            first = 0;
        }
        fprintf(stderr, "]}\n");
    }
    if (getenv("MLVFS_AMD_DROPIN_PROFILE")) {              /* where a bracketed frame's wall time goes (summed over the threads, both passes) */
        double pr[8];
        mlvfs_amd_dropin_profile(pr);
        const double n = pr[7] > 0 ? pr[7] : 1;
        fprintf(stderr, "{\"profile_us_per_frame\": {\"dng_get_image_data\": %.1f, \"of_it_upload_call\": %.1f, \"of_it_wait_for_upload\": %.1f, "
                "\"recorded_stage_calls\": %.1f, \"frame_end\": %.1f, \"of_it_launch_calls\": %.1f, \"of_it_download_and_wait\": %.1f, \"frames\": %.0f}}\n",
                1e3 * pr[0] / n, 1e3 * pr[1] / n, 1e3 * pr[2] / n, 1e3 * pr[3] / n, 1e3 * pr[4] / n, 1e3 * pr[5] / n, 1e3 * pr[6] / n, pr[7]);
    }
    stripes_free_corrections();
    free_focus_pixel_maps();
    return 0;
}
