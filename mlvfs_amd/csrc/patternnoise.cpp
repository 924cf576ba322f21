// patternnoise.cpp -- drop-in fix_pattern_noise (mlvfs/patternnoise.c:357-380) on
// top of the kernels of k_pnoise.hip: stage the int16 frame, run the column pass and
// the row pass (transposed), copy back.  Only debug_flags == 0 (what MLVFS passes,
// main.c:948) is supported; the reference's debug views are not reproduced.
#include "clip.h"

namespace mlv {
size_t pattern_noise_scratch_bytes(int w, int h);
int launch_pattern_noise(void *d_raw, int w, int h, int white, void *d_scratch, hipStream_t stream);
}

using namespace mlv;

extern "C" void fix_pattern_noise(int16_t *raw, int w, int h, int white, int debug_flags)
{
    LibcRandGuard rand_guard;                      // HIP code may run: keep the caller's rand() stream out of its reach
    printf("Fixing pattern noise...\n");                                   // patternnoise.c:359
    if (debug_flags != 0) { set_error("fix_pattern_noise: debug_flags %d not supported (frame left untouched)", debug_flags); return; }
    if (w < 2 || h < 2 || (w & 1) || (h & 1)) { set_error("fix_pattern_noise: %dx%d frame not supported", w, h); return; }
    ThreadCtx *c = thread_ctx();
    if (!c) return;
    if (drop_resident(c, raw)) return;             // this call rewrites the host frame: no resident copy of it (dropin.cpp)
    const size_t bytes = (size_t)w * h * 2;
    if (c->ensure(bytes, pattern_noise_scratch_bytes(w, h))) return;
    if (hipMemcpyAsync(c->d_a, raw, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) { set_error("fix_pattern_noise: upload failed"); return; }
    if (launch_pattern_noise(c->d_a, w, h, white, c->d_b, c->stream)) return;
    if (hipMemcpyAsync(raw, c->d_a, bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess)
        set_error("fix_pattern_noise: download failed");
}
