// patternnoise.cpp -- drop-in fix_pattern_noise (mlvfs/patternnoise.c:357-380) on
// top of the kernels of k_pnoise.hip: the int16 frame on the device -- the copy the unpack left there, or an upload --,
// the column pass and the row pass (transposed) in place, back to the host unless a frame bracket is open (dropin.cpp).  debug_flags != 0 (MLVFS
// passes 0, main.c:948): one direction only and the reference's debug views (patternnoise.c:215-240, 363-379), reproduced as well.
#include "clip.h"

namespace mlv {
size_t pattern_noise_scratch_bytes(int w, int h);
int launch_pattern_noise(void *d_raw, int w, int h, int white, void *d_scratch, hipStream_t stream, int flags);
}

using namespace mlv;

extern "C" void fix_pattern_noise(int16_t *raw, int w, int h, int white, int debug_flags)
{
    LibcRandGuard rand_guard;                      // HIP code may run: keep the caller's rand() stream out of its reach
    printf("Fixing pattern noise...\n");                                   // patternnoise.c:359
    if (w < 2 || h < 2 || (w & 1) || (h & 1)) { set_error("fix_pattern_noise: %dx%d frame not supported", w, h); return; }
    ThreadCtx *c = thread_ctx();
    if (!c) return;
    const size_t bytes = (size_t)w * h * 2;
    void *d_frame = nullptr;
    int which = 0;
    bool was_dirty = false;
    if (inplace_stage_begin(c, STAGE_PNOISE, raw, bytes, &d_frame, &which, &was_dirty)) return;       // the unpack's device copy, or an upload
    const bool done = c->ensure(0, pattern_noise_scratch_bytes(w, h)) == MLVFS_AMD_OK &&
                      launch_pattern_noise(d_frame, w, h, white, c->d_b, c->stream, debug_flags) == MLVFS_AMD_OK;
    inplace_stage_end(c, STAGE_PNOISE, raw, bytes, which, was_dirty, done, true);                              // downloads unless a frame bracket is open
}
