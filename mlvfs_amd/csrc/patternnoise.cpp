// patternnoise.cpp -- drop-in fix_pattern_noise (mlvfs/patternnoise.c:357-380).
// GPU implementation pending; until then the symbol reports that
// loudly instead of silently returning unprocessed data as if it were processed.
#include "clip.h"

extern "C" void fix_pattern_noise(int16_t *raw, int w, int h, int white, int debug_flags)
{
    (void)raw; (void)w; (void)h; (void)white; (void)debug_flags;
    mlv::set_error("fix_pattern_noise: not implemented in this build (frame left untouched)");
}
