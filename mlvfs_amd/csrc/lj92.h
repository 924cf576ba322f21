// lj92.h -- shared between csrc/lj92.cpp (host: stream parsing, tables, buffers) and csrc/k_lj92.hip (kernels)
#pragma once
#include "common.h"

namespace mlv {

constexpr uint32_t LJ_TAIL = 128;          // zero bytes kept behind the unstuffed data (look-ahead of the last symbols)
constexpr int LJ_ERR_CODE = 1;             // a bit pattern that is no Huffman code / a pixel decoded from bits behind the data
constexpr int LJ_ERR_SHORT = 2;            // fewer symbols than pixels
constexpr int LJ_L1_BITS = 11;             // first-level Huffman look-up: 2048 entries
constexpr int LJ_LUT_MAX = 3584;           // both levels together (7 KiB of LDS): 48 second-level tables of 32 entries

// one frame's view of the work buffers (all pointers are device memory)
struct LjFrame {
    const uint8_t *raw;        // entropy-coded segment as stored (0xFF 0x00 stuffing inside)
    uint32_t raw_len;
    uint8_t *ust;              // unstuffed bytes, 16-byte aligned, followed by LJ_TAIL zero bytes
    uint32_t *ust_len;
    uint32_t *blk_drop;        // stuffed zeros per 4 KiB block, then their exclusive prefix sums
    const uint16_t *lut;       // two-level table, see lut_entry() in k_lj92.hip
    int huffbits, lut_entries;
    uint16_t *cmap;            // [nwg * 256][32] chunk maps: exit offset | symbols << 5
    uint2 *wmap, *gmap;        // [nwg][32], [ngrp][32]: {exit offset, symbols}
    uint2 *smap;               // [nwg][16][32]: the same for the 16 segments of 16 chunks inside each workgroup
    uint2 *wstart, *gstart;    // [nwg], [ngrp]: {true entry offset, index of the first symbol}
    int32_t *diff;             // [W * H] differences, then the per-row recurrence values
    int32_t *colsum;           // [16][W] sums of e over segments of rows
    uint16_t *out;             // xres x yres pixels, untiled
    int W, H, bits, pred;      // the JPEG's own dimensions, sample precision, predictor (0..7)
    int xres, yres;            // the video frame the decoded values are re-read as (main.c:646-667)
    uint32_t nwg, ngrp;        // 8 KiB workgroup windows over raw_len bytes, groups of 32 of them
    int *err;
};

constexpr int LJ_WAVE_MAX_H = 8192;        // predictor 7 keeps two anti-diagonals (indexed by row) in LDS

// preds: bit p set when some frame of the batch uses predictor p (the rarely used ones have kernels of their own)
int lj92_launch(const LjFrame *d_frames, int nframes, uint32_t max_raw, uint32_t max_nwg, uint32_t max_ngrp, int max_w, int max_h,
                unsigned preds, hipStream_t s);

}  // namespace mlv
