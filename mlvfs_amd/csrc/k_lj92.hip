// k_lj92.hip -- lossless-JPEG ("LJ92") frame payloads decoded on the GPU.
//
// Replaces, per frame of a compressed clip (mlvfs/main.c:617-681):
//   lj92_open / lj92_decode    mlvfs/lj92.c:344-406 (nextdiff), 408-593 (parsePred6, parseScan), 650-702
//   the untiling loop          mlvfs/main.c:646-667
// Header parsing and the Huffman look-up table stay on the host (csrc/lj92.cpp).
//
// The reference decodes one symbol after the other: a Huffman code, then `ssss` raw bits, then a prediction from the pixel
// just reconstructed.  Both dependencies are taken apart here, exactly:
//
// 1. WHERE THE SYMBOLS START.  The unstuffed bit stream is cut into chunks of 256 bits.  A symbol is at most 32 bits long,
//    so whatever came before, the first symbol that starts inside a chunk starts at one of its first 32 bits.  For each of
//    those 32 entry offsets a thread decodes to the end of its chunk and records {offset at which the next chunk is
//    entered, symbols decoded}: a 32-entry map per chunk (k_lj_chunk_maps).  Maps compose (entry -> exit of one chunk is
//    the entry of the next), so 32 lanes walking 16 chunk maps give the map of a segment, 32 lanes walking the 16 segment
//    maps the map of a workgroup's 8 KiB, 32 lanes walking 32 of those the map of a group (k_lj_group_maps), and a single
//    walk over the few dozen groups from offset 0 (the scan starts on a symbol) gives every group's true entry offset and
//    first symbol index (k_lj_top); three more walks hand these down to workgroups, segments and chunks (k_lj_group_starts,
//    k_lj_decode).  Every chunk then decodes only its true symbols
//    and stores each difference at its index.
// 2. THE PREDICTOR.  Predictor 6, the one MLV files use (lj92.c:951), is  x = above + ((left - above_left) >> 1) + d.
//    With e = x - above it reads  e[c] = (e[c-1] >> 1) + d[c]  along a row, and nested floor divisions collapse:
//        e[c0 + i] = (e[c0] + sum_{j=1..i} d[c0+j] * 2^j) >> i        (exact in 64 bits for i <= 32)
//    so a row is 32-column blocks with one carried value between them (k_lj_rows), rows are independent of each other,
//    and x is the running sum of e down each column (k_lj_columns), which also writes the pixel to its untiled position.
//    Predictor 1 (left) is a prefix sum along each row plus one down the first column.  The first row is always a prefix
//    sum (lj92.c:533-537), the first pixel is predicted by 2^(bits-1).  The other predictors of lj92.c:546-563 reduce to
//    the same pieces: 2 and 4 are column sums (4 after a prefix sum of each row), 0 needs the first column only, 5 is the
//    halving recurrence DOWN the columns followed by predictor 1's sums, 3 sums along diagonals, and 7 -- the one that is
//    not a composition of one-dimensional recurrences -- walks the anti-diagonals (k_lj_wavefront).
// No MFMA: bit-stream and integer work.
#include "lj92.h"

namespace mlv {

namespace {

constexpr int CHUNK_BITS = 256, CHUNK_BYTES = 32, ENTRIES = 32;
constexpr int WG_CHUNKS = 256;                          // chunks per workgroup (8 KiB of stream)
constexpr int GROUP_WGS = 32;                           // workgroup maps per group map
constexpr int UNSTUFF_BYTES = 16;                       // bytes per thread in the unstuff kernels

__device__ __forceinline__ uint32_t bswap(uint32_t v) { return __builtin_bswap32(v); }

// ---------------------------------------------------------------- byte unstuffing (0xFF 0x00 -> 0xFF)
// a byte is dropped iff it is 0x00 and the byte before it is 0xFF (lj92.c:358-370).  A thread's 16 bytes arrive as ONE 16-byte load
// (the raw segment is 256-byte aligned in the arena and has 16 bytes of room behind its end; byte loads made the two kernels below
// 33 + 75 us per four 3584x1320 frames, a sixth of the whole decode).
__device__ __forceinline__ int dropped_in(const uint8_t *raw, uint32_t len, uint32_t p0, uint32_t *keep_mask, uint4 *bytes)
{
    const uint4 v = *(const uint4 *)(raw + p0);
    *bytes = v;
    const uint32_t w[4] = { v.x, v.y, v.z, v.w };
    int n = 0;
    uint32_t mask = 0;
    uint32_t prev = p0 ? raw[p0 - 1] : 0;
#pragma unroll
    for (int i = 0; i < UNSTUFF_BYTES; i++) {
        const uint32_t b = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
        const bool in = p0 + i < len;
        const bool drop = in && b == 0 && prev == 0xFF;
        n += drop;
        mask |= ((in && !drop) ? 1u : 0u) << i;
        prev = b;
    }
    *keep_mask = mask;
    return n;
}

__global__ __launch_bounds__(256) void k_lj_unstuff_count(const LjFrame *frames)
{
    const LjFrame &f = frames[blockIdx.y];
    const uint32_t p0 = (blockIdx.x * 256u + threadIdx.x) * UNSTUFF_BYTES;
    if (blockIdx.x * 256u * UNSTUFF_BYTES >= f.raw_len) return;
    uint32_t mask;
    uint4 bytes;
    int n = p0 < f.raw_len ? dropped_in(f.raw, f.raw_len, p0, &mask, &bytes) : 0;
    for (int o = 32; o; o >>= 1) n += __shfl_down(n, o);
    __shared__ int wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) f.blk_drop[blockIdx.x] = (uint32_t)(wsum[0] + wsum[1] + wsum[2] + wsum[3]);
}

// exclusive scan of the per-block drop counts (a few hundred to a few thousand values per frame)
__global__ __launch_bounds__(256) void k_lj_unstuff_scan(const LjFrame *frames)
{
    const LjFrame &f = frames[blockIdx.x];
    const uint32_t nblk = (f.raw_len + 256u * UNSTUFF_BYTES - 1) / (256u * UNSTUFF_BYTES);
    __shared__ uint32_t part[256];
    const uint32_t per = (nblk + 255) / 256, b0 = threadIdx.x * per, b1 = min(nblk, b0 + per);
    uint32_t s = 0;
    for (uint32_t b = b0; b < b1; b++) s += f.blk_drop[b];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int i = 0; i < 256; i++) { const uint32_t v = part[i]; part[i] = run; run += v; }
        *f.ust_len = f.raw_len - run;
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (uint32_t b = b0; b < b1; b++) { const uint32_t v = f.blk_drop[b]; f.blk_drop[b] = run; run += v; }
}

// The kept bytes of a workgroup's 4 KiB go through LDS -- placed so that LDS dwords and the dwords of the destination line up (the
// destination starts wherever the drops before this block put it) -- and leave as whole dwords, lane after lane; only a block's
// first and last dword, which it may share with its neighbours, are written byte by byte.
__global__ __launch_bounds__(256) void k_lj_unstuff_scatter(const LjFrame *frames)
{
    const LjFrame &f = frames[blockIdx.y];
    constexpr uint32_t BLK = 256u * UNSTUFF_BYTES;
    if (blockIdx.x * BLK >= f.raw_len) {
        // the first idle workgroup clears the look-ahead tail behind the unstuffed data
        if ((blockIdx.x - 1) * BLK < f.raw_len || f.raw_len == 0) {
            const uint32_t n = *f.ust_len;
            for (uint32_t i = threadIdx.x; i < LJ_TAIL; i += 256) f.ust[n + i] = 0;
        }
        return;
    }
    __shared__ uint32_t stage32[BLK / 4 + 2];
    __shared__ uint32_t wsum[4];
    uint8_t *const stage = (uint8_t *)stage32;
    const uint32_t p0 = (blockIdx.x * 256u + threadIdx.x) * UNSTUFF_BYTES;
    uint32_t mask = 0;
    uint4 bytes = make_uint4(0, 0, 0, 0);
    const int n = p0 < f.raw_len ? dropped_in(f.raw, f.raw_len, p0, &mask, &bytes) : 0;
    uint32_t inc = (uint32_t)n;                          // inclusive scan of the 256 drop counts: within the wave, then over the four waves
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(inc, o); if (lane >= o) inc += up; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    for (int k = 0; k < wave; k++) inc += wsum[k];
    const uint32_t dst_base = blockIdx.x * BLK - f.blk_drop[blockIdx.x];       // where this block's first kept byte goes
    const uint32_t mis = dst_base & 3u;
    uint32_t at = mis + threadIdx.x * UNSTUFF_BYTES - (inc - (uint32_t)n);     // in the staging buffer
    const uint32_t w[4] = { bytes.x, bytes.y, bytes.z, bytes.w };
#pragma unroll
    for (int i = 0; i < UNSTUFF_BYTES; i++)
        if ((mask >> i) & 1u) stage[at++] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
    const uint32_t valid = min(BLK, f.raw_len - blockIdx.x * BLK);
    const uint32_t kept = valid - (wsum[0] + wsum[1] + wsum[2] + wsum[3]);     // bytes this block writes: stage[mis .. mis + kept)
    __syncthreads();
    uint32_t *const out32 = (uint32_t *)(f.ust + (dst_base - mis));
    const uint32_t ndw = (mis + kept + 3) / 4;
    for (uint32_t k = threadIdx.x; k < ndw; k += 256) {
        const uint32_t lo = 4 * k, hi = lo + 4;
        if (lo >= mis && hi <= mis + kept) out32[k] = stage32[k];
        else
            for (uint32_t q = max(lo, mis); q < min(hi, mis + kept); q++) f.ust[dst_base - mis + q] = stage[q];
    }
}

// ---------------------------------------------------------------- symbol walk
// big-endian words of a workgroup's 8 KiB (+ look-ahead) in LDS; the 32 stream bits starting at bit `p` of the workgroup.
// Thread t works on words 8t .. 8t+8: one spare word after every 32 spreads the threads of a wave over the banks (without
// it the 64 lanes of a read hit 4 banks).
__device__ __forceinline__ uint32_t pad_word(uint32_t i) { return i + (i >> 5); }
struct Window {
    const uint32_t *w;
    __device__ __forceinline__ uint32_t at(uint32_t p) const
    {
        const uint32_t i = p >> 5, s = p & 31;
        const uint64_t two = ((uint64_t)w[pad_word(i)] << 32) | w[pad_word(i + 1)];
        return (uint32_t)(two >> (32 - s));
    }
};

// Two-level Huffman look-up (built by lj92.cpp): the first LJ_L1_BITS bits of a code index the first level; codes longer
// than that share an entry with bit 15 set that points to a second-level table indexed by the remaining bits.  Entries are
// (ssss << 8) | code length, 0 = no such code.  The whole table is at most LJ_LUT_MAX entries and lives in LDS.
__device__ __forceinline__ uint32_t lut_entry(const uint16_t *lut, int huffbits, uint32_t bits)
{
    const int b1 = huffbits < LJ_L1_BITS ? huffbits : LJ_L1_BITS;
    uint32_t e = lut[bits >> (32 - b1)];
    if (e & 0x8000u) e = lut[(e & 0x7FFFu) + ((bits << b1) >> (32 - (huffbits - b1)))];
    return e;
}

// one symbol at bit p: returns its length in bits (code + ssss raw bits) and the decoded difference
__device__ __forceinline__ uint32_t symbol(const Window &win, const uint16_t *lut, int huffbits, uint32_t p, int *diff, bool *bad)
{
    const uint32_t bits = win.at(p);
    const uint32_t e = lut_entry(lut, huffbits, bits);
    const uint32_t used = e & 0xFFu, t = e >> 8;
    if (used == 0 || t > 16) { *bad = true; *diff = 0; return 1; }           // no such code: step on, the frame is reported corrupt
    int d = 0;
    if (t) {
        d = (int)((bits << used) >> (32 - t));
        if (d < (1 << (t - 1))) d += (int)(0xFFFFFFFFu << t) + 1;
    }
    *diff = d;
    return used + t;
}

constexpr int WG_WORDS = WG_CHUNKS * CHUNK_BYTES / 4;   // 2048

constexpr int CMAP_PITCH = ENTRIES + 2;                 // 17 words per row: the threads' rows start in different banks

// A workgroup's 8 KiB of stream (+ 16 bytes of look-ahead) and the Huffman table into LDS: 16-byte loads, all of a thread's loads issued
// before its first store.  (As loops of 4- and 2-byte loads -- eight and fourteen dependent passes of global-memory latency -- this
// prologue was most of the decode kernel's 58 k cycles per wave: profiles/r05/lj_steps.txt.)
// LENGTHS: the table's plain entries, (ssss << 8) | code length, become the symbol's whole length code + ssss (1 for "no such code", as
// symbol() steps) -- all that k_lj_chunk_maps wants of a symbol; entries that point to a second-level table stay as they are.
template <bool LENGTHS>
__device__ __forceinline__ void load_stream_and_table(const LjFrame &f, uint32_t wg, uint32_t *words, uint16_t *lut)
{
    static_assert(WG_WORDS + 4 == 4 * 513 && LJ_LUT_MAX == 8 * 448, "two passes of 256 threads each");
    const uint4 *src = (const uint4 *)(f.ust + (size_t)wg * WG_CHUNKS * CHUNK_BYTES);      // ust is 16-byte aligned, LJ_TAIL bytes of zeros behind it
    const uint4 *l4 = (const uint4 *)f.lut;                                                // LJ_LUT_MAX entries, 256-byte aligned
    const uint32_t t = threadIdx.x;
    const uint4 zero = make_uint4(0, 0, 0, 0);
    const uint4 w0 = src[t], w1 = src[t + 256], w2 = t == 0 ? src[512] : zero;
    uint4 t0 = l4[t], t1 = t < 192 ? l4[t + 256] : zero;
    if (LENGTHS) {
        auto len1 = [](uint32_t e) -> uint32_t {
            if (e & 0x8000u) return e;
            const uint32_t used = e & 0xFFu, ss = e >> 8;
            return (used == 0 || ss > 16) ? 1u : used + ss;
        };
        auto len2 = [&](uint32_t v) { return len1(v & 0xFFFFu) | (len1(v >> 16) << 16); };
        t0 = make_uint4(len2(t0.x), len2(t0.y), len2(t0.z), len2(t0.w));
        t1 = make_uint4(len2(t1.x), len2(t1.y), len2(t1.z), len2(t1.w));
    }
    auto put = [&](uint32_t q, const uint4 &v) {
        words[pad_word(4 * q)] = bswap(v.x); words[pad_word(4 * q + 1)] = bswap(v.y);
        words[pad_word(4 * q + 2)] = bswap(v.z); words[pad_word(4 * q + 3)] = bswap(v.w);
    };
    put(t, w0); put(t + 256, w1);
    if (t == 0) put(512, w2);
    ((uint4 *)lut)[t] = t0;
    if (t < 192) ((uint4 *)lut)[t + 256] = t1;
}

constexpr int SEGS = 16, SEG_CHUNKS = WG_CHUNKS / SEGS;     // the decode kernel walks 16 + 16 maps instead of 256

// chunk maps of one workgroup + their composition.  33 KiB of LDS -- four workgroups per CU: the maps take the place of the stream
// words and the Huffman table once the walk is over, the segment maps that of the ring (all of them side by side were 53 KiB, three
// workgroups per CU, in a kernel that waits on dependent LDS reads most of the time).
struct alignas(16) MapSmem {
    union {
        struct { uint32_t words[WG_WORDS + 4 + (WG_WORDS + 4) / 32 + 1]; alignas(16) uint16_t lut[LJ_LUT_MAX]; } in;
        uint16_t cmap[WG_CHUNKS][CMAP_PITCH];           // exit offset | symbols << 5
    };
    union {
        uint16_t ring[32][WG_CHUNKS];                   // {exit, symbols} of the 32 positions ahead, per thread
        uint2 segm[SEGS][ENTRIES];
    };
};
static_assert(sizeof(MapSmem) * 4 <= 160 * 1024, "four workgroups per CU");

__global__ __launch_bounds__(256) void k_lj_chunk_maps(const LjFrame *frames)
{
    const LjFrame &f = frames[blockIdx.y];
    const uint32_t wg = blockIdx.x;
    if (wg >= f.nwg) return;
    __shared__ MapSmem sm;
    load_stream_and_table<true>(f, wg, sm.in.words, sm.in.lut);
    __syncthreads();
    const uint32_t c0 = threadIdx.x * CHUNK_BITS;
    // From the last bit of the chunk backwards: a walk that starts at bit q continues at q + length(q), which is at most 32
    // bits ahead, so {exit offset, symbols} of the 32 positions ahead (a ring in LDS) are all that is needed.  256 symbol
    // look-ups per chunk instead of one walk per entry offset (~30 symbols each, 32 of them).
    // (a thread's column of the ring: lanes 0..31 of a wave take the low halves of 32 consecutive dwords, lanes 32..63 the high halves --
    // LDS serves a wave's 64 lanes in two halves of 32, and in thread order each half had two lanes per bank)
    const uint32_t rcol = (threadIdx.x & ~63u) + 2u * (threadIdx.x & 31u) + ((threadIdx.x >> 5) & 1u);
    for (int q = 0; q < 32; q++) sm.ring[q][rcol] = (uint16_t)q;                     // positions 256..287: already outside
    // The symbol lengths of different positions do not depend on each other: 16 look-ups are issued together (their LDS
    // latencies overlap), then the 16 dependent ring steps follow.  32 positions share two stream words, their bit offsets are
    // compile-time constants (one v_alignbit_b32 per window), and the table holds whole lengths (load_stream_and_table<true>):
    // 9 vector instructions per position where the general symbol() path took 22.
    const int b1 = f.huffbits < LJ_L1_BITS ? f.huffbits : LJ_L1_BITS, b2 = f.huffbits - b1;
    const uint16_t *lut = sm.in.lut;
    uint16_t *ring_t = &sm.ring[0][rcol];
    for (int qq = CHUNK_BITS - 32; qq >= 0; qq -= 32) {
        const uint32_t wi = (c0 + (uint32_t)qq) >> 5;
        const uint32_t w0 = sm.in.words[pad_word(wi)], w1 = sm.in.words[pad_word(wi + 1)];
#pragma unroll
        for (int half = 1; half >= 0; half--) {
            uint32_t bits[16], len[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int sft = 16 * half + i;                                           // bit offset in w0: a constant
                bits[i] = sft == 0 ? w0 : __builtin_amdgcn_alignbit(w0, w1, 32 - sft);
                len[i] = lut[bits[i] >> (32 - b1)];
            }
#pragma unroll
            for (int i = 0; i < 16; i++)
                if (len[i] & 0x8000u) len[i] = lut[(len[i] & 0x7FFFu) + ((bits[i] << b1) >> (32 - b2))];
#pragma unroll
            for (int i = 15; i >= 0; i--) {
                const int q = qq + 16 * half + i;
                const uint16_t v = (uint16_t)(ring_t[(((uint32_t)q + len[i]) & 31u) * WG_CHUNKS] + 32);   // one more symbol on that walk
                ring_t[(q & 31) * WG_CHUNKS] = v;       // (q + 32) & 31 == q & 31: read above before it is overwritten here
            }
        }
    }
    __syncthreads();                                     // every thread is done with the stream words and the table: the maps take their place
    // the ring now holds positions 0..31, the chunk's map: thread t's column becomes row t (conflict-free both ways)
    uint16_t mine[ENTRIES];
#pragma unroll
    for (int q = 0; q < ENTRIES; q++) mine[q] = sm.ring[q][rcol];
#pragma unroll
    for (int q = 0; q < ENTRIES; q++) sm.cmap[threadIdx.x][q] = mine[q];
    __syncthreads();                                     // (and the ring is free for the segment maps)
    uint32_t *dst = (uint32_t *)(f.cmap + (size_t)wg * WG_CHUNKS * ENTRIES);
    for (int i = threadIdx.x; i < WG_CHUNKS * ENTRIES / 2; i += blockDim.x)
        dst[i] = *(const uint32_t *)&sm.cmap[i / (ENTRIES / 2)][2 * (i % (ENTRIES / 2))];
    // maps of the 16 segments of 16 chunks (32 lanes each, two rounds), then 32 lanes compose those into the workgroup's map
    uint2 *smap = f.smap + (size_t)wg * SEGS * ENTRIES;
    for (int round = 0; round < 2; round++) {
        const int seg = round * 8 + (threadIdx.x >> 5);
        uint32_t e = threadIdx.x & 31, n = 0;
        for (int c = seg * SEG_CHUNKS; c < (seg + 1) * SEG_CHUNKS; c++) {
            const uint32_t m = sm.cmap[c][e];
            e = m & 31u;
            n += m >> 5;
        }
        sm.segm[seg][threadIdx.x & 31] = make_uint2(e, n);
        smap[seg * ENTRIES + (threadIdx.x & 31)] = make_uint2(e, n);
    }
    __syncthreads();
    if (threadIdx.x < ENTRIES) {
        uint32_t e = threadIdx.x, n = 0;
        for (int sgi = 0; sgi < SEGS; sgi++) {
            const uint2 m = sm.segm[sgi][e];
            e = m.x;
            n += m.y;
        }
        f.wmap[(size_t)wg * ENTRIES + threadIdx.x] = make_uint2(e, n);
    }
}

// map of GROUP_WGS consecutive workgroup maps.  The maps (8 KiB) are fetched into LDS in one go and walked there: walked in global
// memory, this kernel, k_lj_top and k_lj_group_starts were chains of 32, ~18 and 32 dependent loads of memory latency -- 12 + 11 + 20 us
// per sub-batch for a few kilobytes of work (profiles/r05/lj_steps.txt).
__device__ __forceinline__ void load_group_wmaps(const LjFrame &f, uint32_t g, uint2 (*wm)[ENTRIES])
{
    const uint32_t w0 = g * GROUP_WGS, n = min(f.nwg, w0 + GROUP_WGS) - w0;               // wmap is 256-byte aligned, a map 256 bytes
    const uint4 *src = (const uint4 *)(f.wmap + (size_t)w0 * ENTRIES);
    uint4 *dst = (uint4 *)&wm[0][0];
    for (uint32_t i = threadIdx.x; i < n * (ENTRIES / 2); i += blockDim.x) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void k_lj_group_maps(const LjFrame *frames)
{
    const LjFrame &f = frames[blockIdx.y];
    const uint32_t g = blockIdx.x;
    if (g >= f.ngrp) return;
    __shared__ __align__(16) uint2 wm[GROUP_WGS][ENTRIES];
    load_group_wmaps(f, g, wm);
    __syncthreads();
    if (threadIdx.x >= ENTRIES) return;
    uint32_t e = threadIdx.x, n = 0;
    const uint32_t cnt = min(f.nwg, (g + 1) * GROUP_WGS) - g * GROUP_WGS;
    for (uint32_t w = 0; w < cnt; w++) {
        const uint2 m = wm[w][e];
        e = m.x;
        n += m.y;
    }
    f.gmap[(size_t)g * ENTRIES + threadIdx.x] = make_uint2(e, n);
}

// The scan starts on a symbol: offset 0 of group 0.  Every group's workgroup walks the group maps before it (a few dozen steps in
// LDS) for its own entry offset and first symbol index, then its workgroups' maps for theirs.  The last group also knows the total.
constexpr int GMAP_TILE = 32;                           // group maps walked per pass (8 KiB of LDS)
__global__ __launch_bounds__(256) void k_lj_group_starts(const LjFrame *frames)
{
    const LjFrame &f = frames[blockIdx.y];
    const uint32_t g = blockIdx.x;
    if (g >= f.ngrp) return;
    __shared__ __align__(16) uint2 wm[GROUP_WGS][ENTRIES];
    __shared__ __align__(16) uint2 gm[GMAP_TILE][ENTRIES];
    __shared__ uint2 at;
    if (threadIdx.x == 0) at = make_uint2(0, 0);
    load_group_wmaps(f, g, wm);
    for (uint32_t g0 = 0; g0 < g; g0 += GMAP_TILE) {
        const uint32_t cnt = min(g - g0, (uint32_t)GMAP_TILE);
        __syncthreads();
        const uint4 *src = (const uint4 *)(f.gmap + (size_t)g0 * ENTRIES);
        for (uint32_t i = threadIdx.x; i < cnt * (ENTRIES / 2); i += blockDim.x) ((uint4 *)&gm[0][0])[i] = src[i];
        __syncthreads();
        if (threadIdx.x == 0) {
            uint2 s = at;
            for (uint32_t k = 0; k < cnt; k++) {
                const uint2 m = gm[k][s.x];
                s.x = m.x;
                s.y += m.y;
            }
            at = s;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint2 s = at;
        const uint32_t w0 = g * GROUP_WGS, cnt = min(f.nwg, w0 + GROUP_WGS) - w0;
        for (uint32_t w = 0; w < cnt; w++) {
            f.wstart[w0 + w] = s;
            const uint2 m = wm[w][s.x];
            s.x = m.x;
            s.y += m.y;
        }
        if (g == f.ngrp - 1 && s.y < (uint32_t)f.W * (uint32_t)f.H) atomicOr(f.err, LJ_ERR_SHORT);      // fewer symbols than pixels
    }
}

// every chunk decodes its true symbols; the workgroup's differences -- one contiguous range of pixel indices -- are collected in LDS
// and leave as whole lines.  (Stored by the decoding threads themselves, 16 bytes at a time, each of a wave's 64 lanes wrote into a
// line of its own: WRITE_SIZE said 357 MB per four 3584x1320 frames for 76 MB of differences, profiles/r05/lj_pmc_before.txt.)
constexpr int DEC_STAGE = 9184;                         // differences a workgroup can hold (8 KiB of stream at >= 7.14 bits per symbol); the rest goes out directly
struct alignas(16) DecSmem {
    uint32_t words[WG_WORDS + 4 + (WG_WORDS + 4) / 32 + 1];
    alignas(16) uint16_t lut[LJ_LUT_MAX];
    uint32_t cstart[WG_CHUNKS][2];                      // true entry offset, index of the first symbol
    uint2 segstart[SEGS];
    uint32_t end_idx;
    alignas(16) union {
        struct { uint16_t cmap[WG_CHUNKS][CMAP_PITCH]; alignas(16) uint2 segm[SEGS][ENTRIES]; } m;      // until the chunks' starts are known
        int stage[DEC_STAGE];
    };
};
static_assert(sizeof(DecSmem) * 3 <= 160 * 1024, "three workgroups per CU");

__global__ __launch_bounds__(256) void k_lj_decode(const LjFrame *frames)
{
    const LjFrame &f = frames[blockIdx.y];
    const uint32_t wg = blockIdx.x;
    if (wg >= f.nwg) return;
    const uint32_t npx = (uint32_t)f.W * (uint32_t)f.H;
    const uint2 start = f.wstart[wg];
    if (start.y >= npx) return;                          // everything behind the last pixel is padding / the EOI marker
    __shared__ DecSmem sm;
    {
        // the workgroup's chunk maps (16 KiB) and segment maps (4 KiB): 16-byte loads issued together, like the stream's
        static_assert(WG_CHUNKS * ENTRIES * 2 == 1024 * 16 && SEGS * ENTRIES * 8 == 256 * 16, "four passes + one");
        const uint4 *cm = (const uint4 *)(f.cmap + (size_t)wg * WG_CHUNKS * ENTRIES);
        const uint4 *sg = (const uint4 *)(f.smap + (size_t)wg * SEGS * ENTRIES);
        const uint32_t t = threadIdx.x;
        const uint4 c0 = cm[t], c1 = cm[t + 256], c2 = cm[t + 512], c3 = cm[t + 768], s0 = sg[t];
        load_stream_and_table<false>(f, wg, sm.words, sm.lut);
        auto put = [&](uint32_t u, const uint4 &v) {        // piece u: eight entries of chunk u / 4 (rows of 34: dword stores)
            uint32_t *row = (uint32_t *)&sm.m.cmap[u >> 2][0] + 4 * (u & 3);
            row[0] = v.x; row[1] = v.y; row[2] = v.z; row[3] = v.w;
        };
        put(t, c0); put(t + 256, c1); put(t + 512, c2); put(t + 768, c3);
        ((uint4 *)&sm.m.segm[0][0])[t] = s0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {                              // 16 segment maps from the workgroup's true start ...
        uint32_t e = start.x, n = start.y;
        for (int sgi = 0; sgi < SEGS; sgi++) {
            sm.segstart[sgi] = make_uint2(e, n);
            const uint2 m = sm.m.segm[sgi][e];
            e = m.x;
            n += m.y;
        }
    }
    __syncthreads();
    if (threadIdx.x < SEGS) {                            // ... then 16 lanes walk the 16 chunk maps of their segment
        uint32_t e = sm.segstart[threadIdx.x].x, n = sm.segstart[threadIdx.x].y;
        for (int c = threadIdx.x * SEG_CHUNKS; c < (threadIdx.x + 1) * SEG_CHUNKS; c++) {
            sm.cstart[c][0] = e;
            sm.cstart[c][1] = n;
            const uint32_t m = sm.m.cmap[c][e];
            e = m & 31u;
            n += m >> 5;
        }
    }
    __syncthreads();                                     // (the maps are done with: the stage takes their place)
    const uint32_t c0 = threadIdx.x * CHUNK_BITS;
    uint32_t p = c0 + sm.cstart[threadIdx.x][0], idx = sm.cstart[threadIdx.x][1];
    const uint64_t end_bit = (uint64_t)*f.ust_len * 8, wg_bit0 = (uint64_t)wg * WG_CHUNKS * CHUNK_BITS;
    const uint32_t base = start.y;
    bool bad_any = false;
    // The thread's stream runs through a 64-bit register window (the next symbol's bits on top) that is refilled a word at a time from a
    // word fetched one refill ahead: a symbol's dependent chain is the table look-up and a shift, not two reads of the stream first
    // (profiles/r05/lj_pmc_before.txt: two thirds of the kernel's wave cycles were waits, a thousand cycles per symbol).
    uint32_t wi = p >> 5;
    uint64_t buf = (((uint64_t)sm.words[pad_word(wi)] << 32) | sm.words[pad_word(wi + 1)]) << (p & 31);
    int avail = 64 - (int)(p & 31);
    wi += 2;
    uint32_t nextw = sm.words[pad_word(min(wi, (uint32_t)(WG_WORDS + 3)))];
    const int huffbits = f.huffbits;
    while (p < c0 + CHUNK_BITS && idx < npx) {
        const uint32_t bits = (uint32_t)(buf >> 32);
        const uint32_t e = lut_entry(sm.lut, huffbits, bits);
        const uint32_t used = e & 0xFFu, t = e >> 8;
        bool bad = false;
        uint32_t len;
        int d = 0;
        if (used == 0 || t > 16) { bad = true; len = 1; }                   // no such code: step on, the frame is reported corrupt
        else {
            if (t) {
                d = (int)((bits << used) >> (32 - t));
                if (d < (1 << (t - 1))) d += (int)(0xFFFFFFFFu << t) + 1;
            }
            len = used + t;
        }
        if (wg_bit0 + p + len > end_bit) bad = true;      // a pixel decoded from bits behind the end of the data
        bad_any |= bad;
        const uint32_t at = idx - base;
        if (at < (uint32_t)DEC_STAGE) sm.stage[at] = d;
        else f.diff[idx] = d;
        idx++;
        p += len;
        buf <<= len;
        avail -= (int)len;
        if (avail < 32) {                                 // (len <= 32: one word per refill is enough)
            buf |= (uint64_t)nextw << (32 - avail);
            avail += 32;
            wi++;
            nextw = sm.words[pad_word(min(wi, (uint32_t)(WG_WORDS + 3)))];
        }
    }
    if (threadIdx.x == WG_CHUNKS - 1) sm.end_idx = min(idx, npx);      // (the chunks' ranges follow each other: the last one's end is the workgroup's)
    __syncthreads();
    const uint32_t n = min(sm.end_idx - base, (uint32_t)DEC_STAGE);
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) f.diff[base + i] = sm.stage[i];
    if (bad_any) atomicOr(f.err, LJ_ERR_CODE);
}

// ---------------------------------------------------------------- prediction
constexpr int ROW_LDS = 8192;
struct RowView {
    int *p;
    bool padded;
    __device__ __forceinline__ int &operator[](int i) const { return p[padded ? i + (i >> 5) : i]; }
};

// LDS of k_lj_rows for rows of up to max_w values: carries (one per block of 32 columns, + 2), the staged row when it fits
static size_t row_lds_bytes(int max_w)
{
    const size_t carries = (size_t)((max_w + 31) / 32 + 2) * sizeof(long long);
    return carries + (max_w <= ROW_LDS ? (size_t)(max_w + max_w / 32 + 2) * sizeof(int) : 0);
}

// one workgroup per row.  HALVING rows (predictor 6, r >= 1): e[c] = (e[c-1] >> 1) + d[c] in blocks of 32 columns.
// SCAN rows (row 0 always, every row of predictor 1): inclusive prefix sum; row 0 also carries the base 2^(bits-1).
// The row is rewritten in place (diff -> e).
__global__ __launch_bounds__(256) void k_lj_rows(const LjFrame *frames, int max_w)
{
    const LjFrame &f = frames[blockIdx.y];
    const int r = blockIdx.x;
    if (r >= f.H) return;
    // what a row needs below the first one: 1, 5: prefix sums behind the first column; 4: prefix sums of the whole row;
    // 6: the halving recurrence; 0, 2, 3, 7: nothing (their columns / diagonals are summed elsewhere)
    const bool scan_row = r == 0 || f.pred == 1 || f.pred == 5 || f.pred == 4;
    if (!scan_row && f.pred != 6) return;
    int *grow = f.diff + (size_t)r * f.W;
    const int W = f.W;
    // Dynamic LDS, sized by the launcher from the widest frame of the batch (row_lds_bytes): the carries of the row's blocks of 32
    // columns, then the staged row.  (Sized statically for the widest possible row it was 50 KiB -- three rows in flight per CU in a
    // kernel that is all latency: a row's load, 112 dependent carry steps, its store.)
    extern __shared__ long long row_lds[];
    long long *const carry = row_lds;                   // (W + 31) / 32 + 1 entries
    // rows up to ROW_LDS values are staged in LDS (coalesced in, coalesced out; one spare word per 32 keeps the threads,
    // which each work on 32 consecutive values, in different banks); longer rows are worked on in place
    int *const stage = (int *)(row_lds + ((max_w + 31) / 32 + 2));
    const bool staged = W <= ROW_LDS;
    if (staged) {
        for (int i = threadIdx.x; i < W; i += blockDim.x) stage[i + (i >> 5)] = grow[i];
        __syncthreads();
    }
    RowView row{ staged ? stage : grow, staged };
    if (scan_row) {
        // predictors 1 and 5, r >= 1: e[0] = d[0] stays, e[c] = d[1] + .. + d[c];  predictor 4: e[c] = d[0] + .. + d[c];
        // row 0: x[c] = base + d[0] + .. + d[c]
        const int first = (r == 0 || f.pred == 4) ? 0 : 1;
        const int nblk = (W - first + 31) / 32;
        for (int b = threadIdx.x; b < nblk; b += blockDim.x) {
            long long s = 0;
            const int c0 = first + b * 32;
            for (int i = 0; i < 32 && c0 + i < W; i++) s += row[c0 + i];
            carry[b + 1] = s;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            long long run = r == 0 ? (1ll << (f.bits - 1)) : 0;
            for (int b = 0; b < nblk; b++) { const long long v = carry[b + 1]; carry[b] = run; run += v; }
        }
        __syncthreads();
        for (int b = threadIdx.x; b < nblk; b += blockDim.x) {
            long long s = carry[b];
            const int c0 = first + b * 32;
            for (int i = 0; i < 32 && c0 + i < W; i++) { s += row[c0 + i]; row[c0 + i] = (int)s; }
        }
        if (staged) {
            __syncthreads();
            for (int i = threadIdx.x; i < W; i += blockDim.x) grow[i] = stage[i + (i >> 5)];
        }
        return;
    }
    // A[b] = sum_{i=1..n} d[c0+i] << i over block b = columns c0+1 .. c0+n (c0 = 32 b)
    const int nblk = (W - 1 + 31) / 32;
    for (int b = threadIdx.x; b < nblk; b += blockDim.x) {
        long long a = 0;
        const int c0 = b * 32;
        for (int i = 1; i <= 32 && c0 + i < W; i++) a += (long long)row[c0 + i] << i;
        carry[b + 1] = a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long e = row[0];                            // e[0] = d[0]: the first column is predicted by the pixel above
        for (int b = 0; b < nblk; b++) {
            const long long a = carry[b + 1];
            carry[b] = e;                                // e at column 32 b
            const int n = min(32, W - 1 - b * 32);
            e = (e + a) >> n;
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nblk; b += blockDim.x) {
        const long long e0 = carry[b];
        long long s = 0;
        const int c0 = b * 32;
        for (int i = 1; i <= 32 && c0 + i < W; i++) {
            s += (long long)row[c0 + i] << i;
            row[c0 + i] = (int)((e0 + s) >> i);
        }
    }
    if (staged) {
        __syncthreads();
        for (int i = threadIdx.x; i < W; i += blockDim.x) grow[i] = stage[i + (i >> 5)];
    }
}

// Column sums in SEGMENTS of rows so that a single frame still fills the chip: k_lj_column_sums adds up each segment of
// each column, k_lj_columns starts every segment from the sum of the segments above it and writes the pixels (as 16 bits)
// to their untiled positions.  Predictor 6 sums e down every column; predictor 1 only down the first one.
constexpr int COL_SEGS = 16;
// predictors whose rows hang off the first column (0: nothing else is predicted; 1, 5: prefix sums along the row), as opposed
// to 2, 4, 6, whose values are sums down every column
__device__ __forceinline__ bool first_column_only(int pred) { return pred == 0 || pred == 1 || pred == 5 || pred == 3; }

// main.c:646-667: the decoded values, read as yres rows of xres, hold the even rows / columns first
__device__ __forceinline__ void emit_untiled(const LjFrame &f, int r, int c, int px)
{
    const uint32_t xres = (uint32_t)f.xres, yres = (uint32_t)f.yres;
    const uint32_t i = (uint32_t)r * (uint32_t)f.W + (uint32_t)c;
    if (xres == 0) { f.out[i] = (uint16_t)px; return; }                  // the decoder's own order (the drop-in lj92_decode: main.c untiles itself)
    const uint32_t sy = i / xres, sx = i - sy * xres;
    const uint32_t dy = 2 * sy < yres ? 2 * sy : 2 * sy - yres + 1, dx = 2 * sx < xres ? 2 * sx : 2 * sx - xres + 1;
    f.out[(size_t)dy * xres + dx] = (uint16_t)px;
}

// predictor 5: x = left + ((above - above_left) >> 1) + d.  With h = x - left: h[r][c] = (h[r-1][c] >> 1) + d[r][c] down every
// column c >= 1 (row 0: h = d), after which the rows are prefix sums of h like predictor 1.  One thread per column.
__global__ __launch_bounds__(256) void k_lj_vhalve(const LjFrame *frames)
{
    const LjFrame &f = frames[blockIdx.y];
    const int c = 1 + blockIdx.x * 256 + threadIdx.x;
    if (f.pred != 5 || c >= f.W) return;
    int h = f.diff[c];
    for (int r = 1; r < f.H; r++) {
        h = (h >> 1) + f.diff[(size_t)r * f.W + c];
        f.diff[(size_t)r * f.W + c] = h;
    }
}

// predictor 3: x = above_left + d: sums along the diagonals, which start in row 0 (already pixel values) or in column 0
// (pixel (0,0) plus the differences down to that row).  One thread per diagonal.
__global__ __launch_bounds__(256) void k_lj_diagonals(const LjFrame *frames)
{
    const LjFrame &f = frames[blockIdx.y];
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (f.pred != 3 || t >= f.W + f.H - 1) return;
    int r = t < f.W ? 0 : t - f.W + 1, c = t < f.W ? t : 0;
    int x = f.diff[c];                                    // row 0 holds pixel values (k_lj_rows)
    if (r > 0) {
        x = f.diff[0];
        for (int k = 1; k <= r; k++) x += f.diff[(size_t)k * f.W];
    }
    emit_untiled(f, r, c, x);
    for (r++, c++; r < f.H && c < f.W; r++, c++) {
        x += f.diff[(size_t)r * f.W + c];
        emit_untiled(f, r, c, x);
    }
}

// predictor 7: x = ((left + above) >> 1) + d depends on two neighbours non-linearly: anti-diagonals in order, one workgroup
// per frame, the previous anti-diagonal in LDS (indexed by row).  Slow and only there for completeness; H <= LJ_WAVE_MAX_H.
__global__ __launch_bounds__(1024) void k_lj_wavefront(const LjFrame *frames)
{
    const LjFrame &f = frames[blockIdx.x];
    if (f.pred != 7) return;
    __shared__ int diag[2][LJ_WAVE_MAX_H];
    const int W = f.W, H = f.H;
    for (int c = threadIdx.x; c < W; c += blockDim.x) emit_untiled(f, 0, c, f.diff[c]);       // row 0: pixel values already
    if (threadIdx.x == 0) diag[0][0] = f.diff[0];
    __syncthreads();
    for (int k = 1; k <= W + H - 2; k++) {
        const int *prev = diag[(k - 1) & 1];
        int *cur = diag[k & 1];
        const int r_lo = max(0, k - W + 1), r_hi = min(H - 1, k);
        for (int r = r_lo + threadIdx.x; r <= r_hi; r += blockDim.x) {
            const int c = k - r;
            int x;
            if (r == 0) x = f.diff[c];                                                       // first row
            else {
                const int d = f.diff[(size_t)r * W + c];
                x = c == 0 ? prev[r - 1] + d : ((prev[r] + prev[r - 1]) >> 1) + d;            // left is on the previous anti-diagonal in row r, above in row r-1
                emit_untiled(f, r, c, x);
            }
            cur[r] = x;
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void seg_rows(const LjFrame &f, int seg, int *r0, int *r1)
{
    const int per = (f.H - 1 + COL_SEGS - 1) / COL_SEGS;          // rows 1 .. H-1 (row 0 holds pixel values already)
    *r0 = 1 + seg * per;
    *r1 = min(f.H, *r0 + per);
}

__global__ __launch_bounds__(256) void k_lj_column_sums(const LjFrame *frames)
{
    const LjFrame &f = frames[blockIdx.y];
    const int seg = blockIdx.x % COL_SEGS, c = (blockIdx.x / COL_SEGS) * 256 + threadIdx.x;
    if (c >= f.W || (first_column_only(f.pred) && c != 0)) return;
    int r0, r1, s = 0;
    seg_rows(f, seg, &r0, &r1);
    for (int r = r0; r < r1; r++) s += f.diff[(size_t)r * f.W + c];
    f.colsum[(size_t)seg * f.W + c] = s;
}

__global__ __launch_bounds__(256) void k_lj_columns(const LjFrame *frames)
{
    const LjFrame &f = frames[blockIdx.y];
    const int seg = blockIdx.x % COL_SEGS, c = (blockIdx.x / COL_SEGS) * 256 + threadIdx.x;
    if (c >= f.W) return;
    const uint32_t xres = (uint32_t)f.xres, yres = (uint32_t)f.yres, W = (uint32_t)f.W;
    int r0, r1;
    seg_rows(f, seg, &r0, &r1);
    if (seg == 0) r0 = 0;                                   // the first segment also emits row 0
    if (f.pred == 3 || f.pred == 7) return;                 // written by k_lj_diagonals / k_lj_wavefront
    const bool fco = first_column_only(f.pred);
    // value carried into this segment: row 0's pixel plus the sums of the segments above
    const int cc = fco ? 0 : c;
    int x = f.diff[cc];
    for (int k = 0; k < seg; k++) x += f.colsum[(size_t)k * W + cc];
    // main.c:646-667 reads the decoded values as yres rows of xres, whatever the JPEG's own dimensions are: position of
    // element (r, c) in that reading, advanced row by row without divisions
    // (xres == 0: the values stay in the decoder's own order -- the drop-in lj92_decode --, handled as one row per JPEG row)
    const bool raw_order = xres == 0;
    const uint32_t xr = raw_order ? W : xres;
    const uint32_t i0 = (uint32_t)r0 * W + (uint32_t)c, qW = W / xr, mW = W - qW * xr;
    uint32_t sy = i0 / xr, sx = i0 - sy * xr;
    for (int r = r0; r < r1; r++) {
        int px;
        if (r == 0) px = f.diff[c];
        else {
            const int e = f.diff[(size_t)r * W + c];
            if (fco) {                                          // the first column carries; 1, 5: the row's prefix sum rides on it,
                x += c ? f.diff[(size_t)r * W] : e;             // 0: the difference is the pixel
                px = c ? (f.pred == 0 ? e : x + e) : x;
            } else { x += e; px = x; }                          // 2, 4, 6: sums down every column
        }
        const uint32_t dy = raw_order ? sy : (2 * sy < yres ? 2 * sy : 2 * sy - yres + 1), dx = raw_order ? sx : (2 * sx < xres ? 2 * sx : 2 * sx - xres + 1);
        f.out[(size_t)dy * xr + dx] = (uint16_t)px;
        sy += qW; sx += mW;
        if (sx >= xr) { sx -= xr; sy++; }
    }
}

}  // namespace

int lj92_launch(const LjFrame *d_frames, int nframes, uint32_t max_raw, uint32_t max_nwg, uint32_t max_ngrp, int max_w, int max_h,
                unsigned preds, hipStream_t s)
{
    if (nframes <= 0) return MLVFS_AMD_OK;
    const uint32_t ublk = (max_raw + 256u * UNSTUFF_BYTES - 1) / (256u * UNSTUFF_BYTES);
    hipLaunchKernelGGL(k_lj_unstuff_count, dim3(ublk, nframes), dim3(256), 0, s, d_frames);
    hipLaunchKernelGGL(k_lj_unstuff_scan, dim3(nframes), dim3(256), 0, s, d_frames);
    hipLaunchKernelGGL(k_lj_unstuff_scatter, dim3(ublk + 1, nframes), dim3(256), 0, s, d_frames);
    hipLaunchKernelGGL(k_lj_chunk_maps, dim3(max_nwg, nframes), dim3(256), 0, s, d_frames);
    hipLaunchKernelGGL(k_lj_group_maps, dim3(max_ngrp, nframes), dim3(256), 0, s, d_frames);
    hipLaunchKernelGGL(k_lj_group_starts, dim3(max_ngrp, nframes), dim3(256), 0, s, d_frames);
    hipLaunchKernelGGL(k_lj_decode, dim3(max_nwg, nframes), dim3(256), 0, s, d_frames);
    if (preds & (1u << 5)) hipLaunchKernelGGL(k_lj_vhalve, dim3((max_w + 255) / 256, nframes), dim3(256), 0, s, d_frames);
    hipLaunchKernelGGL(k_lj_rows, dim3(max_h, nframes), dim3(256), row_lds_bytes(max_w), s, d_frames, max_w);
    if (preds & (1u << 3)) hipLaunchKernelGGL(k_lj_diagonals, dim3((max_w + max_h + 255) / 256, nframes), dim3(256), 0, s, d_frames);
    if (preds & (1u << 7)) hipLaunchKernelGGL(k_lj_wavefront, dim3(nframes), dim3(1024), 0, s, d_frames);
    hipLaunchKernelGGL(k_lj_column_sums, dim3((max_w + 255) / 256 * COL_SEGS, nframes), dim3(256), 0, s, d_frames);
    hipLaunchKernelGGL(k_lj_columns, dim3((max_w + 255) / 256 * COL_SEGS, nframes), dim3(256), 0, s, d_frames);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

}  // namespace mlv
