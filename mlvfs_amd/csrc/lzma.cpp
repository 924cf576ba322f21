// lzma.cpp -- LZMA payloads of compressed clips (MLV_VIDEO_CLASS_FLAG_LZMA).
//
// Replaces, for this path, LzmaUncompress as mlvfs/main.c:598-616 calls it (mlvfs/LZMA/LzmaLib.c:42-48 -> LzmaDecode,
// LzmaDec.c:970-993, with LZMA_FINISH_ANY): one frame = [u32 size of the packed frame][5 property bytes][LZMA stream], decoded
// into a flat buffer that is the dictionary.  Written from the published format (Igor Pavlov's LZMA specification: range coder
// with 11-bit adaptive probabilities, literals with lc / lp context and matched-byte mode, length coders, 6-bit distance slots,
// four repeat distances, 12 states); no code of the reference's vendored LZMA SDK is used.  The entropy decoding of one frame is
// one serial chain, so it runs on the host -- one frame per reader thread, mlvreader.cpp -- and hands the GPU stages the packed
// 14-bit stream they take for uncompressed clips.
//
// Return values follow LzmaDecode's: 0 = SZ_OK (output full, or end marker seen: *dst_len may then be smaller than the
// capacity), 1 = SZ_ERROR_DATA, 4 = SZ_ERROR_UNSUPPORTED (property byte), 6 = SZ_ERROR_INPUT_EOF (the input ends inside a symbol,
// or holds fewer than the 5 bytes the range coder starts with).  Like the reference's decoder a symbol is only decoded when the
// input holds all of it, and the range coder normalises BEFORE a bit is read (LzmaDec.c:22), so the same streams are accepted.
#include "common.h"

#include <cstring>
#include <vector>

namespace mlv {
namespace {

constexpr int kNumStates = 12, kNumPosBitsMax = 4, kLenLow = 8, kLenMid = 8, kLenHigh = 256, kMatchMinLen = 2;
constexpr int kNumLenToPosStates = 4, kStartPosModelIndex = 4, kEndPosModelIndex = 14, kNumFullDistances = 1 << (kEndPosModelIndex >> 1);
constexpr int kNumAlignBits = 4;
constexpr uint32_t kTop = 1u << 24;
typedef uint16_t Prob;
constexpr Prob kProbInit = 1024;

struct LenCoder {
    Prob choice, choice2, low[1 << kNumPosBitsMax][kLenLow], mid[1 << kNumPosBitsMax][kLenMid], high[kLenHigh];
};

struct Rc {
    const uint8_t *p, *end;
    uint32_t range, code;
    bool eof;                                            // a byte was wanted that the input does not have
    uint32_t next() { if (p < end) return *p++; eof = true; return 0; }
    void normalise() { if (range < kTop) { range <<= 8; code = (code << 8) | next(); } }
    int bit(Prob *pr)
    {
        normalise();
        const uint32_t bound = (range >> 11) * *pr;
        if (code < bound) { range = bound; *pr = (Prob)(*pr + ((2048 - *pr) >> 5)); return 0; }
        range -= bound; code -= bound; *pr = (Prob)(*pr - (*pr >> 5));
        return 1;
    }
    uint32_t direct(int nbits)
    {
        uint32_t r = 0;
        while (nbits--) {
            normalise();
            range >>= 1;
            code -= range;
            const uint32_t t = 0u - (code >> 31);
            code += range & t;
            r = (r << 1) + (t + 1);
        }
        return r;
    }
    int tree(Prob *probs, int nbits)
    {
        uint32_t m = 1;
        for (int i = 0; i < nbits; i++) m = (m << 1) + (uint32_t)bit(probs + m);
        return (int)(m - (1u << nbits));
    }
    int tree_rev(Prob *probs, int nbits)
    {
        uint32_t m = 1, sym = 0;
        for (int i = 0; i < nbits; i++) { const int b = bit(probs + m); m = (m << 1) + (uint32_t)b; sym |= (uint32_t)b << i; }
        return (int)sym;
    }
};

int len_decode(Rc &rc, LenCoder &lc, int pos_state)
{
    if (!rc.bit(&lc.choice)) return rc.tree(lc.low[pos_state], 3);
    if (!rc.bit(&lc.choice2)) return kLenLow + rc.tree(lc.mid[pos_state], 3);
    return kLenLow + kLenMid + rc.tree(lc.high, 8);
}

void init_probs(Prob *p, size_t n) { for (size_t i = 0; i < n; i++) p[i] = kProbInit; }

}  // namespace

int lzma_decode(const uint8_t props[5], const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap, size_t *dst_len)
{
    *dst_len = 0;
    if (src_len < 5) return 6;                            // LzmaDec.c:979
    unsigned d = props[0];
    if (d >= 9 * 5 * 5) return 4;                         // LzmaDec.c:878
    const int lc = (int)(d % 9); d /= 9;
    const int lp = (int)(d % 5), pb = (int)(d / 5);
    uint32_t dict = (uint32_t)props[1] | ((uint32_t)props[2] << 8) | ((uint32_t)props[3] << 16) | ((uint32_t)props[4] << 24);
    if (dict < 4096) dict = 4096;

    static_assert(sizeof(LenCoder) % sizeof(Prob) == 0, "probability model is an array of Prob");
    struct Model {
        Prob is_match[kNumStates][1 << kNumPosBitsMax], is_rep[kNumStates], is_rep_g0[kNumStates], is_rep_g1[kNumStates],
            is_rep_g2[kNumStates], is_rep0_long[kNumStates][1 << kNumPosBitsMax], pos_slot[kNumLenToPosStates][64],
            spec_pos[kNumFullDistances - kEndPosModelIndex], align[1 << kNumAlignBits];
        LenCoder len, rep_len;
    };
    std::vector<Prob> store(sizeof(Model) / sizeof(Prob) + ((size_t)0x300 << (lc + lp)));
    init_probs(store.data(), store.size());
    Model &m = *reinterpret_cast<Model *>(store.data());
    Prob *literal = store.data() + sizeof(Model) / sizeof(Prob);

    Rc rc{ src, src + src_len, 0xFFFFFFFFu, 0, false };
    if (rc.next() != 0) return 1;                         // LzmaDec.c:711: the first byte of the range coder's stream is 0
    for (int i = 0; i < 4; i++) rc.code = (rc.code << 8) | rc.next();

    uint32_t rep0 = 0, rep1 = 0, rep2 = 0, rep3 = 0;       // distances - 1
    int state = 0;
    size_t pos = 0;
    const uint32_t pb_mask = (1u << pb) - 1, lp_mask = (1u << lp) - 1;
    while (pos < dst_cap) {
        // A symbol is decoded only if the input holds all of it (LzmaDec.c:775-800, LzmaDec_TryDummy): remember where we
        // are, decode, and treat a read past the end as "needs more input"
        const int pos_state = (int)(pos & pb_mask);
        if (!rc.bit(&m.is_match[state][pos_state])) {
            Prob *pr = literal + (size_t)0x300 * (((pos & lp_mask) << lc) + (pos ? (uint32_t)(dst[pos - 1] >> (8 - lc)) : 0u));
            uint32_t sym = 1;
            if (state >= 7) {
                uint32_t mb = dst[pos - rep0 - 1];
                do {
                    const uint32_t match_bit = (mb >> 7) & 1;
                    mb <<= 1;
                    const uint32_t b = (uint32_t)rc.bit(pr + ((1 + match_bit) << 8) + sym);
                    sym = (sym << 1) | b;
                    if (match_bit != b) break;
                } while (sym < 0x100);
            }
            while (sym < 0x100) sym = (sym << 1) | (uint32_t)rc.bit(pr + sym);
            rc.normalise();                               // (the decoder of the reference ends every symbol like this: LzmaDec.c:445, :693)
            if (rc.eof) return 6;
            dst[pos++] = (uint8_t)sym;
            state = state < 4 ? 0 : (state < 10 ? state - 3 : state - 6);
            continue;
        }
        int len;
        if (rc.bit(&m.is_rep[state])) {
            if (pos == 0) { if (rc.eof) return 6; return 1; }                         // LzmaDec.c:303-304
            if (!rc.bit(&m.is_rep_g0[state])) {
                if (!rc.bit(&m.is_rep0_long[state][pos_state])) {                     // short rep: one byte
                    rc.normalise();
                    if (rc.eof) return 6;
                    dst[pos] = dst[pos - rep0 - 1];
                    pos++;
                    state = state < 7 ? 9 : 11;
                    continue;
                }
            } else {
                uint32_t dist;
                if (!rc.bit(&m.is_rep_g1[state])) dist = rep1;
                else {
                    if (!rc.bit(&m.is_rep_g2[state])) dist = rep2;
                    else { dist = rep3; rep3 = rep2; }
                    rep2 = rep1;
                }
                rep1 = rep0;
                rep0 = dist;
            }
            len = len_decode(rc, m.rep_len, pos_state);
            state = state < 7 ? 8 : 11;
        } else {
            rep3 = rep2; rep2 = rep1; rep1 = rep0;
            len = len_decode(rc, m.len, pos_state);
            state = state < 7 ? 7 : 10;
            const int slot = rc.tree(m.pos_slot[len < kNumLenToPosStates ? len : kNumLenToPosStates - 1], 6);
            if (slot < kStartPosModelIndex) rep0 = (uint32_t)slot;
            else {
                const int nbits = (slot >> 1) - 1;
                rep0 = (2u | ((uint32_t)slot & 1u)) << nbits;
                if (slot < kEndPosModelIndex) rep0 += (uint32_t)rc.tree_rev(m.spec_pos + rep0 - (uint32_t)slot - 1, nbits);
                else {
                    rep0 += rc.direct(nbits - kNumAlignBits) << kNumAlignBits;
                    rep0 += (uint32_t)rc.tree_rev(m.align, kNumAlignBits);
                    if (rep0 == 0xFFFFFFFFu) {                                       // end marker
                        rc.normalise();
                        if (rc.eof) return 6;
                        *dst_len = pos;
                        return rc.code == 0 ? 0 : 1;                                  // LzmaDec.c:841-843
                    }
                }
            }
            if (rc.eof) return 6;
            // LzmaDec.c:392-402: the distance must lie inside what has been produced (and inside the dictionary size of the
            // properties once that much has been produced)
            if (rep0 >= (pos >= dict ? (size_t)dict : pos)) return 1;
        }
        rc.normalise();
        if (rc.eof) return 6;
        len += kMatchMinLen;
        size_t n = (size_t)len;
        if (n > dst_cap - pos) n = dst_cap - pos;            // the rest of the match would follow in a larger buffer
        const size_t from = pos - rep0 - 1;
        for (size_t i = 0; i < n; i++) dst[pos + i] = dst[from + i];
        pos += n;
    }
    *dst_len = pos;
    return 0;
}

}  // namespace mlv

// payload of one VIDF block of an LZMA clip -> the packed frame.  0 = ok; the reference's LZMA error code otherwise (-2: bad argument)
extern "C" int mlvfs_amd_lzma_uncompress(const void *payload, size_t size, void *dst, size_t dst_cap, size_t *out_size)
{
    if (out_size) *out_size = 0;
    if (!payload || !dst || size < 4 + 5) { mlv::set_error("lzma: payload too short"); return MLVFS_AMD_ERR_ARG; }
    const uint8_t *b = (const uint8_t *)payload;
    size_t want = (size_t)b[0] | ((size_t)b[1] << 8) | ((size_t)b[2] << 16) | ((size_t)b[3] << 24);      // main.c:600
    if (want > dst_cap) { mlv::set_error("lzma: the frame decodes to %zu bytes, the buffer holds %zu", want, dst_cap); return MLVFS_AMD_ERR_ARG; }
    size_t got = 0;
    const int rc = mlv::lzma_decode(b + 4, b + 9, size - 9, (uint8_t *)dst, want, &got);
    if (out_size) *out_size = got;
    if (rc) mlv::set_error("lzma: stream not decodable (error %d)", rc);
    return rc;
}
