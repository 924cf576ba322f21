// mlvreader.cpp -- MLV container walk and frame prefetch.
//
// What MLVFS does per frame read, done once per clip here and kept in memory:
//   chunk files   <name>.MLV + <name>.M00 .. M98                 mlvfs/index.c:367-424 (load_chunks)
//   XREF index    every block of every chunk, NULL blocks left out, MLVI at time 0, stable order by timestamp
//                                                                  mlvfs/index.c:78-99, 216-341 (xref_sort, make_index)
//   .IDX file     MLVI (blockSize 52, frame counts 0, fileNum = chunks + 1) followed by the XREF block
//                                                                  mlvfs/index.c:101-214 (load_index, save_index)
//   frame headers the MLVI / RTCI / IDNT / RAWI / EXPO / LENS / WBAL blocks that precede the n-th VIDF in XREF order,
//                 each copied over the previous one of its kind     mlvfs/main.c:429-558 (mlv_get_frame_headers)
//   payload       position + sizeof(VIDF header) + frameSpace       mlvfs/main.c:688-702 (get_image_data, uncompressed)
// and, what MLVFS's README lists as missing (`--prefetch`): payloads of a whole batch of frames are read by a pool of
// threads into page-locked memory while the previous batch is on the GPU (mlvfs_amd_mlv_process).
//
// The reference opens, indexes and walks the clip again for every frame it serves; results are the same, the cost is not.
// LJ92 payloads (main.c:617-681) are decoded on the GPU (csrc/lj92.cpp) inside mlvfs_amd_mlv_process; LZMA payloads
// (main.c:598-616) by the reader threads that fetch them (csrc/lzma.cpp: one frame per thread), so that an LZMA clip looks like
// an uncompressed one to everything behind read_frames.  LJ92 payloads are refused in mlvfs_amd_mlv_read_frames (which hands
// out packed pixels).
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <exception>
#include <mutex>
#include <string>
#include <thread>
#include <memory>
#include <vector>

#include "clip.h"

namespace {

using mlv::set_error;

#pragma pack(push, 1)
struct BlockHead { uint8_t type[4]; uint32_t size; uint64_t timestamp; };                       // mlv.h:42-46
struct XrefHead { uint8_t type[4]; uint32_t size; uint64_t timestamp; uint32_t frame_type, entries; };   // mlv.h:166-173
struct XrefEntry { uint16_t file; uint8_t empty, kind; uint64_t offset; };                       // mlv.h:159-164
#pragma pack(pop)
static_assert(sizeof(BlockHead) == 16 && sizeof(XrefHead) == 24 && sizeof(XrefEntry) == 12, "MLV v2.0 layouts");

enum : uint8_t { KIND_OTHER = 0, KIND_VIDF = 1, KIND_AUDF = 2 };
constexpr uint16_t CLASS_LZMA = 0x80, CLASS_LJ92 = 0x100;                                         // mlv.h:30-31 (the reference's values: LJ92 is 0x100 there)

bool read_at(int fd, void *dst, size_t n, uint64_t off)
{
    uint8_t *p = (uint8_t *)dst;
    while (n) {
        const ssize_t r = pread(fd, p, n, (off_t)off);
        if (r <= 0) return false;
        p += r; off += (uint64_t)r; n -= (size_t)r;
    }
    return true;
}

struct Reader {
    std::string path;
    std::vector<int> fds;
    std::vector<XrefEntry> xref;
    std::vector<uint32_t> vidf;                 // XREF position of the n-th video frame
    // frame_headers as mlv_get_frame_headers leaves them, without the VIDF part: one snapshot per run of frames between
    // two metadata blocks (a clip has a handful; a clip with per-frame EXPO/LENS blocks one per frame)
    std::vector<frame_headers> snap;
    std::vector<uint32_t> snap_of;              // per frame
    std::vector<uint8_t> has_rawi;              // per snapshot
    frame_headers last;                         // state after every block (what an index past the last frame leaves behind)
    // page-locked staging of mlvfs_amd_mlv_process, kept between calls (locking pages costs about as much as reading them)
    mutable uint8_t *stage[2] = { nullptr, nullptr };
    mutable size_t stage_bytes = 0;
    // LJ92 clips: decoded frames of one batch and two buffers of finished frames in HBM; the download of one batch runs on
    // its own stream while the next batch is decoded
    mutable void *d_dec = nullptr, *d_fin[2] = { nullptr, nullptr };
    mutable size_t dev_bytes = 0;
    mutable hipStream_t s_run = nullptr, s_down = nullptr;
    mutable hipEvent_t fin_done[2] = { nullptr, nullptr }, down_done[2] = { nullptr, nullptr };
    mutable std::mutex stage_mu;
    // dual-ISO clips (mlvfs_amd_mlv_process_dualiso): packed payloads of one batch and two buffers of 16-bit frames in HBM
    mutable void *d_di_packed = nullptr, *d_di_frames[2] = { nullptr, nullptr };
    mutable size_t di_packed_bytes = 0, di_frames_bytes = 0;

    ~Reader()
    {
        if (d_di_packed) (void)hipFree(d_di_packed);
        for (int k = 0; k < 2; k++) if (d_di_frames[k]) (void)hipFree(d_di_frames[k]);
        for (int fd : fds) if (fd >= 0) close(fd);
        mlvfs_amd_host_free(stage[0]);
        mlvfs_amd_host_free(stage[1]);
        if (d_dec) (void)hipFree(d_dec);
        for (int k = 0; k < 2; k++) {
            if (d_fin[k]) (void)hipFree(d_fin[k]);
            if (fin_done[k]) (void)hipEventDestroy(fin_done[k]);
            if (down_done[k]) (void)hipEventDestroy(down_done[k]);
        }
        if (s_run) (void)hipStreamDestroy(s_run);
        if (s_down) (void)hipStreamDestroy(s_down);
    }
};

// ---- chunks ---------------------------------------------------------------------------------------------------------
bool open_chunks(Reader &r)
{
    int fd = open(r.path.c_str(), O_RDONLY);
    if (fd < 0) { set_error("mlv: cannot open %s: %s", r.path.c_str(), strerror(errno)); return false; }
    r.fds.push_back(fd);
    if (r.path.size() < 3) return true;
    std::string name = r.path;
    for (int seq = 0; seq < 99; seq++) {                            // .M00, .M01, ... until one is missing
        char two[8];
        snprintf(two, sizeof two, "%02d", seq);
        name.replace(name.size() - 2, 2, two);
        fd = open(name.c_str(), O_RDONLY);
        if (fd < 0) break;
        r.fds.push_back(fd);
    }
    return true;
}

// ---- index ----------------------------------------------------------------------------------------------------------
struct Timed { uint64_t time; XrefEntry e; };

void scan_chunks(Reader &r)
{
    std::vector<Timed> all;
    mlv_file_hdr_t first{};                                         // the MLVI with fileNum 0; all zero until one is seen
    for (size_t c = 0; c < r.fds.size(); c++) {
        uint64_t pos = 0;
        for (;;) {
            BlockHead h;
            if (!read_at(r.fds[c], &h, sizeof h, pos)) break;
            if (h.size < sizeof h || h.size > (1u << 30)) {
                fprintf(stderr, "Invalid header size: %d bytes at 0x%08llX\n", (int)h.size, (unsigned long long)pos);
                break;
            }
            uint64_t t = h.timestamp;
            if (!memcmp(h.type, "MLVI", 4)) {
                mlv_file_hdr_t fh{};
                if (!read_at(r.fds[c], &fh, std::min<size_t>(sizeof fh, h.size), pos)) break;
                if (fh.fileNum == 0) first = fh;
                else if (first.fileGuid != fh.fileGuid) break;      // a chunk of another recording: stop reading it
                t = 0;                                               // the bytes at the timestamp's place are the version string
            }
            if (memcmp(h.type, "NULL", 4)) {
                Timed x{};
                x.time = t;
                x.e.file = (uint16_t)c;
                x.e.kind = !memcmp(h.type, "VIDF", 4) ? KIND_VIDF : !memcmp(h.type, "AUDF", 4) ? KIND_AUDF : KIND_OTHER;
                x.e.offset = pos;
                all.push_back(x);
            }
            pos += h.size;
        }
    }
    // the reference bubble-sorts on `>`: equal timestamps keep their scan order
    std::stable_sort(all.begin(), all.end(), [](const Timed &a, const Timed &b) { return a.time < b.time; });
    r.xref.resize(all.size());
    for (size_t i = 0; i < all.size(); i++) r.xref[i] = all[i].e;
}

std::string idx_name(const std::string &path)
{
    std::string n = path;
    if (n.size() >= 3) n.replace(n.size() - 3, 3, "IDX");
    return n;
}

bool load_idx(Reader &r)
{
    const int fd = open(idx_name(r.path).c_str(), O_RDONLY);
    if (fd < 0) return false;
    bool ok = false;
    uint64_t pos = 0;
    for (;;) {                                                      // walk the blocks of the .IDX until an XREF is found
        BlockHead h;
        if (!read_at(fd, &h, sizeof h, pos) || h.size < sizeof h) break;
        if (!memcmp(h.type, "XREF", 4)) {
            XrefHead xh;
            if (h.size >= sizeof xh && read_at(fd, &xh, sizeof xh, pos) &&
                (uint64_t)xh.entries * sizeof(XrefEntry) + sizeof xh <= h.size) {
                r.xref.resize(xh.entries);
                ok = xh.entries == 0 || read_at(fd, r.xref.data(), xh.entries * sizeof(XrefEntry), pos + sizeof xh);
            }
            break;
        }
        pos += h.size;
    }
    close(fd);
    if (ok)
        for (const XrefEntry &e : r.xref)
            if (e.file >= r.fds.size()) ok = false;                 // an index of a clip with more chunks than are here
    if (!ok) r.xref.clear();
    return ok;
}

std::vector<uint8_t> xref_block(const Reader &r)
{
    std::vector<uint8_t> b(sizeof(XrefHead) + r.xref.size() * sizeof(XrefEntry), 0);
    XrefHead h{};
    memcpy(h.type, "XREF", 4);
    h.size = (uint32_t)b.size();
    h.entries = (uint32_t)r.xref.size();
    memcpy(b.data(), &h, sizeof h);
    if (!r.xref.empty()) memcpy(b.data() + sizeof h, r.xref.data(), r.xref.size() * sizeof(XrefEntry));
    return b;
}

bool save_idx(const Reader &r)
{
    mlv_file_hdr_t fh{};
    (void)read_at(r.fds[0], &fh, sizeof fh, 0);                     // the first chunk's first 52 bytes, whatever they are
    fh.blockSize = sizeof fh;
    fh.videoFrameCount = 0;
    fh.audioFrameCount = 0;
    fh.fileNum = (uint16_t)(r.fds.size() + 1);
    const std::vector<uint8_t> x = xref_block(r);
    FILE *f = fopen(idx_name(r.path).c_str(), "wb+");
    if (!f) return false;
    const bool ok = fwrite(&fh, sizeof fh, 1, f) == 1 && fwrite(x.data(), x.size(), 1, f) == 1;
    fclose(f);
    return ok;
}

// ---- frame headers ----------------------------------------------------------------------------------------------------
// One pass over the index in XREF order with the running state of mlv_get_frame_headers; a snapshot is taken whenever a
// video frame follows a change.
void gather_headers(Reader &r)
{
    frame_headers cur;
    memset(&cur, 0, sizeof cur);
    bool rawi = false, dirty = true;
    for (uint32_t i = 0; i < r.xref.size(); i++) {
        const XrefEntry &e = r.xref[i];
        const int fd = r.fds[e.file];
        if (e.kind == KIND_VIDF) {
            if (dirty) { r.snap.push_back(cur); r.has_rawi.push_back(rawi); dirty = false; }
            r.vidf.push_back(i);
            r.snap_of.push_back((uint32_t)r.snap.size() - 1);
            continue;
        }
        if (e.kind == KIND_AUDF) continue;
        BlockHead h;
        if (!read_at(fd, &h, sizeof h, e.offset)) continue;
        struct { const char *tag; void *dst; size_t cap; } kinds[] = {
            { "MLVI", &cur.file_hdr, sizeof cur.file_hdr }, { "RTCI", &cur.rtci_hdr, sizeof cur.rtci_hdr },
            { "IDNT", &cur.idnt_hdr, sizeof cur.idnt_hdr }, { "RAWI", &cur.rawi_hdr, sizeof cur.rawi_hdr },
            { "EXPO", &cur.expo_hdr, sizeof cur.expo_hdr }, { "LENS", &cur.lens_hdr, sizeof cur.lens_hdr },
            { "WBAL", &cur.wbal_hdr, sizeof cur.wbal_hdr },
        };
        for (auto &k : kinds) {
            if (memcmp(h.type, k.tag, 4)) continue;
            const bool got = read_at(fd, k.dst, std::min<size_t>(k.cap, h.size), e.offset);
            if (k.dst == (void *)&cur.rawi_hdr && got) rawi = true;
            dirty = true;
            break;
        }
    }
    r.last = cur;
}

int frame_headers_of(const Reader &r, int index, frame_headers *out)
{
    if (index < 0 || (size_t)index >= r.vidf.size()) {
        *out = r.last;                                              // the reference has walked every block by now
        fprintf(stderr, "%s: Error reading frame headers: vidf block for frame %d was not found\n", r.path.c_str(), index);
        return 0;
    }
    *out = r.snap[r.snap_of[index]];
    const XrefEntry &e = r.xref[r.vidf[index]];
    out->fileNumber = e.file;
    out->position = e.offset;
    BlockHead h;
    if (read_at(r.fds[e.file], &h, sizeof h, e.offset))
        (void)read_at(r.fds[e.file], &out->vidf_hdr, std::min<size_t>(sizeof out->vidf_hdr, h.size), e.offset);
    if (!r.has_rawi[r.snap_of[index]]) {
        fprintf(stderr, "%s: Error reading frame headers: no rawi block was found\n", r.path.c_str());
        return 0;
    }
    return 1;
}

// ---- payloads ---------------------------------------------------------------------------------------------------------
struct Span { int fd; uint64_t off; size_t bytes; size_t lzma_bytes = 0; };      // lzma_bytes != 0: the block's compressed payload

bool payload_span(const Reader &r, int index, Span *s, bool lj92 = false)
{
    frame_headers fh;
    if (!frame_headers_of(r, index, &fh)) { set_error("mlv: frame %d has no usable headers", index); return false; }
    uint64_t room = fh.vidf_hdr.blockSize > sizeof(mlv_vidf_hdr_t) + fh.vidf_hdr.frameSpace
                        ? fh.vidf_hdr.blockSize - sizeof(mlv_vidf_hdr_t) - fh.vidf_hdr.frameSpace : 0;
    {
        // blockSize is a field of the file: a payload cannot be longer than what the chunk holds behind its start (buffers
        // are sized from it)
        struct stat st;
        const uint64_t start = fh.position + sizeof(mlv_vidf_hdr_t) + fh.vidf_hdr.frameSpace;
        const uint64_t left = fstat(r.fds[fh.fileNumber], &st) == 0 && (uint64_t)st.st_size > start ? (uint64_t)st.st_size - start : 0;
        room = std::min(room, left);
    }
    if (lj92 && (fh.file_hdr.videoClass & CLASS_LJ92) && !(fh.file_hdr.videoClass & CLASS_LZMA)) {
        // main.c:587-589: everything behind the VIDF header and its frameSpace is the compressed frame (size word + JPEG)
        if (room <= 4) { set_error("mlv: frame %d: empty LJ92 payload", index); return false; }
        s->fd = r.fds[fh.fileNumber];
        s->off = fh.position + sizeof(mlv_vidf_hdr_t) + fh.vidf_hdr.frameSpace;
        s->bytes = (size_t)room;
        return true;
    }
    if ((fh.file_hdr.videoClass & CLASS_LJ92) && !(fh.file_hdr.videoClass & CLASS_LZMA)) {
        set_error("mlv: LJ92 payloads (video class 0x%x) are decoded only by mlvfs_amd_mlv_process", fh.file_hdr.videoClass);
        return false;
    }
    const uint64_t bits = (uint64_t)fh.rawi_hdr.xRes * fh.rawi_hdr.yRes * (uint64_t)fh.rawi_hdr.raw_info.bits_per_pixel;
    s->fd = r.fds[fh.fileNumber];
    s->off = fh.position + sizeof(mlv_vidf_hdr_t) + fh.vidf_hdr.frameSpace;
    s->bytes = (size_t)((bits + 7) / 8);
    if (fh.file_hdr.videoClass & CLASS_LZMA) {           // main.c:573 tests this flag first
        if (room < 4 + 5 + 5) { set_error("mlv: frame %d: empty LZMA payload", index); return false; }
        s->lzma_bytes = (size_t)room;                     // [size][properties][stream], decoded by the thread that reads it
        return true;
    }
    const uint64_t in_block = fh.vidf_hdr.blockSize > sizeof(mlv_vidf_hdr_t) + fh.vidf_hdr.frameSpace
                                  ? fh.vidf_hdr.blockSize - sizeof(mlv_vidf_hdr_t) - fh.vidf_hdr.frameSpace : 0;
    if (in_block < s->bytes) { set_error("mlv: frame %d: VIDF block holds %llu payload bytes, geometry needs %zu", index,
                                         (unsigned long long)in_block, s->bytes); return false; }
    return true;
}

int read_frames(const Reader &r, int first, int count, uint8_t *dst, size_t stride, int threads, bool lj92 = false,
                size_t *sizes = nullptr)
{
    if (count <= 0) return MLVFS_AMD_OK;
    std::vector<Span> spans(count);
    for (int k = 0; k < count; k++) {
        if (!payload_span(r, first + k, &spans[k], lj92)) return MLVFS_AMD_ERR_ARG;
        if (sizes) sizes[k] = spans[k].bytes;
        if (spans[k].bytes > stride) { set_error("mlv: stride %zu smaller than a frame payload (%zu)", stride, spans[k].bytes); return MLVFS_AMD_ERR_ARG; }
    }
    threads = std::max(1, std::min(threads <= 0 ? 8 : threads, count));
    std::atomic<int> next{ 0 }, failed{ -1 }, failed_lzma{ -1 };
    auto work = [&]() {
        for (int k; (k = next.fetch_add(1)) < count;) try {
            uint8_t *d = dst + (size_t)k * stride;
            if (spans[k].lzma_bytes) {
                // main.c:598-616: the size word says how much LzmaUncompress may produce; what dng_get_image_data then reads
                // is the frame's packed size (a shorter result leaves the rest of the reference's buffer undefined: zeros here)
                std::vector<uint8_t> comp(spans[k].lzma_bytes);
                size_t got = 0;
                if (!read_at(spans[k].fd, comp.data(), comp.size(), spans[k].off)) { failed = first + k; continue; }
                const size_t want = (size_t)comp[0] | ((size_t)comp[1] << 8) | ((size_t)comp[2] << 16) | ((size_t)comp[3] << 24);
                // the size word is a field of the file too: a frame cannot decode to much more than its packed size (the
                // reference would malloc up to 4 GiB per worker here)
                if (want > 8 * spans[k].bytes + (1u << 20)) { failed_lzma = first + k; continue; }
                std::vector<uint8_t> big;
                uint8_t *out = d;
                if (want > stride) { big.resize(want); out = big.data(); }      // a size word larger than the frame: decode all, keep the frame
                if (mlv::lzma_decode(comp.data() + 4, comp.data() + 9, comp.size() - 9, out, want, &got) != 0) { failed_lzma = first + k; continue; }
                if (out != d) memcpy(d, out, std::min(got, spans[k].bytes));
                if (got < spans[k].bytes) memset(d + got, 0, spans[k].bytes - got);
            } else if (!read_at(spans[k].fd, d, spans[k].bytes, spans[k].off)) { failed = first + k; continue; }
            if (stride > spans[k].bytes) memset(d + spans[k].bytes, 0, std::min<size_t>(stride - spans[k].bytes, 64));   // the 2-pixel over-read of main.c:579 sees zeros
        } catch (const std::exception &) {               // out of memory in a reader thread: the frame fails, the host lives
            if (spans[k].lzma_bytes) failed_lzma = first + k; else failed = first + k;
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    if (failed >= 0) { set_error("mlv: short read in the payload of frame %d", (int)failed); return MLVFS_AMD_ERR_ARG; }
    if (failed_lzma >= 0) { set_error("mlv: the LZMA payload of frame %d cannot be decoded", (int)failed_lzma); return MLVFS_AMD_ERR_ARG; }   // main.c:614: "LZMA Failed!"
    return MLVFS_AMD_OK;
}

}  // namespace

extern "C" {

void *mlvfs_amd_mlv_open(const char *mlv_path, int use_idx_file)
{
    if (!mlv_path) { set_error("mlv: no path"); return nullptr; }
    Reader *r = new Reader;
    r->path = mlv_path;
    if (!open_chunks(*r)) { delete r; return nullptr; }
    if (!(use_idx_file && load_idx(*r))) {
        scan_chunks(*r);
        if (use_idx_file) (void)save_idx(*r);
    }
    gather_headers(*r);
    return r;
}

void mlvfs_amd_mlv_close(void *reader) { delete (Reader *)reader; }

int mlvfs_amd_mlv_frame_count(const void *reader) { return reader ? (int)((const Reader *)reader)->vidf.size() : 0; }

int mlvfs_amd_mlv_chunk_count(const void *reader) { return reader ? (int)((const Reader *)reader)->fds.size() : 0; }

size_t mlvfs_amd_mlv_xref(const void *reader, void *dst, size_t cap)
{
    if (!reader) return 0;
    const std::vector<uint8_t> b = xref_block(*(const Reader *)reader);
    if (dst && cap >= b.size()) memcpy(dst, b.data(), b.size());
    return b.size();
}

int mlvfs_amd_mlv_frame_headers(const void *reader, int index, struct frame_headers *out)
{
    if (!reader || !out) return 0;
    return frame_headers_of(*(const Reader *)reader, index, out);
}

int mlvfs_amd_mlv_read_frames(const void *reader, int first, int count, void *dst, size_t stride, int io_threads)
{
    if (!reader || !dst) { set_error("mlv: null argument"); return MLVFS_AMD_ERR_ARG; }
    return read_frames(*(const Reader *)reader, first, count, (uint8_t *)dst, stride, io_threads);
}

int mlvfs_amd_mlv_process(const void *reader, mlvfs_amd_clip_t *clip, int first, int count, void *h_out, size_t out_stride,
                          int cs_method, int fix_pixels, int apply_stripes, int batch_frames, int io_threads)
{
    if (!reader || !clip || !h_out) { set_error("mlv: null argument"); return MLVFS_AMD_ERR_ARG; }
    const Reader &r = *(const Reader *)reader;
    if (count <= 0) return MLVFS_AMD_OK;
    if (first < 0 || (size_t)first + (size_t)count > r.vidf.size()) { set_error("mlv: frames %d..%d outside the clip (%zu frames)", first, first + count - 1, r.vidf.size()); return MLVFS_AMD_ERR_ARG; }
    frame_headers fh0;
    if (!frame_headers_of(r, first, &fh0)) { set_error("mlv: frame %d has no usable headers", first); return MLVFS_AMD_ERR_ARG; }
    const bool lj92 = (fh0.file_hdr.videoClass & CLASS_LJ92) && !(fh0.file_hdr.videoClass & CLASS_LZMA);
    {
        // the clip state must describe these frames: the kernels take their geometry from it
        const mlv::Geom &g = reinterpret_cast<const mlv::Clip *>(clip)->g;
        if (g.w != fh0.rawi_hdr.xRes || g.h != fh0.rawi_hdr.yRes || (!lj92 && g.bpp != fh0.rawi_hdr.raw_info.bits_per_pixel)) {
            set_error("mlv: the clip state is %dx%d at %d bits, the file's frames are %dx%d at %d bits", g.w, g.h, g.bpp,
                      fh0.rawi_hdr.xRes, fh0.rawi_hdr.yRes, fh0.rawi_hdr.raw_info.bits_per_pixel);
            return MLVFS_AMD_ERR_ARG;
        }
    }
    if (batch_frames <= 0) batch_frames = 32;
    batch_frames = std::min(batch_frames, count);
    size_t stride = 0;
    if (lj92) {                                                     // payload sizes vary: the largest one sets the staging pitch
        for (int k = 0; k < count; k++) {
            Span sp;
            if (!payload_span(r, first + k, &sp, true)) return MLVFS_AMD_ERR_ARG;
            stride = std::max(stride, sp.bytes);
        }
        stride = (stride + 15) / 16 * 16;
    } else {
        Span s0;
        if (!payload_span(r, first, &s0)) return MLVFS_AMD_ERR_ARG;
        stride = (s0.bytes + 2 + 15) / 16 * 16;
    }
    std::lock_guard<std::mutex> lk(r.stage_mu);                     // one streaming call per reader at a time
    if (r.stage_bytes < stride * batch_frames) {
        mlvfs_amd_host_free(r.stage[0]);
        mlvfs_amd_host_free(r.stage[1]);
        r.stage[0] = (uint8_t *)mlvfs_amd_host_alloc(stride * batch_frames);
        r.stage[1] = (uint8_t *)mlvfs_amd_host_alloc(stride * batch_frames);
        r.stage_bytes = (r.stage[0] && r.stage[1]) ? stride * batch_frames : 0;
    }
    uint8_t *const *stage = r.stage;
    int rc = r.stage_bytes ? MLVFS_AMD_OK : MLVFS_AMD_ERR_NOMEM;
    std::vector<size_t> sizes[2] = { std::vector<size_t>(batch_frames), std::vector<size_t>(batch_frames) };
    const size_t px_bytes = (size_t)fh0.rawi_hdr.xRes * fh0.rawi_hdr.yRes * 2, dstride = (px_bytes + 255) / 256 * 256;
    if (rc == MLVFS_AMD_OK && lj92 && r.dev_bytes < dstride * batch_frames) {
        if (!mlv::thread_ctx()) return MLVFS_AMD_ERR_HIP;
        if (r.d_dec) (void)hipFree(r.d_dec);
        r.d_dec = nullptr; r.dev_bytes = 0;
        MLV_HIP(hipMalloc(&r.d_dec, dstride * batch_frames));
        for (int k = 0; k < 2; k++) {
            if (r.d_fin[k]) (void)hipFree(r.d_fin[k]);
            r.d_fin[k] = nullptr;
            MLV_HIP(hipMalloc(&r.d_fin[k], dstride * batch_frames));
        }
        r.dev_bytes = dstride * batch_frames;
    }
    if (rc == MLVFS_AMD_OK && lj92 && !r.s_run) {
        MLV_HIP(hipStreamCreateWithFlags(&r.s_run, hipStreamNonBlocking));
        MLV_HIP(hipStreamCreateWithFlags(&r.s_down, hipStreamNonBlocking));
        for (int k = 0; k < 2; k++) {
            MLV_HIP(hipEventCreateWithFlags(&r.fin_done[k], hipEventDisableTiming));
            MLV_HIP(hipEventCreateWithFlags(&r.down_done[k], hipEventDisableTiming));
        }
    }
    if (rc == MLVFS_AMD_OK) rc = read_frames(r, first, batch_frames, stage[0], stride, io_threads, lj92, sizes[0].data());
    for (int f0 = 0, k = 0; rc == MLVFS_AMD_OK && f0 < count; f0 += batch_frames, k++) {
        const int n = std::min(batch_frames, count - f0), n_next = std::min(batch_frames, count - f0 - n);
        int rc_io = MLVFS_AMD_OK;
        std::thread io;                                             // the next batch is read while this one is on the GPU
        if (n_next > 0) io = std::thread([&, k, f0, n, n_next]() { rc_io = read_frames(r, first + f0 + n, n_next, stage[(k + 1) & 1], stride, io_threads, lj92, sizes[(k + 1) & 1].data()); });
        if (lj92) {
            // payload = 32-bit decoded size, then the JPEG stream (main.c:628-633); decode, run the stages in HBM, copy out
            std::vector<const void *> ptr(n);
            std::vector<size_t> len(n);
            for (int i = 0; i < n; i++) { ptr[i] = stage[k & 1] + (size_t)i * stride + 4; len[i] = sizes[k & 1][i] - 4; }
            const int slot = k & 1;
            rc = mlvfs_amd_lj92_decode_dev(ptr.data(), len.data(), n, fh0.rawi_hdr.xRes, fh0.rawi_hdr.yRes, r.d_dec, dstride, r.s_run);
            // (no early returns in here: the reader thread of the next batch is running)
            auto ok = [&](hipError_t e) { if (e != hipSuccess && rc == MLVFS_AMD_OK) { set_error("mlv: %s", hipGetErrorString(e)); rc = MLVFS_AMD_ERR_HIP; } };
            if (rc == MLVFS_AMD_OK && k >= 2) ok(hipStreamWaitEvent(r.s_run, r.down_done[slot], 0));        // d_fin[slot] has been downloaded
            if (rc == MLVFS_AMD_OK) rc = mlvfs_amd_process_unpacked_dev(clip, r.d_dec, dstride, r.d_fin[slot], dstride, n, cs_method, fix_pixels, apply_stripes, r.s_run);
            if (rc == MLVFS_AMD_OK) {
                ok(hipEventRecord(r.fin_done[slot], r.s_run));
                ok(hipStreamWaitEvent(r.s_down, r.fin_done[slot], 0));
                ok(hipMemcpy2DAsync((uint8_t *)h_out + (size_t)f0 * out_stride, out_stride, r.d_fin[slot], dstride, px_bytes, n, hipMemcpyDeviceToHost, r.s_down));
                ok(hipEventRecord(r.down_done[slot], r.s_down));
            }
        } else
        rc = mlvfs_amd_process_frames_host(clip, stage[k & 1], stride, (uint8_t *)h_out + (size_t)f0 * out_stride, out_stride, n,
                                           cs_method, fix_pixels, apply_stripes, 0);
        if (io.joinable()) io.join();
        if (rc == MLVFS_AMD_OK && rc_io != MLVFS_AMD_OK) {          // the reader thread's message lives in its own thread
            set_error("mlv: prefetch of frames %d..%d failed", first + f0 + n, first + f0 + n + n_next - 1);
            rc = rc_io;
        }
    }
    if (lj92 && r.s_down) {                                         // the last downloads
        if (hipStreamSynchronize(r.s_down) != hipSuccess || hipStreamSynchronize(r.s_run) != hipSuccess) {
            set_error("mlv: copying finished frames to the host failed");
            if (rc == MLVFS_AMD_OK) rc = MLVFS_AMD_ERR_HIP;
        }
    }
    return rc;
}

// = gif_get_data (gif.c:82-221) on an opened clip: the animated preview of 10 frames spread over the clip, 1/4 x 1/4 size; copies
// min(max_size, size - offset) bytes of the file from `offset` on and returns max_size like the reference, 0 on failure.
// A dual-ISO clip, file -> host: main.c:942 + 956-959 for every frame of a clip (get_image_data, then cr2hdr20_convert_data with the
// frame headers' levels), batched: the payloads of batch k + 1 are read into page-locked staging by the reader threads while batch k
// is on the GPU -- upload, unpack (k_unpack), ONE submission of the batched conversion (mlvfs_amd_cr2hdr20_batch_dev) --, and the
// finished frames of batch k travel back on their own stream under the next batch's kernels.  results[i] = 1: frame first + i was
// converted (its black and white level are 4x the headers', hdr.c:1951-1952); 0: it is no dual ISO frame (or the detection
// failed) and h_out holds it unpacked, as process_frame would go on with it (main.c:961-973).
int mlvfs_amd_mlv_process_dualiso(const void *reader, int first, int count, void *h_out, size_t out_stride, int interp_method, int fullres,
                                  int use_alias_map, int chroma_smooth, int batch_frames, int io_threads, int *results)
{
    if (!reader || !h_out || !results) { set_error("mlv: null argument"); return MLVFS_AMD_ERR_ARG; }
    const Reader &r = *(const Reader *)reader;
    if (count <= 0) return MLVFS_AMD_OK;
    if (first < 0 || (size_t)first + (size_t)count > r.vidf.size()) { set_error("mlv: frames %d..%d outside the clip (%zu frames)", first, first + count - 1, r.vidf.size()); return MLVFS_AMD_ERR_ARG; }
    frame_headers fh0;
    if (!frame_headers_of(r, first, &fh0)) { set_error("mlv: frame %d has no usable headers", first); return MLVFS_AMD_ERR_ARG; }
    if ((fh0.file_hdr.videoClass & CLASS_LJ92) && !(fh0.file_hdr.videoClass & CLASS_LZMA)) { set_error("mlv: dual-ISO batches take plain and LZMA clips"); return MLVFS_AMD_ERR_ARG; }
    const int w = fh0.rawi_hdr.xRes, h = fh0.rawi_hdr.yRes, bpp = fh0.rawi_hdr.raw_info.bits_per_pixel;
    const mlvfs_amd_geom_t geom{ w, h, bpp, (int32_t)fh0.rawi_hdr.raw_info.black_level, (int32_t)fh0.rawi_hdr.raw_info.white_level, 0, 0 };
    const size_t px_bytes = (size_t)w * h * 2;
    if (out_stride < px_bytes) { set_error("mlv: out_stride smaller than a frame"); return MLVFS_AMD_ERR_ARG; }
    if (batch_frames <= 0) batch_frames = 8;
    batch_frames = std::min(batch_frames, count);
    Span s0;
    if (!payload_span(r, first, &s0)) return MLVFS_AMD_ERR_ARG;
    const size_t stride = (s0.bytes + 2 + 15) / 16 * 16, dstride = (px_bytes + 255) / 256 * 256;
    mlv::ThreadCtx *c = mlv::thread_ctx();
    if (!c) return MLVFS_AMD_ERR_HIP;
    std::lock_guard<std::mutex> lk(r.stage_mu);                     // one streaming call per reader at a time
    if (r.stage_bytes < stride * batch_frames) {
        mlvfs_amd_host_free(r.stage[0]);
        mlvfs_amd_host_free(r.stage[1]);
        r.stage[0] = (uint8_t *)mlvfs_amd_host_alloc(stride * batch_frames);
        r.stage[1] = (uint8_t *)mlvfs_amd_host_alloc(stride * batch_frames);
        r.stage_bytes = (r.stage[0] && r.stage[1]) ? stride * batch_frames : 0;
        if (!r.stage_bytes) return MLVFS_AMD_ERR_NOMEM;
    }
    if (r.di_packed_bytes < stride * batch_frames) {
        if (r.d_di_packed) (void)hipFree(r.d_di_packed);
        r.d_di_packed = nullptr; r.di_packed_bytes = 0;
        MLV_HIP(hipMalloc(&r.d_di_packed, stride * batch_frames));
        r.di_packed_bytes = stride * batch_frames;
    }
    if (r.di_frames_bytes < dstride * batch_frames) {
        for (int k = 0; k < 2; k++) {
            if (r.d_di_frames[k]) (void)hipFree(r.d_di_frames[k]);
            r.d_di_frames[k] = nullptr;
        }
        r.di_frames_bytes = 0;
        for (int k = 0; k < 2; k++) MLV_HIP(hipMalloc(&r.d_di_frames[k], dstride * batch_frames));
        r.di_frames_bytes = dstride * batch_frames;
    }
    if (!r.s_down) {
        MLV_HIP(hipStreamCreateWithFlags(&r.s_run, hipStreamNonBlocking));
        MLV_HIP(hipStreamCreateWithFlags(&r.s_down, hipStreamNonBlocking));
        for (int k = 0; k < 2; k++) {
            MLV_HIP(hipEventCreateWithFlags(&r.fin_done[k], hipEventDisableTiming));
            MLV_HIP(hipEventCreateWithFlags(&r.down_done[k], hipEventDisableTiming));
        }
    }
    uint8_t *const *stage = r.stage;
    int rc = read_frames(r, first, batch_frames, stage[0], stride, io_threads);
    for (int f0 = 0, k = 0; rc == MLVFS_AMD_OK && f0 < count; f0 += batch_frames, k++) {
        const int n = std::min(batch_frames, count - f0), n_next = std::min(batch_frames, count - f0 - n), slot = k & 1;
        int rc_io = MLVFS_AMD_OK;
        std::thread io;                                             // the next batch is read while this one is on the GPU
        if (n_next > 0) io = std::thread([&, k, f0, n, n_next]() { rc_io = read_frames(r, first + f0 + n, n_next, stage[(k + 1) & 1], stride, io_threads); });
        // (no early returns in here: the reader thread of the next batch is running)
        auto ok = [&](hipError_t e) { if (e != hipSuccess && rc == MLVFS_AMD_OK) { set_error("mlv: %s", hipGetErrorString(e)); rc = MLVFS_AMD_ERR_HIP; } };
        ok(hipMemcpyAsync(r.d_di_packed, stage[slot], stride * n, hipMemcpyHostToDevice, r.s_run));
        if (rc == MLVFS_AMD_OK && k >= 2) ok(hipStreamWaitEvent(r.s_run, r.down_done[slot], 0));          // this slot's frames of two batches ago are out
        if (rc == MLVFS_AMD_OK) rc = mlvfs_amd_unpack_dev(&geom, r.d_di_packed, stride, r.d_di_frames[slot], dstride, n, r.s_run);
        if (rc == MLVFS_AMD_OK) rc = mlvfs_amd_cr2hdr20_batch_dev(&geom, r.d_di_frames[slot], dstride, n, interp_method, fullres, use_alias_map, chroma_smooth,
                                                                  results + f0, r.s_run);          // (returns with the stream drained)
        if (rc == MLVFS_AMD_OK) {
            ok(hipEventRecord(r.fin_done[slot], r.s_run));
            ok(hipStreamWaitEvent(r.s_down, r.fin_done[slot], 0));
            ok(hipMemcpy2DAsync((uint8_t *)h_out + (size_t)f0 * out_stride, out_stride, r.d_di_frames[slot], dstride, px_bytes, n, hipMemcpyDeviceToHost, r.s_down));
            ok(hipEventRecord(r.down_done[slot], r.s_down));
        }
        if (io.joinable()) io.join();
        if (rc == MLVFS_AMD_OK && rc_io != MLVFS_AMD_OK) {
            set_error("mlv: prefetch of frames %d..%d failed", first + f0 + n, first + f0 + n + n_next - 1);
            rc = rc_io;
        }
    }
    if (hipStreamSynchronize(r.s_down) != hipSuccess || hipStreamSynchronize(r.s_run) != hipSuccess) {
        set_error("mlv: copying finished frames to the host failed");
        if (rc == MLVFS_AMD_OK) rc = MLVFS_AMD_ERR_HIP;
    }
    return rc;
}

static size_t gif_data(const void *reader, uint8_t *output_buffer, off_t offset, size_t max_size);
size_t mlvfs_amd_mlv_gif_data(const void *reader, uint8_t *output_buffer, off_t offset, size_t max_size)
{
    try { return gif_data(reader, output_buffer, offset, max_size); }
    catch (const std::exception &e) { set_error("mlv: preview: %s", e.what()); return 0; }       // (allocation sized from the file)
}

static size_t gif_data(const void *reader, uint8_t *output_buffer, off_t offset, size_t max_size)
{
    if (!reader || !output_buffer || offset < 0) { set_error("mlv: null argument"); return 0; }
    const Reader &r = *(const Reader *)reader;
    frame_headers fh0;
    if (!frame_headers_of(r, 0, &fh0)) return 0;                                                  // gif.c:85
    const int frame_count = (int)r.vidf.size();
    const int xres = fh0.rawi_hdr.xRes, yres = fh0.rawi_hdr.yRes, bpp = fh0.rawi_hdr.raw_info.bits_per_pixel;
    const bool lj92 = (fh0.file_hdr.videoClass & CLASS_LJ92) && !(fh0.file_hdr.videoClass & CLASS_LZMA);
    const mlvfs_amd_geom_t geom{ xres, yres, lj92 ? 16 : bpp, (int32_t)fh0.rawi_hdr.raw_info.black_level, (int32_t)fh0.rawi_hdr.raw_info.white_level, 0, 0 };
    const size_t total = mlvfs_amd_gif_size(&fh0);
    std::vector<uint8_t> file(total);
    constexpr int NF = 10;                                                                        // gif.c:37
    int rc = MLVFS_AMD_OK;
    if (lj92) {
        // the frames' JPEG streams -> 16-bit frames in HBM (the GPU decoder) -> back to the host for the renderer's upload: ten frames
        std::vector<std::vector<uint8_t>> comp(NF);
        std::vector<const void *> ptr(NF);
        std::vector<size_t> len(NF);
        for (int k = 0; k < NF && rc == MLVFS_AMD_OK; k++) {
            Span sp;
            if (!payload_span(r, k * frame_count / NF, &sp, true)) return 0;
            comp[k].resize(sp.bytes);
            if (!read_at(sp.fd, comp[k].data(), sp.bytes, sp.off)) { set_error("mlv: short read"); return 0; }
            ptr[k] = comp[k].data() + 4; len[k] = sp.bytes - 4;
        }
        mlv::ThreadCtx *c = mlv::thread_ctx();
        if (!c) return 0;
        const size_t fbytes = (size_t)xres * yres * 2, dstride = (fbytes + 255) / 256 * 256;
        void *d = nullptr;
        if (hipMalloc(&d, dstride * NF) != hipSuccess) { set_error("mlv: out of device memory"); return 0; }
        std::vector<uint8_t> frames(dstride * NF);
        rc = mlvfs_amd_lj92_decode_dev(ptr.data(), len.data(), NF, xres, yres, d, dstride, c->stream);
        if (rc == MLVFS_AMD_OK && (hipMemcpyAsync(frames.data(), d, dstride * NF, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                                   hipStreamSynchronize(c->stream) != hipSuccess)) rc = MLVFS_AMD_ERR_HIP;
        (void)hipFree(d);
        if (rc == MLVFS_AMD_OK) rc = mlvfs_amd_gif_render(&geom, frames.data(), dstride, 0, NF, file.data());
    } else if (!(fh0.file_hdr.videoClass & CLASS_LZMA) && xres / 4 > 0 && yres / 4 > 0) {
        // An uncompressed clip: the preview picks ONE pixel of every 4x4 block (gif.c:197: flat index y * 4 * P + x * 4 + 1 with P =
        // xRes / 4 * 4), i.e. every fourth row -- only those pieces are read, a quarter of ten payloads (3584x1320: 21 MB instead of
        // 83 MB from the page cache and over the link; the preview 25 -> 8 ms, the reference's gif_get_data: 33 ms).  Piece y of a
        // frame: from the 16-bit word that holds its first pixel to the word behind the one that holds its last.
        const int ow = xres / 4, oh = yres / 4;
        const uint64_t P = (uint64_t)ow * 4;
        const size_t piece = ((((uint64_t)(ow - 1) * 4 * bpp + bpp + 15) / 16 + 2) * 2 + 15) / 16 * 16;
        const size_t stride = piece * (size_t)oh;
        std::unique_ptr<uint8_t[]> rows(new uint8_t[stride * NF]);
        for (int k = 0; k < NF && rc == MLVFS_AMD_OK; k++) {                                       // gif.c:160: frame k * count / 10
            Span sp;
            if (!payload_span(r, k * frame_count / NF, &sp)) return 0;
            for (int y = 0; y < oh; y++) {
                uint8_t *dst = rows.get() + (size_t)k * stride + (size_t)y * piece;
                const uint64_t at = ((((uint64_t)y * 4 * P + 1) * bpp) >> 4) * 2;                  // byte offset of the piece in the payload
                const size_t have = at < sp.bytes ? (size_t)std::min<uint64_t>(piece, sp.bytes - at) : 0;
                if (have && !read_at(sp.fd, dst, have, sp.off + at)) { set_error("mlv: short read"); return 0; }
                if (have < piece) memset(dst + have, 0, piece - have);                             // (the slack word behind a payload)
            }
        }
        if (rc == MLVFS_AMD_OK) rc = mlvfs_amd_gif_render(&geom, rows.get(), stride, 2, NF, file.data());
    } else {
        Span s0;
        if (!payload_span(r, 0, &s0)) return 0;
        const size_t stride = (s0.bytes + 2 + 15) / 16 * 16;
        std::vector<uint8_t> frames(stride * NF, 0);
        for (int k = 0; k < NF && rc == MLVFS_AMD_OK; k++)                                         // gif.c:160: frame k * count / 10
            rc = read_frames(r, k * frame_count / NF, 1, frames.data() + (size_t)k * stride, stride, 1);
        if (rc == MLVFS_AMD_OK) rc = mlvfs_amd_gif_render(&geom, frames.data(), stride, 1, NF, file.data());
    }
    if (rc != MLVFS_AMD_OK) return 0;
    if ((size_t)offset < total) memcpy(output_buffer, file.data() + offset, std::min(max_size, total - (size_t)offset));   // gif.c:215
    return max_size;
}

}  // extern "C"


// ---------------------------------------------------------------- gif.h: the preview's two calls (main.c:1018-1022, 1212)
// With them `gif.o` can leave MLVFS's link as well: the clip is opened by the library's own reader (index walk, LZMA and LJ92
// payloads included), the ten frames are picked and gamma-mapped on the GPU, the file is byte for byte the reference's.
extern "C" size_t gif_get_size(struct frame_headers *frame_headers) { return mlvfs_amd_gif_size(frame_headers); }      // gif.c:63-80

extern "C" size_t gif_get_data(const char *path, uint8_t *output_buffer, off_t offset, size_t max_size)               // gif.c:82-221
{
    if (!path || !output_buffer) return 0;
    void *r = mlvfs_amd_mlv_open(path, 0);
    if (!r) return 0;
    const size_t n = mlvfs_amd_mlv_gif_data(r, output_buffer, offset, max_size);
    mlvfs_amd_mlv_close(r);
    return n;
}
