// dropin.cpp -- PART 1 of include/mlvfs_amd.h: the symbols MLVFS's main.c links
// against (dng.h / cs.h / stripes.h / histogram.h), working in place on HOST
// memory exactly like the reference objects they replace.  Each call stages the
// frame into HBM, runs the HIP kernels on the calling thread's stream and copies
// the result back before returning (synchronous, caller keeps ownership).
//
// Error behaviour: never abort, never errno; diagnostics on
// stderr; on a HIP failure the frame is left untouched and the function returns
// the reference's failure value.  There is deliberately no CPU fallback.
#include "clip.h"

#include <atomic>
#include <chrono>

#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

using namespace mlv;

namespace {

struct FrameView {
    int w, h, bpp, black, white, pan_x, pan_y, frame_size;
    uint64_t guid;
};

FrameView view_of(const struct frame_headers *fh)
{
    FrameView v;
    v.w = fh->rawi_hdr.xRes;
    v.h = fh->rawi_hdr.yRes;
    v.bpp = fh->rawi_hdr.raw_info.bits_per_pixel;
    v.black = fh->rawi_hdr.raw_info.black_level;
    v.white = fh->rawi_hdr.raw_info.white_level;
    v.frame_size = fh->rawi_hdr.raw_info.frame_size;
    v.pan_x = fh->vidf_hdr.panPosX;
    v.pan_y = fh->vidf_hdr.panPosY;
    v.guid = fh->file_hdr.fileGuid;
    return v;
}

Clip *make_clip(const FrameView &v, ThreadCtx *c)
{
    Clip *clip = new Clip;
    clip->g = Geom{ v.w, v.h, v.bpp, v.black, v.white };
    clip->pan_x = v.pan_x;
    clip->pan_y = v.pan_y;
    clip->device = c->dev->id;
    return clip;
}

// ---- host <-> device staging of one 16-bit frame on the thread's stream ------------------------------------------------
// process_frame (main.c:942-997) hands the SAME host buffer to up to five of these symbols in a row and does not touch it in
// between.  By default every call uploads the frame, works, and downloads it (75 MB over PCIe per 3584x1320 frame).  With
// MLVFS_AMD_RESIDENT=1 in the environment a call that finds, on its thread, the device copy the previous call left for the same
// host pointer and size works on that copy and skips the upload; every call still downloads what it changed before it returns,
// so host memory is always current.  The copy is only taken up in process_frame's own order -- unpack, focus pixels, bad
// pixels, chroma smooth, stripes --: a stage that is called again, or after a later one, uploads as usual (a new frame in a
// recycled buffer starts with the unpack, or with whatever stage the caller starts with).  A caller that writes to the buffer
// between two calls of one such sequence must not set the variable: as a safeguard 64 words spread over the buffer are compared
// with what they were when the previous call returned, and a difference makes the call upload -- a change that misses all 64
// goes unnoticed.
// (Copies to and from the caller's pageable memory are the runtime's: staging them through a page-locked buffer of the thread's
// own, chunk by chunk with the copying done by the calling thread, was measured and is slower -- 795 against 1093 frames/s
// from 16 threads.)
// FRAME BRACKET ("level 2"): between mlvfs_amd_frame_begin() and mlvfs_amd_frame_end() on one thread the stages do not download
// at all -- the frame crosses the link once in each direction, 17.7 MB instead of 37-75 -- and the host buffer is NOT current
// until mlvfs_amd_frame_end() (or mlvfs_amd_frame_sync(buffer)) returns.  process_frame brackets its stages with
// mlvfs_load_chunks (main.c:923) ... mlvfs_close_chunks (main.c:998) on the calling thread, and nothing of MLVFS reads the frame
// buffer in between except deflicker's hist_add (handled below): integration/mlvfs_amd_wrap.c, linked with
// -Wl,--wrap=mlvfs_load_chunks -Wl,--wrap=mlvfs_close_chunks, turns those two calls into the bracket with main.c byte for byte
// unchanged.  OUTSIDE a bracket every call completes before it returns: gif_get_data (gif.c:90-221) opens its chunks with
// load_chunks / close_chunks directly and reads the frame right after get_image_data (gif.c:164) -- it never sees a deferred unpack.
// (Round 2 selected this behaviour process-wide with MLVFS_AMD_RESIDENT=2, which was wrong for exactly that caller; the value is
// now read as 1.)  MLVFS_AMD_DEFER=0 in the environment makes the bracket calls do nothing.
enum { RANK_UNPACK = 0, RANK_PNOISE = 1, RANK_DUALISO = 2, RANK_FOCUS = 3, RANK_BAD = 4, RANK_CS = 5, RANK_STRIPES_READ = 6, RANK_STRIPES = 7 };      // process_frame's order (main.c:942-997: unpack, pattern noise :946-949, dual ISO :952-959, focus / bad pixels, chroma smoothing, stripes)

thread_local bool t_bracket = false;               // this thread is between mlvfs_amd_frame_begin and mlvfs_amd_frame_end

bool defer_enabled()
{
    static const bool on = [] {
        const char *e = getenv("MLVFS_AMD_DEFER");
        return !(e && e[0] == '0');
    }();
    return on;
}

int resident_level()
{
    static const int level = [] {
        const char *e = getenv("MLVFS_AMD_RESIDENT");
        if (e && e[0] == '2')
            fprintf(stderr, "mlvfs_amd: MLVFS_AMD_RESIDENT=2 is read as 1: deferred stages now need the frame bracket "
                            "(integration/mlvfs_amd_wrap.c, or mlvfs_amd_frame_begin / mlvfs_amd_frame_end)\n");
        return e && (e[0] == '1' || e[0] == '2') ? 1 : 0;
    }();
    return t_bracket ? 2 : level;
}
bool resident_mode() { return resident_level() >= 1; }

void sample_host(const void *host, size_t bytes, uint64_t (&sig)[ThreadCtx::RES_SAMPLES])
{
    const size_t words = bytes / 8;
    const uint8_t *b = (const uint8_t *)host;
    for (int k = 0; k < ThreadCtx::RES_SAMPLES; k++) {
        const size_t at = words ? (words - 1) * (size_t)k / (ThreadCtx::RES_SAMPLES - 1) : 0;
        uint64_t v = 0;
        if (words) memcpy(&v, b + at * 8, 8);
        sig[k] = v;
    }
}

void warn_unsynced(const void *host)
{
    static std::atomic<bool> said{ false };
    if (!said.exchange(true))
        fprintf(stderr, "mlvfs_amd: the frame at %p, processed inside a frame bracket, was never fetched (mlvfs_amd_frame_end / "
                        "mlvfs_amd_frame_sync): its host buffer holds stale pixels\n", host);
}

int download(ThreadCtx *c, void *host, const void *dev, size_t bytes);

// How a worker waits for its stream.  hipStreamSynchronize spins; with as many workers as the process has CPUs (libfuse's pool on a
// 16-CPU cgroup) the spinning workers take the CPUs the runtime's own threads need.  MLVFS_AMD_WAIT=block: the wait is an event
// created with hipEventBlockingSync, i.e. the thread sleeps until the interrupt.  (A/B: tools/dropin_bench_c.sh)
const int g_wait_mode = [] { const char *e = getenv("MLVFS_AMD_WAIT"); return e && !strcmp(e, "block") ? 1 : 0; }();
thread_local hipEvent_t t_wait_ev = nullptr;
hipError_t wait_stream(hipStream_t st)
{
    if (!g_wait_mode) return hipStreamSynchronize(st);
    if (!t_wait_ev) { const hipError_t e = hipEventCreateWithFlags(&t_wait_ev, hipEventDisableTiming | hipEventBlockingSync); if (e != hipSuccess) return e; }
    hipError_t e = hipEventRecord(t_wait_ev, st);
    if (e == hipSuccess) e = hipEventSynchronize(t_wait_ev);
    return e;
}

// FAILURE POLICY ("When the device fails"): there is no CPU path in this library.  dng_get_image_data is the one
// stage whose failure would leave the caller with a buffer nobody has written (process_frame mallocs it: main.c:931) -- the frame is
// then ZEROED (a black DNG, never the heap's old contents) and one line goes to stderr, like the reference's err_printf; a later
// stage that fails leaves the frame as the stage before it left it (abandon_stage).
void unfilled_frame(void *host, size_t bytes, const char *where)
{
    if (host && bytes) memset(host, 0, bytes);
    fprintf(stderr, "mlvfs_amd: %s failed (%s): the frame is served black\n", where, mlvfs_amd_last_error());
}

// ---- frame bracket: recorded stages, one fused launch ------------------------------------------------------------------------
// Inside a bracket nothing has to be in host memory before mlvfs_amd_frame_end, so nothing has to RUN before it either: the unpack,
// the bad-pixel repair (once the clip's map is cached), the chroma smoothing and the stripe correction that process_frame asks
// for are recorded, and run as ONE launch of the fused kernel on the packed payload (what mlvfs_amd_process_frames_dev does for a
// batch) when the frame is fetched -- or earlier, when a call arrives that is not the next stage of process_frame's order or
// needs the pixels (first frame of a clip: bad-pixel detection, stripe histogram; focus-pixel maps; pattern noise; dual ISO).
std::atomic<long long> g_lazy_fused{ 0 }, g_lazy_early{ 0 };
// test hook (mlvfs_amd_test_fail_next): the calling thread's next fused launch of a bracketed frame [1] / next frame download [2]
// reports a HIP error without touching the device -- how tests/test_failure_policy.py shows what MLVFS serves when the device is lost
thread_local int t_fail_next[3] = { 0, 0, 0 };
// where the wall time of a bracketed frame goes, per process, in ns (MLVFS_AMD_DROPIN_PROFILE=1; mlvfs_amd_dropin_profile):
// [0] inside dng_get_image_data, [1] of it the upload call, [2] of it the wait for the upload's end, [3] inside the recorded stage
// calls, [4] inside mlvfs_amd_frame_end, [5] of it the fused launch's calls, [6] of it the download call + the wait for it, [7] frames
std::atomic<long long> g_prof[8] = { { 0 }, { 0 }, { 0 }, { 0 }, { 0 }, { 0 }, { 0 }, { 0 } };
const bool g_prof_on = [] { const char *e = getenv("MLVFS_AMD_DROPIN_PROFILE"); return e && e[0] == '1'; }();
struct ProfScope {
    int slot; std::chrono::steady_clock::time_point t0;
    explicit ProfScope(int s) : slot(s) { if (g_prof_on) t0 = std::chrono::steady_clock::now(); }
    ~ProfScope() { if (g_prof_on) g_prof[slot] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};
// what the drop-in stages moved over the link, process-wide: {uploads, downloads, bytes up, bytes down} (mlvfs_amd_dropin_transfers)
std::atomic<long long> g_xfer[4] = { { 0 }, { 0 }, { 0 }, { 0 } };

int lazy_run(ThreadCtx *c, bool at_sync)
{
    LazyFrame &z = c->lazy;
    if (!z.active) return MLVFS_AMD_OK;
    ProfScope prof_launch(5);
    z.active = false;
    (at_sync ? g_lazy_fused : g_lazy_early)++;
    std::shared_ptr<Clip> pix;
    pix.swap(z.pix);
    const Geom g{ z.w, z.h, 14, z.black, z.white };
    const bool patch = pix && pix->n_entries > 0;
    int rc = c->ensure_res((z.bytes + 15) / 16 * 16);
    // At the fetch, a frame buffer that comes from mlvfs_amd_host_alloc (page-locked, mapped: integration/mlvfs_amd_wrap_alloc.c
    // makes process_frame's malloc one) is written by the fused kernel itself, over the link: no download, no copy engine.  Sixteen
    // workers' downloads queue up behind each other on the engines (1.6-2.5 ms of waiting per frame for 0.19 ms of transfer); the
    // kernel's stores do not: 2 420 -> 3 618 fps from 16 threads, 1 550 -> 1 615 from one (tools/dropin_bench_c.sh, round 4).
    // MLVFS_AMD_ZEROCOPY=0 switches it off.
    static const bool zc_on = [] { const char *e = getenv("MLVFS_AMD_ZEROCOPY"); return !(e && e[0] == '0'); }();
    const bool zero_copy = zc_on && at_sync && ((uintptr_t)z.host % 16) == 0 && mlvfs_amd_host_owns(z.host, z.bytes) != 0;
    void *const d_out = zero_copy ? z.host : c->d_res[0];
    if (!rc && t_fail_next[1]) { t_fail_next[1] = 0; set_error("injected failure of the fused launch (test hook)"); rc = MLVFS_AMD_ERR_HIP; }
    if (!rc) {
        if (!patch && z.cs == 0 && !z.stripes)
            rc = launch_unpack(c->d_a, 0, d_out, 0, 0, (uint32_t)(z.bytes / 2), 14, 1, c->stream);
        else {
            PatchView pv{};
            if (patch) {
                const int geo = frame_geo_of(z.cs);
                rc = c->ensure_patch(pix->patch_buffer_bytes(1));
                if (!rc)
                    rc = launch_pixfix_for_frame_kernel(true, c->d_a, 0, z.w, z.h, z.black, pix->d_entries, pix->d_level_off, pix->n_levels,
                                                        pix->n_level0, pix->n_entries, c->d_patch, pix->d_tile_rec[geo], pix->n_rec[geo],
                                                        Clip::cells_of(c->d_patch, pix->n_entries, 1), 1, c->dev->luts, c->stream);
                pv = pix->patch_view(c->d_patch, 1, geo);
            }
            if (!rc)
                rc = launch_frame(c->dev, g, true, c->d_a, 0, d_out, 0, 1, z.cs, patch ? &pv : nullptr, z.stripes, z.coef, c->stream);
        }
    }
    if (!rc && zero_copy) {                        // the frame is in the caller's buffer once the stream has drained; no device copy is kept
        ProfScope prof_down(6);
        if (wait_stream(c->stream) != hipSuccess) { set_error("fused launch (zero copy): stream wait failed"); rc = MLVFS_AMD_ERR_HIP; }
        c->res_host = nullptr; c->res_dirty = false;
        return rc;
    }
    if (rc) {
        // A recorded frame that had to run EARLY (a stage that cannot join the launch: pattern noise, dual ISO, a focus map, a clip's
        // first frame) and failed: the host buffer is still what process_frame's malloc returned, and the stages that follow would
        // take it for the frame -- it is served black like a failed unpack.  At the fetch the caller does the same.
        if (!at_sync) unfilled_frame(z.host, z.bytes, "a recorded frame's launch");
        c->res_host = nullptr; c->res_dirty = false;
        return rc;
    }
    c->res_cur = 0;
    c->res_rank = z.rank;
    c->res_host = z.host;
    c->res_bytes = z.bytes;
    c->res_dirty = true;                           // a resident copy newer than the host buffer, like any deferred stage's
    return MLVFS_AMD_OK;
}

// may this call be recorded instead of run?  (the next stage, in process_frame's order, of the frame whose payload is waiting)
bool lazy_next(ThreadCtx *c, const void *host, size_t bytes, int rank)
{
    const LazyFrame &z = c->lazy;
    return z.active && z.host == host && z.bytes == bytes && rank > z.rank;
}

// host -> device on the thread's stream.  (Copies of the sixteen threads through ONE upload and ONE download stream per device,
// for page-locked buffers, were tried: no faster at 16 threads -- 1 200 frames/s, the level a bare copy loop of 16 threads
// reaches 1 840 at, tools/zerocopy_probe.hip -- and slower without MLVFS_AMD_RESIDENT.)
int upload(ThreadCtx *c, void *dev, const void *host, size_t bytes)
{
    ProfScope prof_up(1);
    g_xfer[0]++; g_xfer[2] += (long long)bytes;
    MLV_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, c->stream));
    return MLVFS_AMD_OK;
}

// device buffer that holds the frame at `host`: the resident copy, or a fresh upload
int stage_frame(ThreadCtx *c, const void *host, size_t bytes, int rank, void **d_cur, void **d_other)
{
    int rc = lazy_run(c, false);                   // recorded stages this call cannot join: run them now
    if (rc) return rc;
    rc = c->ensure_res((bytes + 15) / 16 * 16);
    if (rc) return rc;
    bool have = false;
    if (resident_mode() && c->res_host == host && c->res_bytes == bytes && rank > c->res_rank) {
        if (c->res_dirty) have = true;             // the host buffer is behind by design: nothing to compare
        else {
            uint64_t now[ThreadCtx::RES_SAMPLES];
            sample_host(host, bytes, now);
            have = memcmp(now, c->res_sig, sizeof now) == 0;
        }
    }
    if (!have) {
        if (c->res_dirty) {                        // a deferred result is pending and this call does not continue it
            if (c->res_host == host) {             // (waits: the upload below reads the host buffer when it is queued)
                rc = download(c, const_cast<void *>(host), c->d_res[c->res_cur], c->res_bytes);
                if (rc) return rc;
            } else warn_unsynced(c->res_host);
        }
        c->res_dirty = false;
        c->res_host = nullptr;
        c->res_cur = 0;
        rc = upload(c, c->d_res[0], host, bytes);
        if (rc) return rc;
    }
    *d_cur = c->d_res[c->res_cur];
    if (d_other) *d_other = c->d_res[c->res_cur ^ 1];
    return MLVFS_AMD_OK;
}

// the frame at `host` now equals device buffer `which` (call after the host copy is complete)
void commit_frame(ThreadCtx *c, const void *host, size_t bytes, int rank, int which)
{
    c->res_cur = which;
    c->res_rank = rank;
    c->res_dirty = false;
    if (!resident_mode()) { c->res_host = nullptr; return; }
    c->res_host = host;
    c->res_bytes = bytes;
    sample_host(host, bytes, c->res_sig);
}

int download(ThreadCtx *c, void *host, const void *dev, size_t bytes)
{
    if (t_fail_next[2]) { t_fail_next[2] = 0; set_error("injected failure of the frame's download (test hook)"); return MLVFS_AMD_ERR_HIP; }
    ProfScope prof_down(6);
    g_xfer[1]++; g_xfer[3] += (long long)bytes;
    MLV_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    MLV_HIP(wait_stream(c->stream));
    return MLVFS_AMD_OK;
}

// A stage is done with the frame in device buffer `which`: download it (all of it, or -- pixel repairs -- the n_patched
// entries of the thread's patch list) and remember the copy; inside a frame bracket: remember it as newer than the host buffer
int download_patches(ThreadCtx *c, uint16_t *image, size_t npix, int n_entries);
int finish_frame(ThreadCtx *c, void *host, size_t bytes, int rank, int which, int n_patched = -1)
{
    if (resident_level() == 2) {
        c->res_cur = which;
        c->res_rank = rank;
        c->res_host = host;
        c->res_bytes = bytes;
        c->res_dirty = true;
        return MLVFS_AMD_OK;
    }
    int rc = MLVFS_AMD_OK;
    if (n_patched < 0) rc = download(c, host, c->d_res[which], bytes);
    else if (n_patched > 0) rc = download_patches(c, (uint16_t *)host, bytes / 2, n_patched);
    else if (hipStreamSynchronize(c->stream) != hipSuccess) rc = MLVFS_AMD_ERR_HIP;
    if (rc) return rc;
    commit_frame(c, host, bytes, rank, which);
    return MLVFS_AMD_OK;
}

// A stage failed after stage_frame handed it device buffer `which`: that buffer still holds what the previous stage left.  Inside
// a frame bracket it may be the only up-to-date copy of the frame (`was_dirty`): the host buffer gets it before the copy is
// forgotten, so the failed stage is skipped and nothing earlier is lost.
void abandon_stage(ThreadCtx *c, void *host, size_t bytes, int which, bool was_dirty)
{
    // (the only up-to-date copy could not be brought back either: black, not the caller's old bytes)
    if (was_dirty && download(c, host, c->d_res[which], bytes) != MLVFS_AMD_OK) unfilled_frame(host, bytes, "the frame's download after a failed stage");
    c->res_dirty = false;
    c->res_host = nullptr;
}

// pixel repairs: the kernel has left {position, value} per map entry in the thread's patch list (position -1: not the final
// value of its pixel) and applied them to the device frame; the host frame gets the same few pixels instead of the whole frame
int download_patches(ThreadCtx *c, uint16_t *image, size_t npix, int n_entries)
{
    if (n_entries <= 0) return MLVFS_AMD_OK;
    std::vector<int32_t> pl((size_t)n_entries * 2);
    MLV_HIP(hipMemcpyAsync(pl.data(), c->d_patch, pl.size() * 4, hipMemcpyDeviceToHost, c->stream));
    MLV_HIP(hipStreamSynchronize(c->stream));
    for (int m = 0; m < n_entries; m++) {
        const int pos = pl[2 * m];
        if (pos >= 0 && (size_t)pos < npix) image[pos] = (uint16_t)pl[2 * m + 1];
    }
    return MLVFS_AMD_OK;
}

// ---- per-clip caches of the reference (cs.c:215-217, 333-334) ---------------
// A map is the list of sensor coordinates; what the kernels need (dependency levels, per-tile lists: a Clip) is DERIVED from it
// per (device, crop, dual-ISO mode) and never changed afterwards, so that the worker threads that serve frames of one clip
// can share it: each thread launches on its own stream with its own patch buffer (ThreadCtx), the derived Clip is read-only.
// shared_ptr: a slot recycled (or free_focus_pixel_maps) while another thread still holds the Clip must not free it.
using ClipRef = std::shared_ptr<Clip>;

std::string derived_key(const FrameView &v, const ThreadCtx *c, int dual_iso)
{
    char key[128];
    snprintf(key, sizeof key, "%d:%dx%d:%d,%d:%d", c->dev->id, v.w, v.h, v.pan_x, v.pan_y, dual_iso);
    return key;
}

struct BadMap {
    uint64_t guid = 0;
    int aggressive = 0;
    int w = 0, h = 0;
    bool valid = false;
    std::vector<int32_t> xy;                 // sensor coordinates (crop offsets included), list order = application order
    std::map<std::string, ClipRef> clips;
};
constexpr int BAD_PIXEL_MAP_COUNT = 8;
BadMap g_bad_maps[BAD_PIXEL_MAP_COUNT];
int g_bad_next = 0;
std::mutex g_bad_mutex;

struct FocusMap {
    uint32_t camera;
    int raw_w, raw_h;
    std::vector<int32_t> xy;             // as read from the .fpm file
    std::map<std::string, ClipRef> clips; // per device / frame geometry / crop / mode
};
std::vector<FocusMap *> g_focus_maps;
std::mutex g_focus_mutex;

}  // namespace

// The full dual-ISO conversion (dualiso.cpp) and fix_pattern_noise (patternnoise.cpp) as stages like the others: they come right
// behind the unpack in process_frame's order and work in place on the 16-bit frame -- on the copy the unpack left on the device
// when there is one (inside a frame bracket nothing crosses the link for them; until the end of round 3 they downloaded the
// unpacked frame, uploaded it again and downloaded their result).  d_other: the second frame buffer, for a stage that works out of
// place (it then ends with `which ^ 1`).  changed = false: the stage left the frame as it was.
int mlv::inplace_stage_begin(ThreadCtx *c, InplaceStage st, void *host, size_t bytes, void **d_frame, int *which, bool *was_dirty, void **d_other)
{
    const int rc = stage_frame(c, host, bytes, st == STAGE_DUALISO ? RANK_DUALISO : RANK_PNOISE, d_frame, d_other);
    *which = c->res_cur;
    *was_dirty = c->res_dirty;
    return rc;
}
void mlv::inplace_stage_end(ThreadCtx *c, InplaceStage st, void *host, size_t bytes, int which, bool was_dirty, bool done, bool changed)
{
    if (!done) abandon_stage(c, host, bytes, which, was_dirty);
    else if (finish_frame(c, host, bytes, st == STAGE_DUALISO ? RANK_DUALISO : RANK_PNOISE, which, changed ? -1 : 0))
        abandon_stage(c, host, bytes, which, false);
}

int mlv::drop_resident(ThreadCtx *c, void *host)
{
    int rc = lazy_run(c, false);
    if (rc) return rc;
    if (c->res_dirty) {
        if (c->res_host == host) rc = download(c, host, c->d_res[c->res_cur], c->res_bytes);
        else warn_unsynced(c->res_host);
    }
    c->res_dirty = false;
    c->res_host = nullptr;
    return rc;
}

// everything this thread has deferred reaches its host buffer: recorded stages run (one fused launch), a device copy newer than
// the host buffer is downloaded.  The copy stays known as the buffer's mirror, so later stages may still take it up.
int mlv::flush_pending(ThreadCtx *c)
{
    if (c->lazy.active) {
        // (inside a bracket the host buffer has not been written at all yet: a failure here must not hand the caller its malloc'ed bytes)
        void *lazy_host = c->lazy.host;
        const size_t lazy_bytes = c->lazy.bytes;
        int rc = lazy_run(c, true);
        if (rc) { unfilled_frame(lazy_host, lazy_bytes, "the frame's fused launch"); return rc; }
    }
    if (!c->res_dirty) return MLVFS_AMD_OK;
    void *host = const_cast<void *>(c->res_host);
    const int rc = download(c, host, c->d_res[c->res_cur], c->res_bytes);
    c->res_dirty = false;
    if (rc) { c->res_host = nullptr; unfilled_frame(host, c->res_bytes, "the frame's download"); return rc; }
    sample_host(host, c->res_bytes, c->res_sig);
    return MLVFS_AMD_OK;
}

extern "C" {

// Frame bracket (see the top of this file).  mlvfs_amd_frame_begin: from here on this thread's drop-in stages are recorded or
// leave their result on the GPU.  Costs nothing and touches no GPU: mlv_get_frame_headers (main.c:434,555) brackets a mere header
// walk with the same two calls.  mlvfs_amd_frame_end: everything this thread deferred reaches its host buffer (recorded stages
// run as one fused launch, one download), then calls complete at once again.  0 = the host buffer is current.
int mlvfs_amd_frame_begin(void)
{
    if (defer_enabled()) t_bracket = true;
    return MLVFS_AMD_OK;
}

int mlvfs_amd_frame_end(void)
{
    if (!t_bracket) return MLVFS_AMD_OK;
    t_bracket = false;
    ThreadCtx *c = thread_ctx_if_any();             // a bracket without pixel work never creates a stream
    if (!c || (!c->lazy.active && !c->res_dirty)) return MLVFS_AMD_OK;
    LibcRandGuard rand_guard;
    ProfScope prof_end(4);
    if (g_prof_on) g_prof[7]++;
    return flush_pending(c);
}

// Inside a bracket: fetch the frame at `image_data` now (what mlvfs_amd_frame_end does for whatever is pending).  A no-op outside
// a bracket and for a buffer nothing is pending for.  0 = the host buffer is current.
int mlvfs_amd_frame_sync(void *image_data)
{
    ThreadCtx *c = thread_ctx_if_any();
    if (!c) return MLVFS_AMD_OK;
    const bool mine = (c->lazy.active && c->lazy.host == image_data) || (c->res_dirty && c->res_host == image_data);
    if (!mine) return MLVFS_AMD_OK;
    LibcRandGuard rand_guard;
    return flush_pending(c);
}

// MLVFS_AMD_DROPIN_PROFILE=1: where the wall time of the bracketed frames went, in milliseconds summed over all threads: {inside
// dng_get_image_data, of it the upload call, of it the wait for the upload, inside the recorded stage calls, inside
// mlvfs_amd_frame_end, of it the fused launch's calls, of it download + wait, number of frames}
void mlvfs_amd_dropin_profile(double out[8]) { for (int i = 0; i < 8; i++) out[i] = i < 7 ? g_prof[i].load() * 1e-6 : (double)g_prof[i].load(); }

// whole-frame transfers of the drop-in stages since the process started: {uploads, downloads, bytes up, bytes down}
void mlvfs_amd_dropin_transfers(long long out[4]) { for (int i = 0; i < 4; i++) out[i] = g_xfer[i].load(); }

// test hook: see t_fail_next (1: the next fused launch of a bracketed frame, 2: the next frame download, on the calling thread)
void mlvfs_amd_test_fail_next(int what) { if (what == 1 || what == 2) t_fail_next[what] = 1; }

// how often this process ran recorded stages as one fused launch at the fetch [0], and how often earlier because a call could not
// be recorded [1] (for tests and tuning)
void mlvfs_amd_dropin_stats(long long out[2])
{
    out[0] = g_lazy_fused.load();
    out[1] = g_lazy_early.load();
}

// ============================================================== dng.h
size_t dng_get_header_size(void) { return 65536; }                                   // dng.c:797-800 (HEADER_SIZE)

size_t dng_get_image_size(struct frame_headers *fh)                                  // dng.c:879-882
{
    return (size_t)fh->rawi_hdr.xRes * fh->rawi_hdr.yRes * 2;
}

size_t dng_get_size(struct frame_headers *fh) { return dng_get_header_size() + dng_get_image_size(fh); }

size_t dng_get_image_data(struct frame_headers *fh, uint16_t *packed_bits, uint8_t *output_buffer, off_t offset,
                          size_t max_size)
{
    LibcRandGuard rand_guard;                      // HIP code may run: keep the caller's rand() stream out of its reach
    ProfScope prof_unpack(0);
    const int bpp = fh->rawi_hdr.raw_info.bits_per_pixel;
    // window arithmetic of dng.c:815-826
    const uint32_t first_px = (uint32_t)(offset > 0 ? offset : 0) / 2;
    const size_t lead = offset < 0 ? (size_t)(-offset) : 0;
    if (max_size <= lead) return max_size;
    const size_t out_bytes = max_size - lead;
    const uint32_t npix = (uint32_t)(out_bytes / 2);
    if (npix == 0) return max_size;
    // (a failure from here on zeroes what was asked for: see unfilled_frame)
    uint8_t *const want = output_buffer + lead + offset % 2;
    auto fail = [&]() -> size_t { unfilled_frame(want, out_bytes / 2 * 2, "dng_get_image_data"); return 0; };
    if (bpp < 1 || bpp > 16) { set_error("dng_get_image_data: unsupported bits_per_pixel %d", bpp); return fail(); }
    ThreadCtx *c = thread_ctx();
    if (!c) return fail();
    // the reference fetches, per pixel, the two 16-bit words that hold it; packed_bits
    // starts at the word of the first requested pixel
    const uint32_t first_word = first_px * (uint32_t)bpp / 16;
    const uint64_t last_bit = (uint64_t)(first_px + npix - 1) * bpp;
    const size_t words = (size_t)(last_bit / 16 - first_word) + 2;
    const size_t in_bytes = words * 2, out_b = (size_t)npix * 2;
    // a new frame starts: whatever this thread still holds for an earlier one was never fetched (a bracket that was not closed);
    // its host buffer may be gone by now, so it is dropped, not written
    if (c->res_dirty) warn_unsynced(c->res_host);
    if (c->lazy.active) { warn_unsynced(c->lazy.host); c->lazy.active = false; c->lazy.pix.reset(); }
    c->res_host = nullptr;
    c->res_dirty = false;
    if (c->ensure((in_bytes + 15) / 16 * 16, 0) || c->ensure_res((out_b + 15) / 16 * 16)) return fail();
    // (The same for the way in -- the payload fetched by a kernel instead of a copy engine -- was measured and is not there: 2 481
    // instead of 3 618 fps at 16 threads, 1 410 instead of 1 615 from one; reads over the link stall the whole launch.)
    if (upload(c, c->d_a, packed_bits, in_bytes)) {
        set_error("dng_get_image_data: upload failed");
        return fail();
    }
    // inside a frame bracket the call returns without waiting for the stream, but packed_bits is the caller's again on return: if it is
    // page-locked memory the copy above is still under way then (from pageable memory it is not), so its end gets an event
    const bool deferred = resident_level() == 2 && offset == 0 && out_b == dng_get_image_size(fh);
    if (deferred) {
        if (!c->ev_up && hipEventCreateWithFlags(&c->ev_up, hipEventDisableTiming | (g_wait_mode ? hipEventBlockingSync : 0)) != hipSuccess) return fail();
        if (hipEventRecord(c->ev_up, c->stream) != hipSuccess) return fail();
    }
    uint8_t *dst = output_buffer + lead + offset % 2;
    if (deferred && bpp == 14 && fh->rawi_hdr.raw_info.black_level >= 0) {
        // recorded, not run: the payload waits in d_a for the stages that follow (lazy_run)
        LazyFrame &z = c->lazy;
        z.active = true; z.host = dst; z.bytes = out_b;
        z.w = fh->rawi_hdr.xRes; z.h = fh->rawi_hdr.yRes;
        z.black = fh->rawi_hdr.raw_info.black_level; z.white = fh->rawi_hdr.raw_info.white_level;
        z.rank = RANK_UNPACK; z.pix.reset(); z.cs = 0; z.stripes = false;
        {
            ProfScope prof_wait(2);
            if (hipEventSynchronize(c->ev_up) != hipSuccess) { z.active = false; return fail(); }
        }
        return max_size;
    }
    if (launch_unpack(c->d_a, 0, c->d_res[0], 0, first_px, npix, bpp, 1, c->stream)) return fail();
    // process_frame's call -- the whole frame (main.c:942) --: the next stage on this buffer finds it on the device
    // (MLVFS_AMD_RESIDENT=1, frame bracket); a window of the frame is delivered at once in every mode
    if (offset == 0 && out_b == dng_get_image_size(fh)) {
        if (finish_frame(c, dst, out_b, RANK_UNPACK, 0)) return fail();
        if (deferred && hipEventSynchronize(c->ev_up) != hipSuccess) { c->res_dirty = false; c->res_host = nullptr; return fail(); }
    } else {
        if (download(c, dst, c->d_res[0], out_b)) return fail();
        commit_frame(c, dst, out_b, RANK_UNPACK, 0);
    }
    return max_size;
}

// ============================================================== cs.h
void chroma_smooth(struct frame_headers *fh, uint16_t *image_data, int method)
{
    ProfScope prof_stage(3);
    LibcRandGuard rand_guard;                      // HIP code may run: keep the caller's rand() stream out of its reach
    const FrameView v = view_of(fh);
    if (v.black > 16384) { fprintf(stderr, "Black level too large for processing\n"); return; }   // main.c:170-174
    if (method != 2 && method != 3 && method != 5) { fprintf(stderr, "Unsupported chroma smooth method\n"); return; }
    ThreadCtx *c = thread_ctx();
    if (!c) return;
    const size_t bytes = (size_t)v.w * v.h * 2;
    if (lazy_next(c, image_data, bytes, RANK_CS) && v.black == c->lazy.black) {
        c->lazy.cs = method;
        c->lazy.rank = RANK_CS;
        return;
    }
    void *d_in = nullptr, *d_out = nullptr;
    if (stage_frame(c, image_data, bytes, RANK_CS, &d_in, &d_out)) return;
    const int which = c->res_cur;
    const bool was_dirty = c->res_dirty;
    if (launch_frame(c->dev, Geom{ v.w, v.h, v.bpp, v.black, v.white }, false, d_in, bytes, d_out, bytes, 1, method,
                     nullptr, false, nullptr, c->stream)) {
        abandon_stage(c, image_data, bytes, which, was_dirty);
        return;
    }
    if (finish_frame(c, image_data, bytes, RANK_CS, which ^ 1)) abandon_stage(c, image_data, bytes, which, false);
}

// device-level forms (frame already in HBM at d_frame): shared with the dual-ISO path
}  // extern "C"

namespace mlv {

// the clip's bad-pixel map if it has been detected already (derived for this device / crop on first use); no GPU work on a frame
bool cached_bad_clip(struct frame_headers *fh, ThreadCtx *c, int aggressive, std::shared_ptr<Clip> *out)
{
    const FrameView v = view_of(fh);
    if (v.black > 16384 || !v.guid) return false;
    std::lock_guard<std::mutex> lk(g_bad_mutex);
    for (int i = 0; i < BAD_PIXEL_MAP_COUNT; i++) {
        BadMap &m = g_bad_maps[i];
        if (!(m.valid && v.guid == m.guid && aggressive == m.aggressive && m.w == v.w && m.h == v.h)) continue;
        ClipRef &slot = m.clips[derived_key(v, c, 0)];
        if (!slot) {
            ClipRef fresh(make_clip(v, c));
            if (fresh->set_pixel_map(m.xy.data(), m.xy.size() / 2, 0, 0)) { m.clips.erase(derived_key(v, c, 0)); return false; }
            slot = fresh;
        }
        *out = slot;
        return true;
    }
    return false;
}

int bad_pixels_device(struct frame_headers *fh, ThreadCtx *c, void *d_frame, int aggressive, int dual_iso, bool *changed, int *n_patched)
{
    if (n_patched) *n_patched = 0;
    const FrameView v = view_of(fh);
    if (changed) *changed = false;
    if (v.black > 16384) { fprintf(stderr, "Black level too large for processing\n"); return MLVFS_AMD_OK; }
    const size_t bytes = (size_t)v.w * v.h * 2;
    ClipRef clip;
    {
        // held over the detection: the other workers of this clip wait for the map instead of detecting it again
        std::lock_guard<std::mutex> lk(g_bad_mutex);
        BadMap *map = nullptr;
        for (int i = 0; i < BAD_PIXEL_MAP_COUNT; i++)                                // cs.c:233-239
            if (g_bad_maps[i].valid && v.guid && v.guid == g_bad_maps[i].guid && aggressive == g_bad_maps[i].aggressive &&
                g_bad_maps[i].w == v.w && g_bad_maps[i].h == v.h)
                map = &g_bad_maps[i];
        const std::string key = derived_key(v, c, dual_iso);
        if (!map) {
            map = &g_bad_maps[g_bad_next];
            g_bad_next = (g_bad_next + 1) % BAD_PIXEL_MAP_COUNT;
            *map = BadMap();
            clip.reset(make_clip(v, c));
            int rc = clip->detect_bad_pixels(d_frame, aggressive, dual_iso, c->stream);
            if (rc) return rc;
            map->guid = v.guid; map->aggressive = aggressive; map->w = v.w; map->h = v.h;
            map->xy = clip->xy;
            map->clips[key] = clip;
            map->valid = true;
            const int crop_x = (v.pan_x + 7) & ~7, crop_y = v.pan_y & ~1;
            const size_t n = map->xy.size() / 2;
            printf("%zu bad pixels found for %llx (crop: %d, %d):\n", n, (unsigned long long)v.guid, crop_x, crop_y);   // cs.c:307-311
            for (size_t m = 0; m < n; m++) printf("%d %d\n", map->xy[2 * m], map->xy[2 * m + 1]);
        } else {
            ClipRef &slot = map->clips[key];
            if (!slot) {                            // this crop / mode / device for the first time: derive, then never change
                ClipRef fresh(make_clip(v, c));
                int rc = fresh->set_pixel_map(map->xy.data(), map->xy.size() / 2, 0, dual_iso);
                if (rc) { map->clips.erase(key); return rc; }
                slot = fresh;
            }
            clip = slot;
        }
    }
    if (clip->n_entries == 0) return MLVFS_AMD_OK;                                   // nothing to repair
    int rc = clip->fix_pixels_shared(d_frame, bytes, v.black, c);
    if (rc == MLVFS_AMD_OK && changed) *changed = true;
    if (rc == MLVFS_AMD_OK && n_patched) *n_patched = clip->n_entries;
    return rc;
}

static FocusMap *load_focus_map(uint32_t camera, int raw_w, int raw_h)               // cs.c:356-401
{
    FocusMap *m = new FocusMap;
    m->camera = camera; m->raw_w = raw_w; m->raw_h = raw_h;
    g_focus_maps.push_back(m);
    char filename[1024];
    snprintf(filename, sizeof filename, "%x_%ix%i.fpm", camera, raw_w, raw_h);
    FILE *f = fopen(filename, "r+");                                                 // relative to the CWD, like the reference
    if (!f) return m;
    printf("Loading focus pixel map '%s'...\n", filename);
    // The reference reads pairs with fscanf("%i %i") until EOF (cs.c:380-392): %i is strtol with base 0 -- decimal, 0x.., 0.. --
    // behind any white space.  The same on the whole file in memory (151 200 lines: 12 ms through fscanf, 3 through strtol); a
    // token that is no number ends the list (the reference would spin on it).
    std::vector<char> text;
    char chunk[1 << 16];
    for (size_t got; (got = fread(chunk, 1, sizeof chunk, f)) > 0;) text.insert(text.end(), chunk, chunk + got);
    if (ferror(f)) fprintf(stderr, "file error: %s\n", strerror(errno));
    fclose(f);
    text.push_back('\0');
    for (char &ch : text) if (ch == '\0' && &ch != &text.back()) ch = ' ';          // (an embedded NUL would end strtol's view of the file)
    const char *p = text.data();
    for (;;) {
        char *e = nullptr;
        const long x = strtol(p, &e, 0);
        if (e == p) break;
        p = e;
        const long y = strtol(p, &e, 0);
        if (e == p) break;
        p = e;
        m->xy.push_back((int)x); m->xy.push_back((int)y);
    }
    return m;
}

// the clip (pixel map) for this frame's camera / geometry, or null when no .fpm file exists
static ClipRef focus_clip_ref(struct frame_headers *fh, ThreadCtx *c, int dual_iso)
{
    const FrameView v = view_of(fh);
    const uint32_t camera = fh->idnt_hdr.cameraModel;
    const int raw_w = fh->rawi_hdr.raw_info.width, raw_h = fh->rawi_hdr.raw_info.height;
    std::lock_guard<std::mutex> lk(g_focus_mutex);
    FocusMap *fm = nullptr;
    for (FocusMap *m : g_focus_maps)
        if (m->camera == camera && m->raw_w == raw_w && m->raw_h == raw_h) fm = m;
    if (!fm) fm = load_focus_map(camera, raw_w, raw_h);
    if (fm->xy.empty()) return nullptr;                                              // no map for this camera: no-op
    if (v.black > 16384) { fprintf(stderr, "raw2ev LUT error\n"); return nullptr; }
    const std::string key = derived_key(v, c, dual_iso);
    ClipRef &slot = fm->clips[key];
    if (!slot) {
        ClipRef fresh(make_clip(v, c));
        if (fresh->set_pixel_map(fm->xy.data(), fm->xy.size() / 2, 1, dual_iso)) { fm->clips.erase(key); return nullptr; }
        slot = fresh;
    }
    return slot;
}

bool focus_map_applies(struct frame_headers *fh, ThreadCtx *c, int dual_iso)
{
    ClipRef clip = focus_clip_ref(fh, c, dual_iso);
    return clip && clip->n_entries != 0;
}

int focus_pixels_device(struct frame_headers *fh, ThreadCtx *c, void *d_frame, int dual_iso, bool *changed, int *n_patched)
{
    const FrameView v = view_of(fh);
    if (changed) *changed = false;
    if (n_patched) *n_patched = 0;
    ClipRef clip = focus_clip_ref(fh, c, dual_iso);
    if (!clip || clip->n_entries == 0) return MLVFS_AMD_OK;
    int rc = clip->fix_pixels_shared(d_frame, (size_t)v.w * v.h * 2, v.black, c);
    if (rc == MLVFS_AMD_OK && changed) *changed = true;
    if (rc == MLVFS_AMD_OK && n_patched) *n_patched = clip->n_entries;
    return rc;
}

}  // namespace mlv

extern "C" {

void fix_bad_pixels(struct frame_headers *fh, uint16_t *image_data, int aggressive, int dual_iso)
{
    ProfScope prof_stage(3);
    LibcRandGuard rand_guard;                      // HIP code may run: keep the caller's rand() stream out of its reach
    ThreadCtx *c = thread_ctx();
    if (!c) return;
    const size_t npix = (size_t)fh->rawi_hdr.xRes * fh->rawi_hdr.yRes, bytes = npix * 2;
    if (lazy_next(c, image_data, bytes, RANK_BAD) && dual_iso == 0) {
        ClipRef clip;
        if (cached_bad_clip(fh, c, aggressive, &clip)) {      // (the first frame of a clip detects: that needs the pixels)
            c->lazy.pix = clip;
            c->lazy.rank = RANK_BAD;
            return;
        }
    }
    void *d_in = nullptr;
    if (stage_frame(c, image_data, bytes, RANK_BAD, &d_in, nullptr)) return;
    const int which = c->res_cur;
    const bool was_dirty = c->res_dirty;
    int n_patched = 0;
    // (a repair that fails half way may have rewritten part of the device copy: the frame is then as good as the reference's
    // after a failed malloc in the middle of cs.c:220-331 -- partly repaired)
    if (bad_pixels_device(fh, c, d_in, aggressive, dual_iso, nullptr, &n_patched)) { abandon_stage(c, image_data, bytes, which, was_dirty); return; }
    if (finish_frame(c, image_data, bytes, RANK_BAD, which, n_patched)) abandon_stage(c, image_data, bytes, which, false);
}

void fix_focus_pixels(struct frame_headers *fh, uint16_t *image_data, int dual_iso)
{
    ProfScope prof_stage(3);
    LibcRandGuard rand_guard;                      // HIP code may run: keep the caller's rand() stream out of its reach
    ThreadCtx *c = nullptr;
    // no map file for this camera is the common case: decide before touching the GPU
    {
        ThreadCtx *probe = thread_ctx();
        if (!probe) return;
        c = probe;
    }
    if (!focus_map_applies(fh, c, dual_iso)) return;
    const size_t npix = (size_t)fh->rawi_hdr.xRes * fh->rawi_hdr.yRes, bytes = npix * 2;
    void *d_in = nullptr;
    if (stage_frame(c, image_data, bytes, RANK_FOCUS, &d_in, nullptr)) return;
    const int which = c->res_cur;
    const bool was_dirty = c->res_dirty;
    int n_patched = 0;
    if (focus_pixels_device(fh, c, d_in, dual_iso, nullptr, &n_patched)) { abandon_stage(c, image_data, bytes, which, was_dirty); return; }
    if (finish_frame(c, image_data, bytes, RANK_FOCUS, which, n_patched)) abandon_stage(c, image_data, bytes, which, false);
}

void free_focus_pixel_maps(void)                                                     // cs.c:403-418
{
    LibcRandGuard rand_guard;                      // HIP code may run: keep the caller's rand() stream out of its reach
    {
        std::lock_guard<std::mutex> lk(g_focus_mutex);
        for (FocusMap *m : g_focus_maps) delete m;               // derived clips: freed with their last user
        g_focus_maps.clear();
    }
    std::lock_guard<std::mutex> lk(g_bad_mutex);
    for (auto &bm : g_bad_maps) bm = BadMap();
}

// ============================================================== stripes.h
static struct stripes_correction *g_corrections = nullptr;
static std::mutex g_corr_mutex;

struct stripes_correction *stripes_get_correction(const char *mlv_filename)          // stripes.c:31-38
{
    std::lock_guard<std::mutex> lk(g_corr_mutex);
    for (struct stripes_correction *cur = g_corrections; cur; cur = cur->next)
        if (!strcmp(cur->mlv_filename, mlv_filename)) return cur;
    return nullptr;
}

struct stripes_correction *stripes_new_correction(const char *mlv_filename)          // stripes.c:40-69
{
    // the reference leaves the coefficients uninitialised; they start at 0 here
    // (a 0 coefficient means "leave this column alone", stripes.c:261)
    struct stripes_correction *n = (struct stripes_correction *)calloc(1, sizeof *n);
    if (!n) return nullptr;
    n->mlv_filename = (char *)malloc(strlen(mlv_filename) + 2);
    if (!n->mlv_filename) { free(n); return nullptr; }
    strcpy(n->mlv_filename, mlv_filename);
    std::lock_guard<std::mutex> lk(g_corr_mutex);
    if (!g_corrections) g_corrections = n;
    else {
        struct stripes_correction *cur = g_corrections;
        while (cur->next) cur = cur->next;
        cur->next = n;
    }
    return n;
}

void stripes_free_corrections(void)                                                  // stripes.c:71-83
{
    LibcRandGuard rand_guard;                      // HIP code may run: keep the caller's rand() stream out of its reach
    std::lock_guard<std::mutex> lk(g_corr_mutex);
    struct stripes_correction *cur = g_corrections;
    while (cur) {
        struct stripes_correction *next = cur->next;
        free(cur->mlv_filename);
        free(cur);
        cur = next;
    }
    g_corrections = nullptr;
}

void stripes_compute_correction(struct frame_headers *fh, struct stripes_correction *correction, uint16_t *image_data,
                                off_t offset, size_t size)
{
    LibcRandGuard rand_guard;                      // HIP code may run: keep the caller's rand() stream out of its reach
    (void)offset; (void)size;                       // the reference ignores both and walks xRes x yRes (stripes.c:153-156)
    if (!correction) return;
    const FrameView v = view_of(fh);
    ThreadCtx *c = thread_ctx();
    if (!c) return;
    const size_t bytes = (size_t)v.w * v.h * 2;
    void *d_in = nullptr;
    if (stage_frame(c, image_data, bytes, RANK_STRIPES_READ, &d_in, nullptr)) return;      // read only: a resident copy stays what it is
    Clip *clip = make_clip(v, c);
    memcpy(clip->coef, correction->coeffficients, sizeof clip->coef);
    if (clip->stripes_compute(d_in, v.frame_size, /*rand_mode=*/0, c->stream) == MLVFS_AMD_OK) {
        memcpy(correction->coeffficients, clip->coef, sizeof clip->coef);
        correction->correction_needed = clip->needed;
    }
    delete clip;
}

void stripes_apply_correction(struct frame_headers *fh, struct stripes_correction *correction, uint16_t *image_data,
                              off_t offset, size_t size)
{
    ProfScope prof_stage(3);
    LibcRandGuard rand_guard;                      // HIP code may run: keep the caller's rand() stream out of its reach
    if (!correction || !correction->correction_needed) return;                       // stripes.c:252-253
    const FrameView v = view_of(fh);
    if (v.w % 8 != 0 || size == 0) return;
    ThreadCtx *c = thread_ctx();
    if (!c) return;
    // pixel i uses coefficient (i + offset % 8) % 8: rotate the table instead
    int32_t coef[8];
    const int start = (int)(((offset % 8) + 8) % 8);
    for (int k = 0; k < 8; k++) coef[k] = correction->coeffficients[(k + start) % 8];
    const size_t padded = (size + 7) / 8 * 8, bytes = padded * 2;
    if (padded == size && start == 0 && lazy_next(c, image_data, bytes, RANK_STRIPES) && v.black == c->lazy.black && v.white == c->lazy.white) {
        memcpy(c->lazy.coef, coef, sizeof coef);
        c->lazy.stripes = true;
        c->lazy.rank = RANK_STRIPES;
        return;
    }
    void *d_in = nullptr;
    if (padded == size) {
        if (stage_frame(c, image_data, bytes, RANK_STRIPES, &d_in, nullptr)) return;
    } else {                                        // a window that is not a whole number of 8-pixel groups: zero-padded copy
        if (drop_resident(c, image_data) || c->ensure_res(bytes)) return;
        c->res_cur = 0;
        d_in = c->d_res[0];
        if (hipMemsetAsync(d_in, 0, bytes, c->stream) != hipSuccess) return;
        if (hipMemcpyAsync(d_in, image_data, size * 2, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
            set_error("stripes_apply_correction: upload failed");
            return;
        }
    }
    const int which = c->res_cur;
    const bool was_dirty = c->res_dirty;
    if (launch_stripes_apply(d_in, bytes, padded, v.w, v.black, v.white, coef, 1, c->stream)) {
        abandon_stage(c, image_data, padded == size ? bytes : 0, which, padded == size && was_dirty);
        return;
    }
    if (padded == size) { if (finish_frame(c, image_data, bytes, RANK_STRIPES, which)) abandon_stage(c, image_data, bytes, which, false); }
    else { (void)download(c, image_data, d_in, size * 2); c->res_host = nullptr; c->res_dirty = false; }
}

// ============================================================== histogram.h (host only)
struct histogram *hist_create(uint16_t white)                                        // histogram.c:33-47
{
    struct histogram *h = (struct histogram *)malloc(sizeof *h);
    if (!h) return nullptr;
    h->white = white;
    h->count = 0;
    h->data = (uint16_t *)calloc((size_t)white + 1, sizeof(uint16_t));
    return h;
}

void hist_add(struct histogram *h, uint16_t *data, uint32_t size, uint16_t skip)     // histogram.c:52-59
{
    // main.c:943 runs deflicker() -- hist_add on the frame buffer -- right after the unpack: inside a frame bracket the pixels
    // are still on the GPU then, and that is where they are counted (64 KB of counts cross the link instead of the frame, and
    // the 2.4 M-sample loop below does not run); anything that goes wrong there falls back to fetching the frame
    if (resident_level() == 2 && h && h->data && size > 0) {
        ThreadCtx *c = thread_ctx_if_any();
        const uint8_t *p = (const uint8_t *)data;
        const bool in_lazy = c && c->lazy.active && p >= (const uint8_t *)c->lazy.host && p + (size_t)size * 2 <= (const uint8_t *)c->lazy.host + c->lazy.bytes;
        const bool in_res = c && !c->lazy.active && c->res_dirty && p >= (const uint8_t *)c->res_host &&
                            p + (size_t)size * 2 <= (const uint8_t *)c->res_host + c->res_bytes;
        if (in_lazy || in_res) {
            const uint8_t *base = (const uint8_t *)(in_lazy ? c->lazy.host : c->res_host);
            const uint32_t step = (uint32_t)skip + 1, samples = (size + step - 1) / step;
            const size_t hbytes = sizeof(unsigned) * ((size_t)h->white + 1);
            bool counted = false;
            if (((p - base) & 1) == 0 && lazy_run(c, false) == MLVFS_AMD_OK && c->res_dirty && c->ensure(0, hbytes) == MLVFS_AMD_OK) {
                const uint16_t *d = (const uint16_t *)c->d_res[c->res_cur];
                std::vector<unsigned> cnt((size_t)h->white + 1);
                counted = launch_hist_add(d, (uint32_t)((p - base) / 2), step, samples, h->white, (unsigned *)c->d_b, c->stream) == MLVFS_AMD_OK &&
                          hipMemcpyAsync(cnt.data(), c->d_b, hbytes, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
                          hipStreamSynchronize(c->stream) == hipSuccess;
                if (counted) {
                    for (uint32_t i = 0; i <= h->white; i++) h->data[i] = (uint16_t)(h->data[i] + cnt[i]);      // 16-bit counters wrap
                    h->count += size / step;
                    return;
                }
            }
            (void)hipGetLastError();
        }
        // (a range that only starts inside the frame, an odd offset, a failure above: the frame comes to the host and is counted there)
        if (c && c->lazy.active && p >= (const uint8_t *)c->lazy.host && p < (const uint8_t *)c->lazy.host + c->lazy.bytes)
            (void)mlvfs_amd_frame_sync(c->lazy.host);
        else if (c && c->res_dirty && p >= (const uint8_t *)c->res_host && p < (const uint8_t *)c->res_host + c->res_bytes)
            (void)mlvfs_amd_frame_sync(const_cast<void *>(c->res_host));
    }
    const uint32_t step = (uint32_t)skip + 1;
    for (uint32_t i = 0; i < size; i += step) h->data[data[i] < h->white ? data[i] : h->white]++;   // 16-bit counters wrap
    h->count += size / step;
}

uint16_t hist_median(struct histogram *h)                                            // histogram.c:64-75
{
    const uint32_t middle = h->count / 2;
    uint32_t acc = 0;
    for (uint32_t i = 0; i <= h->white; i++) {
        acc += h->data[i];
        if (acc > middle) return (uint16_t)i;
    }
    return 0;
}

void hist_destroy(struct histogram *h) { if (h) { free(h->data); free(h); } }

}  // extern "C"
