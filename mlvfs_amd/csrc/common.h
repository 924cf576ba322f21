// common.h -- shared declarations of libmlvfs_amd.so (host + device).
//
// Device-side EV arithmetic.  The reference works in "EV space" through two
// host tables (mlvfs/main.c:128-196):
//     raw2ev[p]  = (int)(log2(p - black) * 32768)          (INT_MIN at p == black, 0 below)
//     ev2raw[ev] = (int)pow(2, (float)ev / 32768)           ev in [-10*32768, 14*32768)
// On the GPU both are served from two small 16-bit tables that are exact
// re-encodings of those host tables (checked entry by entry when the context is
// created, see luts.cpp):
//     T16[j - 8192] = raw2ev_lin[j] - 13*32768   for j in [8192, 16384)   (16 KiB)
//        raw2ev_lin[i] = T16[(i << (13 - e)) - 8192] + e*32768,  e = floor(log2 i)
//     U16[f]        = ev2raw[13*32768 + f]       for f in [0, 32768)       (64 KiB)
//        ev2raw[q*32768 + f] = U16[f] >> (13 - q)                 for q in [0, 14)
// T16 is staged into LDS by the stencil kernels; U16 is gathered from L2.
#pragma once
#include <cstdlib>

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <memory>

#include "../../include/mlvfs_amd.h"

#define MLV_EV_RES 32768
#define MLV_EV_MAX (14 * MLV_EV_RES - 1)

#define MLV_T16_N 8192
#define MLV_U16_N 32768

namespace mlv {

// ---------------------------------------------------------------- error plumbing
void set_error(const char *fmt, ...);
#define MLV_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            mlv::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return MLVFS_AMD_ERR_HIP;                                                   \
        }                                                                               \
    } while (0)

// ---------------------------------------------------------------- device state
struct DeviceLuts {
    const uint16_t *t16;   // [8192]   mantissa-normalised raw2ev (sparse kernels, global memory)
    const uint16_t *t16d;  // [16384]  direct raw2ev: raw2ev_lin[i] - (floor(log2 i) << 15) (LDS copy in k_frame)
    const uint16_t *u16;   // [32768]
};

struct Device {            // one per GPU, created once
    int id = -1;
    DeviceLuts luts{};
    int num_cu = 0;
};

// The reference draws its stripes dither from the process-global libc rand() stream (stripes.c:129-130) and nothing else in
// a CPU-only MLVFS touches that stream.  The HIP runtime does (measured: a plain C host that calls the drop-in symbols gets
// other stripe coefficients than the reference because rand() values disappear during runtime initialisation), so every
// drop-in entry point parks the application's generator state while HIP code may run and puts it back on return; the dither
// itself is drawn from the application's state.  Nests across threads (first in switches, last out restores).
struct LibcRandGuard {
    LibcRandGuard();
    ~LibcRandGuard();
    LibcRandGuard(const LibcRandGuard &) = delete;
    LibcRandGuard &operator=(const LibcRandGuard &) = delete;
    static void draw_mod1024(uint16_t *out, long long n);      // n values of rand() % 1024 from the APPLICATION's stream
    // The application's generator itself, while it is parked (i.e. inside a guard): its 31 words, oldest first (next value =
    // x[0] + x[28], output = next >> 1), if it is glibc's default TYPE_3 generator -- what rand() / srand() use unless the application
    // called initstate() with another size.  put_app_state stores the words back (after the library has produced the values in
    // between some faster way than by calling rand() millions of times).  false: not parked, or another generator type.
    static bool take_app_state(uint32_t x[31]);
    static bool put_app_state(const uint32_t x[31]);
};

struct Clip;
// Frame bracket (dropin.cpp): the stages process_frame has asked for so far on the frame whose packed payload sits in d_a.  They run as
// ONE launch of the fused kernel -- the batch path's -- when the frame is fetched (mlvfs_amd_frame_end / _sync) or when a call comes
// that cannot be recorded (dropin.cpp)
struct LazyFrame {
    bool active = false;
    void *host = nullptr;
    size_t bytes = 0;
    int w = 0, h = 0, black = 0, white = 0;
    int rank = 0;                        // last stage recorded (process_frame's order only)
    std::shared_ptr<Clip> pix;           // bad-pixel map to apply (shared, read-only)
    int cs = 0;
    bool stripes = false;
    int32_t coef[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
};

struct ThreadCtx {         // one per (host thread, device)
    Device *dev = nullptr;
    hipStream_t stream = nullptr;
    // grow-only staging buffers for the host-memory (drop-in) entry points
    void *d_a = nullptr, *d_b = nullptr;
    size_t cap_a = 0, cap_b = 0;
    void *h_pin = nullptr;
    size_t cap_pin = 0;
    void *d_patch = nullptr;             // this thread's patch list for pixel maps shared between threads (Clip::fix_pixels_shared)
    size_t cap_patch = 0;
    // The 16-bit frame the drop-in stages work on: two buffers (a stage reads one and writes the other, or works in place), and --
    // MLVFS_AMD_RESIDENT=1 -- which host buffer the current one mirrors, so that the next stage called on the same host pointer
    // need not upload it again (dropin.cpp)
    static constexpr int RES_SAMPLES = 64;
    void *d_res[2] = { nullptr, nullptr };
    size_t cap_res = 0;
    const void *res_host = nullptr;
    size_t res_bytes = 0;
    int res_cur = 0;
    int res_rank = 0;                    // which stage left the resident copy (dropin.cpp: stages only continue in process_frame's order)
    bool res_dirty = false;              // frame bracket: the resident copy is newer than the host buffer (mlvfs_amd_frame_end / _sync)
    LazyFrame lazy;
    hipEvent_t ev_up = nullptr;          // end of the upload of dng_get_image_data's input (dropin.cpp, frame bracket)
    uint64_t res_sig[RES_SAMPLES];
    int ensure_res(size_t bytes);
    int ensure(size_t need_a, size_t need_b);
    int ensure_patch(size_t need);
    ~ThreadCtx();
};

// what the library keeps per (device, stream) -- the fused kernel's ticket counters -- is dropped with the stream
// (k_frame.hip); call before destroying a stream the library created
void release_stream_state(int device, hipStream_t stream);

// returns nullptr (and sets the error string) on failure
ThreadCtx *thread_ctx();
// the calling thread's context if it has one already (never creates a stream, never touches the GPU)
ThreadCtx *thread_ctx_if_any();
// everything the thread has deferred inside a frame bracket reaches its host buffer (dropin.cpp)
int flush_pending(ThreadCtx *c);
// a symbol that reads or rewrites the host frame itself: bring the host copy up to date if a deferred result of this thread is
// pending for it (frame bracket), then forget the resident copy (dropin.cpp)
int drop_resident(ThreadCtx *c, void *host);
// the in-place stages that live outside dropin.cpp as stages of the drop-in sequence: the device frame to work on, then what
// became of it (dropin.cpp)
enum InplaceStage { STAGE_DUALISO, STAGE_PNOISE };
int inplace_stage_begin(ThreadCtx *c, InplaceStage st, void *host, size_t bytes, void **d_frame, int *which, bool *was_dirty, void **d_other = nullptr);
void inplace_stage_end(ThreadCtx *c, InplaceStage st, void *host, size_t bytes, int which, bool was_dirty, bool done, bool changed);

int bind_device(int device);

// host copies of the reference tables (built once; luts.cpp)
const int32_t *host_raw2ev_lin();   // [16384]  index = p - black
const int32_t *host_ev2raw();       // [24*32768] index 0 is ev = -10*32768
const uint16_t *host_t16();
const uint16_t *host_t16d();
const uint16_t *host_u16();
// code objects of the frame path, loaded when a device context is created (defined in the .hip files they belong to)
void preload_k_unpack();
void preload_k_pixfix();
void preload_k_stripes();
void preload_k_frame();
void preload_k_frame_p();
void preload_k_frame_s();

// The streaming kernels' (k_frame_s, k_frame_p5) columns: a wave's lanes 1 .. S write S items of a row, lanes 0 and S + 1 are the halo.
// S = 62.  MLVFS_AMD_KF_COLW=60 makes a column's output rows whole 64-byte pieces (columns of 62 items share a piece with their
// neighbour at every edge, written by two waves at different times): the bare traffic of 3584x1320 takes 3.61 instead of 3.94 us
// per frame that way (tools/stream_floor.hip), the kernels themselves gain nothing -- k_frame_s +-0, k_frame_p5 +3.5 % (7.5 instead
// of 7.25 columns of steps; profiles/r05/ab_colw.log).
static inline int frame_stream_colw()
{
    static const int colw = [] { const char *e = getenv("MLVFS_AMD_KF_COLW"); const int v = e ? atoi(e) : 0; return v >= 8 && v <= 62 ? v : 62; }();
    return colw;
}
static inline int frame_stream_cols(int w) { return (w / 8 + frame_stream_colw() - 1) / frame_stream_colw(); }
// A last column of at most 14 (30) items is folded: a wave takes four (two) of its segments side by side, one per group of 16 (32)
// lanes, each group with its own two halo lanes.  MLVFS_AMD_KF_FOLD=1: never (A/B).
static inline int frame_stream_fold(int w, int cols, int segs)
{
    static const int env_fold = [] { const char *e = getenv("MLVFS_AMD_KF_FOLD"); return e ? atoi(e) : 0; }();
    const int last_items = w / 8 - (cols - 1) * frame_stream_colw();
    if (segs < 2 || env_fold == 1) return 1;
    return last_items + 2 <= 16 ? 4 : last_items + 2 <= 32 ? 2 : 1;
}

int luts_ok();

struct Geom {
    int w, h, bpp, black, white;
};

// LZMA payloads (lzma.cpp): 0 = ok, else LzmaDecode's error code
int lzma_decode(const uint8_t props[5], const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap, size_t *dst_len);

// device API: the caller's stream; NULL is HIP's default (null) stream, which orders
// against everything else the caller enqueued
inline hipStream_t pick_stream(void *s, ThreadCtx *) { return (hipStream_t)s; }

// ---------------------------------------------------------------- device helpers
#if defined(__HIPCC__)

// wrap-around arithmetic: the reference's signed overflow is de-facto two's
// complement (SURVEY.md 8a notes); unsigned arithmetic makes that defined.
__device__ __forceinline__ int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
__device__ __forceinline__ int wsub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
__device__ __forceinline__ int wmul(int a, int b) { return (int)((unsigned)a * (unsigned)b); }
__device__ __forceinline__ int wabs(int a) { return a > 0 ? a : (int)(0u - (unsigned)a); }
// C's truncating /2 on a possibly negative int
__device__ __forceinline__ int half_trunc(int s) { return (s + (int)((unsigned)s >> 31)) >> 1; }

// raw2ev of a pixel value (table pointer may be LDS or global)
template <typename TablePtr>
__device__ __forceinline__ int ev_of_pixel(int p, int black, TablePtr t16)
{
    const int lin = p - black;
    if (lin <= 0 || lin >= 16384) return lin == 0 ? (int)0x80000000 : 0;
    const int e = 31 - __clz(lin);
    const int j = lin << (13 - e);
    return (int)t16[j - 8192] + (e << 15);
}

// ev2raw[clamp(ev)] + black, as the uint16 the reference stores
__device__ __forceinline__ uint16_t pixel_of_ev(int ev, int black, const uint16_t *__restrict__ u16)
{
    ev = ev < 0 ? 0 : (ev > MLV_EV_MAX ? MLV_EV_MAX : ev);
    const int q = ev >> 15, f = ev & 32767;
    return (uint16_t)((((int)u16[f]) >> (13 - q)) + black);
}

#endif  // __HIPCC__

}  // namespace mlv
