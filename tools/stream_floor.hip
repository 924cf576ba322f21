// tools/stream_floor.hip -- what HBM gives the frame path's traffic when nothing is computed: per 8-pixel item two 8-byte loads from the
// 14-bit stream (at the item's dword-aligned offset, like the loaders of k_frame*) and one 16-byte non-temporal store, items in
// linear order, 400 frames of 3584x1320.  Build: hipcc -O3 --offload-arch=gfx950 tools/stream_floor.hip -o build/stream_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>   // 0: read + write, 1: read only (sum kept), 2: write only
__global__ __launch_bounds__(256) void k_copy(const uint8_t *__restrict__ src, uint4 *__restrict__ dst, long long items, int per_thread)
{
    long long i0 = ((long long)blockIdx.x * per_thread) * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (int k = 0; k < per_thread; k++) {
        const long long i = i0 + (long long)k * 256;
        if (i >= items) break;
        uint4 v = make_uint4((uint32_t)i, 1, 2, 3);
        if (MODE != 2) {
            const uint8_t *p = src + ((i * 14) & ~3ll);
            const uint2 a = *(const uint2 *)p, b = *(const uint2 *)(p + 8);
            v = make_uint4(a.x, a.y, b.x, b.y);
        }
        if (MODE == 1) acc += v.x ^ v.y ^ v.z ^ v.w;
        else { const u32x4 vv = { v.x, v.y, v.z, v.w }; __builtin_nontemporal_store(vv, (u32x4 *)&dst[i]); }
    }
    if (MODE == 1 && acc == 0x12345678u) dst[0] = make_uint4(acc, 0, 0, 0);
}
// the streaming kernels' pattern: a wave owns a column of 62 items (64 lanes load: two halo items), walks down a segment of SEG row
// pairs, DEPTH row pairs of loads under way, stores what it loaded (two rows of 16 bytes per lane); tasks from a ticket counter
template <int DEPTH>
__global__ __launch_bounds__(256, 4) void k_walk(const uint8_t *__restrict__ src, uint4 *__restrict__ dst, int F, int W, int H, int seg, int *ticket, int order, int S = 62, int halo = 1)
{
    const int lane = threadIdx.x & 63, items = W / 8, cols = (items + S - 1) / S, rows = H / 2, segs = (rows + seg - 1) / seg;
    const int per_frame = cols * segs, ntasks = F * per_frame;
    const long long pitch = (long long)items * 14;
    for (;;) {
        int task = 0;
        if (lane == 0) task = atomicAdd(ticket, 1);
        task = __builtin_amdgcn_readfirstlane(task);
        if (task >= ntasks) break;
        const int f = task / per_frame, rem = task - f * per_frame;
        const int c = order ? rem % cols : rem / segs, sg = order ? rem / cols : rem % segs;
        const int g_true = c * S + lane - halo, g = min(max(g_true, 0), items - 1);
        const bool writes = lane >= halo && lane < S + halo && g_true < items;
        const uint8_t *fr = src + (long long)f * pitch * H;
        uint4 *out = dst + (long long)f * items * H;
        const int j0 = sg * seg, j1 = min(j0 + seg, rows);
        uint2 q[DEPTH][4];
        auto issue = [&](int r, uint2 (&d)[4]) {
            const int rr = min(max(r, 0), rows - 1);
            const uint8_t *p0 = fr + ((2ll * rr * pitch + 14ll * g) & ~3ll), *p1 = fr + (((2ll * rr + 1) * pitch + 14ll * g) & ~3ll);
            d[0] = *(const uint2 *)p0; d[1] = *(const uint2 *)(p0 + 8); d[2] = *(const uint2 *)p1; d[3] = *(const uint2 *)(p1 + 8);
        };
#pragma unroll
        for (int k = 0; k < DEPTH; k++) issue(j0 - 2 + k, q[k]);
        for (int r = j0 - 2; r <= j1 + 1; r += DEPTH) {
#pragma unroll
            for (int k = 0; k < DEPTH; k++) {
                const int rk = r + k;
                if (rk > j1 + 1) break;
                const u32x4 v0 = { q[k][0].x, q[k][0].y, q[k][1].x, q[k][1].y }, v1 = { q[k][2].x, q[k][2].y, q[k][3].x, q[k][3].y };
                if (rk + DEPTH <= j1 + 1) issue(rk + DEPTH, q[k]);
                const int jr = rk - 2;
                if (jr >= j0 && jr < j1 && writes) {
                    __builtin_nontemporal_store(v0, (u32x4 *)&out[(2ll * jr) * items + g]);
                    __builtin_nontemporal_store(v1, (u32x4 *)&out[(2ll * jr + 1) * items + g]);
                }
            }
        }
    }
}
// ... and with the NW waves of a workgroup on NW adjacent columns of the same segment, a barrier per row pair (SYNC) or none: the workgroup's
// rows are NW x 868 bytes of contiguous stream at (about) the same time
template <int NW, bool SYNC>
__global__ __launch_bounds__(64 * NW) void k_walk_wg(const uint8_t *__restrict__ src, uint4 *__restrict__ dst, int F, int W, int H, int seg, int *ticket)
{
    __shared__ int s_task;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, items = W / 8, cols = (items + 61) / 62, rows = H / 2, segs = (rows + seg - 1) / seg;
    const int cgs = (cols + NW - 1) / NW, per_frame = cgs * segs, ntasks = F * per_frame;
    const long long pitch = (long long)items * 14;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_task = atomicAdd(ticket, 1);
        __syncthreads();
        const int task = s_task;
        if (task >= ntasks) break;
        const int f = task / per_frame, rem = task - f * per_frame;
        const int cg = rem % cgs, sg = rem / cgs, c = cg * NW + wv;
        const int g_true = c * 62 + lane - 1, g = min(max(g_true, 0), items - 1);
        const bool writes = lane >= 1 && lane <= 62 && g_true < items && c < cols;
        const uint8_t *fr = src + (long long)f * pitch * H;
        uint4 *out = dst + (long long)f * items * H;
        const int j0 = sg * seg, j1 = min(j0 + seg, rows);
        uint2 q[2][4];
        auto issue = [&](int r, uint2 (&d)[4]) {
            const int rr = min(max(r, 0), rows - 1);
            const uint8_t *p0 = fr + ((2ll * rr * pitch + 14ll * g) & ~3ll), *p1 = fr + (((2ll * rr + 1) * pitch + 14ll * g) & ~3ll);
            d[0] = *(const uint2 *)p0; d[1] = *(const uint2 *)(p0 + 8); d[2] = *(const uint2 *)p1; d[3] = *(const uint2 *)(p1 + 8);
        };
        issue(j0 - 2, q[0]); issue(j0 - 1, q[1]);
        for (int r = j0 - 2; r <= j1 + 1; r += 2) {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int rk = r + k;
                if (rk > j1 + 1) break;
                const u32x4 v0 = { q[k][0].x, q[k][0].y, q[k][1].x, q[k][1].y }, v1 = { q[k][2].x, q[k][2].y, q[k][3].x, q[k][3].y };
                if (SYNC) __syncthreads();
                if (rk + 2 <= j1 + 1) issue(rk + 2, q[k]);
                const int jr = rk - 2;
                if (jr >= j0 && jr < j1 && writes) {
                    __builtin_nontemporal_store(v0, (u32x4 *)&out[(2ll * jr) * items + g]);
                    __builtin_nontemporal_store(v1, (u32x4 *)&out[(2ll * jr + 1) * items + g]);
                }
            }
        }
    }
}
int main(int argc, char **argv)
{
    const int F = argc > 1 ? atoi(argv[1]) : 400, W = 3584, H = 1320;
    const long long items = (long long)F * W * H / 8;
    uint8_t *src; uint4 *dst;
    CK(hipMalloc(&src, items * 14 + 64)); CK(hipMalloc(&dst, items * 16));
    CK(hipMemset(src, 0x5a, items * 14 + 64)); CK(hipMemset(dst, 0, items * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; mode++)
        for (int pt : { 1, 4, 16 }) {
            const long long per_wg = 256ll * pt;
            const int grid = (int)((items + per_wg - 1) / per_wg);
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipEventRecord(e0, 0));
                if (mode == 0) hipLaunchKernelGGL(k_copy<0>, dim3(grid), dim3(256), 0, 0, src, dst, items, pt);
                else if (mode == 1) hipLaunchKernelGGL(k_copy<1>, dim3(grid), dim3(256), 0, 0, src, dst, items, pt);
                else hipLaunchKernelGGL(k_copy<2>, dim3(grid), dim3(256), 0, 0, src, dst, items, pt);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double bytes = (mode != 2 ? items * 14.0 : 0) + (mode != 1 ? items * 16.0 : 0);
            printf("mode %d (%s) items/thread %2d: %.3f ms, %.2f us per frame, %.2f TB/s\n", mode, mode == 0 ? "read+write" : mode == 1 ? "read" : "write", pt, best,
                   best * 1000.0 / F, bytes / best / 1e9);
        }
    int *ticket; CK(hipMalloc(&ticket, 4));
    for (int order = 0; order < 2; order++)
        for (int depth : { 2, 4 })
            for (int wgs : { 4, 8 }) {
                float best = 1e9f;
                for (int rep = 0; rep < 4; rep++) {
                    CK(hipMemsetAsync(ticket, 0, 4, 0));
                    CK(hipEventRecord(e0, 0));
                    if (depth == 2) hipLaunchKernelGGL(k_walk<2>, dim3(256 * wgs), dim3(256), 0, 0, src, dst, F, W, H, 60, ticket, order);
                    else hipLaunchKernelGGL(k_walk<4>, dim3(256 * wgs), dim3(256), 0, 0, src, dst, F, W, H, 60, ticket, order);
                    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                printf("column walk, %s, %d row pairs under way, %d workgroups per CU: %.3f ms, %.2f us per frame, %.2f TB/s algorithmic\n",
                       order ? "columns on consecutive tickets" : "segments on consecutive tickets", depth, wgs, best, best * 1000.0 / F, items * 30.0 / best / 1e9);
            }
#define WALK_WG(NW, SYNC, WGS) do { \
        float best = 1e9f; \
        for (int rep = 0; rep < 4; rep++) { \
            CK(hipMemsetAsync(ticket, 0, 4, 0)); CK(hipEventRecord(e0, 0)); \
            hipLaunchKernelGGL((k_walk_wg<NW, SYNC>), dim3(256 * WGS), dim3(64 * NW), 0, 0, src, dst, F, W, H, 60, ticket); \
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); \
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } \
        printf("column walk, %d waves of a workgroup side by side, %s, %d workgroups per CU: %.3f ms, %.2f us per frame\n", NW, SYNC ? "barrier per row pair" : "no barrier", WGS, best, best * 1000.0 / F); } while (0)
    for (int S : { 62, 60, 58, 56, 52, 64 }) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipMemsetAsync(ticket, 0, 4, 0)); CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_walk<2>, dim3(256 * 4), dim3(256), 0, 0, src, dst, F, W, H, 60, ticket, 0, S, S == 64 ? 0 : 1);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("column walk, independent waves, columns of %d items (%d halo lanes): %.3f ms, %.2f us per frame\n", S, S == 64 ? 0 : 2, best, best * 1000.0 / F);
    }
    WALK_WG(4, false, 4); WALK_WG(4, true, 4); WALK_WG(8, false, 2); WALK_WG(8, true, 2); WALK_WG(8, true, 4); WALK_WG(2, true, 8);
    return 0;
}
