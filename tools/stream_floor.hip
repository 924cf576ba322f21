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
    return 0;
}
