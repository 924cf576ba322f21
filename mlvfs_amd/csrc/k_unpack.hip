// k_unpack.hip -- packed MLV payload -> 16-bit pixels (replaces the loop of
// dng_get_image_data_inline, mlvfs/dng.c:813-843).
//
// Format (mlvfs/raw.h:41-79): pixel i is bits [i*bpp, (i+1)*bpp) of an MSB-first
// bit stream stored as little-endian 16-bit words.
//
// HBM-bound: 14/8 B read + 2 B written per pixel = 3.75 B/px.
//   k_unpack_x16<14 | 12 | 10> : one lane = 16 pixels = 7 / 6 / 5 coalesced dword loads
//                (always 4-byte aligned: 16 px * bpp bits) and two 128-bit stores.
//                Frames are batched in grid.y.
//   k_unpack_generic : any bpp in 1..16, any length; one lane = one pixel.
#include "clip.h"

namespace mlv {

// Swap the two 16-bit words of a little-endian dword: gives 32 stream bits in
// MSB-first order.
__device__ __forceinline__ uint32_t stream_word(uint32_t le_dword) { return (le_dword << 16) | (le_dword >> 16); }

// 16 pixels of BPP bits (BPP even) from BPP / 2 MSB-first 32-bit stream words
template <int BPP>
__device__ __forceinline__ void unpack_x16(const uint32_t (&s)[BPP / 2], uint32_t (&px)[16])
{
    constexpr uint32_t mask = (1u << BPP) - 1u;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int bit = BPP * k, wi = bit >> 5, sh = bit & 31;
        if (sh + BPP <= 32) {
            px[k] = (s[wi] >> (32 - BPP - sh)) & mask;
        } else {
            const uint64_t two = ((uint64_t)s[wi] << 32) | s[wi + 1];
            px[k] = (uint32_t)(two >> (64 - BPP - sh)) & mask;
        }
    }
}

// one lane = 16 pixels = BPP / 2 coalesced dword loads (always 4-byte aligned: 16 px * BPP bits) and two 128-bit stores:
// 14 bits, and the reduced depths of ML's raw video (12, 10), which took the pixel-per-lane kernel until the end of round 3
template <int BPP>
__global__ __launch_bounds__(256) void k_unpack_x16(const uint8_t *__restrict__ packed, size_t packed_stride,
                                                    uint8_t *__restrict__ out, size_t out_stride, uint32_t groups)
{
    constexpr int NW = BPP / 2;
    const uint32_t *src = (const uint32_t *)(packed + (size_t)blockIdx.y * packed_stride);
    uint4 *dst = (uint4 *)(out + (size_t)blockIdx.y * out_stride);
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += gridDim.x * blockDim.x) {
        uint32_t s[NW], px[16];
#pragma unroll
        for (int i = 0; i < NW; i++) s[i] = stream_word(src[(size_t)g * NW + i]);
        unpack_x16<BPP>(s, px);
        uint4 lo, hi;
        lo.x = px[0] | (px[1] << 16);   lo.y = px[2] | (px[3] << 16);
        lo.z = px[4] | (px[5] << 16);   lo.w = px[6] | (px[7] << 16);
        hi.x = px[8] | (px[9] << 16);   hi.y = px[10] | (px[11] << 16);
        hi.z = px[12] | (px[13] << 16); hi.w = px[14] | (px[15] << 16);
        dst[(size_t)g * 2] = lo;
        dst[(size_t)g * 2 + 1] = hi;
    }
}

__global__ __launch_bounds__(256) void k_unpack_generic(const uint8_t *__restrict__ packed, size_t packed_stride,
                                                        uint8_t *__restrict__ out, size_t out_stride,
                                                        uint32_t first_px, uint32_t npix, int bpp)
{
    const uint16_t *src = (const uint16_t *)(packed + (size_t)blockIdx.y * packed_stride);
    uint16_t *dst = (uint16_t *)(out + (size_t)blockIdx.y * out_stride);
    const uint32_t mask = (1u << bpp) - 1u;
    const uint32_t first_word = first_px * (uint32_t)bpp / 16;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < npix; k += gridDim.x * blockDim.x) {
        const uint32_t bit = (first_px + k) * (uint32_t)bpp;
        const uint32_t wi = bit / 16 - first_word, sh = bit % 16;
        const uint32_t two = ((uint32_t)src[wi] << 16) | src[wi + 1];
        dst[k] = (uint16_t)((two >> (32 - bpp - sh)) & mask);
    }
}

// host launcher: d_packed points at the word that holds the first requested pixel
int launch_unpack(const void *d_packed, size_t packed_stride, void *d_out, size_t out_stride, uint32_t first_px,
                  uint32_t npix, int bpp, int nframes, hipStream_t stream)
{
    if (npix == 0 || nframes <= 0) return MLVFS_AMD_OK;
    if (bpp < 1 || bpp > 16) { set_error("unsupported bits_per_pixel %d", bpp); return MLVFS_AMD_ERR_ARG; }
    const bool fast = (bpp == 14 || bpp == 12 || bpp == 10) && first_px == 0 && npix % 16 == 0 && ((uintptr_t)d_packed % 4 == 0) &&
                      ((uintptr_t)d_out % 16 == 0) && (nframes == 1 || (packed_stride % 4 == 0 && out_stride % 16 == 0));
    if (fast) {
        const uint32_t groups = npix / 16;
        dim3 grid((groups + 255) / 256, nframes);
        if (grid.x > 8192) grid.x = 8192;
        auto kern = bpp == 14 ? k_unpack_x16<14> : (bpp == 12 ? k_unpack_x16<12> : k_unpack_x16<10>);
        hipLaunchKernelGGL(kern, grid, dim3(256), 0, stream, (const uint8_t *)d_packed, packed_stride, (uint8_t *)d_out, out_stride, groups);
    } else {
        dim3 grid((npix + 255) / 256, nframes);
        if (grid.x > 16384) grid.x = 16384;
        hipLaunchKernelGGL(k_unpack_generic, grid, dim3(256), 0, stream, (const uint8_t *)d_packed, packed_stride,
                           (uint8_t *)d_out, out_stride, first_px, npix, bpp);
    }
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}


// the first launch of any kernel of this file loads the file's code object (HIP loads them lazily): the device context asks for a
// kernel's attributes when it is created, so that a clip's first frame does not pay for it (runtime.cpp: get_device)
void preload_k_unpack() { hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, (const void *)k_unpack_x16<14>); (void)hipGetLastError(); }

}  // namespace mlv
