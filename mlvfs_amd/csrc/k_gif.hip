// k_gif.hip -- the animated GIF preview of a clip (second half): replaces mlvfs/gif.c:82-244.
//
// gif_get_data renders 10 frames of the clip at 1/4 x 1/4 size: output pixel (x, y) is gamma[p >> 4] of the ONE pixel
// p = image[y * 4 * (width * 4) + x * 4 + 1] (the green next to the cell's red; the row pitch is (xRes / 4) * 4, which is xRes
// only when xRes is a multiple of 4: kept), with gamma[i] = g * g / 255 / 2, g = (int)(log2f(i - (black >> 4)) * 255 / 10)
// above black and 0 below -- and wraps the bytes as "uncompressed GIF": 7-bit codes, a clear code in front of every 125 pixels.
//   device: k_gif_pixels -- pick and map the pixels of all 10 frames, from the packed payloads (no unpacked frame is ever
//           made: one pixel in 16 is read) or from 16-bit frames (decoded LJ92 payloads);
//   host:   the gamma table (log2f of the node's libm, like every table of this path), the constant bytes of the file and
//           the placing of the pixel bytes into the 126-byte sub-blocks.
#include "clip.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace mlv {

constexpr int GIF_BPP = 7, GIF_COLOR_TABLE = (1 << GIF_BPP) * 3, GIF_CC = 1 << GIF_BPP, GIF_EOI = GIF_CC + 1;
constexpr int GIF_SUB = (1 << GIF_BPP) - 2, GIF_FRAMES = 10, GIF_DOWN = 4;                     // gif.c:29-38

__global__ __launch_bounds__(256) void k_gif_pixels(const uint8_t *frames, size_t stride, int bpp, int packed, int out_w, int out_h,
                                                    const uint8_t *gamma, uint8_t *out)
{
    const uint8_t *f = frames + (size_t)blockIdx.y * stride;
    uint8_t *o = out + (size_t)blockIdx.y * out_w * out_h;
    const int n = out_w * out_h;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const int y = k / out_w, x = k - y * out_w;
        const uint32_t i = (uint32_t)y * GIF_DOWN * (uint32_t)out_w * GIF_DOWN + (uint32_t)x * GIF_DOWN + 1;      // gif.c:197
        uint32_t p;
        if (packed) {                                     // dng.c:813-843 for this one pixel
            const uint64_t bit = (uint64_t)i * (uint32_t)bpp;
            const uint16_t *s = (const uint16_t *)f;
            uint64_t word = bit >> 4;
            if (packed == 2) {                            // row pieces: piece y starts at the word of its first pixel
                const uint64_t bit0 = ((uint64_t)y * GIF_DOWN * (uint32_t)out_w * GIF_DOWN + 1) * (uint32_t)bpp;
                s = (const uint16_t *)(f + (size_t)y * (stride / (size_t)out_h));
                word -= bit0 >> 4;
            }
            const uint32_t two = ((uint32_t)s[word] << 16) | s[word + 1];
            p = (two >> (32 - bpp - (int)(bit & 15))) & ((1u << bpp) - 1u);
        } else p = ((const uint16_t *)f)[i];
        o[k] = gamma[min(p >> 4, 4095u) & 1023u];         // a 16-bit container never exceeds 4095 here; 14-bit data stays below 1024
    }
}

void gif_gamma(int black, uint8_t (&gamma)[1024])                                                // gif.c:103-107
{
    const int b4 = (int)(uint16_t)black >> 4;
    for (int i = 0; i < 1024; i++) {
        const int g = (i > b4) ? (int)(log2f((float)(i - b4)) * 255 / 10) : 0;
        gamma[i] = (uint8_t)(g * g / 255 / 2);
    }
}

size_t gif_size(int xres, int yres)                                                              // gif.c:222-234
{
    const uint16_t width = (uint16_t)(xres / GIF_DOWN), height = (uint16_t)(yres / GIF_DOWN);
    const size_t header = 13 + GIF_COLOR_TABLE + 19, frame_header = 8 + 11;
    const size_t pixels = (size_t)(int)(width * height) + 1;
    const size_t lzw = (pixels / (GIF_SUB - 1) + 1) * 2;
    return header + GIF_FRAMES * (frame_header + pixels + lzw + 1) + 1;
}

// pixel bytes of the 10 frames (device, [10][width * height]) -> the file (host, gif_size bytes)
void gif_assemble(int xres, int yres, const uint8_t *pixels, uint8_t *file)
{
    const uint16_t width = (uint16_t)(xres / GIF_DOWN), height = (uint16_t)(yres / GIF_DOWN);
    size_t pos = 0;
    auto put = [&](const void *p, size_t n) { memcpy(file + pos, p, n); pos += n; };
    auto put16 = [&](uint16_t v) { file[pos++] = (uint8_t)v; file[pos++] = (uint8_t)(v >> 8); };
    put("GIF89a", 6);
    put16(width); put16(height);
    file[pos++] = 0xF6; file[pos++] = 0; file[pos++] = 0;                                          // gif.c:109-118
    for (int i = 0, c = 0; i < GIF_COLOR_TABLE; i += 3, c += 2) { file[pos] = file[pos + 1] = file[pos + 2] = (uint8_t)c; pos += 3; }      // gif.c:119-128: greys 0, 2, ... 254
    static const uint8_t app[19] = { 0x21, 0xFF, 0x0B, 0x4E, 0x45, 0x54, 0x53, 0x43, 0x41, 0x50, 0x45, 0x32, 0x2E, 0x30, 0x03, 0x01, 0x00, 0x00, 0x00 };
    static const uint8_t gfx[8] = { 0x21, 0xF9, 0x04, 0x00, 0x32, 0x00, 0x00, 0x00 };              // gif.c:79-80
    put(app, sizeof app);
    const size_t n = (size_t)width * height;
    for (int fr = 0; fr < GIF_FRAMES; fr++) {
        put(gfx, sizeof gfx);
        file[pos++] = 0x2C; put16(0); put16(0); put16(width); put16(height); file[pos++] = 0x00; file[pos++] = GIF_BPP;      // gif.c:130-139
        const uint8_t *px = pixels + (size_t)fr * n;
        size_t done = 0;
        for (; done + (GIF_SUB - 1) <= n; done += GIF_SUB - 1) {          // full sub-blocks: size, clear code, 125 pixels (gif.c:193-206)
            file[pos++] = GIF_SUB; file[pos++] = GIF_CC;
            put(px + done, GIF_SUB - 1);
        }
        const size_t rest = n - done;                                      // the last one ends with the end-of-information code (gif.c:208-212)
        file[pos++] = (uint8_t)(rest + 2); file[pos++] = GIF_CC;
        put(px + done, rest);
        file[pos++] = GIF_EOI;
        file[pos++] = 0x00;
    }
    file[pos++] = 0x3B;
}

int launch_gif_pixels(const void *d_frames, size_t stride, int bpp, int packed, int xres, int yres, int nframes, const uint8_t *d_gamma,
                      void *d_out, hipStream_t stream)
{
    const int ow = xres / GIF_DOWN, oh = yres / GIF_DOWN;
    if (ow <= 0 || oh <= 0 || nframes <= 0) return MLVFS_AMD_OK;
    dim3 grid((unsigned)std::min((ow * oh + 255) / 256, 4096), (unsigned)nframes);
    hipLaunchKernelGGL(k_gif_pixels, grid, dim3(256), 0, stream, (const uint8_t *)d_frames, stride, bpp, packed, ow, oh, d_gamma,
                       (uint8_t *)d_out);
    MLV_HIP(hipGetLastError());
    return MLVFS_AMD_OK;
}

}  // namespace mlv

using namespace mlv;

extern "C" size_t mlvfs_amd_gif_size(const struct frame_headers *fh)
{
    return fh ? gif_size(fh->rawi_hdr.xRes, fh->rawi_hdr.yRes) : 0;
}

// h_frames: `nframes` (= 10) frames in HOST memory, `stride` bytes apart: packed payloads (packed != 0, geom->bpp bits per pixel;
// one 16-bit word of slack behind each, like dng_get_image_data's input) or 16-bit frames.  file: mlvfs_amd_gif_size bytes.
extern "C" int mlvfs_amd_gif_render(const mlvfs_amd_geom_t *geom, const void *h_frames, size_t stride, int packed, int nframes, uint8_t *file)
{
    if (!geom || !h_frames || !file || nframes != GIF_FRAMES) { set_error("gif: needs %d frames", GIF_FRAMES); return MLVFS_AMD_ERR_ARG; }
    if (packed == 2 && (geom->height / GIF_DOWN <= 0 || stride % (size_t)(geom->height / GIF_DOWN) != 0 || (stride / (size_t)(geom->height / GIF_DOWN)) % 2 != 0)) {
        set_error("gif: row pieces need a stride that is a multiple of the %d rows", geom->height / GIF_DOWN);
        return MLVFS_AMD_ERR_ARG;
    }
    ThreadCtx *c = thread_ctx();
    if (!c) return MLVFS_AMD_ERR_HIP;
    const int ow = geom->width / GIF_DOWN, oh = geom->height / GIF_DOWN;
    const size_t n = (size_t)ow * oh, in_bytes = stride * nframes, out_bytes = n * nframes;
    int rc = c->ensure(in_bytes + 16, out_bytes + 1024 + 16);
    if (rc) return rc;
    uint8_t gamma[1024];
    gif_gamma(geom->black, gamma);
    uint8_t *d_gamma = (uint8_t *)c->d_b + ((out_bytes + 15) & ~(size_t)15);
    MLV_HIP(hipMemcpyAsync(c->d_a, h_frames, in_bytes, hipMemcpyHostToDevice, c->stream));
    MLV_HIP(hipMemcpyAsync(d_gamma, gamma, sizeof gamma, hipMemcpyHostToDevice, c->stream));
    rc = launch_gif_pixels(c->d_a, stride, geom->bpp, packed == 2 ? 2 : (packed != 0 ? 1 : 0), geom->width, geom->height, nframes, d_gamma, c->d_b, c->stream);
    if (rc) return rc;
    std::vector<uint8_t> px(out_bytes ? out_bytes : 1);
    if (out_bytes) MLV_HIP(hipMemcpyAsync(px.data(), c->d_b, out_bytes, hipMemcpyDeviceToHost, c->stream));
    MLV_HIP(hipStreamSynchronize(c->stream));
    gif_assemble(geom->width, geom->height, px.data(), file);
    return MLVFS_AMD_OK;
}
