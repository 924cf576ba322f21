// tools/zerocopy_probe.hip -- can kernels move a frame across PCIe themselves faster than T host threads calling hipMemcpyAsync?
// Per thread and stream, N times: 8.3 MB page-locked host -> device, a small kernel, 9.46 MB device -> page-locked host; either with
// hipMemcpyAsync (mode 0) or with copy kernels that read / write the host buffers directly (mode 1).  Prints frames/s.
//   hipcc --offload-arch=gfx950 -O2 tools/zerocopy_probe.hip -o build/zerocopy_probe -lpthread ; build/zerocopy_probe T N mode
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void k_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void k_touch(uint4 *p, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i].x ^= 1u;
}

static const size_t IN = 8279040, OUT = 9461760;

static void worker(int n, int mode, int grid)
{
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    void *h_in, *h_out, *d_in, *d_out;
    hipHostMalloc(&h_in, IN, hipHostMallocDefault);
    hipHostMalloc(&h_out, OUT, hipHostMallocDefault);
    hipMalloc(&d_in, IN);
    hipMalloc(&d_out, OUT);
    for (int k = 0; k < n; k++) {
        if (mode == 0) hipMemcpyAsync(d_in, h_in, IN, hipMemcpyHostToDevice, s);
        else hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, s, (const uint4 *)h_in, (uint4 *)d_in, IN / 16);
        hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, s, (uint4 *)d_out, OUT / 16);
        if (mode == 0) hipMemcpyAsync(h_out, d_out, OUT, hipMemcpyDeviceToHost, s);
        else hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, s, (const uint4 *)d_out, (uint4 *)h_out, OUT / 16);
        hipStreamSynchronize(s);
    }
    hipFree(d_in); hipFree(d_out); hipHostFree(h_in); hipHostFree(h_out);
    hipStreamDestroy(s);
}

int main(int argc, char **argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 16, N = argc > 2 ? atoi(argv[2]) : 32, mode = argc > 3 ? atoi(argv[3]) : 0;
    const int grid = argc > 4 ? atoi(argv[4]) : 256;
    hipSetDevice(0);
    worker(2, mode, grid);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int i = 0; i < T; i++) th.emplace_back(worker, N, mode, grid);
    for (auto &t : th) t.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("threads %d mode %s grid %d: %.0f frames/s (%.1f GB/s each way)\n", T, mode ? "copy kernels" : "hipMemcpyAsync", grid, T * N / dt,
           T * N / dt * OUT / 1e9);
    return 0;
}
