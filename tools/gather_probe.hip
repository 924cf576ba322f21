// Probe for the table look-ups of k_frame as buffer loads with idxen (index * stride added by the address unit, no VALU):
//  1. semantics: stride / num_records / out-of-range behaviour of buffer_load_dword / _ushort ... idxen on gfx950
//  2. rate: wave-loads per clock per CU for 32 gathers in flight per lane from a 64 KiB int32 table (raw2ev by pixel value)
//     and a 917 KiB u16 table (ev2raw by EV), with the index locality of an image (smooth gradient + noise)
// build: hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o tools/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int int32x4_t __attribute__((ext_vector_type(4)));
__device__ int llvm_sbl_i32(int32x4_t rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.i32");
__device__ unsigned short llvm_sbl_u16(int32x4_t rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.i16");
__device__ inline int32x4_t make_rsrc(const void *p, unsigned stride, unsigned num_records)
{
    const unsigned long long a = (unsigned long long)p;
    int32x4_t r;
    r.x = (int)(unsigned)a;
    r.y = (int)(((unsigned)(a >> 32) & 0xFFFFu) | (stride << 16));
    r.z = (int)num_records;
    r.w = 0x00020000;
    return r;
}
__global__ void k_sem(const int *tab, const unsigned short *tab2, int n, const int *idx, int *out)
{
    const int32x4_t rs = make_rsrc(tab, 4, n), rs2 = make_rsrc(tab2, 2, n);
    const int i = idx[threadIdx.x];
    out[threadIdx.x] = llvm_sbl_i32(rs, i, 0, 0, 0);
    out[64 + threadIdx.x] = llvm_sbl_u16(rs2, i, 0, 0, 0);
}
template <int MODE> __global__ __launch_bounds__(256, 4) void k_rate(const int *tab, const unsigned short *tab2, int n1, int n2, int iters, int *out)
{
    const int32x4_t rs = make_rsrc(tab, 4, n1), rs2 = make_rsrc(tab2, 2, n2);
    unsigned s = blockIdx.x * 977u + threadIdx.x * 40503u;
    const int base = 2300 + (int)((blockIdx.x * 131 + (threadIdx.x & 7) * 90) % 9000);      // "image": a value level per tile / column group
    int acc = 0;
    for (int it = 0; it < iters; it++) {
        int v[32];
#pragma unroll
        for (int k = 0; k < 32; k++) {
            s = s * 1664525u + 1013904223u;
            v[k] = base + (int)((s >> 20) & 63) + ((k & 1) ? 700 : 0);
        }
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 32; k++) v[k] = llvm_sbl_i32(rs, v[k], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 32; k++) v[k] = llvm_sbl_u16(rs2, v[k] * 20, 0, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 32; k++) v[k] = tab[v[k]];
        }
#pragma unroll
        for (int k = 0; k < 32; k++) acc += v[k];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main()
{
    const int n1 = 16384, n2 = 14 * 32768;
    std::vector<int> h1(n1); std::vector<unsigned short> h2(n2);
    for (int i = 0; i < n1; i++) h1[i] = i * 3 + 1;
    for (int i = 0; i < n2; i++) h2[i] = (unsigned short)(i * 7 + 3);
    int *d1; unsigned short *d2; int *didx, *dout;
    hipMalloc(&d1, n1 * 4); hipMalloc(&d2, n2 * 2); hipMalloc(&didx, 64 * 4); hipMalloc(&dout, 1024 * 256 * 4);
    hipMemcpy(d1, h1.data(), n1 * 4, hipMemcpyHostToDevice); hipMemcpy(d2, h2.data(), n2 * 2, hipMemcpyHostToDevice);
    // 1. semantics with num_records = 1000 entries: indices 0, 1, 999 in range; 1000, 5000 out of range -> 0
    int hidx[64];
    for (int i = 0; i < 64; i++) hidx[i] = i;
    hidx[60] = 999; hidx[61] = 1000; hidx[62] = 5000; hidx[63] = -1;
    hipMemcpy(didx, hidx, sizeof hidx, hipMemcpyHostToDevice);
    k_sem<<<1, 64>>>(d1, d2, 1000, didx, dout);
    int ho[128]; hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; i++) {
        const bool in = hidx[i] >= 0 && hidx[i] < 1000;
        const int w1 = in ? h1[hidx[i]] : 0, w2 = in ? h2[hidx[i]] : 0;
        if (ho[i] != w1 || ho[64 + i] != w2) { bad++; printf("idx %d: dword %d (want %d) ushort %d (want %d)\n", hidx[i], ho[i], w1, ho[64 + i], w2); }
    }
    printf("idxen semantics (stride 4 / 2, num_records in entries, out of range reads 0): %s\n", bad ? "MISMATCH" : "ok");
    // 2. rates
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 1024, iters = 200;
    const char *names[3] = { "buffer_load_dword idxen, 64 KiB table", "buffer_load_ushort idxen, 917 KiB table", "global_load_dword (64-bit address), 64 KiB table" };
    for (int m = 0; m < 3; m++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (m == 0) k_rate<0><<<blocks, 256>>>(d1, d2, n1, n2, iters, dout);
            if (m == 1) k_rate<1><<<blocks, 256>>>(d1, d2, n1, n2, iters, dout);
            if (m == 2) k_rate<2><<<blocks, 256>>>(d1, d2, n1, n2, iters, dout);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double loads = (double)blocks * 4 * iters * 32;
        printf("%-52s %.3f wave-loads/clk/CU @2.4GHz  (%.1f ns per wave-load per CU)\n", names[m], loads / (ms * 1e-3) / (256.0 * 2.4e9), ms * 1e6 * 256 / loads);
    }
    return bad != 0;
}
