// Where do the waves of resident workgroups sit?  (HW_REG_HW_ID of every wave of a k_frame-shaped launch:
// 256 threads, 40 KiB LDS, 4 workgroups per CU.)  Used to decide how to balance per-wave roles across SIMDs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 4) void probe(unsigned *out, int spin)
{
    __shared__ int pad[10000];
    pad[threadIdx.x] = spin;
    __syncthreads();
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = clock64();
    while (clock64() - t0 < spin + pad[threadIdx.x] * 0) {}        // keep every workgroup resident for a while
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
}
int main()
{
    const int blocks = 1024;
    unsigned *d; hipMalloc(&d, blocks * 8 * sizeof(unsigned));
    probe<<<blocks, 256>>>(d, 2000000);
    std::vector<unsigned> h(blocks * 8);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> per_cu;
    for (int b = 0; b < blocks; b++) {
        unsigned hw = h[b * 8], xcc = h[b * 8 + 1] & 15;
        unsigned key = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
        per_cu[key].push_back(b);
    }
    printf("%zu distinct (xcc,se,sh,cu); first 6:\n", per_cu.size());
    int shown = 0;
    int same_map = 0, total = 0, distinct4 = 0;
    for (auto &kv : per_cu) {
        for (int b : kv.second) {
            unsigned s[4], wslot[4], tg = (h[b * 8] >> 16) & 15;
            bool d4 = true;
            for (int w = 0; w < 4; w++) { s[w] = (h[(b * 4 + w) * 2] >> 4) & 3; wslot[w] = h[(b * 4 + w) * 2] & 15; }
            for (int a = 0; a < 4; a++) for (int c = a + 1; c < 4; c++) if (s[a] == s[c]) d4 = false;
            total++; distinct4 += d4; same_map += (s[0] == 0 && s[1] == 1 && s[2] == 2 && s[3] == 3);
            if (shown < 6) printf("  cu %06x block %4d tg %2u  simd of waves 0..3 = %u %u %u %u   wave slots = %u %u %u %u\n", kv.first, b, tg, s[0], s[1], s[2], s[3], wslot[0], wslot[1], wslot[2], wslot[3]);
        }
        shown++;
    }
    printf("workgroups %d: four distinct SIMDs %d, wave w on SIMD w %d\n", total, distinct4, same_map);
    return 0;
}
