// tools/dpp_probe.hip -- what the DPP controls that k_frame.hip relies on do on gfx950: prints, per lane, the lane whose value arrives.
//   wave_shl:1 (0x130, bound_ctrl)      lane i reads lane i + 1, the wave's last lane reads 0            (chain_fetch_*)
//   row_shl:1  (0x101, no bound_ctrl)   lane i reads lane i + 1 of its row of 16, lane 15 keeps its own  (rejected variant)
//   row_share:n (0x150 + n)             every lane of a row of 16 reads lane n of that row                (robust_ref)
//   hipcc --offload-arch=gfx950 -O2 tools/dpp_probe.hip -o tools/dpp_probe && tools/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_probe(int *out)
{
    const int lane = threadIdx.x;
    const int v = lane + 100;
    out[lane] = __builtin_amdgcn_mov_dpp(v, 0x130, 0xf, 0xf, true);
    out[64 + lane] = __builtin_amdgcn_update_dpp(v, v, 0x101, 0xf, 0xf, false);
    out[128 + lane] = __builtin_amdgcn_mov_dpp(v, 0x150 + 4, 0xf, 0xf, true);
    out[192 + lane] = __builtin_amdgcn_mov_dpp(v, 0x150 + 14, 0xf, 0xf, true);
}

int main()
{
    int *d = nullptr, h[256];
    if (hipMalloc(&d, sizeof h) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[4] = { "wave_shl:1 bound_ctrl", "row_shl:1 keep", "row_share:4", "row_share:14" };
    for (int t = 0; t < 4; t++) {
        printf("%-22s", names[t]);
        for (int i = 0; i < 64; i++) printf(" %d", h[64 * t + i] ? h[64 * t + i] - 100 : -1);
        printf("\n");
    }
    hipFree(d);
    return 0;
}
