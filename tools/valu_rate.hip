// Micro-benchmark: issue rate of the integer VALU ops the median networks use (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
template <int OP> __global__ void k(int *out, int n, int seed)
{
    int a[8], b = seed + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed * (i + 3) + threadIdx.x;
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 1) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 2) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(a[(i + 1) & 7]));
                if (OP == 3) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(a[(i + 1) & 7]));
                if (OP == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 5) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 6) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 7) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]));
                if (OP == 8) asm volatile("v_alignbit_b32 %0, %0, %1, 5" : "+v"(a[i]) : "v"(b));
                if (OP == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
                if (OP == 10) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 11) asm volatile("v_ffbh_u32 %0, %0" : "+v"(a[i]));
                if (OP == 12) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 13) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 14) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 15) asm volatile("v_med3_i16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(a[(i + 1) & 7]));
                if (OP == 16) asm volatile("v_cvt_pk_i16_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 17) asm volatile("v_bfe_i32 %0, %0, 16, 16" : "+v"(a[i]));
                if (OP == 18) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 19) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(a[(i + 1) & 7]));
                if (OP == 20) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 21) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 22) asm volatile("v_ashrrev_i32 %0, 16, %0" : "+v"(a[i]));
            }
        }
    }
    int s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char *name, int blocks, int threads)
{
    int *d; hipMalloc(&d, sizeof(int) * blocks * threads);
    const int n = 200;
    k<OP><<<blocks, threads>>>(d, 2, 1);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<OP><<<blocks, threads>>>(d, n, 1); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)blocks * threads / 64 * n * REP * 8;
    printf("%-14s blocks=%4d thr=%3d  %.2f wave-instr/clk/SIMD @2.4GHz  (%.1f T lane-ops/s)\n", name, blocks, threads,
           wave_instr / (ms * 1e-3) / (1024.0 * 2.4e9), wave_instr * 64 / (ms * 1e-3) / 1e12);
    hipFree(d);
}
int main()
{
    for (int thr : {256, 1024}) {
        int blocks = thr == 256 ? 2048 : 512;
        run<0>("v_min_i32", blocks, thr); run<1>("v_max_i32", blocks, thr); run<2>("v_med3_i32", blocks, thr);
        run<3>("v_min3_i32", blocks, thr); run<4>("v_add_u32", blocks, thr); run<5>("v_fma_f32", blocks, thr);
        run<6>("v_and_b32", blocks, thr); run<7>("v_lshlrev_b32", blocks, thr); run<8>("v_alignbit", blocks, thr);
        run<9>("v_cndmask", blocks, thr); run<10>("v_mul_i32_i24", blocks, thr); run<11>("v_ffbh_u32", blocks, thr);
        run<12>("v_pk_min_i16", blocks, thr); run<13>("v_lshl_add_u32", blocks, thr);
        run<14>("v_pk_max_i16", blocks, thr); run<15>("v_med3_i16", blocks, thr); run<16>("v_cvt_pk_i16_i32", blocks, thr);
        run<17>("v_bfe_i32", blocks, thr); run<18>("v_pk_add_i16", blocks, thr); run<19>("v_perm_b32", blocks, thr);
        run<20>("v_pk_min_u16", blocks, thr); run<21>("v_sub_u32", blocks, thr); run<22>("v_ashrrev_i32", blocks, thr);
    }
    return 0;
}
