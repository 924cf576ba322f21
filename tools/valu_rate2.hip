// Micro-benchmark 2: issue rate of candidate replacements for the integer min/max networks (gfx950):
// float min/max (exact on integers < 2^24), gfx950's three-input minimum/maximum, packed f16/f32 forms, DPP operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
#define OPS(X) \
 X(0, "v_min_i32 %0, %0, %1") X(1, "v_add_u32 %0, %0, %1") \
 X(2, "v_min_f32 %0, %0, %1") X(3, "v_max_f32 %0, %0, %1") X(4, "v_med3_f32 %0, %0, %1, %2") \
 X(5, "v_min3_f32 %0, %0, %1, %2") X(6, "v_max3_f32 %0, %0, %1, %2") \
 X(7, "v_minimum3_f32 %0, %0, %1, %2") X(8, "v_maximum3_f32 %0, %0, %1, %2") \
 X(9, "v_pk_min_f16 %0, %0, %1") X(10, "v_pk_max_f16 %0, %0, %1") \
 X(11, "v_pk_minimum3_f16 %0, %0, %1, %2") X(12, "v_pk_maximum3_f16 %0, %0, %1, %2") \
 X(13, "v_cvt_f32_i32 %0, %0") X(14, "v_cvt_i32_f32 %0, %0") \
 X(15, "v_bfe_u32 %0, %0, 3, 14") X(16, "v_and_or_b32 %0, %0, %1, %2") X(17, "v_lshl_or_b32 %0, %0, 3, %1") \
 X(18, "v_or3_b32 %0, %0, %1, %2") X(19, "v_add3_u32 %0, %0, %1, %2") X(20, "v_mad_u32_u24 %0, %0, %1, %2") \
 X(21, "v_min_u32 %0, %0, %1") X(22, "v_max_u16 %0, %0, %1") X(23, "v_xor_b32 %0, %0, %1") \
 X(24, "v_or_b32 %0, %0, %1") X(25, "v_sub_f32 %0, %0, %1") X(26, "v_mul_f32 %0, %0, %1") \
 X(27, "v_min_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf") \
 X(28, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf") \
 X(29, "v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf") \
 X(30, "v_min_i32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf") \
 X(31, "v_lshrrev_b32 %0, 3, %0") X(32, "v_ashrrev_i32 %0, 1, %0") X(33, "v_mov_b32 %0, %1") \
 X(34, "v_pk_add_u16 %0, %0, %1") X(35, "v_pk_sub_i16 %0, %0, %1 clamp") X(36, "v_pk_lshrrev_b16 %0, 3, %0") \
 X(37, "v_pk_mul_lo_u16 %0, %0, %1") X(38, "v_pk_mad_u16 %0, %0, %1, %2") X(39, "v_cmp_lt_i32 vcc, %0, %1") \
 X(40, "v_cndmask_b32 %0, %0, %1, vcc") X(41, "v_max_i16 %0, %0, %1") X(42, "v_sub_u32 %0, %0, %1") \
 X(43, "v_subrev_u32 %0, %0, %1") X(44, "v_max_f16 %0, %0, %1") X(45, "v_add_f32 %0, %0, %1") \
 X(46, "v_mul_u32_u24 %0, %0, %1") X(47, "v_mul_hi_u32_u24 %0, %0, %1") X(48, "v_mul_lo_u32 %0, %0, %1") \
 X(49, "v_alignbit_b32 %0, %0, %1, 5") X(50, "v_perm_b32 %0, %0, %1, %2") X(51, "v_bfi_b32 %0, %0, %1, %2") \
 X(52, "v_cvt_pk_i16_i32 %0, %0, %1") X(53, "v_cvt_pk_u16_u32 %0, %0, %1") X(54, "v_sat_pk_u8_i16 %0, %0") \
 X(55, "v_min_f32 %0, %0, %1 mul:2") X(56, "v_fma_f32 %0, %0, %1, %1") X(57, "v_cvt_f32_u32 %0, %0") \
 X(58, "v_cvt_f32_ubyte0 %0, %0") X(59, "v_frexp_exp_i32_f32 %0, %0") X(60, "v_ldexp_f32 %0, %0, %1") \
 X(61, "v_lshlrev_b32 %0, %1, %0") X(62, "v_max_i32 %0, %0, %1") X(63, "v_and_b32 %0, %0, %1")
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
#define X(N, S) if (OP == N) asm volatile(S : "+v"(a[i]) : "v"(b), "v"(a[(i + 1) & 7]) : "vcc");
                OPS(X)
#undef X
            }
        }
    }
    int s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// packed-f32 ops need 64-bit register pairs
template <int OP> __global__ void k2(double *out, int n, int seed)
{
    double a[8], b = seed + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed * (i + 3) + threadIdx.x;
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 3) asm volatile("v_pk_mov_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 4) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(a[i]));
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// LDS gathers: 64 KiB dword table / 16 KiB u16 table, addresses with the locality of a smooth image + noise
template <int MODE> __global__ void kl(int *out, int n, int seed)
{
    extern __shared__ int tab[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) tab[i] = i * 7;
    __syncthreads();
    unsigned s = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x, acc = 0;
    unsigned base = (threadIdx.x * 37) & 8191;
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            s = s * 1664525u + 1013904223u;
            unsigned idx = MODE == 0 ? (base + ((s >> 20) & 63)) : MODE == 1 ? ((s >> 12) & 16383) : (base + ((s >> 20) & 63));
            if (MODE == 2) acc += ((unsigned short *)tab)[idx];
            else acc += tab[idx];
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
static void report(const char *name, int blocks, int threads, double wave_instr, float ms)
{
    printf("%-58s thr=%4d  %.3f wave-instr/clk/SIMD @2.4GHz\n", name, threads, wave_instr / (ms * 1e-3) / (1024.0 * 2.4e9));
}
template <int OP> void run(const char *name, int blocks, int threads)
{
    int *d; hipMalloc(&d, sizeof(int) * blocks * threads);
    const int n = 100;
    k<OP><<<blocks, threads>>>(d, 2, 1);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<OP><<<blocks, threads>>>(d, n, 1); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    report(name, blocks, threads, (double)blocks * threads / 64 * n * REP * 8, ms);
    hipFree(d);
}
template <int OP> void run2(const char *name, int blocks, int threads)
{
    double *d; hipMalloc(&d, sizeof(double) * blocks * threads);
    const int n = 100;
    k2<OP><<<blocks, threads>>>(d, 2, 1);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k2<OP><<<blocks, threads>>>(d, n, 1); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    report(name, blocks, threads, (double)blocks * threads / 64 * n * REP * 8, ms);
    hipFree(d);
}
template <int MODE> void runl(const char *name, int blocks, int threads)
{
    int *d; hipMalloc(&d, sizeof(int) * blocks * threads);
    const int n = 2000;
    hipFuncSetAttribute((const void *)kl<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    kl<MODE><<<blocks, threads, 65536>>>(d, 2, 1);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); kl<MODE><<<blocks, threads, 65536>>>(d, n, 1); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double reads = (double)blocks * threads / 64 * n * 16;
    printf("%-58s thr=%4d  %.3f LDS wave-reads/clk/CU @2.4GHz (loop has ~6 VALU per read)\n", name, threads, reads / (ms * 1e-3) / (256.0 * 2.4e9));
    hipFree(d);
}
int main()
{
    for (int thr : {256, 1024}) {
        int blocks = thr == 256 ? 2048 : 512;
#define X(N, S) run<N>(S, blocks, thr);
        OPS(X)
#undef X
        run2<0>("v_pk_add_f32", blocks, thr); run2<1>("v_pk_mul_f32", blocks, thr); run2<2>("v_pk_fma_f32", blocks, thr);
        run2<3>("v_pk_mov_b32", blocks, thr); run2<4>("v_lshlrev_b64", blocks, thr);
    }
    runl<0>("ds_read_b32 local window (64 dwords)", 512, 1024);
    runl<1>("ds_read_b32 random over 64 KiB", 512, 1024);
    runl<2>("ds_read_u16 local window", 512, 1024);
    return 0;
}
