// Micro-benchmark 4 (round 5): issue rates of what the packed-once median chain could use on gfx950 --
// the unpacked 16-bit three-input forms (v_min3/max3/med3_i16, with op_sel on the high halves), SDWA min/max on one half of a
// register, compare -> select sequences on VCC and on an SGPR pair (valu_rate2/3 timed a v_cndmask whose VCC the loop never wrote),
// and the packed f16 three-input forms once more beside v_pk_min_i16.
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate4.hip -o build/valu_rate4
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
#define OPS(X) \
 X(0, "v_pk_min_i16 %0, %0, %1") \
 X(1, "v_min3_i16 %0, %0, %1, %2") X(2, "v_max3_i16 %0, %0, %1, %2") X(3, "v_med3_i16 %0, %0, %1, %2") \
 X(4, "v_med3_i16 %0, %0, %1, %2 op_sel:[1,1,1,1]") X(5, "v_min3_i16 %0, %0, %1, %2 op_sel:[1,1,1,1]") \
 X(6, "v_min_i16 %0, %0, %1") X(7, "v_max_i16 %0, %0, %1") \
 X(8, "v_min_i16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1") \
 X(9, "v_pk_minimum3_f16 %0, %0, %1, %2") X(10, "v_pk_maximum3_f16 %0, %0, %1, %2") X(11, "v_pk_min_f16 %0, %0, %1") \
 X(12, "v_cmp_lt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc") \
 X(13, "v_cmp_lt_i32 s[20:21], %0, %1\n\tv_cndmask_b32 %0, %0, %2, s[20:21]") \
 X(14, "v_cmp_lt_i32 vcc, %0, %1\n\tv_min_i32 %0, %0, %2\n\tv_cndmask_b32 %0, %0, %2, vcc") \
 X(15, "v_med3_i32 %0, %0, %1, %2") X(16, "v_min_i32 %0, %0, %1") \
 X(17, "v_pk_max_i16 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]") \
 X(18, "v_pk_add_i16 %0, %0, %1 clamp") X(19, "v_pk_sub_i16 %0, %0, %1 clamp") \
 X(20, "v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1") \
 X(21, "v_add_u32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0") \
 X(22, "v_add_u32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1") \
 X(23, "v_or_b32 %0, 0x4b000000, %0") X(24, "v_lshrrev_b32 %0, 8, %0") X(25, "v_bcnt_u32_b32 %0, %0, %1") \
 X(26, "v_bfe_u32 %0, %0, 0, %1") X(27, "v_dot2_i32_i16 %0, %0, %1, %2") X(28, "v_mad_i32_i24 %0, %0, %1, %2") \
 X(29, "v_sub_u32 %0, %0, %1") X(30, "v_cvt_pk_i16_i32 %0, %0, %1") X(31, "v_perm_b32 %0, %0, %1, %2")
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
#define X(N, S) if (OP == N) asm volatile(S : "+v"(a[i]) : "v"(b), "v"(a[(i + 1) & 7]) : "vcc", "s20", "s21");
                OPS(X)
#undef X
            }
        }
    }
    int s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char *name, int per, int blocks, int threads)
{
    int *d; (void)hipMalloc(&d, sizeof(int) * blocks * threads);
    const int n = 100;
    k<OP><<<blocks, threads>>>(d, 2, 1);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0); k<OP><<<blocks, threads>>>(d, n, 1); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double seq = (double)blocks * threads / 64 * n * REP * 8 / (ms * 1e-3) / (1024.0 * 2.4e9);
    printf("%-120s  %.3f sequences/clk/SIMD @2.4GHz = %.2f clk per sequence of %d\n", name, seq, 1.0 / seq, per);
    (void)hipFree(d);
}
int main()
{
#define X(N, S) { int per = 1; for (const char *p = S; *p; p++) per += *p == '\n'; run<N>(S, per, 512, 1024); }
    OPS(X)
#undef X
    return 0;
}
