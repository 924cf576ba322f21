// Micro-benchmark 3: issue rate of the instructions of k_frame<5,true,true,false> that tools/valu_rate.hip and valu_rate2.hip
// did not time (tools/isa_classes.py prices every instruction class of the kernel with these three logs).  v_cndmask_b32 is
// timed again: valu_rate2's form read a VCC the loop never wrote and came out at 0.044 per clock, which no kernel confirms.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
#define OPS(X) \
 X(0, "v_min_i32 %0, %0, %1") X(1, "v_cndmask_b32 %0, %0, %1, vcc") X(2, "v_cndmask_b32 %0, %0, %1, s[20:21]") \
 X(3, "v_pk_min_i16 %0, %0, %1") X(4, "v_pk_max_i16 %0, %0, %1") X(5, "v_pk_min_u16 %0, %0, %1") \
 X(6, "v_med3_i32 %0, %0, %1, %2") X(7, "v_min3_i32 %0, %0, %1, %2") X(8, "v_max3_i32 %0, %0, %1, %2") X(9, "v_min3_u32 %0, %0, %1, %2") \
 X(10, "v_sub_i32 %0, %0, %1 clamp") X(11, "v_lshl_add_u32 %0, %0, 3, %1") X(12, "v_add_lshl_u32 %0, %0, %1, 3") \
 X(13, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD") \
 X(14, "v_mul_i32_i24 %0, %0, %1") X(15, "v_mul_i32_i24_sdwa %0, sext(%0), %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD") \
 X(16, "v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1") \
 X(17, "v_mov_b32_dpp %0, %1 row_newbcast:4 row_mask:0xf bank_mask:0xf bound_ctrl:1") \
 X(18, "v_pk_ashrrev_i16 %0, 15, %0") X(19, "v_pk_add_i16 %0, %0, %1 clamp") X(20, "v_pk_sub_i16 %0, %0, %1") \
 X(21, "v_cmp_gt_u32 vcc, %0, %1") X(22, "v_cmp_lt_i32 s[20:21], %0, %1") X(23, "v_add3_u32 %0, %0, %1, %2") \
 X(24, "v_cvt_pk_i16_i32 %0, %0, %1") X(25, "v_readlane_b32 s22, %0, 3") X(26, "v_writelane_b32 %0, s22, 3") \
 X(27, "v_mov_b32 %0, %1") X(28, "v_sub_f32 %0, %0, %1") X(29, "v_bfe_u32 %0, %0, 9, 14") X(30, "v_alignbit_b32 %0, %0, %0, 16") \
 X(31, "v_lshl_or_b32 %0, %0, 16, %1") X(32, "v_and_b32 %0, 0xffff0000, %0") X(33, "v_bfi_b32 %0, %1, %0, %2")
template <int OP> __global__ void k(int *out, int n, int seed)
{
    int a[8], b = seed + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed * (i + 3) + threadIdx.x;
    asm volatile("v_cmp_lt_i32 vcc, %0, %1\n\tv_cmp_gt_i32 s[20:21], %0, %1\n\ts_mov_b32 s22, 7" :: "v"(a[0]), "v"(b) : "vcc", "s20", "s21", "s22");
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
#define X(N, S) if (OP == N) asm volatile(S : "+v"(a[i]) : "v"(b), "v"(a[(i + 1) & 7]) : "vcc", "s20", "s21", "s22");
                OPS(X)
#undef X
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
    const int n = 100;
    k<OP><<<blocks, threads>>>(d, 2, 1);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<OP><<<blocks, threads>>>(d, n, 1); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-110s thr=%4d  %.3f wave-instr/clk/SIMD @2.4GHz\n", name, threads, (double)blocks * threads / 64 * n * REP * 8 / (ms * 1e-3) / (1024.0 * 2.4e9));
    hipFree(d);
}
int main()
{
#define X(N, S) run<N>(S, 512, 1024);
    OPS(X)
#undef X
    return 0;
}
