// Microbenchmark (GPU box only): issue cost in cycles of the VALU instructions the split-operand kernels are made of,
// one and two waves per SIMD.  hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o valu_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X X X X X X X X X X X X X X X X
enum { ADD, PKADD, CVTPK, LSHL, AND, EXP, CNDMASK, PERM, FMA, PKFMA, PKMUL, MAX, SUB_DEP, MOV };

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned long long* cyc, float* out, int iters) {
    float a = threadIdx.x * 0.25f + 1.0f, b = 1.0001f, c = 0.5f, d = 2.0f;
    float r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3, r4 = a + 4, r5 = a + 5, r6 = a + 6, r7 = a + 7;
    float2 p0 = {a, b}, p1 = {c, d}, p2 = {a, c}, p3 = {b, d};
    unsigned u0 = threadIdx.x, u1 = 77, u2 = 0x07060302u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        // 16 x 8 = 128 instructions per iteration on 8 independent destinations
        if (OP == ADD) { REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b));) }
        if (OP == FMA) { REP16(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b));) }
        if (OP == MAX) { REP16(asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b));) }
        if (OP == SUB_DEP) { REP16(asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %0, %0, %8\n v_sub_f32 %0, %0, %8\n v_sub_f32 %0, %0, %8\n v_sub_f32 %0, %0, %8\n v_sub_f32 %0, %0, %8\n v_sub_f32 %0, %0, %8\n v_sub_f32 %0, %0, %8" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b));) }
        if (OP == EXP) { REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b));) }
        if (OP == LSHL) { REP16(asm volatile("v_lshlrev_b32 %0, 16, %0\n v_lshlrev_b32 %1, 16, %1\n v_lshlrev_b32 %2, 16, %2\n v_lshlrev_b32 %3, 16, %3\n v_lshlrev_b32 %4, 16, %4\n v_lshlrev_b32 %5, 16, %5\n v_lshlrev_b32 %6, 16, %6\n v_lshlrev_b32 %7, 16, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b));) }
        if (OP == AND) { REP16(asm volatile("v_and_b32 %0, 0xffff0000, %0\n v_and_b32 %1, 0xffff0000, %1\n v_and_b32 %2, 0xffff0000, %2\n v_and_b32 %3, 0xffff0000, %3\n v_and_b32 %4, 0xffff0000, %4\n v_and_b32 %5, 0xffff0000, %5\n v_and_b32 %6, 0xffff0000, %6\n v_and_b32 %7, 0xffff0000, %7" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b));) }
        if (OP == MOV) { REP16(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b));) }
        if (OP == CNDMASK) { REP16(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b) : "vcc");) }
        if (OP == CVTPK) { REP16(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %8\n v_cvt_pk_bf16_f32 %1, %1, %8\n v_cvt_pk_bf16_f32 %2, %2, %8\n v_cvt_pk_bf16_f32 %3, %3, %8\n v_cvt_pk_bf16_f32 %4, %4, %8\n v_cvt_pk_bf16_f32 %5, %5, %8\n v_cvt_pk_bf16_f32 %6, %6, %8\n v_cvt_pk_bf16_f32 %7, %7, %8" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b));) }
        if (OP == PERM) { REP16(asm volatile("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b), "v"(u2));) }
        if (OP == PKADD) { REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p3));) }
        if (OP == PKMUL) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p3));) }
        if (OP == PKFMA) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p3));) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0.x + p1.y + p2.x + p3.y + (float)(u0 + u1);
}

template <int OP>
void run(const char* name) {
    unsigned long long* cyc; float* out;
    (void)hipMalloc(&cyc, 8 * 2048 * 4); (void)hipMalloc(&out, 256 * 2048 * 4);
    for (int wps = 1; wps <= 2; ++wps) {
        const int grid = 256 * wps, iters = 2000;
        hipLaunchKernelGGL((k<OP>), dim3(grid), dim3(256), 0, 0, cyc, out, iters);
        (void)hipDeviceSynchronize();
        unsigned long long h[8 * 2048];
        (void)hipMemcpy(h, cyc, sizeof(unsigned long long) * grid * 4, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < grid * 4; ++i) s += (double)h[i];
        // s_memtime ticks at 100 MHz constant clock on some parts; also report wall via events
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<OP>), dim3(grid), dim3(256), 0, 0, cyc, out, iters);
        (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-22s waves/SIMD %d: %6.2f s_memtime ticks / instr / wave ; %6.2f ns per instr per SIMD (all waves)\n", name, wps,
               s / (grid * 4) / (iters * 128.0), ms * 1e6 / (iters * 128.0 * wps));
    }
    (void)hipFree(cyc); (void)hipFree(out);
}

int main() {
    run<ADD>("v_add_f32"); run<FMA>("v_fma_f32"); run<MAX>("v_max_f32"); run<SUB_DEP>("v_sub_f32 dependent"); run<MOV>("v_mov_b32");
    run<LSHL>("v_lshlrev_b32"); run<AND>("v_and_b32"); run<CNDMASK>("v_cndmask_b32"); run<PERM>("v_perm_b32");
    run<CVTPK>("v_cvt_pk_bf16_f32"); run<EXP>("v_exp_f32"); run<PKADD>("v_pk_add_f32"); run<PKMUL>("v_pk_mul_f32"); run<PKFMA>("v_pk_fma_f32");
    return 0;
}
