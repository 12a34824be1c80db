// Microbenchmark (GPU box only): do the matrix pipe and the VALU of one SIMD overlap when the work comes from two DIFFERENT
// waves (one issuing MFMAs, the other VALU), and which waves of a 512-thread block share a SIMD?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_cowave.hip -o cowave.bin && ./cowave.bin
// role per wave: 0 idle, 1 MFMA only, 2 VALU only, 3 alternate [48 MFMA | 384 VALU] per iteration, 4 alternate [384 VALU | 48 MFMA]
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Roles { int r[8]; };

__device__ __forceinline__ void mfma48(f32x16& a0, f32x16& a1, f32x16& a2, f32x16& a3, bf16x8 x, bf16x8 y) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
    }
}
#define V8(OPS) asm volatile(OPS : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b));
__device__ __forceinline__ void valu384(float& r0, float& r1, float& r2, float& r3, float& r4, float& r5, float& r6, float& r7, float b) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {      // 24 per round: the split mix (and / sub / perm / fma)
        V8("v_and_b32 %0, 0xffff0000, %0\n v_sub_f32 %1, %1, %8\n v_and_b32 %2, 0xffff0000, %2\n v_sub_f32 %3, %3, %8\n v_fma_f32 %4, %4, %8, %8\n v_sub_f32 %5, %5, %8\n v_and_b32 %6, 0xffff0000, %6\n v_max_f32 %7, %7, %8")
        V8("v_and_b32 %0, 0xffff0000, %0\n v_sub_f32 %1, %1, %8\n v_and_b32 %2, 0xffff0000, %2\n v_sub_f32 %3, %3, %8\n v_fma_f32 %4, %4, %8, %8\n v_sub_f32 %5, %5, %8\n v_and_b32 %6, 0xffff0000, %6\n v_max_f32 %7, %7, %8")
        V8("v_and_b32 %0, 0xffff0000, %0\n v_sub_f32 %1, %1, %8\n v_and_b32 %2, 0xffff0000, %2\n v_sub_f32 %3, %3, %8\n v_fma_f32 %4, %4, %8, %8\n v_sub_f32 %5, %5, %8\n v_and_b32 %6, 0xffff0000, %6\n v_max_f32 %7, %7, %8")
    }
}

__global__ __launch_bounds__(512) void k(Roles roles, float* out, int iters, int use_barrier) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6;
    const int role = roles.r[wave];
    float r0 = threadIdx.x, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7, b = 1.0001f;
    f32x16 a0, a1, a2, a3;
    for (int i = 0; i < 16; ++i) { a0[i] = 0; a1[i] = 0; a2[i] = 0; a3[i] = 0; }
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 0.001f); y[i] = (__bf16)0.5f; }
    for (int it = 0; it < iters; ++it) {
        if (role == 1) mfma48(a0, a1, a2, a3, x, y);
        else if (role == 2) valu384(r0, r1, r2, r3, r4, r5, r6, r7, b);
        else if (role == 3) { mfma48(a0, a1, a2, a3, x, y); valu384(r0, r1, r2, r3, r4, r5, r6, r7, b); }
        else if (role == 4) { valu384(r0, r1, r2, r3, r4, r5, r6, r7, b); mfma48(a0, a1, a2, a3, x, y); }
        if (use_barrier) __syncthreads();
    }
    float s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) lds[0] = s;
}

static void run(const char* name, Roles r, int barrier) {
    static float* out = nullptr;
    if (!out) (void)hipMalloc(&out, 512 * 256 * 4);
    const int iters = 2000, lds = 100 * 1024;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, 0, r, out, iters, barrier);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, 0, r, out, iters, barrier);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %s: %7.1f ns per iteration\n", name, barrier ? "barrier" : "free   ", ms * 1e6 / iters);
}

int main() {
    for (int barrier = 0; barrier <= 1; ++barrier) {
        run("waves 0-3 MFMA(48), 4-7 idle", Roles{{1, 1, 1, 1, 0, 0, 0, 0}}, barrier);
        run("waves 0-3 VALU(384), 4-7 idle", Roles{{2, 2, 2, 2, 0, 0, 0, 0}}, barrier);
        run("all 8 MFMA", Roles{{1, 1, 1, 1, 1, 1, 1, 1}}, barrier);
        run("all 8 VALU", Roles{{2, 2, 2, 2, 2, 2, 2, 2}}, barrier);
        run("waves 0-3 MFMA, 4-7 VALU", Roles{{1, 1, 1, 1, 2, 2, 2, 2}}, barrier);
        run("even waves MFMA, odd waves VALU", Roles{{1, 2, 1, 2, 1, 2, 1, 2}}, barrier);
        run("waves 0,1,4,5 MFMA, 2,3,6,7 VALU", Roles{{1, 1, 2, 2, 1, 1, 2, 2}}, barrier);
        run("waves 0-3 [MFMA|VALU], 4-7 idle", Roles{{3, 3, 3, 3, 0, 0, 0, 0}}, barrier);
        run("all 8 [MFMA|VALU] (lockstep)", Roles{{3, 3, 3, 3, 3, 3, 3, 3}}, barrier);
        run("waves 0-3 [MFMA|VALU], 4-7 [VALU|MFMA] (de-phased)", Roles{{3, 3, 3, 3, 4, 4, 4, 4}}, barrier);
        run("even [MFMA|VALU], odd [VALU|MFMA]", Roles{{3, 4, 3, 4, 3, 4, 3, 4}}, barrier);
    }
    return 0;
}
