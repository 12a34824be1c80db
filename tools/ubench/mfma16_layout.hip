// Microtest (GPU box only): operand / result register layout of v_mfma_f32_16x16x32_bf16 as the split kernels assume it:
//   A (16 x 32): lane l holds row l % 16, k = 8 * (l / 16) + 0..7;   B (32 x 16): lane l holds column l % 16, same k;
//   D (16 x 16): lane l holds column l % 16, rows 4 * (l / 16) + 0..3.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma16_layout.hip -o tools/ubench/mfma16_layout.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const float* A, const float* B, float* D) {
    const int l = threadIdx.x, r = l % 16, g = l / 16;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)A[r * 32 + 8 * g + j];
        b[j] = (__bf16)B[(8 * g + j) * 16 + r];
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(4 * g + i) * 16 + r] = acc[i];
}

int main() {
    float hA[16 * 32], hB[32 * 16], hD[256], ref[256];
    srand(3);
    for (int i = 0; i < 512; ++i) { hA[i] = (float)(rand() % 17 - 8); hB[i] = (float)(rand() % 13 - 6); }
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float s = 0;
            for (int kk = 0; kk < 32; ++kk) s += hA[i * 32 + kk] * hB[kk * 16 + j];
            ref[i * 16 + j] = s;
        }
    float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
    printf("v_mfma_f32_16x16x32_bf16 layout check: %d mismatches of 256 %s\n", bad, bad ? "(ASSUMED LAYOUT WRONG)" : "(layout as assumed)");
    return bad != 0;
}
