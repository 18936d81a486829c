// Probe: operand layout of v_mfma_f32_16x16x16_bf16 (builtin mfma_f32_16x16x16bf16_1k) on gfx950.
// Hypothesis: lane l holds A[row l&15][k = 4(l>>4) + j], B[k = 4(l>>4) + j][col l&15], j = 0..3; D[row 4(l>>4) + r][col l&15].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ uint16_t bf(float f) { return (uint16_t)(__float_as_uint(f) >> 16); }
__global__ void k(const float* A, const float* B, float* D) {
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    s16x4 a, b;
    for (int j = 0; j < 4; ++j) {
        a[j] = (short)bf(A[i * 16 + 4 * g + j]);
        b[j] = (short)bf(B[(4 * g + j) * 16 + i]);
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = c[r];
}
int main() {
    float hA[256], hB[256], hD[256], ref[256];
    for (int x = 0; x < 256; ++x) { hA[x] = (float)((x * 7) % 13 - 6); hB[x] = (float)((x * 5) % 11 - 5); }
    for (int i = 0; i < 16; ++i) for (int n = 0; n < 16; ++n) { float s = 0; for (int kk = 0; kk < 16; ++kk) s += hA[i * 16 + kk] * hB[kk * 16 + n]; ref[i * 16 + n] = s; }
    float *dA, *dB, *dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    double e = 0; for (int x = 0; x < 256; ++x) e = fmax(e, fabs(hD[x] - ref[x]));
    printf("mfma 16x16x16 bf16 layout hypothesis: max abs error %g (%s)\n", e, e == 0 ? "confirmed" : "WRONG");
    return 0;
}
