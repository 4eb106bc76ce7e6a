// mfma32_layout.hip - operand / result layout of v_mfma_f32_32x32x16_f16 on gfx950, checked against a host matmul.
// hypothesis: A[i][k]: lane l holds i = l % 32, k = 8 (l / 32) + j (j = 0..7);  B[k][n]: lane l holds n = l % 32, k = 8 (l / 32) + j;
//             D[i][n]: lane l, register v holds n = l % 32, i = 8 (v / 4) + 4 (l / 32) + v % 4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(const float* A, const float* B, float* D) {   // A [32][16], B [16][32], D [32][32] row-major
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)A[r * 16 + 8 * h + j]; b[j] = (_Float16)B[(8 * h + j) * 32 + r]; }
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    for (int v = 0; v < 16; ++v) D[(8 * (v / 4) + 4 * h + v % 4) * 32 + r] = acc[v];
}
int main() {
    float hA[512], hB[512], hD[1024], ref[1024];
    srand(1);
    for (int i = 0; i < 512; ++i) { hA[i] = (float)(rand() % 17 - 8) / 4.0f; hB[i] = (float)(rand() % 13 - 6) / 2.0f; }   // exact in f16
    for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n) { float s = 0; for (int kk = 0; kk < 16; ++kk) s += hA[i * 16 + kk] * hB[kk * 32 + n]; ref[i * 32 + n] = s; }
    float *dA, *dB, *dD;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    double err = 0; for (int i = 0; i < 1024; ++i) err = fmax(err, fabs(hD[i] - ref[i]));
    printf("v_mfma_f32_32x32x16_f16 layout hypothesis: max |D - ref| = %g  -> %s\n", err, err == 0 ? "CONFIRMED" : "WRONG");
    return err == 0 ? 0 : 1;
}
