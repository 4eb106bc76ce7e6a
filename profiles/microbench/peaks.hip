// peaks.hip - this box's achievable peaks, measured before any roofline fraction is quoted:
//   * HBM: float4 stream copy and read-only sweep over buffers far larger than the 256 MB Infinity Cache;
//   * matrix cores: v_mfma_f32_16x16x32_f16 (the instruction the split GEMMs use), v_mfma_f32_32x32x16_f16 and
//     v_mfma_f32_16x16x4_f32 (the exact path), register-resident operands, 8 independent accumulators per wave.
// build: hipcc --offload-arch=gfx950 -O3 -o peaks peaks.hip        run: ./peaks
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ a, float* __restrict__ out, size_t n) {
    float4 s = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 v = a[i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = s.x;
}

// the accumulators are separate named values (an indexed array makes the compiler rotate them through v_accvgpr moves, which
// then dominate the loop); the loop body is exactly the MFMAs
template <int KIND>
__global__ __launch_bounds__(256) void k_mfma(float* __restrict__ out, int iters) {
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 1e-3f + j); b[j] = (_Float16)(0.25f * j); }
    if (KIND == 0) {
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
        for (int i = 0; i < iters; ++i) {
#define M16(c) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
            M16(c0) M16(c1) M16(c2) M16(c3) M16(c4) M16(c5) M16(c6) M16(c7)
        }
        const float s = c0[0] + c1[0] + c2[0] + c3[0] + c4[0] + c5[0] + c6[0] + c7[0];
        if (s == 1.2345f) out[threadIdx.x] = s;
    } else if (KIND == 1) {
        f32x16 c0, c1, c2, c3;
        for (int j = 0; j < 16; ++j) { c0[j] = 0; c1[j] = 0; c2[j] = 0; c3[j] = 0; }
        for (int i = 0; i < iters; ++i) {
#define M32(c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
            M32(c0) M32(c1) M32(c2) M32(c3)
        }
        const float s = c0[0] + c1[0] + c2[0] + c3[0];
        if (s == 1.2345f) out[threadIdx.x] = s;
    } else {
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
        const float x = threadIdx.x * 1e-3f, y = 0.5f + threadIdx.x;
        for (int i = 0; i < iters; ++i) {
#define M4(c) c = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c, 0, 0, 0);
            M4(c0) M4(c1) M4(c2) M4(c3) M4(c4) M4(c5) M4(c6) M4(c7)
        }
        const float s = c0[0] + c1[0] + c2[0] + c3[0] + c4[0] + c5[0] + c6[0] + c7[0];
        if (s == 1.2345f) out[threadIdx.x] = s;
    }
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s, %d CUs, clock %d MHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t bytes = (size_t)2 << 30;     // 2 GiB per buffer
    float4 *a, *b; float* out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&out, 4096));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    const size_t n = bytes / 16;
    for (int blocks : {2048, 8192, 32768}) {
        float best_c = 1e9f, best_r = 1e9f, ms;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n); CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best_c) best_c = ms;
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, out, n); CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best_r) best_r = ms;
        }
        printf("HBM  float4 copy  %6d blocks: %7.1f GB/s (read + write)   read-only sweep: %7.1f GB/s\n", blocks,
               2.0 * bytes / best_c / 1e6, (double)bytes / best_r / 1e6);
    }
    const int iters = 20000;
    for (int wps : {1, 2, 3, 4, 8}) {
    const int grid = p.multiProcessorCount * wps;      // wps workgroups of 4 waves per CU = wps waves per SIMD
    struct { const char* name; double flop; int kind; int per_iter; } cases[] = {
        {"v_mfma_f32_16x16x32_f16", 2.0 * 16 * 16 * 32, 0, 8}, {"v_mfma_f32_32x32x16_f16", 2.0 * 32 * 32 * 16, 1, 4}, {"v_mfma_f32_16x16x4_f32", 2.0 * 16 * 16 * 4, 2, 8}};
    for (auto& c : cases) {
        float best = 1e9f, ms;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            if (c.kind == 0) hipLaunchKernelGGL(k_mfma<0>, dim3(grid), dim3(256), 0, 0, out, iters);
            else if (c.kind == 1) hipLaunchKernelGGL(k_mfma<1>, dim3(grid), dim3(256), 0, 0, out, iters);
            else hipLaunchKernelGGL(k_mfma<2>, dim3(grid), dim3(256), 0, 0, out, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        const double total = (double)grid * 4 * iters * c.per_iter * c.flop;
        printf("MFMA %-26s %8.1f TFLOP/s dense (%d waves/SIMD, %d independent accumulators)  = %.1f cycles per MFMA per SIMD at 2.4 GHz\n", c.name,
               total / best / 1e9, wps, c.per_iter, best * 1e-3 * 2.4e9 / ((double)wps * iters * c.per_iter));
    }
    }
    return 0;
}
