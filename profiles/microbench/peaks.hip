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

template <int KIND>
__global__ __launch_bounds__(256) void k_mfma(float* __restrict__ out, int iters) {
    const f16x8 a = {(_Float16)1.0f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)1.5f, (_Float16)1.0f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)1.5f};
    const f16x8 b = {(_Float16)0.5f, (_Float16)0.5f, (_Float16)0.5f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)0.25f, (_Float16)0.25f, (_Float16)0.25f};
    if (KIND == 0) {
        f32x4 c[8];
        for (int k = 0; k < 8; ++k) c[k] = f32x4{0, 0, 0, 0};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k) c[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[k], 0, 0, 0);
        float s = 0;
        for (int k = 0; k < 8; ++k) s += c[k][0];
        if (s == 1.2345f) out[threadIdx.x] = s;
    } else if (KIND == 1) {
        f32x16 c[4];
        for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) c[k][j] = 0;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) c[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[k], 0, 0, 0);
        float s = 0;
        for (int k = 0; k < 4; ++k) s += c[k][0];
        if (s == 1.2345f) out[threadIdx.x] = s;
    } else {
        f32x4 c[8];
        for (int k = 0; k < 8; ++k) c[k] = f32x4{0, 0, 0, 0};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k) c[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, 0.5f, c[k], 0, 0, 0);
        float s = 0;
        for (int k = 0; k < 8; ++k) s += c[k][0];
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
    const int grid = p.multiProcessorCount * 8;      // 8 workgroups of 4 waves per CU = 8 waves per SIMD
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
        printf("MFMA %-26s %8.1f TFLOP/s dense (%d waves/SIMD, %d independent accumulators)\n", c.name, total / best / 1e9, 8, c.per_iter);
    }
    return 0;
}
