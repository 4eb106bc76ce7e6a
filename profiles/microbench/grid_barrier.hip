// grid_barrier.hip - what does a grid-wide barrier cost on MI355X next to a kernel boundary? 256 persistent workgroups of 768 threads
// (one per CU, like the layer kernels) run R rounds of { touch 1 MB of a buffer another workgroup wrote in the previous round; barrier }.
// Barrier = one atomic counter in device memory (release: __threadfence, agent scope -> L2 write-back across XCDs; spin with s_sleep;
// acquire: __threadfence). Compared with the same rounds as R separate kernel launches on one stream.
// build: hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(counter, 1u);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
        __threadfence();
    }
    __syncthreads();
}
__global__ __launch_bounds__(768) void k_rounds(float* buf, int n, unsigned* counter, int rounds, int use_barrier) {
    const int nb = gridDim.x;
    for (int r = 0; r < rounds; ++r) {
        // read the slice the NEXT workgroup wrote last round, write my own (cross-CU, mostly cross-XCD traffic)
        const int src = (blockIdx.x + 1) % nb;
        float acc = 0.f;
        for (int i = threadIdx.x; i < n; i += 768) acc += buf[(size_t)src * n + i];
        for (int i = threadIdx.x; i < n; i += 768) buf[(size_t)blockIdx.x * n + i] = acc * 1e-9f + r;
        if (use_barrier) grid_barrier(counter, (unsigned)(nb * (r + 1)));
    }
}
int main() {
    const int nb = 256, n = 1024;      // 4 KB per workgroup and round
    float* buf; unsigned* cnt;
    hipMalloc(&buf, (size_t)nb * n * 4); hipMemset(buf, 0, (size_t)nb * n * 4);
    hipMalloc(&cnt, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int R = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(cnt, 0, 4);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_rounds, dim3(nb), dim3(768), 0, 0, buf, n, cnt, R, 1);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("persistent kernel, %d rounds with a grid barrier : %.2f us per round\n", R, ms * 1e3 / R);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_rounds, dim3(nb), dim3(768), 0, 0, buf, n, cnt, R, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("persistent kernel, %d rounds WITHOUT barrier     : %.2f us per round (the work itself)\n", R, ms * 1e3 / R);
        hipEventRecord(e0);
        for (int r = 0; r < R; ++r) hipLaunchKernelGGL(k_rounds, dim3(nb), dim3(768), 0, 0, buf, n, cnt, 1, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("%d separate launches of one round                : %.2f us per round\n", R, ms * 1e3 / R);
    }
    return 0;
}
