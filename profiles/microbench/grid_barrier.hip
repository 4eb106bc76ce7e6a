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
// XCD-hierarchical form (MI355X_MICROARCH.md "barrier-xcd"): workgroup b runs on XCD b % 8. Arrive on the XCD's own counter (its own
// 128-byte line); the LAST arriver of an XCD (its leader for this round) arrives on the top counter, waits for the eight leaders, and
// publishes the round in the XCD's generation word, which the other workgroups of the XCD poll (relaxed agent-scope loads + s_sleep from
// one lane). Release fence before the arrival, acquire fence behind the wait. bar: [8 x 32 counters | top x 32 | 8 x 32 generations].
__device__ __forceinline__ void grid_barrier_xcd(unsigned* bar, unsigned round1 /* 1-based */) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
        unsigned* cnt = bar + 32 * xcd, *top = bar + 32 * 8, *gen = bar + 32 * 9 + 32 * xcd;
        __atomic_thread_fence(__ATOMIC_RELEASE);      // agent scope: this workgroup's stores are visible before it counts as arrived
        const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == per_xcd * round1 - 1) {
            __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 8u * round1) __builtin_amdgcn_s_sleep(1);
            __hip_atomic_store(gen, round1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round1) __builtin_amdgcn_s_sleep(1);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
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
        if (use_barrier == 1) grid_barrier(counter, (unsigned)(nb * (r + 1)));
        else if (use_barrier == 2) grid_barrier_xcd(counter, (unsigned)(r + 1));
    }
}
int main() {
    const int nb = 256;
    float* buf; unsigned* cnt;
    hipMalloc(&buf, (size_t)nb * 8192 * 4); hipMemset(buf, 0, (size_t)nb * 8192 * 4);
    hipMalloc(&cnt, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int R = 2000;
    for (int n : {1024, 1536, 8192}) {      // floats per workgroup and round: 4 KB, 6 KB (= 1.5 MB per round: one structure's state), 32 KB
    printf("---- %d bytes written and read per workgroup and round (%.2f MB per round)\n", n * 4, nb * n * 4 / 1e6);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(cnt, 0, 4096);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_rounds, dim3(nb), dim3(768), 0, 0, buf, n, cnt, R, 2);
        hipEventRecord(e1); hipEventSynchronize(e1);
        { float ms2; hipEventElapsedTime(&ms2, e0, e1);
          printf("persistent kernel, %d rounds with the XCD-hierarchical barrier : %.2f us per round\n", R, ms2 * 1e3 / R); }
        hipMemset(cnt, 0, 4096);
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
    }
    return 0;
}
