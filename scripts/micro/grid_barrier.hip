// What does a dependent launch cost on this part, and what does a grid-wide barrier inside a persistent kernel cost instead?
// (a) K dependent tiny kernels back to back on one stream, each a chain of `chain` dependent loads + one store per thread (what a
//     latency-bound coarse-level transfer / sweep looks like);  (b) the same phases inside ONE persistent kernel, separated by a
//     grid barrier (agent-scope release / acquire, one arrival counter);  (c) the barrier alone.
// Build: hipcc --offload-arch=gfx950 -O2 grid_barrier.hip -o grid_barrier.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void phase_work(const int* __restrict__ next, double* __restrict__ x, int n, int chain, int p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int j = i;
    for (int c = 0; c < chain; ++c) j = next[j];                     // dependent loads
    x[(size_t)((p + 1) & 1) * n + i] = x[(size_t)(p & 1) * n + j] + 1.0;      // ping-pong: reads what the previous phase wrote
}

__global__ void phase_kernel(const int* __restrict__ next, double* __restrict__ x, int n, int chain, int p) { phase_work(next, x, n, chain, p); }

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = wall_clock64();                 // 100 MHz; give up after 2 s (a grid that is not co-resident must not hang the box)
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 200000000ull) break;
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__global__ void persistent(const int* __restrict__ next, double* __restrict__ x, int n, int chain, int phases, unsigned* counter, unsigned base, int work) {
    for (int p = 0; p < phases; ++p) {
        if (work) phase_work(next, x, n, chain, p);
        grid_barrier(counter, base + (unsigned)(p + 1) * gridDim.x);
    }
}

int main() {
    hipStream_t s; CHK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int phases = 200;
    unsigned* counter; CHK(hipMalloc((void**)&counter, 64)); CHK(hipMemset(counter, 0, 64));
    unsigned base = 0;
    for (int blocks : {1, 64, 256, 512, 1024}) {
        const int n = blocks * 256;
        std::vector<int> h(n);
        for (int i = 0; i < n; ++i) h[i] = (int)(((long long)i * 7919 + 13) % n);
        int* next; double* x;
        CHK(hipMalloc((void**)&next, sizeof(int) * n)); CHK(hipMalloc((void**)&x, sizeof(double) * 2 * n));
        CHK(hipMemcpy(next, h.data(), sizeof(int) * n, hipMemcpyHostToDevice)); CHK(hipMemset(x, 0, sizeof(double) * 2 * n));
        for (int chain : {0, 3}) {
            float ms_k = 0, ms_p = 0, ms_b = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CHK(hipEventRecord(e0, s));
                for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(phase_kernel, dim3(blocks), dim3(256), 0, s, next, x, n, chain, p);
                CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms_k, e0, e1));
            }
            std::vector<double> xk(n); CHK(hipMemcpy(xk.data(), x, sizeof(double) * n, hipMemcpyDeviceToHost));
            CHK(hipMemset(x, 0, sizeof(double) * 2 * n));
            for (int rep = 0; rep < 2; ++rep) {
                CHK(hipEventRecord(e0, s));
                hipLaunchKernelGGL(persistent, dim3(blocks), dim3(256), 0, s, next, x, n, chain, phases, counter, base, 1);
                CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms_p, e0, e1));
                base += (unsigned)phases * blocks;
            }
            std::vector<double> xp(n); CHK(hipMemcpy(xp.data(), x, sizeof(double) * n, hipMemcpyDeviceToHost));
            CHK(hipMemset(x, 0, sizeof(double) * 2 * n));
            int bad = 0;
            for (int i = 0; i < n; ++i) bad += xk[i] != xp[i];
            for (int rep = 0; rep < 2; ++rep) {
                CHK(hipEventRecord(e0, s));
                hipLaunchKernelGGL(persistent, dim3(blocks), dim3(256), 0, s, next, x, n, chain, phases, counter, base, 0);
                CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms_b, e0, e1));
                base += (unsigned)phases * blocks;
            }
            std::printf("blocks %5d chain %d: per phase -- dependent launches %.2f us, persistent + grid barrier %.2f us (barrier alone %.2f us); results %s\n",
                        blocks, chain, 1e3 * ms_k / phases, 1e3 * ms_p / phases, 1e3 * ms_b / phases, bad ? "DIFFER" : "equal");
        }
        CHK(hipFree(next)); CHK(hipFree(x));
    }
    return 0;
}
