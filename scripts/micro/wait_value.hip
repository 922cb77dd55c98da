// Latency of releasing pre-enqueued GPU work from the host: hipStreamWaitValue64 on a host flag vs. launching the kernel after
// the host event.  Build: hipcc --offload-arch=gfx950 -O2 wait_value.hip -o wait_value
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void stamp(volatile unsigned long long* out, unsigned long long v) { *out = v; __threadfence_system(); }
using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
int main() {
    hipStream_t s; CHK(hipStreamCreate(&s));
    unsigned long long *flag = nullptr, *out = nullptr;
    CHK(hipHostMalloc((void**)&out, 64, hipHostMallocCoherent | hipHostMallocMapped));
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 0) CHK(hipHostMalloc((void**)&flag, 64, hipHostMallocCoherent | hipHostMallocMapped));
        else { hipError_t e = hipExtMallocWithFlags((void**)&flag, 64, hipMallocSignalMemory); if (e != hipSuccess) { std::printf("signal memory: %s\n", hipGetErrorString(e)); continue; } }
        *flag = 0; *out = 0;
        double gated = 0, direct = 0; int ok = 1;
        for (int it = 1; it <= 200; ++it) {
            hipError_t e = hipStreamWaitValue64(s, flag, (uint64_t)it, hipStreamWaitValueGte, ~0ull);
            if (e != hipSuccess) { std::printf("mode %d: hipStreamWaitValue64 -> %s\n", mode, hipGetErrorString(e)); ok = 0; break; }
            hipLaunchKernelGGL(stamp, dim3(1), dim3(1), 0, s, out, (unsigned long long)it);
            std::this_thread::sleep_for(std::chrono::microseconds(200));
            auto t0 = clk::now();
            __atomic_store_n(flag, (unsigned long long)it, __ATOMIC_RELEASE);
            while (__atomic_load_n(out, __ATOMIC_ACQUIRE) != (unsigned long long)it) {}
            gated += us(t0, clk::now());
        }
        if (!ok) continue;
        CHK(hipStreamSynchronize(s));
        for (int it = 1; it <= 200; ++it) {
            std::this_thread::sleep_for(std::chrono::microseconds(200));
            auto t0 = clk::now();
            hipLaunchKernelGGL(stamp, dim3(1), dim3(1), 0, s, out, (unsigned long long)(1000 + it));
            while (__atomic_load_n(out, __ATOMIC_ACQUIRE) != (unsigned long long)(1000 + it)) {}
            direct += us(t0, clk::now());
        }
        std::printf("%s flag: host event -> kernel result visible: gated by hipStreamWaitValue64 %.1f us, launched after the event %.1f us\n",
                    mode == 0 ? "pinned host" : "signal-memory", gated / 200, direct / 200);
    }
    return 0;
}
