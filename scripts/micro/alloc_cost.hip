// micro-benchmark: what hipMalloc / hipFree / stream sync / pageable vs pinned copies cost on this box (setup planning)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using clk = std::chrono::steady_clock;
static double ms(clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); }
__global__ void touch(char* p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i * 4096 < n) p[i * 4096] = 1; }
int main() {
    hipStream_t s; hipStreamCreate(&s);
    void* w; hipMalloc(&w, 1 << 20); hipFree(w);
    for (size_t mb : {1, 4, 16, 64, 256}) {
        size_t n = mb << 20;
        for (int rep = 0; rep < 3; ++rep) {
            void* p;
            auto t = clk::now(); hipMalloc(&p, n); double a = ms(t);
            t = clk::now(); hipLaunchKernelGGL(touch, dim3((n / 4096 + 255) / 256), dim3(256), 0, s, (char*)p, n); hipStreamSynchronize(s); double k = ms(t);
            t = clk::now(); hipFree(p); double f = ms(t);
            printf("size %4zu MB rep %d: hipMalloc %.3f ms, first-touch kernel %.3f ms, hipFree %.3f ms\n", mb, rep, a, k, f);
        }
    }
    { auto t = clk::now(); for (int i = 0; i < 100; ++i) hipStreamSynchronize(s); printf("idle hipStreamSynchronize: %.1f us\n", 10 * ms(t)); }
    size_t n = 256u << 20;
    char* d; hipMalloc((void**)&d, n);
    std::vector<char> pageable(n, 1);
    char* pinned; hipHostMalloc((void**)&pinned, n); memset(pinned, 1, n);
    for (int rep = 0; rep < 3; ++rep) {
        auto t = clk::now(); hipMemcpyAsync(d, pageable.data(), n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double a = ms(t);
        t = clk::now(); hipMemcpyAsync(d, pinned, n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double b = ms(t);
        t = clk::now(); hipMemcpyAsync(pageable.data(), d, n, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); double c = ms(t);
        t = clk::now(); hipMemcpyAsync(pinned, d, n, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); double e = ms(t);
        printf("256 MB: H2D pageable %.1f ms (%.1f GB/s), H2D pinned %.1f ms (%.1f GB/s), D2H pageable %.1f ms, D2H pinned %.1f ms\n", a, n / a / 1e6, b, n / b / 1e6, c, e);
    }
    { auto t = clk::now(); std::vector<char> fresh(n); double a = ms(t); t = clk::now(); hipMemcpyAsync(fresh.data(), d, n, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
      printf("fresh std::vector<char>(256MB) %.1f ms, D2H into it %.1f ms\n", a, ms(t)); }
    { auto t = clk::now(); void* hp; hipHostMalloc(&hp, n); printf("hipHostMalloc 256 MB %.1f ms\n", ms(t)); t = clk::now(); hipHostFree(hp); printf("hipHostFree %.1f ms\n", ms(t)); }
    for (size_t kb : {4, 64, 1024}) {
        auto t = clk::now(); for (int i = 0; i < 50; ++i) { hipMemcpyAsync(d, pageable.data(), kb << 10, hipMemcpyHostToDevice, s); } hipStreamSynchronize(s);
        printf("50 x H2D pageable %zu KB: %.1f us each\n", kb, 20 * ms(t));
    }
    return 0;
}
