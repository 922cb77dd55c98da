// micro-benchmark: pageable -> device through pinned bounce buffers filled by threads, for the set-up's LHS upload
// (engine_state.hip.hpp h2d).  Sweeps chunk size, buffer count and copy threads; prints GB/s for 252 MB.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
using clk = std::chrono::steady_clock;
static double ms(clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); }
static void tcopy(char* dst, const char* src, size_t n, int T) {
    if (T <= 1) { memcpy(dst, src, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([=] { size_t lo = n * t / T / 64 * 64, hi = t + 1 == T ? n : n * (t + 1) / T / 64 * 64; memcpy(dst + lo, src + lo, hi - lo); });
    for (auto& x : th) x.join();
}
int main() {
    hipStream_t s; hipStreamCreate(&s);
    const size_t n = 252u << 20;
    char* d; hipMalloc((void**)&d, n);
    std::vector<char> src(n, 1);
    { auto t = clk::now(); hipMemcpyAsync(d, src.data(), n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); printf("direct pageable: %.2f ms\n", ms(t)); }
    { auto t = clk::now(); hipMemcpyAsync(d, src.data(), n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); printf("direct pageable (2nd): %.2f ms\n", ms(t)); }
    for (size_t mb : {4, 8, 16, 32}) for (int nb : {2, 3, 4}) for (int T : {4, 8, 16, 32}) {
        const size_t chunk = mb << 20;
        std::vector<char*> buf(nb); std::vector<hipEvent_t> ev(nb);
        for (int i = 0; i < nb; ++i) { hipHostMalloc((void**)&buf[i], chunk); memset(buf[i], 0, chunk); hipEventCreateWithFlags(&ev[i], hipEventDisableTiming); }
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            auto t = clk::now();
            int f = 0;
            for (size_t off = 0; off < n; off += chunk) {
                size_t len = std::min(chunk, n - off);
                hipEventSynchronize(ev[f]);
                tcopy(buf[f], src.data() + off, len, T);
                hipMemcpyAsync(d + off, buf[f], len, hipMemcpyHostToDevice, s);
                hipEventRecord(ev[f], s);
                f = (f + 1) % nb;
            }
            hipStreamSynchronize(s);
            best = std::min(best, ms(t));
        }
        printf("chunk %2zu MB x %d buffers, %2d threads: %.2f ms (%.1f GB/s)\n", mb, nb, T, best, n / best / 1e6);
        for (int i = 0; i < nb; ++i) { hipHostFree(buf[i]); hipEventDestroy(ev[i]); }
    }
    // memcpy alone
    { char* b; hipHostMalloc((void**)&b, 32u << 20); for (int T : {1, 4, 8, 16, 32}) { auto t = clk::now(); for (int r = 0; r < 8; ++r) tcopy(b, src.data() + ((size_t)r << 25), 32u << 20, T); double a = ms(t); printf("memcpy into pinned, %2d threads: %.1f GB/s\n", T, 256.0 * 1.048576 / a); } }
    return 0;
}
