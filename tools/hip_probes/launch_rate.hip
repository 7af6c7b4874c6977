// Launch throughput of the HIP runtime from T host threads, one stream each: empty kernels, a stream synchronisation every S launches.
// hipcc --offload-arch=gfx950 -O2 -o tools/ubench/launch_rate tools/ubench/launch_rate.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 1000) *p = 1; }
__global__ void busy_kernel(float* p, int n) { float a = 0.f; for (int i = 0; i < n; ++i) a = a * 1.0001f + 1.f; if (a == 123.f) *p = a; }
int main(int argc, char** argv) {
    const int per_thread = argc > 1 ? atoi(argv[1]) : 4000, sync_every = argc > 2 ? atoi(argv[2]) : 4, spin = argc > 3 ? atoi(argv[3]) : 0;
    hipSetDevice(0);
    for (int T : {1, 2, 4, 8, 12}) {
        std::vector<hipStream_t> streams((size_t)T);
        for (auto& s : streams) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        for (auto& s : streams) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, nullptr); hipStreamSynchronize(s); }
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t] {
                hipSetDevice(0);
                for (int i = 0; i < per_thread; ++i) {
                    if (spin) hipLaunchKernelGGL(busy_kernel, dim3(256), dim3(256), 0, streams[(size_t)t], nullptr, spin);
                    else hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, streams[(size_t)t], nullptr);
                    if ((i + 1) % sync_every == 0) hipStreamSynchronize(streams[(size_t)t]);
                }
                hipStreamSynchronize(streams[(size_t)t]);
            });
        for (auto& x : th) x.join();
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("threads %2d: %d launches (sync every %d, kernel %s) in %.1f ms -> %.0f launches/ms total, %.1f us per launch per thread\n", T,
               T * per_thread, sync_every, spin ? "busy" : "empty", s * 1e3, T * per_thread / (s * 1e3), s * 1e6 / per_thread);
        for (auto& s2 : streams) hipStreamDestroy(s2);
    }
    return 0;
}
