// launch_rate.hip -- how many kernel launches per second can T host threads enqueue, each on its own stream?
// The pipelined benchmark needs 3 x 18 000 decode launches + ~2 500 render launches per 3 batches (= 56 500 per 261 ms = 216 000 / s from
// four threads).  Empty kernels with a 256-byte by-value argument (the size of GemvArgs), so the GPU is never the bound.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -pthread scripts/micro/launch_rate.hip -o scripts/micro/launch_rate
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

struct Args { char pad[256]; };
__global__ void empty_kernel(Args a, int* sink) { if (a.pad[0] == 77 && sink) sink[0] = 1; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 18000;
    for (int threads : {1, 2, 3, 4, 6}) {
        std::vector<hipStream_t> st(threads);
        for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        std::vector<double> host_ms(threads);
        auto work = [&](int t) {
            Args a{};
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st[t], a, (int*)nullptr);
            host_ms[t] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        };
        for (int rep = 0; rep < 2; ++rep) {   // second repetition reported
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int t = 0; t < threads; ++t) th.emplace_back(work, t);
            for (auto& x : th) x.join();
            const double enq = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            CK(hipDeviceSynchronize());
            const double all = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (rep == 1) {
                double mx = 0;
                for (double h : host_ms) mx = h > mx ? h : mx;
                printf("%d thread(s) x %d launches: enqueue wall %.1f ms (slowest thread %.1f ms = %.2f us per launch), done %.1f ms; aggregate %.0f launches / ms\n",
                       threads, n, enq, mx, mx * 1e3 / n, all, threads * (double)n / all);
            }
        }
        for (auto& s : st) CK(hipStreamDestroy(s));
    }
    return 0;
}
