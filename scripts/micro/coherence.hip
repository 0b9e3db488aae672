// Does a reader kernel always see what the previous kernel of the SAME stream wrote, when the buffer is
// rewritten every iteration and a second stream keeps the GPU busy (so block -> XCD placement varies)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <atomic>

__global__ void writer(float* x, int n, float v) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] = v + (float)(i & 1023);
}
__global__ void reader(const float* __restrict__ x, int n, float v, unsigned* err) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (x[i] != v + (float)(i & 1023)) atomicAdd(err, 1u);
}
__global__ void busy(float* y, int n, int iters) {
    float acc = 0.f;
    for (int k = 0; k < iters; ++k)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += y[i] * 1.0001f;
    if (acc == 123.456f) y[0] = acc;
}
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 16 * 1024;       // floats in the rewritten buffer
    const int iters = argc > 2 ? atoi(argv[2]) : 20000;
    const int with_load = argc > 3 ? atoi(argv[3]) : 1;
    float *x, *y; unsigned* err;
    CK(hipMalloc(&x, n * sizeof(float))); CK(hipMalloc(&y, 64 << 20)); CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4)); CK(hipMemset(y, 0, 64 << 20));
    hipStream_t sa, sb; CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
    std::atomic<bool> stop{false};
    std::thread load([&] {
        if (!with_load) return;
        int k = 0;
        while (!stop) {
            hipLaunchKernelGGL(busy, dim3(200 + (k * 37) % 400), dim3(256), 0, sb, y, 1 << 20, 1 + k % 3); ++k;
            if (k % 64 == 0) (void)hipStreamSynchronize(sb);
        }
        (void)hipStreamSynchronize(sb);
    });
    for (int it = 0; it < iters; ++it) {
        hipLaunchKernelGGL(writer, dim3(16 + it % 5), dim3(256), 0, sa, x, n, (float)it);
        hipLaunchKernelGGL(reader, dim3(128 + it % 7), dim3(256), 0, sa, x, n, (float)it, err);
    }
    CK(hipStreamSynchronize(sa));
    stop = true; load.join();
    unsigned h = 0; CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
    printf("n=%d iters=%d load=%d: %u stale reads\n", n, iters, with_load, h);
    return 0;
}
