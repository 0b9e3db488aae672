// What does one dependent kernel boundary cost on this box?  N back-to-back launches on one stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void empty_k(float* p) { if (p == nullptr) p[0] = 1.f; }
__global__ void touch_k(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
int main() {
    float* d; hipMalloc(&d, 64 << 20);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = 2000;
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 100; ++i) launch();
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        for (int i = 0; i < N; ++i) launch();
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-40s %.2f us per launch\n", name, ms * 1e3 / N);
    };
    run("empty <<<1,64>>>", [&] { hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, st, d); });
    run("empty <<<256,256>>>", [&] { hipLaunchKernelGGL(empty_k, dim3(256), dim3(256), 0, st, d); });
    run("empty <<<256,512>>>", [&] { hipLaunchKernelGGL(empty_k, dim3(256), dim3(512), 0, st, d); });
    run("touch 32KB <<<32,256>>>", [&] { hipLaunchKernelGGL(touch_k, dim3(32), dim3(256), 0, st, d, 8192); });
    run("touch 4MB <<<4096,256>>>", [&] { hipLaunchKernelGGL(touch_k, dim3(4096), dim3(256), 0, st, d, 1 << 20); });
    // graph of 100 dependent empty kernels
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(touch_k, dim3(32), dim3(256), 0, st, d, 8192);
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < 20; ++i) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %.2f us per kernel node\n", "graph: 100 x touch 32KB", ms * 1e3 / 2000);
    return 0;
}
