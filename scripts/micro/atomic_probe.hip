// atomic_probe.hip -- what does a deterministic (integer) cross-workgroup reduction cost?  W workgroups of 512 threads each add a
// [rows x 1024] slab of int64 values into ONE shared [rows x 1024] accumulator with device-scope atomics (W adds per address), the way a
// row-parallel GEMV would publish its partial sums.   hipcc -O3 --offload-arch=gfx950 atomic_probe.hip -o atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(512) void add_slab(unsigned long long* acc, int n, long long v) {
    for (int i = threadIdx.x; i < n; i += 512) atomicAdd(acc + i, (unsigned long long)(v + i + blockIdx.x));
}
__global__ __launch_bounds__(512) void add_slab_f32(float* acc, int n, float v) {
    for (int i = threadIdx.x; i < n; i += 512) atomicAdd(acc + i, v + (float)i);
}
__global__ void empty_k(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

int main() {
    unsigned long long* acc; float* accf;
    CK(hipMalloc(&acc, 32 * 1024 * 8)); CK(hipMalloc(&accf, 32 * 1024 * 4));
    CK(hipMemset(acc, 0, 32 * 1024 * 8)); CK(hipMemset(accf, 0, 32 * 1024 * 4));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, int wgs, int rows, bool f32) {
        const int n = rows * 1024, reps = 200;
        for (int i = 0; i < 20; ++i) { if (f32) hipLaunchKernelGGL(add_slab_f32, dim3(wgs), dim3(512), 0, st, accf, n, 1.0f); else hipLaunchKernelGGL(add_slab, dim3(wgs), dim3(512), 0, st, acc, n, 3LL); }
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) { if (f32) hipLaunchKernelGGL(add_slab_f32, dim3(wgs), dim3(512), 0, st, accf, n, 1.0f); else hipLaunchKernelGGL(add_slab, dim3(wgs), dim3(512), 0, st, acc, n, 3LL); }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-10s %3d workgroups x %2d rows x 1024 %s atomics (%d per address): %.2f us per launch (back to back, incl. the boundary)\n", name, wgs, rows,
               f32 ? "f32" : "i64", wgs, ms * 1e3 / reps);
    };
    {   // the boundary alone
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(empty_k, dim3(64), dim3(512), 0, st, (int*)nullptr);
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_k, dim3(64), dim3(512), 0, st, (int*)nullptr);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("empty kernel, 64 workgroups: %.2f us per launch\n", ms * 1e3 / 200);
    }
    for (int rows : {8, 2, 32}) {
        timeit("ffn-like", 64, rows, false);
        timeit("ffn-like", 128, rows, false);
        timeit("attn-like", 16, rows, false);
        timeit("f32", 64, rows, true);
    }
    return 0;
}
