#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
__global__ void k(const _Float16* __restrict__ src, _Float16* __restrict__ dst) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // each wave: 2 instructions, instruction j writes 1 KiB at smem + (wid*2 + j) * 1024; lane l sources chunk (63 - l)
    for (int j = 0; j < 2; ++j) {
        const _Float16* g = src + ((wid * 2 + j) * 64 + (63 - lane)) * 8;
        __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)(smem + (wid * 2 + j) * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int j = 0; j < 2; ++j) {
        half8 v = *reinterpret_cast<const half8*>(smem + (wid * 2 + j) * 1024 + lane * 16);
        *reinterpret_cast<half8*>(dst + ((wid * 2 + j) * 64 + lane) * 8) = v;
    }
}
int main() {
    const int n = 4 * 2 * 64 * 8;
    _Float16 *h = (_Float16*)malloc(n * 2), *o = (_Float16*)malloc(n * 2), *d, *e;
    for (int i = 0; i < n; ++i) h[i] = (_Float16)(i / 8);
    hipMalloc(&d, n * 2); hipMalloc(&e, n * 2); hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 8192, 0, d, e);
    hipMemcpy(o, e, n * 2, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 8; ++w) for (int l = 0; l < 64; ++l) for (int q = 0; q < 8; ++q) {
        float want = (float)(w * 64 + (63 - l)); float got = (float)o[(w * 64 + l) * 8 + q];
        if (want != got) { if (bad < 5) printf("w %d l %d q %d want %g got %g\n", w, l, q, want, got); ++bad; }
    }
    printf("glds check: %d mismatches\n", bad);
    return 0;
}
