// boundary_probe.hip -- what stretches the distance between DEPENDENT launches of a small-kernel chain while another stream runs
// big kernels?  Chain: N dependent launches of a tiny kernel (128 workgroups, one load + one store each).  Background, on its own
// stream, back to back: (a) nothing, (b) ALU only (every CU busy, no memory), (c) reads only, (d) plain stores (dirty lines in
// L2), (e) write-through stores (sc1: no dirty lines).  If the kernel-boundary release (L2 write-back) is what the chain pays for,
// (d) stretches the chain and (e) does not.
//   hipcc -O3 --offload-arch=gfx950 boundary_probe.hip -o boundary_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <atomic>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NREG>
__global__ __launch_bounds__(512) void tiny(const float* __restrict__ in, float* __restrict__ out) {
    extern __shared__ float tiny_lds[];                  // dynamic LDS like a decode-step workgroup's (17-66 KB): decides co-residency
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) & (128 * 256 - 1);
    float r[NREG];                                       // NREG live registers (a decode GEMV holds 80-128)
#pragma unroll
    for (int k = 0; k < NREG; ++k) r[k] = in[(i + k * 64) & (128 * 256 - 1)];
    tiny_lds[threadIdx.x] = r[0];
    __syncthreads();
    float s = tiny_lds[threadIdx.x ^ 1];
#pragma unroll
    for (int k = 0; k < NREG; ++k) s = s * 1.0001f + r[k];
    out[i] = s;
}
// MODE 0 ALU, 1 reads, 2 plain stores, 3 sc1 (write-through) stores, 4 nt stores
template <int MODE>
__global__ __launch_bounds__(512) void bg(float4* __restrict__ buf, size_t per_wg_vec, int iters, float* sink) {
    extern __shared__ float lds_pad[];
    if (threadIdx.x == 0) lds_pad[0] = 1.0f;
    float4* p = buf + (size_t)blockIdx.x * per_wg_vec;
    float4 acc = make_float4(threadIdx.x, 1.f, 2.f, 3.f);
    for (int it = 0; it < (MODE >= 5 ? 0 : iters); ++it) {
        for (size_t i = threadIdx.x; i < per_wg_vec; i += 512) {
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < 16; ++k) { acc.x = acc.x * 1.0001f + acc.y; acc.y = acc.y * 0.9999f + acc.z; }
            } else if (MODE == 1) {
                const float4 v = p[i];
                acc.x += v.x + v.w;
            } else if (MODE == 2) {
                p[i] = acc;
            } else if (MODE == 3) {
                typedef float f4v __attribute__((ext_vector_type(4)));
                const f4v vv = {acc.x, acc.y, acc.z, acc.w};
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p + i), "v"(vv) : "memory");
            } else {
                __builtin_nontemporal_store(acc.x, &p[i].x); __builtin_nontemporal_store(acc.y, &p[i].y);
                __builtin_nontemporal_store(acc.z, &p[i].z); __builtin_nontemporal_store(acc.w, &p[i].w);
            }
        }
    }
    if (MODE >= 5) {   // a render-like kernel: ~20 us of arithmetic, then ONE pass of output stores (plain: dirty lines stay in L2; sc1: write-through)
        for (int it = 0; it < iters; ++it)
            for (size_t i = threadIdx.x; i < 4096; i += 512) {
#pragma unroll
                for (int k = 0; k < 16; ++k) { acc.x = acc.x * 1.0001f + acc.y; acc.y = acc.y * 0.9999f + acc.z; }
            }
        for (size_t i = threadIdx.x; i < per_wg_vec; i += 512) {
            if (MODE == 5) p[i] = acc;
            else {
                typedef float f4v __attribute__((ext_vector_type(4)));
                const f4v vv = {acc.x, acc.y, acc.z, acc.w};
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p + i), "v"(vv) : "memory");
            }
        }
    }
    if (acc.x == 12345.678f) *sink = acc.y;
}

// ALU-only background whose waves hold ~200 live VGPRs (2 waves per SIMD = 400 of the 512 registers per lane, like tfm_attn_fused)
__global__ __launch_bounds__(512, 2) void bg_regs_kernel(const float* __restrict__ in, int iters, float* sink) {
    extern __shared__ float lds_pad2[];
    if (threadIdx.x == 0) lds_pad2[0] = 1.0f;
    float r[192];
#pragma unroll
    for (int k = 0; k < 192; ++k) r[k] = in[(threadIdx.x + k * 64) & 32767];
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int k = 0; k < 192; ++k) r[k] = r[k] * 1.0001f + r[(k + 1) % 192];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 192; ++k) s += r[k];
    if (s == 12345.678f) *sink = s;
}

template <int MODE>
static void launch_bg(hipStream_t sb, float4* big, size_t per_wg, int iters, float* sink, size_t lds) {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bg<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    hipLaunchKernelGGL(bg<MODE>, dim3(256), dim3(512), lds, sb, big, per_wg, iters, sink);
}
static const float* g_in = nullptr;
static void launch_mode(int mode, hipStream_t sb, float4* big, size_t per_wg, int alu_iters, int mem_iters, float* sink, size_t lds) {
    if (mode == 7) {
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bg_regs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
        hipLaunchKernelGGL(bg_regs_kernel, dim3(256), dim3(512), lds, sb, g_in, alu_iters * 4, sink);
        return;
    }
    switch (mode) {
        case 0: launch_bg<0>(sb, big, per_wg, alu_iters, sink, lds); break;
        case 1: launch_bg<1>(sb, big, per_wg, mem_iters, sink, lds); break;
        case 2: launch_bg<2>(sb, big, per_wg, mem_iters, sink, lds); break;
        case 3: launch_bg<3>(sb, big, per_wg, mem_iters, sink, lds); break;
        case 4: launch_bg<4>(sb, big, per_wg, mem_iters, sink, lds); break;
        case 5: launch_bg<5>(sb, big, per_wg, alu_iters, sink, lds); break;
        default: launch_bg<6>(sb, big, per_wg, alu_iters, sink, lds); break;
    }
}

int main(int argc, char** argv) {
    const int n_chain = argc > 1 ? atoi(argv[1]) : 4000;
    const size_t mb = argc > 2 ? atoi(argv[2]) : 16;           // MB written / read per background launch
    const size_t lds = (argc > 3 ? atoi(argv[3]) : 150) * 1024; // dynamic LDS of a background workgroup (150 KB: nothing co-resides)
    const int mem_iters = argc > 4 ? atoi(argv[4]) : 4, alu_iters = argc > 5 ? atoi(argv[5]) : 12;
    const size_t tiny_lds_bytes = (argc > 6 ? atoi(argv[6]) : 1) * 1024;     // LDS of a chain workgroup
    const bool big_regs = argc > 7 && atoi(argv[7]) != 0;   // chain waves with ~100 live VGPRs
    const bool bg_regs = argc > 8 && atoi(argv[8]) != 0;    // background waves with ~200 live VGPRs (mode 7)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tiny<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tiny<96>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    float *a, *b, *sink;
    CK(hipMalloc(&a, 128 * 256 * 4)); CK(hipMalloc(&b, 128 * 256 * 4)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 0, 128 * 256 * 4));
    float4* big;
    CK(hipMalloc(&big, mb << 20));
    hipStream_t sc, sb;
    CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const size_t per_wg = (mb << 20) / 16 / 256;
    auto chain = [&]() {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n_chain / 2; ++i) { if (big_regs) { hipLaunchKernelGGL(tiny<96>, dim3(128), dim3(512), tiny_lds_bytes, sc, a, b); hipLaunchKernelGGL(tiny<96>, dim3(128), dim3(512), tiny_lds_bytes, sc, b, a); } else { hipLaunchKernelGGL(tiny<8>, dim3(128), dim3(512), tiny_lds_bytes, sc, a, b); hipLaunchKernelGGL(tiny<8>, dim3(128), dim3(512), tiny_lds_bytes, sc, b, a); } }
        CK(hipStreamSynchronize(sc));
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n_chain;
    };
    chain();
    printf("chain alone: %.2f us per dependent launch (chain workgroups: %zu KB LDS; background workgroups: 512 threads, %zu KB LDS)\n", chain(), tiny_lds_bytes >> 10, lds >> 10);
    g_in = a;
    const char* names[8] = {"ALU only", "reads", "plain stores", "sc1 write-through stores", "nt stores", "ALU + one plain-store pass", "ALU + one sc1-store pass",
                            "ALU only, ~200 VGPRs per wave"};
    for (int mode = 0; mode < 8; ++mode) {
        if (mode == 7 && !bg_regs) continue;
        // background kernel duration alone
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int k = 0; k < 5; ++k) launch_mode(mode, sb, big, per_wg, alu_iters, mem_iters, sink, lds);
        CK(hipEventRecord(e0, sb));
        for (int k = 0; k < 20; ++k) launch_mode(mode, sb, big, per_wg, alu_iters, mem_iters, sink, lds);
        CK(hipEventRecord(e1, sb)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double each = ms * 1e3 / 20;
        // enough background launches queued ahead to cover the chain (the chain is ~n_chain x 10 us at worst)
        const int n_bg = (int)(n_chain * 12.0 / each) + 50;
        std::thread th([&] { for (int k = 0; k < n_bg; ++k) launch_mode(mode, sb, big, per_wg, alu_iters, mem_iters, sink, lds); });
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
        const double us = chain();
        th.join();
        const bool still = hipStreamQuery(sb) == hipErrorNotReady;
        CK(hipStreamSynchronize(sb));
        printf("chain beside %-26s (%zu MB x %d per launch, %.1f us each, background %s): %.2f us per dependent launch\n", names[mode], mb,
               mode ? mem_iters : alu_iters, each, still ? "outlasted the chain" : "ENDED EARLY", us);
    }
    return 0;
}
