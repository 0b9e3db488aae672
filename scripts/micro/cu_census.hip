// cu_census.hip -- which CUs does a hipExtStreamCreateWithCUMask stream really get?
// One short-lived workgroup per slot writes (XCC id, SE, SH, CU) read from the hardware registers; the host prints, per mask
// pattern, how many distinct CUs were seen on every XCC.  Settles what bit i of the mask means on gfx950 (8 XCCs x 32 CUs).
//   hipcc -O2 --offload-arch=gfx950 cu_census.hip -o cu_census
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

__global__ void census(unsigned* out, long long ticks) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2] = hw;
        out[blockIdx.x * 2 + 1] = xcc;
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);     // hold the slot so that the grid spreads over every CU it may use
}

static void run(const char* name, hipStream_t st, unsigned* d, int blocks) {
    std::vector<unsigned> h(blocks * 2);
    CK(hipMemsetAsync(d, 0xff, blocks * 2 * sizeof(unsigned), st));
    hipLaunchKernelGGL(census, dim3(blocks), dim3(1024), 64 * 1024, st, d, 2000LL);      // 1024 threads + 64 KB: at most two per CU
    CK(hipGetLastError());
    CK(hipMemcpyAsync(h.data(), d, blocks * 2 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    std::map<int, std::set<int>> per_xcc;
    std::map<int, int> wg_per_xcc;
    int rr_ok = 0;
    for (int b = 0; b < blocks; ++b) {
        const unsigned hw = h[b * 2], xcc = h[b * 2 + 1] & 0xf;
        const int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        per_xcc[xcc].insert(se * 32 + sh * 16 + cu);
        wg_per_xcc[xcc]++;
        if (b >= 8 && (h[(b - 8) * 2 + 1] & 0xf) == xcc) ++rr_ok;
    }
    int total = 0;
    printf("%-28s", name);
    for (auto& kv : per_xcc) {
        printf(" x%d:%2zu(%d)", kv.first, kv.second.size(), wg_per_xcc[kv.first]);
        total += (int)kv.second.size();
    }
    printf("  total %d CUs; block b and b+8 on one XCC: %d of %d\n", total, rr_ok, blocks - 8);
    if (getenv("CENSUS_VERBOSE")) {
        for (auto& kv : per_xcc) {
            printf("    xcc %d:", kv.first);
            for (int id : kv.second) printf(" %d.%d.%d", id >> 5, (id >> 4) & 1, id & 15);
            printf("\n");
        }
    }
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("%s: %d CUs\n", p.name, p.multiProcessorCount);
    const int blocks = 1024;
    unsigned* d;
    CK(hipMalloc(&d, blocks * 2 * sizeof(unsigned)));
    hipStream_t s0;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    run("unmasked", s0, d, blocks);
    struct Pat { std::string name; std::vector<int> bits; };
    std::vector<Pat> pats;
    for (int n : {224, 192, 128, 64, 32}) {
        Pat lo{"low " + std::to_string(n), {}}, hi{"high " + std::to_string(n), {}};
        for (int i = 0; i < n; ++i) lo.bits.push_back(i), hi.bits.push_back(255 - i);
        pats.push_back(lo);
        pats.push_back(hi);
    }
    for (int k : {7, 6, 4, 2, 1}) {      // bits whose index modulo 8 is below k: whole XCCs if bit b belongs to XCC b % 8
        Pat m{"mod8 < " + std::to_string(k), {}};
        for (int i = 0; i < 256; ++i) if (i % 8 < k) m.bits.push_back(i);
        pats.push_back(m);
    }
    for (int k : {6, 2}) {               // contiguous blocks of 32 bits: whole XCCs if bit b belongs to XCC b / 32
        Pat m{"first " + std::to_string(k) + " words", {}};
        for (int i = 0; i < 32 * k; ++i) m.bits.push_back(i);
        pats.push_back(m);
    }
    for (auto& pt : pats) {
        uint32_t mask[8] = {0};
        for (int b : pt.bits) mask[b >> 5] |= 1u << (b & 31);
        hipStream_t st;
        hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, mask);
        if (e != hipSuccess) {
            printf("%-28s stream creation failed: %s\n", pt.name.c_str(), hipGetErrorString(e));
            continue;
        }
        run(pt.name.c_str(), st, d, blocks);
        CK(hipStreamDestroy(st));
    }
    return 0;
}
