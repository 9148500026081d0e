// Which SIMD does workgroup b of a 4096 x 64-thread launch with 10 KB LDS land on?  (placement probe for the env -> block
// balancing experiment; prints per-XCC/SE/CU/SIMD occupancy statistics and the block ids sharing the first SIMDs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
__global__ void __launch_bounds__(64) k(unsigned* out, int spin) {
    __shared__ float lds[2544];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) { a = a * 1.0001f + 0.5f; lds[(threadIdx.x + i) % 2544] = a; }
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; }
    if (a == 12345.f) out[0] = lds[5];
}
int main() {
    const int N = 4096; unsigned* d; hipMalloc(&d, N * 8);
    std::vector<unsigned> h(N * 2);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(N), dim3(64), 0, 0, d, 20000); hipDeviceSynchronize();
        hipMemcpy(h.data(), d, N * 8, hipMemcpyDeviceToHost);
        std::map<unsigned long long, std::vector<int>> simd;
        for (int b = 0; b < N; ++b) {
            unsigned hw = h[b * 2], xcc = h[b * 2 + 1] & 0xf;
            unsigned wave = hw & 0xf, simd_id = (hw >> 4) & 0x3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
            unsigned long long key = ((unsigned long long)xcc << 24) | (se << 16) | (sh << 12) | (cu << 4) | simd_id;
            simd[key].push_back(b);
        }
        std::map<int, int> hist; for (auto& kv : simd) hist[(int)kv.second.size()]++;
        printf("rep %d: distinct SIMDs %zu; waves-per-SIMD histogram:", rep, simd.size());
        for (auto& kv : hist) printf(" %d:%d", kv.first, kv.second);
        printf("\n");
        int shown = 0;
        for (auto& kv : simd) { if (shown++ >= 6) break; printf("  simd %llx:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); }
        // how often do blocks b and b+1024 share a SIMD? and b, b+1?
        std::map<int, unsigned long long> where; for (auto& kv : simd) for (int b : kv.second) where[b] = kv.first;
        for (int d2 : {1, 8, 256, 512, 1024, 2048}) { int same = 0, tot = 0; for (int b = 0; b + d2 < N; ++b) { ++tot; same += where[b] == where[b + d2]; } printf("  P(same SIMD | delta=%d) = %.3f\n", d2, (double)same / tot); }
    }
    return 0;
}
