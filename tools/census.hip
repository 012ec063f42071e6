// census.hip -- how many 1024-thread workgroups with L bytes of LDS are co-resident per CU?
// Each workgroup idles ~50 us and records its start tick and hardware id.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int LDS, int THREADS>
__global__ __launch_bounds__(THREADS) void k(unsigned long long* t0s, unsigned long long* t1s, unsigned* hw, unsigned* sink) {
    __shared__ unsigned buf[LDS / 4];
    // light workgroups: groups of 64 heavy are followed by 192 that only look at one word and leave
    if ((blockIdx.x >> 6) & 3) {
        if (sink[1 + (blockIdx.x & 1023)] == 12345u) sink[0] = 1;
        if (threadIdx.x == 0) { t0s[blockIdx.x] = 0; t1s[blockIdx.x] = 0; }
        return;
    }
    buf[threadIdx.x % (LDS / 4)] = threadIdx.x;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 5000ull) __builtin_amdgcn_s_sleep(16);
    if (threadIdx.x == 0) {
        t0s[blockIdx.x] = t0;
        t1s[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        unsigned id, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        hw[blockIdx.x] = (id & 0xFFFFF) | (xcc << 24);
        sink[0] = buf[5];
    }
}
template <int LDS, int THREADS>
void run(int nwg) {
    unsigned long long *t0, *t1; unsigned *hw, *sink;
    hipMalloc(&t0, nwg * 8); hipMalloc(&t1, nwg * 8); hipMalloc(&hw, nwg * 4); hipMalloc(&sink, 8192); hipMemset(sink, 0, 8192);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<LDS, THREADS>), dim3(nwg), dim3(THREADS), 0, 0, t0, t1, hw, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<LDS, THREADS>), dim3(nwg), dim3(THREADS), 0, 0, t0, t1, hw, sink);
    hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h0(nwg), h1(nwg);
    hipMemcpy(h0.data(), t0, nwg * 8, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), t1, nwg * 8, hipMemcpyDeviceToHost);
    // max concurrency: sweep
    std::vector<std::pair<unsigned long long, int>> ev;
    for (int i = 0; i < nwg; ++i) { if (!h0[i]) continue; ev.push_back({h0[i], 1}); ev.push_back({h1[i], -1}); }
    std::sort(ev.begin(), ev.end());
    int cur = 0, mx = 0; for (auto& e : ev) { cur += e.second; mx = std::max(mx, cur); }
    printf("LDS %6d B  threads %4d  wgs %5d  time %.3f ms  rounds(50us) %.1f  max concurrent %d\n", LDS, THREADS, nwg, ms, ms / 0.05, mx);
    hipFree(t0); hipFree(t1); hipFree(hw); hipFree(sink);
}
int main() {
    run<4096, 1024>(16384);
    run<75264, 1024>(16384);
    run<75264, 512>(16384);
    run<4096, 256>(16384);
    return 0;
}
