// How many workgroups of the decoder's footprint (384 threads, <=128 VGPRs, 53.5 KB LDS) does a CU hold at once?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
#include <algorithm>
struct Rec { uint64_t t0, t1; uint32_t hwid, xcc; };
template <int NV> __global__ __launch_bounds__(384, 4) void probe(Rec* out, int spin) {
    extern __shared__ char lds[];
    uint64_t t0 = __builtin_readcyclecounter();
    uint32_t hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = threadIdx.x * 0.5f + i;
    for (int s = 0; s < spin; ++s) {
#pragma unroll
        for (int i = 0; i < NV; ++i) acc[i] = acc[i] * 1.0001f + 0.5f;
        if ((s & 63) == 0) { ((float*)lds)[threadIdx.x] = acc[0]; __syncthreads(); }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) r += acc[i];
    ((float*)lds)[threadIdx.x] = r;
    __syncthreads();
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[blockIdx.x] = Rec{t0, t1, hwid, xcc}; if (((float*)lds)[5] == 12345.f) out[0].t0 = 0; }
}
template <int NV> void run(size_t lds, int grid) {
    Rec* d; (void)hipMalloc(&d, grid * sizeof(Rec));
    (void)hipFuncSetAttribute((const void*)probe<NV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(probe<NV>, dim3(grid), dim3(384), lds, 0, d, 3000);
    (void)hipDeviceSynchronize();
    std::vector<Rec> h(grid); (void)hipMemcpy(h.data(), d, grid * sizeof(Rec), hipMemcpyDeviceToHost);
    std::map<uint32_t, std::vector<std::pair<uint64_t,int>>> ev; // key = xcc<<16 | (se,cu bits)
    for (auto& r : h) { uint32_t cu = (r.hwid >> 8) & 0xF, sh = (r.hwid >> 12) & 1, se = (r.hwid >> 13) & 7; uint32_t key = (r.xcc & 0xF) << 16 | se << 8 | sh << 4 | cu;
        ev[key].push_back({r.t0, +1}); ev[key].push_back({r.t1, -1}); }
    int maxc = 0; double avg = 0; int n = 0; std::map<int,int> hist;
    for (auto& kv : ev) { auto& v = kv.second; std::sort(v.begin(), v.end()); int c = 0, m = 0; uint64_t last = v[0].first; double area = 0; for (auto& e : v) { area += (double)c * (e.first - last); last = e.first; c += e.second; m = std::max(m, c); }
        double span = (double)(v.back().first - v[0].first); avg += area / span; ++n; maxc = std::max(maxc, m); hist[m]++; }
    printf("NV=%d lds=%zu grid=%d: distinct CUs seen=%d, max concurrent WG/CU=%d, mean concurrent=%.2f ; per-CU max histogram:", NV, lds, grid, n, maxc, avg / n);
    for (auto& kv : hist) printf(" %d:%d", kv.first, kv.second);
    printf("\n");
    (void)hipFree(d);
}
int main() {
    run<100>(53520, 4096); run<100>(32768, 4096); run<100>(16384, 4096); run<60>(53520, 4096); run<60>(16384, 4096); run<100>(53520, 512);
    return 0;
}
