// hbm_copy.hip -- round 6: what a plain device-to-device copy reaches on the MI355X at the SIZES of the stage kernels (tools/bench_chain.py: 70 ... 1460 MB
// of traffic per launch, launches of 20 ... 330 us), timed the way bench_chain times them (8 launches back to back between one event pair, median of 10 groups).
// The stage kernels' "fraction of 8 TB/s" is to be read against these figures, not against the peak: a short launch pays its ramp and its drain.
//   hipcc --offload-arch=gfx950 -O3 -o hbm_copy tools/ubench/hbm_copy.hip && ./hbm_copy
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ __launch_bounds__(256) void copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    // one 16-byte word per lane and trip, four trips in flight (the stage kernels' shape: short-lived waves, a few loads each)
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = i + 256 * k < n ? src[i + 256 * k] : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i + 256 * k < n) dst[i + 256 * k] = v[k];
}
__global__ __launch_bounds__(256) void copy16_persistent(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
int main() {
    const size_t maxb = (size_t)1 << 30;
    uint4 *a, *b;
    if (hipMalloc(&a, maxb) != hipSuccess || hipMalloc(&b, maxb) != hipSuccess) return 1;
    (void)hipMemset(a, 1, maxb); (void)hipMemset(b, 0, maxb);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    // traffic (read + write) of: CRC check, encode, rate match, rate recovery, channel, rate recovery + HARQ, and a large copy
    const double traffic_mb[] = {69, 141, 207, 352, 628, 1458, 2000};
    for (double mb : traffic_mb) {
        const size_t bytes = (size_t)(mb * 1e6 / 2) & ~(size_t)4095, n = bytes / 16;
        for (int variant = 0; variant < 2; ++variant) {
            std::vector<float> ms;
            for (int rep = 0; rep < 11; ++rep) {
                (void)hipEventRecord(e0);
                for (int k = 0; k < 8; ++k) {
                    if (variant == 0) hipLaunchKernelGGL(copy16, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, a, b, n);
                    else hipLaunchKernelGGL(copy16_persistent, dim3(256 * 8), dim3(256), 0, 0, a, b, n);
                }
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float t; (void)hipEventElapsedTime(&t, e0, e1);
                if (rep) ms.push_back(t / 8);
            }
            std::sort(ms.begin(), ms.end());
            const double t = ms[ms.size() / 2];
            printf("%-28s traffic %7.0f MB  %.4f ms per launch  %6.0f GB/s  %.3f of 8 TB/s\n", variant == 0 ? "copy (4 x 16 B per lane)" : "copy (persistent, 2048 WGs)",
                   2.0 * bytes / 1e6, t, 2.0 * bytes / t / 1e6, 2.0 * bytes / t / 1e6 / 8000.0);
        }
    }
    return 0;
}
