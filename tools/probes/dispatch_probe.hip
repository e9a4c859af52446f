// How fast does the chip start workgroups?  Each workgroup stamps the 100 MHz chip clock at entry and spins ~8 us.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__device__ long long g_t[16384];
template <int NT>
__global__ __launch_bounds__(NT) void k(int spin_ticks)
{
    extern __shared__ char lds[];
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) g_t[blockIdx.x] = t0;
    if (spin_ticks < 0) lds[threadIdx.x] = 1;  // keep the allocation
    while (wall_clock64() - t0 < spin_ticks) {}
}
// same, but the wave needs ~200 VGPRs (like the conv kernels)
__global__ __launch_bounds__(256, 1) void kfat(int spin_ticks, float *sink)
{
    extern __shared__ char lds[];
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) g_t[blockIdx.x] = t0;
    if (spin_ticks < 0) lds[threadIdx.x] = 1;
    float a[192];
#pragma unroll
    for (int i = 0; i < 192; ++i) a[i] = (float)(threadIdx.x * (i + 1));
    while (wall_clock64() - t0 < spin_ticks) {
#pragma unroll
        for (int i = 0; i < 192; ++i) a[i] = a[i] * 1.0001f + a[(i + 7) % 192];
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 192; ++i) s += a[i];
    if (s == 12345.f) *sink = s;
}
void run_fat(int nwg, int ldsb, const char *name)
{
    float *sink; hipMalloc(&sink, 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(kfat), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kfat, dim3(nwg), dim3(256), ldsb, 0, 800, sink); hipDeviceSynchronize(); }
    std::vector<long long> t(nwg);
    hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_t), nwg * 8);
    std::sort(t.begin(), t.end());
    printf("%-34s %5d workgroups: 25%% started after %5.2f us, 50%% %5.2f, 75%% %5.2f, last %5.2f us\n", name, nwg, (t[nwg / 4] - t[0]) / 100.0,
           (t[nwg / 2] - t[0]) / 100.0, (t[3 * nwg / 4] - t[0]) / 100.0, (t[nwg - 1] - t[0]) / 100.0);
}
template <int NT>
void run(int nwg, int ldsb, const char *name)
{
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<NT>, dim3(nwg), dim3(NT), ldsb, 0, 800); hipDeviceSynchronize(); }
    std::vector<long long> t(nwg);
    hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_t), nwg * 8);
    std::sort(t.begin(), t.end());
    printf("%-34s %5d workgroups: 25%% started after %5.2f us, 50%% %5.2f, 75%% %5.2f, last %5.2f us\n", name, nwg, (t[nwg / 4] - t[0]) / 100.0,
           (t[nwg / 2] - t[0]) / 100.0, (t[3 * nwg / 4] - t[0]) / 100.0, (t[nwg - 1] - t[0]) / 100.0);
}
int main()
{
    run<256>(512, 64064, "256 thr, 64 KB LDS");
    run<256>(512, 0, "256 thr, no LDS");
    run<256>(512, 32768, "256 thr, 32 KB LDS");
    run<256>(256, 64064, "256 thr, 64 KB LDS");
    run<512>(256, 131072, "512 thr, 128 KB LDS");
    run<512>(256, 65536, "512 thr, 64 KB LDS");
    run<1024>(256, 65536, "1024 thr, 64 KB LDS");
    run<64>(2048, 16384, "64 thr, 16 KB LDS");
    run<256>(2048, 16384, "256 thr, 16 KB LDS");
    run_fat(512, 64064, "256 thr, 64 KB LDS, ~200 VGPRs");
    run_fat(512, 0, "256 thr, no LDS, ~200 VGPRs");
    run_fat(256, 64064, "256 thr, 64 KB LDS, ~200 VGPRs");
    return 0;
}
