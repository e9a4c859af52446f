// Which clock pair do the start/stop events of hipExtLaunchKernelGGL carry?  A kernel that spins a known time is timed
// (a) with two different events, (b) with the SAME event passed as start and stop and read against itself / a later one,
// (c) by a hipEventRecord bracket, (d) by the kernel itself (wall_clock64); run it under `rocprofv3 --kernel-trace --stats`
// to get the dispatch-packet duration of the same launches.   build: hipcc --offload-arch=gfx950 -O2 -o ext_event_probe ext_event_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
__global__ void spin(unsigned long long ticks, unsigned long long *clk, float *sink)
{
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) atomicMin(clk, t0);
    float x = threadIdx.x;
    while (wall_clock64() - t0 < ticks) x = x * 1.0001f + 1.f;
    if (x == 12345.f) *sink = x;
    if (threadIdx.x == 0) atomicMax(clk + 1, wall_clock64());
}
int main()
{
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    const unsigned long long ticks = (unsigned long long)(18e-6 * khz * 1e3);  // ~18 us
    hipStream_t s;
    hipStreamCreateWithPriority(&s, hipStreamNonBlocking, -1);
    unsigned long long *clk; float *sink;
    const int N = 200;
    hipMalloc(&clk, N * 16); hipMalloc(&sink, 4);
    std::vector<unsigned long long> init(2 * N);
    for (int i = 0; i < N; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0; }
    std::vector<hipEvent_t> a(N), b(N), c(N), d(N), e(N);
    for (int i = 0; i < N; ++i) { hipEventCreate(&a[i]); hipEventCreate(&b[i]); hipEventCreate(&c[i]); hipEventCreate(&d[i]); hipEventCreate(&e[i]); }
    for (int mode = 0; mode < 4; ++mode) {
        hipMemcpy(clk, init.data(), N * 16, hipMemcpyHostToDevice);
        for (int i = 0; i < N; ++i) {
            if (mode == 0) hipExtLaunchKernelGGL(spin, dim3(512), dim3(256), 0, s, a[i], b[i], 0, ticks, clk + 2 * i, sink);
            if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(512), dim3(256), 0, s, (hipEvent_t) nullptr, c[i], 0, ticks, clk + 2 * i, sink);
            if (mode == 2) { hipEventRecord(d[i], s); hipLaunchKernelGGL(spin, dim3(512), dim3(256), 0, s, ticks, clk + 2 * i, sink); hipEventRecord(e[i], s); }
            if (mode == 3) hipExtLaunchKernelGGL(spin, dim3(512), dim3(256), 0, s, a[i], a[i], 0, ticks, clk + 2 * i, sink);
        }
        hipStreamSynchronize(s);
        std::vector<unsigned long long> h(2 * N);
        hipMemcpy(h.data(), clk, N * 16, hipMemcpyDeviceToHost);
        double dev = 0, ev = 0; int nev = 0;
        for (int i = 10; i < N; ++i) {
            dev += (double)(h[2 * i + 1] - h[2 * i]) / khz * 1e3;
            float ms = 0; hipError_t r = hipSuccess;
            if (mode == 0) r = hipEventElapsedTime(&ms, a[i], b[i]);
            if (mode == 1) r = hipEventElapsedTime(&ms, c[i - 1], c[i]);  // end(i-1) .. end(i) = gap + duration
            if (mode == 2) r = hipEventElapsedTime(&ms, d[i], e[i]);
            if (mode == 3) r = hipEventElapsedTime(&ms, a[i - 1], a[i]);
            if (r == hipSuccess) { ev += ms * 1e3; ++nev; }
        }
        const char *names[] = {"ext start!=stop", "ext stop only: end(i-1)..end(i)", "hipEventRecord bracket", "ext same event: (i-1)..(i)"};
        printf("%-34s events %.2f us (n=%d)   in-kernel %.2f us\n", names[mode], nev ? ev / nev : -1.0, nev, dev / (N - 10));
    }
    return 0;
}
