// How long after the end of a kernel on stream A does a dependent kernel on stream B start?  (the join of the forward's lanes)
//   1. same stream (in-order)            2. hipEventRecord(A) + hipStreamWaitEvent(B)
//   3. hipStreamWriteValue32(A) + hipStreamWaitValue32(B) on device memory       4. the same on pinned host memory
// Kernels stamp wall_clock64() (100 MHz): A its end, B its start.  build: hipcc -O2 --offload-arch=gfx950 -o xq_sync_probe xq_sync_probe.hip
// Measured (MI355X, ROCm 7.2): 1.1 / 10.5 / 4.1 / 5.3 us.  The forward's joins built from write / wait values were nevertheless slower
// than the event joins and can stall when two nets' streams share a hardware queue ("The joins").
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void busy(unsigned long long *stamp_end, int spin)
{
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin) {}
    __syncthreads();
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) atomicMax(stamp_end, wall_clock64());
}
__global__ void mark(unsigned long long *stamp_start)
{
    if (threadIdx.x == 0) atomicMin(stamp_start, wall_clock64());
}
int main()
{
    hipStream_t A, B;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    unsigned long long *st;
    CK(hipMalloc((void **)&st, 16));
    unsigned *flag_dev, *flag_host;
    CK(hipMalloc((void **)&flag_dev, 4));
    CK(hipHostMalloc((void **)&flag_host, 4, hipHostMallocDefault));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    int khz = 0;
    CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0));
    const char *names[4] = {"same stream", "event record + stream wait", "write/wait value32, device memory", "write/wait value32, pinned host memory"};
    for (int mode = 0; mode < 4; ++mode) {
        std::vector<double> gaps;
        for (int it = 0; it < 60; ++it) {
            unsigned long long init[2] = {0ull, ~0ull};
            CK(hipMemcpy(st, init, 16, hipMemcpyHostToDevice));
            CK(hipMemset(flag_dev, 0, 4));
            *flag_host = 0;
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, A, st, 2000 + 37 * (it % 7));  // ~20 us
            if (mode == 0) {
                hipLaunchKernelGGL(mark, dim3(256), dim3(256), 0, A, st + 1);
            } else if (mode == 1) {
                CK(hipEventRecord(ev, A));
                CK(hipStreamWaitEvent(B, ev, 0));
                hipLaunchKernelGGL(mark, dim3(256), dim3(256), 0, B, st + 1);
            } else {
                unsigned *f = mode == 2 ? flag_dev : flag_host;
                CK(hipStreamWriteValue32(A, f, 1u, 0));
                CK(hipStreamWaitValue32(B, f, 1u, hipStreamWaitValueEq, 0xffffffffu));
                hipLaunchKernelGGL(mark, dim3(256), dim3(256), 0, B, st + 1);
            }
            CK(hipDeviceSynchronize());
            unsigned long long r[2];
            CK(hipMemcpy(r, st, 16, hipMemcpyDeviceToHost));
            gaps.push_back(((double)r[1] - (double)r[0]) / khz * 1e3);
        }
        std::sort(gaps.begin(), gaps.end());
        printf("%-42s gap end(A) -> start(B): min %6.1f  median %6.1f  p90 %6.1f us\n", names[mode], gaps[0], gaps[gaps.size() / 2], gaps[gaps.size() * 9 / 10]);
    }
    return 0;
}
