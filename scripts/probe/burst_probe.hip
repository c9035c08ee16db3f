// Gap between two dependent kernels of one stream when the second one is enqueued (a) right behind the first, (b) 10 us
// later, while the first is still running (both long before the first ends).  Device timestamps (wall_clock64, 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <thread>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void k_spin(long long ticks, unsigned long long *t_end)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (blockIdx.x == 0 && threadIdx.x == 0) *t_end = wall_clock64();
}
__global__ void k_mark(unsigned long long *t_start)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *t_start = wall_clock64();
}
int main()
{
    unsigned long long *d; CK(hipMalloc(&d, 64));
    hipStream_t s; CK(hipStreamCreate(&s));
    using clk = std::chrono::steady_clock;
    for (int delay_us : {0, 10, 20}) {
        double sum = 0; const int reps = 300;
        for (int r = 0; r < reps; r++) {
            hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s, 4000LL, d);           // 40 us
            if (delay_us) { auto t0 = clk::now(); while (std::chrono::duration<double, std::micro>(clk::now() - t0).count() < delay_us) { } }
            hipLaunchKernelGGL(k_mark, dim3(64), dim3(256), 0, s, d + 1);
            CK(hipStreamSynchronize(s));
            unsigned long long t[2]; CK(hipMemcpy(t, d, 16, hipMemcpyDeviceToHost));
            sum += (double)(t[1] - t[0]) * 0.01;
        }
        printf("second kernel enqueued %2d us after the first (which runs 40 us): start of second - end of first = %.2f us\n", delay_us, sum / reps);
    }
    return 0;
}
