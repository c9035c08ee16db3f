// When does the host see a flag that a kernel writes to mapped pinned memory at its START: at once, at the end of that
// kernel, or only when the stream drains?  (The poly engine's host follows the device through such a mailbox.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct Mail { volatile int seq; int pad[15]; };
__global__ void k_pub_then_spin(Mail *m, int seq, long long spin_ticks)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) { m->pad[0] = seq; __threadfence_system(); m->seq = seq; }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin_ticks) { }
}
__global__ void k_spin(long long spin_ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin_ticks) { }
}
int main()
{
    Mail *mh, *md;
    CK(hipHostMalloc(&mh, sizeof(Mail), hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void **)&md, mh, 0));
    mh->seq = 0;
    hipStream_t s; CK(hipStreamCreate(&s));
    using clk = std::chrono::steady_clock;
    for (int follow : {0, 1, 3}) {
        double sum_seen = 0, sum_done = 0;
        const int reps = 200;
        for (int r = 1; r <= reps; r++) {
            const int seq = follow * 1000 + r;
            auto t0 = clk::now();
            hipLaunchKernelGGL(k_pub_then_spin, dim3(64), dim3(256), 0, s, md, seq, 1500LL);          // 15 us
            for (int f = 0; f < follow; f++) hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s, 1500LL);
            while (mh->seq != seq) { }
            auto t1 = clk::now();
            CK(hipStreamSynchronize(s));
            auto t2 = clk::now();
            sum_seen += std::chrono::duration<double, std::micro>(t1 - t0).count();
            sum_done += std::chrono::duration<double, std::micro>(t2 - t0).count();
        }
        printf("publisher (15 us) followed by %d kernels of 15 us: flag seen after %.1f us, stream done after %.1f us\n", follow, sum_seen / reps, sum_done / reps);
    }
    return 0;
}
