// Dependent-load latency on MI355X for random gathers in arrays of various sizes (design probe: the per-cut
// kernels of the polyhedron engine are chains of 3-4 dependent gathers).  One wave, each lane chases its own chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void chase(const unsigned *next, int steps, unsigned *out, unsigned long long *ticks, unsigned stride)
{
    unsigned p = (blockIdx.x * blockDim.x + threadIdx.x) * stride;
    unsigned long long t0 = wall_clock64();
    for (int s = 0; s < steps; s++) p = next[p];
    unsigned long long t1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = p;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
int main()
{
    unsigned long long *ticks; unsigned *out;
    CK(hipMalloc(&ticks, 64)); CK(hipMalloc(&out, 1 << 20));
    for (size_t mb : {1, 16, 256, 4096, 32768}) {
        size_t n = mb * 1024 * 1024 / 4;
        std::vector<unsigned> h(n);
        // random cyclic permutation in blocks (cheap to build): p -> (a*p + c) mod n with odd a is a permutation for n = 2^k
        unsigned a = 1664525u, c = 1013904223u;
        for (size_t i = 0; i < n; i++) h[i] = (unsigned)((a * (unsigned long long)i + c) & (n - 1));
        unsigned *d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
        for (int threads : {64, 4096, 65536}) {
            const int steps = 64;
            unsigned long long t = 0;
            for (int rep = 0; rep < 3; rep++) {
                hipLaunchKernelGGL(chase, dim3((threads + 255) / 256), dim3(threads < 256 ? threads : 256), 0, 0, d, steps, out, ticks, (unsigned)(n / threads / 2 + 1));
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
            }
            printf("array %6zu MB, %6d lanes chasing: %.0f ns per dependent load\n", mb, threads, t * 10.0 / steps);
        }
        CK(hipFree(d));
    }
    return 0;
}
