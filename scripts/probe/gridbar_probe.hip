// Latency of a grid-wide barrier on MI355X for persistent cooperative kernels (design probe for the
// one-launch-per-batch cut pipeline).  Usage: gridbar_probe
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdlib>
namespace cg = cooperative_groups;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void grid_bar(unsigned *ctr, unsigned &target, unsigned G)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        target += G;
        __threadfence();
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 50000000) break;          // never hang the box
        }
        __threadfence();
    }
    __syncthreads();
}
__global__ void k_mine(unsigned *ctr, int iters, int *data, int n)
{
    unsigned target = 0;
    const unsigned G = gridDim.x;
    for (int it = 0; it < iters; it++) {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += G * blockDim.x) data[i] += 1;
        grid_bar(ctr, target, G);
    }
}
__global__ void k_cg(int iters, int *data, int n)
{
    cg::grid_group g = cg::this_grid();
    for (int it = 0; it < iters; it++) {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) data[i] += 1;
        g.sync();
    }
}
__global__ void k_tiny(int *data, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) data[i] += 1;
}
int main()
{
    unsigned *ctr; int *data; const int n = 1 << 16;
    CK(hipMalloc(&ctr, 64)); CK(hipMalloc(&data, n * 4)); CK(hipMemset(data, 0, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    int Gs[] = {8, 16, 32, 64, 128, 256, 512};
    for (int T : {256, 1024}) for (int G : Gs) {
        if (G * T > 256 * 2048) continue;
        float ms1 = 0, ms2 = 0;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemset(ctr, 0, 64));
            int it = iters, nn = n;
            void *a1[] = {&ctr, &it, &data, &nn};
            CK(hipEventRecord(e0));
            CK(hipLaunchCooperativeKernel((void *)k_mine, dim3(G), dim3(T), a1, 0, 0));
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms1, e0, e1));
            void *a2[] = {&it, &data, &nn};
            CK(hipEventRecord(e0));
            CK(hipLaunchCooperativeKernel((void *)k_cg, dim3(G), dim3(T), a2, 0, 0));
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms2, e0, e1));
        }
        printf("G=%4d T=%4d  own barrier %.2f us/iter   cg grid.sync %.2f us/iter\n", G, T, ms1 * 1e3 / iters, ms2 * 1e3 / iters);
    }
    float ms = 0;
    CK(hipEventRecord(e0));
    for (int it = 0; it < iters; it++) hipLaunchKernelGGL(k_tiny, dim3(n / 256), dim3(256), 0, 0, data, n);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("back-to-back tiny kernels: %.2f us/launch\n", ms * 1e3 / iters);
    return 0;
}
