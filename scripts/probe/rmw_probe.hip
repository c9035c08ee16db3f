// Practical HBM ceilings on MI355X for the access pattern of the tableau update: streaming read-modify-write in place
// (every double2 read, changed, written back), next to a pure read and a copy.  4 GiB working set.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void k_rmw(double2 *a, size_t n, double f)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = a[i]; v.x = fma(-f, 1.5, v.x); v.y = fma(-f, 2.5, v.y); a[i] = v; }
}
__global__ void k_copy(const double2 *a, double2 *b, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_read(const double2 *a, size_t n, double *out)
{
    double s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = a[i]; s += v.x + v.y; }
    if (s == 123.456) out[0] = s;
}
int main()
{
    const size_t n = (size_t)4 << 30 >> 4;      // double2 elements in 4 GiB
    double2 *a, *b; double *o;
    CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&o, 8));
    CK(hipMemset(a, 0, n * 16)); CK(hipMemset(b, 0, n * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {2048, 8192, 32768, 131072}) {
        float ms[3];
        for (int k = 0; k < 3; k++) {
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(e0));
                if (k == 0) hipLaunchKernelGGL(k_rmw, dim3(grid), dim3(256), 0, 0, a, n, 1e-9);
                if (k == 1) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n);
                if (k == 2) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, o);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[k], e0, e1));
            }
        }
        printf("grid %6d x 256: in-place RMW %.2f TB/s (R+W), copy %.2f TB/s (R+W), read %.2f TB/s\n", grid, 2.0 * n * 16 / ms[0] / 1e9, 2.0 * n * 16 / ms[1] / 1e9, 1.0 * n * 16 / ms[2] / 1e9);
    }
    return 0;
}
