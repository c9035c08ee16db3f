// MEASUREMENT: do v_mfma_f64_16x16x4 and the fp64 vector instructions of ANOTHER wave on the same SIMD run side by side on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probe/_bin/f64_pipes scripts/probe/f64_pipes.hip && scripts/probe/_bin/f64_pipes
// One workgroup of 8 waves per CU (waves w and w + 4 share a SIMD).  Role of a wave: 0 idle, 1 = ITER x 8 independent chained
// f64 MFMAs, 2 = ITER x 32 independent v_fma_f64, 3 = ITER x 32 v_cmp_gt_f64 (into SGPR pairs) + v_writelane_b32.
// Each wave times itself with s_memtime (100 MHz constant clock -> ns) and wall_clock64; the host prints ns per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef double d4_t __attribute__((ext_vector_type(4)));
#define ITER 4096
__global__ __launch_bounds__(512) void k_probe(int role_lo, int role_hi, double *sink, long long *ticks)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int role = wave < 4 ? role_lo : role_hi;
    double x = 1.0 + lane * 1e-3, y = 0.5 + lane * 1e-4;
    __syncthreads();
    const long long t0 = wall_clock64();
    double out = 0.0;
    if (role == 1) {
        d4_t acc[8];
        for (int j = 0; j < 8; j++) acc[j] = d4_t{0.0, 0.0, 0.0, 0.0};
        for (int it = 0; it < ITER; it++) {
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[j], 0, 0, 0);
        }
        for (int j = 0; j < 8; j++) out += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    } else if (role == 2) {
        double a[16];
        for (int j = 0; j < 16; j++) a[j] = j;
        for (int it = 0; it < ITER; it++) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int j = 0; j < 16; j++) a[j] = fma(a[j], x, y);
        }
        for (int j = 0; j < 16; j++) out += a[j];
    } else if (role == 3) {
        int m = 0;
        double th = 0.25 + lane;
        for (int it = 0; it < ITER; it++) {
#pragma unroll
            for (int j = 0; j < 32; j++) {
                unsigned long long b;
                asm volatile("v_cmp_gt_f64 %0, %1, %2" : "=s"(b) : "v"(x), "v"(th));
                asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(m) : "s"((unsigned)b));
            }
            th += 1.0;
        }
        out = m;
    }
    const long long t1 = wall_clock64();
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
    if (out == 123.456) sink[0] = out;
}
int main()
{
    double *sink; long long *ticks;
    hipMalloc(&sink, 8); hipMalloc(&ticks, 256 * 8 * 8);
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0);
    std::vector<long long> t(256 * 8);
    const char *names[4] = {"idle", "mfma_f64_16x16x4 x8", "v_fma_f64 x32", "v_cmp_f64+writelane x32"};
    const int per_iter[4] = {0, 8, 32, 32};
    const int combos[][2] = {{1, 0}, {2, 0}, {3, 0}, {1, 1}, {2, 2}, {1, 2}, {1, 3}, {2, 3}};
    printf("{\"wall_clock_khz\": %d, \"rows\": [\n", clk_khz);
    for (unsigned c = 0; c < sizeof(combos) / sizeof(combos[0]); c++) {
        const int lo = combos[c][0], hi = combos[c][1];
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_probe, dim3(256), dim3(512), 0, 0, lo, hi, sink, ticks);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_probe, dim3(256), dim3(512), 0, 0, lo, hi, sink, ticks);
        hipEventRecord(e1, 0);
        if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(t.data(), ticks, t.size() * 8, hipMemcpyDeviceToHost);
        double slo = 0, shi = 0;
        for (int b = 0; b < 256; b++) for (int w = 0; w < 8; w++) (w < 4 ? slo : shi) += (double)t[b * 8 + w];
        slo /= 1024.0; shi /= 1024.0;
        const double ns_lo = per_iter[lo] ? slo / clk_khz * 1e6 / ((double)ITER * per_iter[lo]) : 0.0;
        const double ns_hi = per_iter[hi] ? shi / clk_khz * 1e6 / ((double)ITER * per_iter[hi]) : 0.0;
        printf(" {\"waves_0_3\": \"%s\", \"waves_4_7\": \"%s\", \"kernel_ms\": %.4f, \"ns_per_instr_0_3\": %.2f, \"ns_per_instr_4_7\": %.2f}%s\n",
               names[lo], names[hi], ms, ns_lo, ns_hi, c + 1 < sizeof(combos) / sizeof(combos[0]) ? "," : "");
    }
    printf("]}\n");
    return 0;
}
