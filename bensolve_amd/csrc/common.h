// common.h -- shared helpers for the HIP engines (error handling, device reductions)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <cmath>
#include "../../include/bslv_hip.h"

namespace bslv {

void set_error(const char *fmt, ...);

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            bslv::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                            __LINE__);                                                        \
            return (_e == hipErrorOutOfMemory) ? BSLV_E_NOMEM : BSLV_E_NODEVICE;              \
        }                                                                                     \
    } while (0)

// Device memory of the engines starts out as ZEROS.  A fresh process gets zero pages from the driver, a process that has destroyed
// an engine gets that engine's memory back: without this the two differ (seen as a memory access fault in the third engine of a
// bench run, never in a test process), and "reproducible bit for bit" would depend on who used the memory before.
template <class T>
static inline hipError_t malloc0(T **p, size_t bytes)
{
    hipError_t e = hipMalloc((void **)p, bytes);
#ifndef BSLV_NO_MALLOC0            // (diagnostic builds only: the allocations as they were before this was found)
    if (e == hipSuccess && bytes) e = hipMemset((void *)*p, 0, bytes);
#endif
    return e;
}

constexpr int WAVE = 64;

// wave-level reductions (64 lanes)
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, WAVE));
    return v;
}
__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, WAVE));
    return v;
}

// (value, index) argmax with deterministic tie-break on the smaller index
struct ValIdx { double v; int i; };
__device__ __forceinline__ ValIdx better_max(ValIdx a, ValIdx b)
{
    if (b.v > a.v || (b.v == a.v && b.i >= 0 && (a.i < 0 || b.i < a.i))) return b;
    return a;
}
__device__ __forceinline__ ValIdx wave_argmax(ValIdx x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ValIdx y;
        y.v = __shfl_xor(x.v, o, WAVE);
        y.i = __shfl_xor(x.i, o, WAVE);
        x = better_max(x, y);
    }
    return x;
}

}  // namespace bslv
