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
// an engine gets that engine's memory back: without this the two differ, and "reproducible bit for bit" would depend on who used
// the memory before.  (Belt, not braces: the one read of an unwritten word that round 3 met as a memory access fault is fixed where
// it happened -- DESIGN.md section 6 -- and tests/test_fill_gpu.py runs the engines on memory filled with OTHER bytes.)
// Debugging aids: BSLV_FILL=<byte> (decimal or 0x..) fills fresh allocations and the uncopied tail of every grow() with that byte
// instead of zero; BSLV_ALLOC_LOG=<file> appends "name pointer bytes file:line" per allocation (to match a fault address).
int debug_fill();
void debug_note_alloc(const char *name, const void *p, size_t bytes, const char *file, int line);
template <class T>
static inline hipError_t malloc0_impl(T **p, size_t bytes, hipStream_t s, bool on_stream, const char *name, const char *file, int line)
{
    hipError_t e = hipMalloc((void **)p, bytes);
    if (e == hipSuccess && bytes) {
        debug_note_alloc(name, (const void *)*p, bytes, file, line);
        // (creation-time allocations: the null stream, which orders against every stream of the process; allocations made while an
        // engine runs name their stream, so that the other engine stream of a pipelined driver is not held up)
        e = on_stream ? hipMemsetAsync((void *)*p, debug_fill(), bytes, s) : hipMemset((void *)*p, debug_fill(), bytes);
    }
    return e;
}
#define malloc0(p, bytes) bslv::malloc0_impl(p, bytes, nullptr, false, #p, __FILE__, __LINE__)
#define malloc0s(p, bytes, stream) bslv::malloc0_impl(p, bytes, stream, true, #p, __FILE__, __LINE__)

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
