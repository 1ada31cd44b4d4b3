// Shared device helpers for the gfx950 kernels of libesr_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/esr_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// float -> bf16 bits, round to nearest even (finite inputs; NaN stays NaN-ish, never produced on this path)
__device__ __forceinline__ uint32_t f2bf(float x) {
    uint32_t u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float bf2f(uint32_t h) { return __uint_as_float(h << 16); }

// split x into hi = bf16(x), lo = bf16(x - hi); hi + lo carries ~16 mantissa bits of x
__device__ __forceinline__ void split_bf16(float x, uint32_t& hi, uint32_t& lo) {
    hi = f2bf(x);
    lo = f2bf(x - bf2f(hi));
}

// fp16 element <-> float (round to nearest even)
__device__ __forceinline__ uint32_t f2h(float x) { return (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)x); }
__device__ __forceinline__ float h2f(uint32_t h) { return (float)__builtin_bit_cast(_Float16, (unsigned short)h); }
// 16-bit element of format FMT (0 bf16, 1 f16) <-> float
template <int FMT> __device__ __forceinline__ float e2f(uint32_t e) { return FMT ? h2f(e) : bf2f(e); }
template <int FMT> __device__ __forceinline__ uint32_t f2e(float x) { return FMT ? f2h(x) : f2bf(x); }

// device-side mirror of esr_act_view (strides in 16-byte vectors)
struct DView {
    const uint4* hi;
    const uint4* lo;
    long long bs, cs;
    int ncg;
    int fmt;
};

static inline DView to_dview(const esr_act_view& v) {
    DView d;
    d.hi = (const uint4*)v.hi;
    d.lo = (const uint4*)v.lo;
    d.bs = v.batch_stride;
    d.cs = v.cg_stride;
    d.ncg = v.hi ? v.ncg : 0;
    d.fmt = v.fmt;
    return d;
}

// hipGetLastError() is sticky-until-read and shared with the host framework: drop whatever an earlier, unrelated call left
// behind before launching, so that ESR_CHECK_LAUNCH reports only our own launch
#define ESR_CLEAR_ERR() (void)hipGetLastError()
#define ESR_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return ESR_E_LAUNCH;          \
    } while (0)

// Raise a kernel's dynamic-LDS limit to the full 160 KiB once per (kernel, device): the attribute is per device, and one process may
// drive several (host threads under nn.DataParallel-style use).  `mask` is a function-local static of the caller.
#define ESR_ALLOW_160K_LDS(kernel_ptr)                                                                                      \
    do {                                                                                                                    \
        static unsigned long long mask__ = 0;   /* benign race: setting the attribute twice is harmless */                  \
        int dev__ = 0;                                                                                                      \
        (void)hipGetDevice(&dev__);                                                                                         \
        if (!(mask__ >> (dev__ & 63) & 1ull)) {                                                                             \
            (void)hipFuncSetAttribute((const void*)(kernel_ptr), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);   \
            mask__ |= 1ull << (dev__ & 63);                                                                                 \
        }                                                                                                                   \
    } while (0)
