// Shared device/host helpers for libmakani_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/makani_amd.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

#define MK_NUM_XCD 8

// ---- error plumbing -----------------------------------------------------------
void mk_set_error(const char* fmt, ...);

#define MK_REQUIRE(cond, ...)          \
    do {                               \
        if (!(cond)) {                 \
            mk_set_error(__VA_ARGS__); \
            return MK_EINVAL;          \
        }                              \
    } while (0)

static inline int mk_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        mk_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

// ---- bf16 <-> f32 (round-to-nearest-even, matches torch's .to(bfloat16)) -------
__device__ __forceinline__ float bf16_to_f32(u16 v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ u16 f32_to_bf16(float f) {
    // gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32: RNE, NaN quieted); the integer emulation of it
    // cost 13 % of the instance-norm kernels
    const __bf16 h = (__bf16)f;
    return __builtin_bit_cast(u16, h);
}

// two fp32 values -> one dword of two bf16 (a in the low half), ONE v_cvt_pk_bf16_f32 (the element-at-a-time form
// (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16) compiles to two conversions with an unused source each plus a
// v_or_b32_sdwa: the epilogues of the channel GEMMs pack 48-96 pairs per lane and tile); same rounding, same bits
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    typedef float mk_f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 mk_bf16x2_t __attribute__((ext_vector_type(2)));
    const mk_f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, mk_bf16x2_t));
}

// XCD-aware remap of a 1-D block id: consecutive ids handed to one XCD (blocks b, b+8, b+16 ...
// run on XCD b%8 — speed only, never correctness).  Bijective for any n (guide §5 "XCD swizzle").
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk / MK_NUM_XCD, r = nblk % MK_NUM_XCD;
    const int xcd = bid % MK_NUM_XCD, j = bid / MK_NUM_XCD;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + j;
}

// exact-erf GELU and its derivative (nn.GELU default, approximate='none')
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// GELU / GELU' for GEMM epilogues, where the transcendental work is NOT hidden behind memory traffic (128 evaluations per
// thread and output tile): erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 rounding of the
// result) on the hardware exp and reciprocal; ~12 VALU instructions instead of ~30 for erff.  The streaming kernels of
// pointwise.hip keep the exact forms above.
__device__ __forceinline__ void erf_as_f(float x, float& erfv, float& expv) {      // erf(x / sqrt 2) and exp(-x^2 / 2)
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    const float e = __expf(-z * z);
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float r = 1.0f - poly * t * e;
    erfv = copysignf(r, x);
    expv = e;
}
__device__ __forceinline__ float gelu_fast_f(float x) {
    float er, ex;
    erf_as_f(x, er, ex);
    return 0.5f * x * (1.0f + er);
}
__device__ __forceinline__ float gelu_grad_fast_f(float x) {
    float er, ex;
    erf_as_f(x, er, ex);
    return fmaf(x * 0.39894228040143268f, ex, 0.5f * (1.0f + er));
}
