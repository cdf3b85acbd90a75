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

// GELU' (and, with -DMK_GELU_EXP2=0, GELU) of the bf16 kernels, where the transcendental work is NOT hidden behind memory traffic
// (128 evaluations per thread and output tile of a GEMM epilogue): erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7) on the
// hardware exp and reciprocal; ~12 VALU instructions instead of ~30 for erff.  The fp32 kernels keep the exact forms above.
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
#ifndef MK_GELU_EXP2            // forward GELU of the bf16 kernels: 0 = the erf of erf_as_f, 7 / 9 = gelu_exp2_x2 with a polynomial of that degree
#define MK_GELU_EXP2 9
#endif
// Forward GELU of the bf16 kernels (round 6):  x Phi(x) = relu(x) - |x| Phi(-|x|),  Phi(-a) = erfc(a / sqrt 2) / 2 = 2^-(1 + a S(a)) with S a
// polynomial on [0, 9] (weighted minimax fit of -log2(erfc(a / sqrt 2)) / a: tools/gelu_fit.py; relative error of Phi(-a) 6.5e-6 / 3.1e-6
// at degree 7 / 9 over the WHOLE range).  No cancellation on either side of zero: over all 31 002 bf16 arguments with |GELU| >= 1e-17
// the bf16 result equals the correctly rounded one (fp64, erfc form) — also at x < -4, where x (1 + erf) / 2 has lost every digit
// (the form above: 6 % of the results of a sigma = 3 input one or more bf16 steps off; this one 0.02 % of fp32 arguments by one step) —
// tests/test_gpu_kernels.py::test_bf16_forward_gelu_is_exact_to_the_rounding_including_the_tails.  One v_exp_f32, no reciprocal, the
// Horner chain as v_pk_fma_f32 on value pairs with the coefficient pairs in scalar registers: 10.7 vector instructions per value in a GEMM
// epilogue against 14.7, one of them quarter-rate instead of two (static issue cycles of the loop -34 %).  Beyond |x| = 9, Phi(-9) = 1e-19
// stands in for smaller values.
typedef float mk_f32x2 __attribute__((ext_vector_type(2)));
// max(x, 0) as ONE instruction (fmaxf on a value unpacked from bf16 bits puts a canonicalising v_max_f32 x, x in front)
__device__ __forceinline__ float mk_relu_f(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ mk_f32x2 mk_splat2(float v) { return mk_f32x2{v, v}; }
// S(a): -log2(erfc(a / sqrt 2)) / a on [0, 9]
__device__ __forceinline__ mk_f32x2 mk_phi_poly(mk_f32x2 a) {
#if MK_GELU_EXP2 == 9
    mk_f32x2 s = mk_splat2(1.188001586e-09f);
    s = __builtin_elementwise_fma(s, a, mk_splat2(-5.194742769e-08f));
    s = __builtin_elementwise_fma(s, a, mk_splat2(8.870460420e-07f));
    s = __builtin_elementwise_fma(s, a, mk_splat2(-6.154555649e-06f));
    s = __builtin_elementwise_fma(s, a, mk_splat2(-1.931048610e-05f));
    s = __builtin_elementwise_fma(s, a, mk_splat2(7.936366601e-04f));
    s = __builtin_elementwise_fma(s, a, mk_splat2(-8.236761205e-03f));
    s = __builtin_elementwise_fma(s, a, mk_splat2(5.364274606e-02f));
    s = __builtin_elementwise_fma(s, a, mk_splat2(4.586661756e-01f));
    s = __builtin_elementwise_fma(s, a, mk_splat2(1.151193023e+00f));
#else
    mk_f32x2 s = mk_splat2(-7.347856723e-08f);
    s = __builtin_elementwise_fma(s, a, mk_splat2(3.615617288e-06f));
    s = __builtin_elementwise_fma(s, a, mk_splat2(-7.886793173e-05f));
    s = __builtin_elementwise_fma(s, a, mk_splat2(1.015813905e-03f));
    s = __builtin_elementwise_fma(s, a, mk_splat2(-8.733216673e-03f));
    s = __builtin_elementwise_fma(s, a, mk_splat2(5.426375940e-02f));
    s = __builtin_elementwise_fma(s, a, mk_splat2(4.582903981e-01f));
    s = __builtin_elementwise_fma(s, a, mk_splat2(1.151269913e+00f));
#endif
    return s;
}
// two values at a time: the Horner chain as v_pk_fma_f32 (coefficient pairs in scalar registers)
__device__ __forceinline__ mk_f32x2 gelu_exp2_x2(mk_f32x2 x) {
    const mk_f32x2 a = {__builtin_amdgcn_fmed3f(fabsf(x.x), 0.0f, 9.0f), __builtin_amdgcn_fmed3f(fabsf(x.y), 0.0f, 9.0f)};     // min(|x|, 9): one v_med3_f32
    const mk_f32x2 e = __builtin_elementwise_fma(a, mk_phi_poly(a), mk_splat2(1.0f));
    const mk_f32x2 h = {__builtin_amdgcn_exp2f(-e.x), __builtin_amdgcn_exp2f(-e.y)};      // Phi(-|x|)
    const mk_f32x2 r = {mk_relu_f(x.x), mk_relu_f(x.y)};
    return __builtin_elementwise_fma(-a, h, r);               // x Phi(x) = relu(x) - |x| Phi(-|x|) on both sides of zero, no cancellation
}
__device__ __forceinline__ float gelu_exp2_f(float x) { return gelu_exp2_x2(mk_f32x2{x, x}).x; }
__device__ __forceinline__ float gelu_fast_f(float x) {
#if MK_GELU_EXP2
    return gelu_exp2_f(x);
#else
    float er, ex;
    erf_as_f(x, er, ex);
    return 0.5f * x * (1.0f + er);
#endif
}
#ifndef MK_GELU_GRAD_EXP2       // GELU' of the bf16 kernels: 0 = erf_as_f (default), 1 = the exp2 form below (two exponentials, no reciprocal).
#define MK_GELU_GRAD_EXP2 0     // Measured equal (gpurun_out/r07j: norm backward, GEMM epilogues and the step within noise): GELU' is hidden behind
#endif                          // the two operand streams of the kernels that evaluate it, so the form with the smaller error around zero stays
// GELU'(x) = Phi(x) + x phi(x) with the same Phi (selected, not subtracted: relative error 1e-6 at x = -5 where the erf form has 3 %)
// and phi(x) = 2^(-x^2 log2(e) / 2) / sqrt(2 pi): absolute error 1.5e-6 around zero (the 3e-6 of Phi), tools/gelu_fit.py
__device__ __forceinline__ mk_f32x2 gelu_grad_exp2_x2(mk_f32x2 x) {
    const mk_f32x2 a = {__builtin_amdgcn_fmed3f(fabsf(x.x), 0.0f, 9.0f), __builtin_amdgcn_fmed3f(fabsf(x.y), 0.0f, 9.0f)};
    const mk_f32x2 e = __builtin_elementwise_fma(a, mk_phi_poly(a), mk_splat2(1.0f));
    const mk_f32x2 h = {__builtin_amdgcn_exp2f(-e.x), __builtin_amdgcn_exp2f(-e.y)};      // Phi(-|x|)
    const mk_f32x2 q = (x * x) * mk_splat2(0.72134752044448170f);
    const mk_f32x2 t = {__builtin_amdgcn_exp2f(-q.x), __builtin_amdgcn_exp2f(-q.y)};      // exp(-x^2 / 2)
    const mk_f32x2 u = mk_splat2(1.0f) - h;
    const mk_f32x2 phi = {x.x < 0.f ? h.x : u.x, x.y < 0.f ? h.y : u.y};
    return __builtin_elementwise_fma(x * mk_splat2(0.39894228040143268f), t, phi);
}
__device__ __forceinline__ float gelu_grad_fast_f(float x) {
#if MK_GELU_GRAD_EXP2
    return gelu_grad_exp2_x2(mk_f32x2{x, x}).x;
#else
    float er, ex;
    erf_as_f(x, er, ex);
    return fmaf(x * 0.39894228040143268f, ex, 0.5f * (1.0f + er));
#endif
}
// out[i] = GELU'(arg[i]) for N_ values (pairs through the packed form)
template <int N_>
__device__ __forceinline__ void gelu_grad_fast_n(const float* arg, float* out) {
#if MK_GELU_GRAD_EXP2
#pragma unroll
    for (int e = 0; e + 1 < N_; e += 2) {
        const mk_f32x2 r = gelu_grad_exp2_x2(mk_f32x2{arg[e], arg[e + 1]});
        out[e] = r.x;
        out[e + 1] = r.y;
    }
    if (N_ & 1) out[N_ - 1] = gelu_grad_fast_f(arg[N_ - 1]);
#else
#pragma unroll
    for (int e = 0; e < N_; ++e) out[e] = gelu_grad_fast_f(arg[e]);
#endif
}
// N_ values in place (pairs through the packed form)
template <int N_>
__device__ __forceinline__ void gelu_fast_n(float* v) {
#if MK_GELU_EXP2
#pragma unroll
    for (int e = 0; e + 1 < N_; e += 2) {
        const mk_f32x2 r = gelu_exp2_x2(mk_f32x2{v[e], v[e + 1]});
        v[e] = r.x;
        v[e + 1] = r.y;
    }
    if (N_ & 1) v[N_ - 1] = gelu_exp2_f(v[N_ - 1]);
#else
#pragma unroll
    for (int e = 0; e < N_; ++e) v[e] = gelu_fast_f(v[e]);
#endif
}

