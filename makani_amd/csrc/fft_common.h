// Shared device helpers of the FFT kernels (gfx950).
#pragma once
#include "common.h"

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ float2 mul_neg_i(float2 a) { return make_float2(a.y, -a.x); }  // a * (-i)

// forward DFT_R (sign -1) in registers
template <int R>
__device__ __forceinline__ void dft_small(float2* v);

template <>
__device__ __forceinline__ void dft_small<2>(float2* v) {
    const float2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}
template <>
__device__ __forceinline__ void dft_small<3>(float2* v) {
    const float c = 0.86602540378443865f;
    const float2 s = cadd(v[1], v[2]), d = csub(v[1], v[2]);
    const float2 m = make_float2(v[0].x - 0.5f * s.x, v[0].y - 0.5f * s.y);
    const float2 q = make_float2(c * d.y, -c * d.x);
    v[0] = cadd(v[0], s);
    v[1] = cadd(m, q);
    v[2] = csub(m, q);
}
template <>
__device__ __forceinline__ void dft_small<4>(float2* v) {
    const float2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
    const float2 t2 = cadd(v[1], v[3]), t3 = mul_neg_i(csub(v[1], v[3]));
    v[0] = cadd(t0, t2);
    v[1] = cadd(t1, t3);
    v[2] = csub(t0, t2);
    v[3] = csub(t1, t3);
}
template <>
__device__ __forceinline__ void dft_small<5>(float2* v) {
    const float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;
    const float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;
    const float2 a1 = cadd(v[1], v[4]), a2 = cadd(v[2], v[3]);
    const float2 b1 = csub(v[1], v[4]), b2 = csub(v[2], v[3]);
    const float2 p1 = make_float2(v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y);
    const float2 p2 = make_float2(v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y);
    const float2 q1 = make_float2(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y);
    const float2 q2 = make_float2(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y);
    const float2 iq1 = mul_neg_i(q1), iq2 = mul_neg_i(q2);
    v[0] = make_float2(v[0].x + a1.x + a2.x, v[0].y + a1.y + a2.y);
    v[1] = cadd(p1, iq1);
    v[4] = csub(p1, iq1);
    v[2] = cadd(p2, iq2);
    v[3] = csub(p2, iq2);
}

template <typename T>
__device__ __forceinline__ float2 load_pair(const T* p);
template <>
__device__ __forceinline__ float2 load_pair<float>(const float* p) {
    return *reinterpret_cast<const float2*>(p);
}
template <>
__device__ __forceinline__ float2 load_pair<u16>(const u16* p) {
    const uint32_t u = *reinterpret_cast<const uint32_t*>(p);
    return make_float2(__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u));
}
template <typename T>
__device__ __forceinline__ void store_pair(T* p, float a, float b);
#ifndef MK_FFT_ST_NT           // A/B knob: the transforms' outputs with the streaming (nt) store policy
#define MK_FFT_ST_NT 0
#endif
typedef float mk_fft_f2 __attribute__((ext_vector_type(2)));
typedef float mk_fft_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void fft_st4(float* p, float4 v) {
#if MK_FFT_ST_NT
    __builtin_nontemporal_store(mk_fft_f4{v.x, v.y, v.z, v.w}, reinterpret_cast<mk_fft_f4*>(p));
#else
    *reinterpret_cast<float4*>(p) = v;
#endif
}
template <>
__device__ __forceinline__ void store_pair<float>(float* p, float a, float b) {
#if MK_FFT_ST_NT
    __builtin_nontemporal_store(mk_fft_f2{a, b}, reinterpret_cast<mk_fft_f2*>(p));
#else
    *reinterpret_cast<float2*>(p) = make_float2(a, b);
#endif
}
template <>
__device__ __forceinline__ void store_pair<u16>(u16* p, float a, float b) {
#if MK_FFT_ST_NT
    __builtin_nontemporal_store(pack_bf16x2(a, b), reinterpret_cast<uint32_t*>(p));
#else
    *reinterpret_cast<uint32_t*>(p) = pack_bf16x2(a, b);
#endif
}
