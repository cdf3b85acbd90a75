// Complex arithmetic of the specialised FFT kernels on PACKED fp32 instructions (gfx950: v_pk_add_f32 / v_pk_mul_f32 /
// v_pk_fma_f32 work on a 64-bit register pair at the issue rate of their scalar forms).
//
// A complex number is a two-element vector that lives in an aligned register pair for its whole life; every butterfly
// below is written so that one instruction acts on (re, im) at once:
//   * a +- b                      one v_pk_add_f32 (the subtraction is a source modifier);
//   * a * w, both run-time        v_pk_mul_f32 (w.re broadcast by op_sel) + v_pk_fma_f32 with the operand halves swapped
//                                 by op_sel and the sign of the w.im * a.im term as a neg_lo modifier;
//   * a +- (-i) d                 one v_pk_fma_f32 of the swapped d with the constant pair (1, -1) / (-1, 1);
//   * real constants (radix-3 / radix-5 butterflies) multiply both halves in one instruction.
// hipcc folds whole-vector negations and half swaps of a two-float vector into the instruction's modifiers, but not the
// negation of ONE half, so the two products that need it are inline assembly; everything else is plain vector code.
// The struct-of-two-floats form of the same butterflies (fft_common.h: the generic kernels' radix 2-5) reaches the
// packed instructions only through the SLP vectoriser, which pairs unrelated scalars and pays for it in register moves:
// 1 003 packed + 400 scalar floating-point instructions + 676 v_mov in the 1440-point forward kernel.
#pragma once
#include "fft_common.h"
#include "fft_consts.h"

typedef float cf __attribute__((ext_vector_type(2)));       // (re, im)

__device__ __forceinline__ cf cf_make(float re, float im) {
    cf r = {re, im};
    return r;
}
__device__ __forceinline__ cf pk_fma(cf a, cf b, cf c) { return __builtin_elementwise_fma(a, b, c); }

// a * w
__device__ __forceinline__ cf cmul(cf a, cf w) {
    const cf t = a * w.xx;
    cf r;       // lo: t.re - a.im w.im   hi: t.im + a.re w.im
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// a * conj(w)
__device__ __forceinline__ cf cmulc(cf a, cf w) {
    const cf t = a * w.xx;
    cf r;       // lo: t.re + a.im w.im   hi: t.im - a.re w.im
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// a * (wr + i wi) with compile-time wr, wi (the constant pairs end up in scalar registers)
__device__ __forceinline__ cf cmul_const(cf a, float wr, float wi) {
    if (wr == 1.f && wi == 0.f) return a;
    if (wr == 0.f && wi == -1.f) return a.yx * cf_make(1.f, -1.f);
    if (wr == 0.f && wi == 1.f) return a.yx * cf_make(-1.f, 1.f);
    if (wr == -1.f && wi == 0.f) return -a;
    const cf t = a * cf_make(wr, wr);
    return pk_fma(a.yx, cf_make(-wi, wi), t);
}
// a + (-i) d,  a - (-i) d
__device__ __forceinline__ cf add_mi(cf a, cf d) { return pk_fma(d.yx, cf_make(1.f, -1.f), a); }
__device__ __forceinline__ cf sub_mi(cf a, cf d) { return pk_fma(d.yx, cf_make(-1.f, 1.f), a); }
// a - b.yx = (a.re - b.im, a.im - b.re).  Written by hand because hipcc's form of it — v_pk_add_f32 a, b with op_sel / neg on
// src1 — is one of the packed forms that return wrong results on gfx950 while a matrix-core kernel shares the compute unit
// (a VGPR src1 read through op_sel in the two-operand packed ops; tools/pk_hazard_probe.py, docs/LAB_NOTEBOOK.md round 6):
// the swapped operand goes first, where the same modifiers are reliable.
__device__ __forceinline__ cf sub_yx(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0] neg_hi:[1,0]" : "=v"(r) : "v"(b), "v"(a));
    return r;
}
// (w.re a, w.im b) from two scalars that live in unrelated registers: two plain multiplies, pinned — left to itself hipcc pairs the
// operands of such products into v_pk_mul_f32 with op_sel on a VGPR src1, one of the unreliable forms (see sub_yx)
__device__ __forceinline__ cf mul_parts(cf w, float a, float b) {
    float x, y;
    asm("v_mul_f32 %0, %1, %2" : "=v"(x) : "v"(w.x), "v"(a));
    asm("v_mul_f32 %0, %1, %2" : "=v"(y) : "v"(w.y), "v"(b));
    return cf_make(x, y);
}
// a + conj(b),  a - conj(b)
__device__ __forceinline__ cf add_conj(cf a, cf b) { return pk_fma(b, cf_make(1.f, -1.f), a); }
__device__ __forceinline__ cf sub_conj(cf a, cf b) { return pk_fma(b, cf_make(-1.f, 1.f), a); }

// forward DFT_R (sign -1) in registers
template <int R>
__device__ __forceinline__ void pdft_small(cf* v);

template <>
__device__ __forceinline__ void pdft_small<2>(cf* v) {
    const cf a = v[0], b = v[1];
    v[0] = a + b;
    v[1] = a - b;
}
template <>
__device__ __forceinline__ void pdft_small<3>(cf* v) {
    const float c = 0.86602540378443865f;
    const cf s = v[1] + v[2], d = v[1] - v[2];
    const cf m = pk_fma(s, cf_make(-0.5f, -0.5f), v[0]);
    v[0] = v[0] + s;
    v[1] = pk_fma(d.yx, cf_make(c, -c), m);          // m + (-i) c d
    v[2] = pk_fma(d.yx, cf_make(-c, c), m);
}
template <>
__device__ __forceinline__ void pdft_small<4>(cf* v) {
    const cf t0 = v[0] + v[2], t1 = v[0] - v[2];
    const cf t2 = v[1] + v[3], d = v[1] - v[3];
    v[0] = t0 + t2;
    v[2] = t0 - t2;
    v[1] = add_mi(t1, d);
    v[3] = sub_mi(t1, d);
}
template <>
__device__ __forceinline__ void pdft_small<5>(cf* v) {
    const float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;
    const float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;
    const cf a1 = v[1] + v[4], a2 = v[2] + v[3];
    const cf b1 = v[1] - v[4], b2 = v[2] - v[3];
    const cf p1 = pk_fma(a2, cf_make(c2, c2), pk_fma(a1, cf_make(c1, c1), v[0]));
    const cf p2 = pk_fma(a2, cf_make(c1, c1), pk_fma(a1, cf_make(c2, c2), v[0]));
    const cf q1 = pk_fma(b2, cf_make(s2, s2), b1 * cf_make(s1, s1));
    const cf q2 = pk_fma(b2, cf_make(-s1, -s1), b1 * cf_make(s2, s2));
    v[0] = v[0] + a1 + a2;
    v[1] = add_mi(p1, q1);
    v[4] = sub_mi(p1, q1);
    v[2] = add_mi(p2, q2);
    v[3] = sub_mi(p2, q2);
}

// ---- in-register DFTs of composite size (Cooley-Tukey on two factors) ------------------------------------------------
// PDft<R>::run(v) transforms v[0..R) in place; output bin o ends up at v[PDft<R>::loc(o)].
template <int R>
struct PDft {
    __device__ static __forceinline__ void run(cf* v) { pdft_small<R>(v); }
    __host__ __device__ static constexpr int loc(int o) { return o; }
};

// RA and RB may themselves be composite (PDft<RA> / PDft<RB> report where their output bins end up)
template <int RA, int RB>
struct PDftComp {
    static constexpr int R = RA * RB;
    __host__ __device__ static constexpr int loc(int o) { return RB * (o % RA) + PDft<RB>::loc(o / RA); }
    __device__ static __forceinline__ void run(cf* v) {
        // n = RB*a + b, k = k1 + RA*k2
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            cf t[RA];
#pragma unroll
            for (int a = 0; a < RA; ++a) t[a] = v[RB * a + b];
            PDft<RA>::run(t);
#pragma unroll
            for (int k1 = 0; k1 < RA; ++k1) v[RB * k1 + b] = t[PDft<RA>::loc(k1)];
        }
#pragma unroll
        for (int k1 = 1; k1 < RA; ++k1)
#pragma unroll
            for (int b = 1; b < RB; ++b)
                v[RB * k1 + b] = cmul_const(v[RB * k1 + b], RootTable<R>::re[(b * k1) % R], RootTable<R>::im[(b * k1) % R]);
#pragma unroll
        for (int k1 = 0; k1 < RA; ++k1) PDft<RB>::run(v + RB * k1);
    }
};
template <> struct PDft<6> : PDftComp<2, 3> {};
template <> struct PDft<10> : PDftComp<2, 5> {};
template <> struct PDft<24> : PDftComp<4, 6> {};
template <> struct PDft<30> : PDftComp<5, 6> {};
