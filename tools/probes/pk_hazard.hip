// Probe library (NOT part of the product), round 6.  Which packed-fp32 instruction FORMS return wrong results on gfx950 while a
// matrix-core kernel runs on another stream?  (tools/pk_hazard_probe.py; docs/LAB_NOTEBOOK.md round 6.)  Every kernel runs, in ONE
// asm block, ITERS times on changing inputs:   v_mov_b32 v100, x0 ; v_mov_b32 v101, x1 ; [s_nop 3] ; <form under test>
// and counts results whose bits differ from the same arithmetic in plain (unpacked) instructions.  one `if constexpr (FORM == n)` block per form.
// Build: hipcc --offload-arch=gfx950 -O2 -fno-slp-vectorize -shared -fPIC tools/probes/pk_hazard.hip -o tools/probes/libpk_hazard.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define PRE "v_mov_b32 v100, %1\n\tv_mov_b32 v101, %2\n\t"
#define G0 ""
#define G1 "s_nop 3\n\t"

template <int FORM, int GAP>
__global__ __launch_bounds__(256) void hz_kernel(const float* __restrict__ in, unsigned* __restrict__ out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float x0 = in[2 * t], x1 = in[2 * t + 1];
    const float c0 = 1.5f, c1 = -0.75f, k0 = 1.25f, k1 = -2.5f;
    const f2 cv = {c0, c1};
    const f2 kp = {k0, k1};
    unsigned bad = 0;
    for (int i = 0; i < iters; ++i) {
        x0 = x0 * 1.0001f + 0.37f;
        x1 = x1 * 0.9999f - 0.21f;
        const float y0 = x1 * 0.5f + 3.f, y1 = x0 * 0.25f - 2.f;
        const float a0 = x0, a1 = x1, d0 = y0, d1 = y1;
        const f2 dv = {d0, d1};
        float e0 = 0.f, e1 = 0.f;
        f2 r = {0.f, 0.f};
        if constexpr (FORM == 0) {
            e0 = a0+c0; e1 = a1+c1;
            if constexpr (GAP == 0) asm volatile(PRE G0 "v_pk_add_f32 %0, v[100:101], %3" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
            else asm volatile(PRE G1 "v_pk_add_f32 %0, v[100:101], %3" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
        }
        if constexpr (FORM == 1) {
            e0 = k0*a0; e1 = k1*a1;
            if constexpr (GAP == 0) asm volatile(PRE G0 "v_pk_mul_f32 %0, %7, v[100:101]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
            else asm volatile(PRE G1 "v_pk_mul_f32 %0, %7, v[100:101]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
        }
        if constexpr (FORM == 2) {
            e0 = a0*c1; e1 = a1*c0;
            if constexpr (GAP == 0) asm volatile(PRE G0 "v_pk_mul_f32 %0, v[100:101], %3 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
            else asm volatile(PRE G1 "v_pk_mul_f32 %0, v[100:101], %3 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
        }
        if constexpr (FORM == 3) {
            e0 = a1*c0; e1 = a0*c1;
            if constexpr (GAP == 0) asm volatile(PRE G0 "v_pk_mul_f32 %0, v[100:101], %3 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
            else asm volatile(PRE G1 "v_pk_mul_f32 %0, v[100:101], %3 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
        }
        if constexpr (FORM == 4) {
            e0 = a0*c0; e1 = a1*c0;
            if constexpr (GAP == 0) asm volatile(PRE G0 "v_pk_mul_f32 %0, v[100:101], %3 op_sel_hi:[1,0]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
            else asm volatile(PRE G1 "v_pk_mul_f32 %0, v[100:101], %3 op_sel_hi:[1,0]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
        }
        if constexpr (FORM == 5) {
            e0 = a0*c1; e1 = a1*c1;
            if constexpr (GAP == 0) asm volatile(PRE G0 "v_pk_mul_f32 %0, v[100:101], %3 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
            else asm volatile(PRE G1 "v_pk_mul_f32 %0, v[100:101], %3 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
        }
        if constexpr (FORM == 6) {
            e0 = a0+c1; e1 = a1+c0;
            if constexpr (GAP == 0) asm volatile(PRE G0 "v_pk_add_f32 %0, v[100:101], %3 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
            else asm volatile(PRE G1 "v_pk_add_f32 %0, v[100:101], %3 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
        }
        if constexpr (FORM == 7) {
            e0 = fmaf(a1,-c1,d0); e1 = fmaf(a0,c1,d1);
            if constexpr (GAP == 0) asm volatile(PRE G0 "v_pk_fma_f32 %0, v[100:101], %3, %8 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
            else asm volatile(PRE G1 "v_pk_fma_f32 %0, v[100:101], %3, %8 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
        }
        if constexpr (FORM == 8) {
            e0 = fmaf(a1,k0,d0); e1 = fmaf(a0,k1,d1);
            if constexpr (GAP == 0) asm volatile(PRE G0 "v_pk_fma_f32 %0, v[100:101], %7, %8 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
            else asm volatile(PRE G1 "v_pk_fma_f32 %0, v[100:101], %7, %8 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
        }
        if constexpr (FORM == 9) {
            e0 = fmaf(a0,c0,d0); e1 = fmaf(a1,c1,d1);
            if constexpr (GAP == 0) asm volatile(PRE G0 "v_pk_fma_f32 %0, v[100:101], %3, %8" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
            else asm volatile(PRE G1 "v_pk_fma_f32 %0, v[100:101], %3, %8" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
        }
        if constexpr (FORM == 10) {
            e0 = a0*k1; e1 = a1*k0;
            if constexpr (GAP == 0) asm volatile(PRE G0 "v_pk_mul_f32 %0, v[100:101], %7 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
            else asm volatile(PRE G1 "v_pk_mul_f32 %0, v[100:101], %7 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
        }
        if constexpr (FORM == 11) {
            e0 = fmaf(a0,c0,d0); e1 = fmaf(a1,c0,d1);
            if constexpr (GAP == 0) asm volatile(PRE G0 "v_pk_fma_f32 %0, v[100:101], %3, %8 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
            else asm volatile(PRE G1 "v_pk_fma_f32 %0, v[100:101], %3, %8 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
        }
        if constexpr (FORM == 12) {
            e0 = a1; e1 = a0;
            if constexpr (GAP == 0) asm volatile(PRE G0 "v_pk_mov_b32 %0, v[100:101], v[100:101] op_sel:[1,0]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
            else asm volatile(PRE G1 "v_pk_mov_b32 %0, v[100:101], v[100:101] op_sel:[1,0]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
        }
        if constexpr (FORM == 13) {
            e0 = d0 - a1; e1 = d1 - a0;          // the form of fft_packed.h sub_yx: swizzle + neg on src0
            if constexpr (GAP == 0) asm volatile(PRE G0 "v_pk_add_f32 %0, v[100:101], %8 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0] neg_hi:[1,0]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
            else asm volatile(PRE G1 "v_pk_add_f32 %0, v[100:101], %8 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0] neg_hi:[1,0]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
        }
        if constexpr (FORM == 14) {
            e0 = fmaf(a0, c1, d0); e1 = fmaf(a1, c0, d1);
            if constexpr (GAP == 0) asm volatile(PRE G0 "v_pk_fma_f32 %0, v[100:101], %3, %8 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
            else asm volatile(PRE G1 "v_pk_fma_f32 %0, v[100:101], %3, %8 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
        }
        if constexpr (FORM == 15) {
            e0 = fmaf(a0, c0, d1); e1 = fmaf(a1, c1, d0);       // src2 (VGPR) read through op_sel
            if constexpr (GAP == 0) asm volatile(PRE G0 "v_pk_fma_f32 %0, v[100:101], %3, %8 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
            else asm volatile(PRE G1 "v_pk_fma_f32 %0, v[100:101], %3, %8 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
        }
        if constexpr (FORM == 16) {
            e0 = fmaf(a0, c1, d0); e1 = fmaf(a1, c1, d1);       // src1 (VGPR) high half broadcast, src0 straight
            if constexpr (GAP == 0) asm volatile(PRE G0 "v_pk_fma_f32 %0, v[100:101], %3, %8 op_sel:[0,1,0]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
            else asm volatile(PRE G1 "v_pk_fma_f32 %0, v[100:101], %3, %8 op_sel:[0,1,0]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
        }
        if constexpr (FORM == 17) {
            e0 = a1 * c1; e1 = a0 * c0;                          // BOTH sources swapped (op_sel:[1,1] op_sel_hi:[0,0])
            if constexpr (GAP == 0) asm volatile(PRE G0 "v_pk_mul_f32 %0, v[100:101], %3 op_sel:[1,1] op_sel_hi:[0,0]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
            else asm volatile(PRE G1 "v_pk_mul_f32 %0, v[100:101], %3 op_sel:[1,1] op_sel_hi:[0,0]" : "=v"(r) : "v"(x0), "v"(x1), "v"(cv), "v"(y0), "v"(y1), "s"(0), "s"(kp), "v"(dv) : "v100", "v101");
        }
        asm volatile("s_nop 3" ::: "memory");
        const bool w0 = __float_as_uint(r.x) != __float_as_uint(e0), w1 = __float_as_uint(r.y) != __float_as_uint(e1);
        bad += (w0 ? 1u : 0u) + (w1 ? 1u : 0u);
    }
    out[t] = bad;
}

template <int FORM, int GAP>
static int go(const float* in, unsigned* out, int blocks, int iters, hipStream_t s) {
    hipLaunchKernelGGL((hz_kernel<FORM, GAP>), dim3(blocks), dim3(256), 0, s, in, out, iters);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int mk_probe_pk_forms() { return 18; }
extern "C" const char* mk_probe_pk_form_name(int f) {
    static const char* names[] = {"pk_add plain", "pk_mul sgpr-pair", "pk_mul swap src1", "pk_mul swap src0", "pk_mul bcast src1.lo", "pk_mul bcast src1.hi", "pk_add swap src1", "pk_fma cmul form", "pk_fma swap src0, sgpr src1", "pk_fma plain", "pk_mul sgpr swap", "pk_fma bcast src1.lo", "pk_mov swap (v_pk_mov_b32)", "pk_add swap+neg src0 (sub_yx)", "pk_fma swap src1 vgpr", "pk_fma swap src2 vgpr", "pk_fma bcast src1.hi vgpr", "pk_mul swap BOTH"};
    return (f >= 0 && f < 18) ? names[f] : "?";
}
// in: 2 * blocks * 256 floats, out: blocks * 256 counters
extern "C" int mk_probe_pk_hazard(int form, int gap, const void* in, void* out, int blocks, int iters, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const float* i = (const float*)in;
    unsigned* o = (unsigned*)out;
    switch (form * 2 + gap) {
        case 0: return go<0, 0>(i, o, blocks, iters, s);
        case 1: return go<0, 1>(i, o, blocks, iters, s);
        case 2: return go<1, 0>(i, o, blocks, iters, s);
        case 3: return go<1, 1>(i, o, blocks, iters, s);
        case 4: return go<2, 0>(i, o, blocks, iters, s);
        case 5: return go<2, 1>(i, o, blocks, iters, s);
        case 6: return go<3, 0>(i, o, blocks, iters, s);
        case 7: return go<3, 1>(i, o, blocks, iters, s);
        case 8: return go<4, 0>(i, o, blocks, iters, s);
        case 9: return go<4, 1>(i, o, blocks, iters, s);
        case 10: return go<5, 0>(i, o, blocks, iters, s);
        case 11: return go<5, 1>(i, o, blocks, iters, s);
        case 12: return go<6, 0>(i, o, blocks, iters, s);
        case 13: return go<6, 1>(i, o, blocks, iters, s);
        case 14: return go<7, 0>(i, o, blocks, iters, s);
        case 15: return go<7, 1>(i, o, blocks, iters, s);
        case 16: return go<8, 0>(i, o, blocks, iters, s);
        case 17: return go<8, 1>(i, o, blocks, iters, s);
        case 18: return go<9, 0>(i, o, blocks, iters, s);
        case 19: return go<9, 1>(i, o, blocks, iters, s);
        case 20: return go<10, 0>(i, o, blocks, iters, s);
        case 21: return go<10, 1>(i, o, blocks, iters, s);
        case 22: return go<11, 0>(i, o, blocks, iters, s);
        case 23: return go<11, 1>(i, o, blocks, iters, s);
        case 24: return go<12, 0>(i, o, blocks, iters, s);
        case 25: return go<12, 1>(i, o, blocks, iters, s);
        case 26: return go<13, 0>(i, o, blocks, iters, s);
        case 27: return go<13, 1>(i, o, blocks, iters, s);
        case 28: return go<14, 0>(i, o, blocks, iters, s);
        case 29: return go<14, 1>(i, o, blocks, iters, s);
        case 30: return go<15, 0>(i, o, blocks, iters, s);
        case 31: return go<15, 1>(i, o, blocks, iters, s);
        case 32: return go<16, 0>(i, o, blocks, iters, s);
        case 33: return go<16, 1>(i, o, blocks, iters, s);
        case 34: return go<17, 0>(i, o, blocks, iters, s);
        case 35: return go<17, 1>(i, o, blocks, iters, s);
        default: return 1;
    }
}
