// Fused AdamW step over one flat fp32 tensor (complex64 parameters are passed as their
// interleaved real view).  One HBM pass: reads p, g, m, v and writes p, m, v (28 B / element);
// the global-norm clipping coefficient is read from device memory so the host never syncs.
// Update rule = torch.optim.AdamW (decoupled weight decay, bias-corrected moments).
#include "common.h"

// The step streams 28 bytes per element through the chip exactly once (16 GB per step for the SFNO's spectral weights)
// and nothing of it is read again before the caches have turned over many times: non-temporal loads and stores, and a grid
// of (nearly) one 16-byte vector per lane instead of 4 096 resident blocks walking a grid-stride loop.  Same box, 70.8 M
// floats: 394-414 us (4.8-5.0 TB/s) -> 322 us = 6.15 TB/s (profiles/r03_ab_adamw.txt; the chip's copy rate is 6.3)
#ifndef MK_ADAMW_NT
#define MK_ADAMW_NT 1
#endif
#ifndef MK_ADAMW_BLOCKS       // cap of the grid, in blocks per CU
#define MK_ADAMW_BLOCKS 256
#endif

namespace {
constexpr int NT = 256;

__global__ __launch_bounds__(NT) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long n,
                                                   const float* __restrict__ grad_scale, float lr, float beta1,
                                                   float beta2, float eps, float weight_decay, float bc1_inv,
                                                   float bc2_rsqrt, const float* __restrict__ bc) {
    if (bc) bc1_inv = bc[1], bc2_rsqrt = bc[2];      // bias corrections of a device-side step counter (mk_adamw_advance)
    const float gs = grad_scale ? *grad_scale : 1.f;
    const float decay = 1.f - lr * weight_decay;
    const float step = lr * bc1_inv;
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * NT;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n4; i += stride) {
#if MK_ADAMW_NT
        f32x4 pv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(p) + i);
        const f32x4 gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + i) * gs;
        f32x4 mv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(m) + i);
        f32x4 vv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(v) + i);
#else
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i] * gs;
        f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
        f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#endif
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mv[e] = beta1 * mv[e] + (1.f - beta1) * gv[e];
            vv[e] = beta2 * vv[e] + (1.f - beta2) * gv[e] * gv[e];
            const float denom = sqrtf(vv[e]) * bc2_rsqrt + eps;
            pv[e] = pv[e] * decay - step * (mv[e] / denom);
        }
#if MK_ADAMW_NT
        __builtin_nontemporal_store(pv, reinterpret_cast<f32x4*>(p) + i);
        __builtin_nontemporal_store(mv, reinterpret_cast<f32x4*>(m) + i);
        __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v) + i);
#else
        reinterpret_cast<f32x4*>(p)[i] = pv;
        reinterpret_cast<f32x4*>(m)[i] = mv;
        reinterpret_cast<f32x4*>(v)[i] = vv;
#endif
    }
    // tail
    if (blockIdx.x == 0) {
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += NT) {
            const float gi = g[i] * gs;
            const float mi = beta1 * m[i] + (1.f - beta1) * gi;
            const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
            m[i] = mi;
            v[i] = vi;
            p[i] = p[i] * decay - step * (mi / (sqrtf(vi) * bc2_rsqrt + eps));
        }
    }
}

// ---- many small tensors in one launch ---------------------------------------------------------
// The SFNO has 8 spectral weights of 70.8 M floats and ~80 tensors of a few hundred to 300 K floats; one
// launch per small tensor costs more in dispatch gaps and host time than in HBM time.  Up to MULTI_MAX tensors
// ride in the kernel-argument block; block b works on tensor t with first[t] <= b < first[t+1].
constexpr int MULTI_MAX = 48;
constexpr int MULTI_CHUNK = NT * 4 * 4;       // elements per block

struct AdamMulti {
    u16* pb[MULTI_MAX];          // optional bf16 copy of the updated parameter (the autocast operand of the next step)
    u16* pbt[MULTI_MAX];         // optional bf16 copy of its transpose (the data-gradient GEMM's operand)
    int cols[MULTI_MAX];         // > 0: the parameter is a (n / cols, cols) matrix; pb has row pitch ld, pbt row pitch ldt
    int ld[MULTI_MAX];
    int ldt[MULTI_MAX];
    float* p[MULTI_MAX];
    const float* g[MULTI_MAX];
    float* m[MULTI_MAX];
    float* v[MULTI_MAX];
    long long n[MULTI_MAX];
    int first[MULTI_MAX + 1];
    int count;
};

__global__ __launch_bounds__(NT) void adamw_multi_kernel(const AdamMulti a, const float* __restrict__ grad_scale, float lr,
                                                         float beta1, float beta2, float eps, float weight_decay,
                                                         float bc1_inv, float bc2_rsqrt, const float* __restrict__ bc) {
    if (bc) bc1_inv = bc[1], bc2_rsqrt = bc[2];
    int t = 0;
    while (t + 1 < a.count && (int)blockIdx.x >= a.first[t + 1]) ++t;
    float* __restrict__ p = a.p[t];
    u16* __restrict__ pb = a.pb[t];
    u16* __restrict__ pbt = a.pbt[t];
    const int cols = a.cols[t], ld = a.ld[t], ldt = a.ldt[t];
    const float* __restrict__ g = a.g[t];
    float* __restrict__ m = a.m[t];
    float* __restrict__ v = a.v[t];
    const long long n = a.n[t];
    const long long e0 = (long long)((int)blockIdx.x - a.first[t]) * MULTI_CHUNK;
    const long long e1 = min(n, e0 + MULTI_CHUNK);
    const float gs = grad_scale ? *grad_scale : 1.f;
    const float decay = 1.f - lr * weight_decay;
    const float step = lr * bc1_inv;
    for (long long i = e0 + threadIdx.x; i < e1; i += NT) {       // small tensors: scalar accesses, any alignment
        const float gi = g[i] * gs;
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float pn = p[i] * decay - step * (mi / (sqrtf(vi) * bc2_rsqrt + eps));
        p[i] = pn;
        if (pb || pbt) {
            const u16 h = f32_to_bf16(pn);
            if (cols > 0) {
                const long long r = i / cols, c = i - r * cols;
                if (pb) pb[r * ld + c] = h;
                if (pbt) pbt[c * ldt + r] = h;
            } else if (pb) {
                pb[i] = h;
            }
        }
    }
}

// step counter and bias corrections in device memory: st = {step, 1 / (1 - beta1^step), 1 / sqrt(1 - beta2^step)};
// one thread advances it, every update kernel of the step reads it (nothing about the step number is baked into a
// launch, so a captured hipGraph of the train step replays correctly)
__global__ void adamw_advance_kernel(float* __restrict__ st, float beta1, float beta2) {
    const double step = (double)st[0] + 1.0;
    st[0] = (float)step;
    st[1] = (float)(1.0 / (1.0 - pow((double)beta1, step)));
    st[2] = (float)(1.0 / sqrt(1.0 - pow((double)beta2, step)));
}

// ---- global gradient norm -> clipping coefficient, over all tensors in 2-3 launches ---------------
constexpr int SQ_CHUNK = NT * 4 * 16;         // elements per block

struct SumsqMulti {
    const float* g[MULTI_MAX];
    long long n[MULTI_MAX];
    int first[MULTI_MAX + 1];
    int count;
};

__global__ __launch_bounds__(NT) void sumsq_multi_kernel(const SumsqMulti a, float* __restrict__ partial) {
    __shared__ float red[NT / 64];
    int t = 0;
    while (t + 1 < a.count && (int)blockIdx.x >= a.first[t + 1]) ++t;
    const float* __restrict__ g = a.g[t];
    const long long n = a.n[t];
    const long long e0 = (long long)((int)blockIdx.x - a.first[t]) * SQ_CHUNK;
    const long long e1 = min(n, e0 + SQ_CHUNK);
    float s = 0.f;
    if ((((uintptr_t)g) & 15) == 0) {
        const long long v1 = e0 + ((e1 - e0) & ~3ll);
        for (long long i = e0 + (long long)threadIdx.x * 4; i < v1; i += NT * 4) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(g + i);
            s += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
        }
        for (long long i = v1 + threadIdx.x; i < e1; i += NT) s += g[i] * g[i];
    } else {
        for (long long i = e0 + threadIdx.x; i < e1; i += NT) s += g[i] * g[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tsum = 0.f;
        for (int i = 0; i < NT / 64; ++i) tsum += red[i];
        partial[blockIdx.x] = tsum;
    }
}

// out[0] = clip coefficient min(1, max_norm / (norm + 1e-6)), out[1] = norm   (fixed summation order)
__global__ __launch_bounds__(NT) void clip_coef_kernel(const float* __restrict__ partial, int nparts, float max_norm,
                                                       float* __restrict__ out) {
    __shared__ double red[NT];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += NT) s += (double)partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = NT / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(red[0]);
        out[1] = norm;
        out[0] = max_norm > 0.f ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f;
    }
}
}  // namespace

extern "C" int mk_adamw_advance(float* state, float beta1, float beta2, void* stream) {
    MK_REQUIRE(state, "adamw_advance: null pointer");
    hipLaunchKernelGGL(adamw_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, beta1, beta2);
    return mk_check_launch("mk_adamw_advance");
}

extern "C" int mk_adamw_step(float* p, const float* g, float* m, float* v, long long n, const float* grad_scale,
                             float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                             const float* step_state, void* stream) {
    MK_REQUIRE(p && g && m && v && n > 0 && (step >= 1 || step_state), "adamw: bad args");
    if (step < 1) step = 1;
    MK_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adamw: pointers must be 16-byte aligned");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    long long blocks = (n / 4 + NT - 1) / NT;
    if (blocks > 256 * MK_ADAMW_BLOCKS) blocks = 256 * MK_ADAMW_BLOCKS;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, p, g, m, v, n, grad_scale,
                       lr, beta1, beta2, eps, weight_decay, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), step_state);
    return mk_check_launch("mk_adamw_step");
}

extern "C" int mk_adamw_multi(const MkAdamTensor* tensors, int count, const float* grad_scale, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, const float* step_state, void* stream) {
    MK_REQUIRE(tensors && count > 0 && (step >= 1 || step_state), "adamw_multi: bad args");
    if (step < 1) step = 1;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    for (int base = 0; base < count; base += MULTI_MAX) {
        AdamMulti a;
        a.count = count - base < MULTI_MAX ? count - base : MULTI_MAX;
        int blocks = 0;
        for (int t = 0; t < a.count; ++t) {
            const MkAdamTensor& s = tensors[base + t];
            MK_REQUIRE(s.p && s.g && s.m && s.v && s.n > 0, "adamw_multi: tensor %d has a null pointer or no elements", base + t);
            MK_REQUIRE(s.n < (1ll << 40), "adamw_multi: tensor %d too large for the multi-tensor path", base + t);
            a.p[t] = s.p, a.g[t] = s.g, a.m[t] = s.m, a.v[t] = s.v, a.n[t] = s.n, a.pb[t] = (u16*)s.p_bf16;
            a.pbt[t] = (u16*)s.p_bf16_t, a.cols[t] = s.cols, a.ld[t] = s.ld, a.ldt[t] = s.ld_t;
            MK_REQUIRE(s.cols >= 0 && (s.cols == 0 || (s.n % s.cols == 0 && s.ld >= s.cols && s.ld_t >= s.n / s.cols)) && (s.cols > 0 || !s.p_bf16_t),
                       "adamw_multi: tensor %d: bad matrix shape for the bf16 copies", base + t);
            a.first[t] = blocks;
            blocks += (int)((s.n + MULTI_CHUNK - 1) / MULTI_CHUNK);
        }
        a.first[a.count] = blocks;
        hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, a, grad_scale, lr,
                           beta1, beta2, eps, weight_decay, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), step_state);
    }
    return mk_check_launch("mk_adamw_multi");
}

extern "C" long long mk_grad_norm_workspace(const MkAdamTensor* tensors, int count) {
    long long blocks = 0;
    for (int t = 0; t < count; ++t) blocks += (tensors[t].n + SQ_CHUNK - 1) / SQ_CHUNK;
    return blocks;
}

// partial sums of squares that producers already formed (mk_cgemm_split2_batched_ssq): summed NT at a time into the array behind the
// partials of the tensors that are read here, so that clip_coef_kernel sums ONE array in ONE fixed order
static __global__ __launch_bounds__(NT) void gather_partials_kernel(const SumsqMulti a, float* __restrict__ partial) {
    __shared__ float red[NT / 64];
    int t = 0;
    while (t + 1 < a.count && (int)blockIdx.x >= a.first[t + 1]) ++t;
    const long long i = (long long)((int)blockIdx.x - a.first[t]) * NT + threadIdx.x;
    float s = i < a.n[t] ? a.g[t][i] : 0.f;                 // NT of the producer's partials -> one (fixed order)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tsum = 0.f;
        for (int w = 0; w < NT / 64; ++w) tsum += red[w];
        partial[blockIdx.x] = tsum;
    }
}

static long long pre_slots(const MkAdamTensor* pre, int npre) {
    long long blocks = 0;
    for (int t = 0; t < npre; ++t) blocks += (pre[t].n + NT - 1) / NT;
    return blocks;
}

extern "C" long long mk_grad_norm_workspace_pre(const MkAdamTensor* tensors, int count, const MkAdamTensor* pre, int npre) {
    return mk_grad_norm_workspace(tensors, count) + pre_slots(pre, npre);
}

static int grad_clip_coef_impl(const MkAdamTensor* tensors, int count, const MkAdamTensor* pre, int npre, float max_norm,
                               float* partial, float* out, void* stream);

extern "C" int mk_grad_clip_coef(const MkAdamTensor* tensors, int count, float max_norm, float* partial, float* out,
                                 void* stream) {
    MK_REQUIRE(tensors && count > 0, "grad_clip_coef: bad args");
    return grad_clip_coef_impl(tensors, count, nullptr, 0, max_norm, partial, out, stream);
}

extern "C" int mk_grad_clip_coef_pre(const MkAdamTensor* tensors, int count, const MkAdamTensor* pre, int npre, float max_norm,
                                     float* partial, float* out, void* stream) {
    MK_REQUIRE((count > 0 || npre > 0) && (count == 0 || tensors) && (npre == 0 || pre), "grad_clip_coef_pre: bad args");
    return grad_clip_coef_impl(tensors, count, pre, npre, max_norm, partial, out, stream);
}

static int grad_clip_coef_impl(const MkAdamTensor* tensors, int count, const MkAdamTensor* pre, int npre, float max_norm,
                               float* partial, float* out, void* stream) {
    MK_REQUIRE(partial && out, "grad_clip_coef: bad args");
    hipStream_t s = (hipStream_t)stream;
    long long done = 0;
    for (int base = 0; base < count; base += MULTI_MAX) {
        SumsqMulti a;
        a.count = count - base < MULTI_MAX ? count - base : MULTI_MAX;
        long long blocks = 0;
        for (int t = 0; t < a.count; ++t) {
            const MkAdamTensor& d = tensors[base + t];
            MK_REQUIRE(d.g && d.n > 0, "grad_clip_coef: tensor %d has no gradient", base + t);
            a.g[t] = d.g, a.n[t] = d.n;
            a.first[t] = (int)blocks;
            blocks += (d.n + SQ_CHUNK - 1) / SQ_CHUNK;
            MK_REQUIRE(blocks < (1ll << 30), "grad_clip_coef: too many blocks");
        }
        a.first[a.count] = (int)blocks;
        hipLaunchKernelGGL(sumsq_multi_kernel, dim3((unsigned)blocks), dim3(NT), 0, s, a, partial + done);
        done += blocks;
    }
    for (int base = 0; base < npre; base += MULTI_MAX) {
        SumsqMulti a;
        a.count = npre - base < MULTI_MAX ? npre - base : MULTI_MAX;
        long long blocks = 0;
        for (int t = 0; t < a.count; ++t) {
            const MkAdamTensor& d = pre[base + t];
            MK_REQUIRE(d.g && d.n > 0, "grad_clip_coef_pre: partial buffer %d is empty", base + t);
            a.g[t] = d.g, a.n[t] = d.n;
            a.first[t] = (int)blocks;
            blocks += (d.n + NT - 1) / NT;
            MK_REQUIRE(blocks < (1ll << 22), "grad_clip_coef_pre: too many partials");
        }
        a.first[a.count] = (int)blocks;
        hipLaunchKernelGGL(gather_partials_kernel, dim3((unsigned)blocks), dim3(NT), 0, s, a, partial + done);
        done += blocks;
    }
    MK_REQUIRE(done < (1ll << 31), "grad_clip_coef: too many partials");
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(NT), 0, s, partial, (int)done, max_norm, out);
    return mk_check_launch("mk_grad_clip_coef");
}
