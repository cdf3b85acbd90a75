// MK_HIPCC_FLAGS: -fno-slp-vectorize
// (gfx950: v_pk_mul_f32 / v_pk_add_f32 whose src1 is a VGPR pair read through op_sel return wrong results while certain
//  matrix-core kernels run on the same compute unit — tools/pk_hazard_probe.py, docs/LAB_NOTEBOOK.md round 6.  The SLP vectoriser
//  emits exactly those forms from plain scalar code, so this file is compiled without it; tools/pk_opsel_scan.py checks the ISA.)
// HBM-streaming pointwise kernels on NCHW planes (gfx950):
//   instance norm (stats / apply / backward) with optional fused exact-erf GELU,
//   bias + GELU forward / backward.
// All kernels move 16 B per lane (4 x f32 or 8 x bf16), compute in fp32, and are laid out as
// (plane, chunk) work items: one block streams CHUNK contiguous elements of one (batch, channel)
// plane, so per-plane scalars (mean, rstd, gamma, beta, bias) are block-uniform.
#include "common.h"
#include <type_traits>

// non-temporal loads in the passes that read their inputs for the last time (for_chunk<..., LAST>): the backward apply pass
// of the instance norm 91.6 -> 80.5 us (cold 88 MB planes, same box, profiles/r03_ab_pointwise_nt.txt); the forward apply pass
// (one input, already cached by the statistics pass) is unchanged
#ifndef MK_PW_NT
#define MK_PW_NT 1
#endif
#ifndef MK_PW_ST_NT            // A/B knob: plane-sized outputs with the streaming (nt) store policy
#define MK_PW_ST_NT 0
#endif
#ifndef MK_PW_DIAG_VGPR
#define MK_PW_DIAG_VGPR 0
#endif

namespace {

constexpr int NT = 256;
constexpr int UNROLL = 4;

template <typename T>
struct VecIO;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <>
struct VecIO<float> {
    static constexpr int N = 4;
    typedef f32x4 Raw;
    __device__ static __forceinline__ Raw load_raw(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    __device__ static __forceinline__ Raw load_raw_nt(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }
    __device__ static __forceinline__ void unpack(const Raw& r, float* v) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = r[i];
    }
    __device__ static __forceinline__ void load(const float* p, float* v) { unpack(load_raw(p), v); }
    __device__ static __forceinline__ void store(float* p, const float* v) {
        f32x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = v[i];
        if constexpr (MK_PW_ST_NT) __builtin_nontemporal_store(r, reinterpret_cast<f32x4*>(p));
        else *reinterpret_cast<f32x4*>(p) = r;
    }
    __device__ static __forceinline__ float load1(const float* p) { return *p; }
    __device__ static __forceinline__ void store1(float* p, float v) { *p = v; }
};
template <>
struct VecIO<u16> {
    static constexpr int N = 8;
    typedef u32x4 Raw;
    __device__ static __forceinline__ Raw load_raw(const u16* p) { return *reinterpret_cast<const u32x4*>(p); }
    __device__ static __forceinline__ Raw load_raw_nt(const u16* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
    __device__ static __forceinline__ void unpack(const Raw& r, float* v) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(r[i] << 16);
            v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u);
        }
    }
    __device__ static __forceinline__ void load(const u16* p, float* v) { unpack(load_raw(p), v); }
    __device__ static __forceinline__ void store(u16* p, const float* v) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
        if constexpr (MK_PW_ST_NT) __builtin_nontemporal_store(u32x4{w[0], w[1], w[2], w[3]}, reinterpret_cast<u32x4*>(p));
        else *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __device__ static __forceinline__ float load1(const u16* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void store1(u16* p, float v) { *p = f32_to_bf16(v); }
};

// one pass of a block: NT lanes x 16 B x UNROLL loads in flight per lane
template <typename T>
__host__ __device__ constexpr long long pass_elems() {
    return (long long)NT * VecIO<T>::N * UNROLL;
}

// A plane of hw elements is cut into `chunks` equal pieces (a multiple of the vector width each); the launch picks
// `chunks` so that all planes * chunks blocks are resident at once (chunks_for): a block then streams its piece in
// passes instead of paying launch + prologue latency per 16 KB.
__host__ __device__ inline long long chunk_len(long long hw, int chunks, int vec) {
    const long long per = (hw + chunks - 1) / chunks;
    return (per + vec - 1) / vec * vec;
}

// Visit every element of this block's chunk of NIN (1 or 2) same-shaped input planes p[0], p[1].
// f(index_in_plane, cnt, a, b): cnt (an integral_constant) values of each input, already converted to fp32; the vector
// path (cnt = VEC) needs 16-byte aligned plane bases (hw % VEC == 0).  All UNROLL x NIN 16-byte loads of a pass are issued
// before the first value is used: lanes beyond the end of the chunk read the chunk's first vector instead (a load under a
// lane condition makes hipcc wait for it on the spot, which left ONE load in flight per lane), and only the arithmetic and
// the stores of `f` sit under the lane condition.
template <int C>
using cnt_t = std::integral_constant<int, C>;

// LAST: the inputs are not read again before the caches have turned over -> non-temporal loads (MK_PW_NT)
template <typename T, int NIN, bool LAST = false, typename F>
__device__ __forceinline__ void for_chunk(long long hw, int chunks, int chunk, const T* p0, const T* p1, F&& f) {
    constexpr int VEC = VecIO<T>::N;
    typedef typename VecIO<T>::Raw Raw;
    const long long len = chunk_len(hw, chunks, VEC);
    const long long c0 = (long long)chunk * len;
    const long long c1 = min(hw, c0 + len);
    if ((hw % VEC) == 0) {
        constexpr int U = UNROLL / NIN;          // the same number of loads in flight per lane for one and two inputs
        for (long long base = c0; base < c1; base += (long long)NT * VEC * U) {
            Raw ra[U], rb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long e = base + ((long long)u * NT + threadIdx.x) * VEC;
                const long long a = e < c1 ? e : c0;
                if constexpr (LAST && MK_PW_NT) {
                    ra[u] = VecIO<T>::load_raw_nt(p0 + a);
                    if (NIN > 1) rb[u] = VecIO<T>::load_raw_nt(p1 + a);
                } else {
                    ra[u] = VecIO<T>::load_raw(p0 + a);
                    if (NIN > 1) rb[u] = VecIO<T>::load_raw(p1 + a);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long e = base + ((long long)u * NT + threadIdx.x) * VEC;
                if (e < c1) {
                    float va[VEC], vb[VEC];
                    VecIO<T>::unpack(ra[u], va);
                    if (NIN > 1) VecIO<T>::unpack(rb[u], vb);
                    f(e, cnt_t<VEC>{}, va, vb);
                }
            }
        }
    } else {
        for (long long e = c0 + threadIdx.x; e < c1; e += NT) {
            float va[1], vb[1] = {0.f};
            va[0] = VecIO<T>::load1(p0 + e);
            if (NIN > 1) vb[0] = VecIO<T>::load1(p1 + e);
            f(e, cnt_t<1>{}, va, vb);
        }
    }
}

// store cnt (VEC or 1) values
template <typename T, int C>
__device__ __forceinline__ void store_n(T* p, const float* v, cnt_t<C>) {
    if constexpr (C == 1) VecIO<T>::store1(p, v[0]);
    else VecIO<T>::store(p, v);
}

// GELU of a bf16 tensor: the packed exp2 form of common.h (relative error 4e-6 at any x, one quarter-rate instruction); GELU' keeps
// the A&S 7.1.26 erf, which also yields the exp(-x^2 / 2) it needs; fp32 tensors take the exact erff
template <typename T>
__device__ __forceinline__ float gelu_t(float a) {
    if constexpr (sizeof(T) == 2) return gelu_fast_f(a);
    else return gelu_f(a);
}
template <typename T, int N_>
__device__ __forceinline__ void gelu_n(float* v) {
    if constexpr (sizeof(T) == 2) gelu_fast_n<N_>(v);
    else {
#pragma unroll
        for (int i = 0; i < N_; ++i) v[i] = gelu_f(v[i]);
    }
}
template <typename T>
__device__ __forceinline__ float gelu_grad_t(float a) {
    if constexpr (sizeof(T) == 2) return gelu_grad_fast_f(a);
    else return gelu_grad_f(a);
}
// out[i] = GELU'(n[i] g + b): the ONE form every backward kernel of the norm uses (the statistics pass, the apply pass and the one-pass
// kernel must multiply by the same bits)
template <typename T, int N_>
__device__ __forceinline__ void gelu_grad_affine_n(const float* n, float g, float b, float* out) {
    float arg[N_];
#pragma unroll
    for (int i = 0; i < N_; ++i) arg[i] = fmaf(n[i], g, b);
    if constexpr (sizeof(T) == 2) gelu_grad_fast_n<N_>(arg, out);
    else {
#pragma unroll
        for (int i = 0; i < N_; ++i) out[i] = gelu_grad_f(arg[i]);
    }
}

__device__ __forceinline__ void block_reduce2(float& a, float& b, float* red /*[2*NT/64]*/) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_down(a, o, 64);
        b += __shfl_down(b, o, 64);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        red[2 * w] = a;
        red[2 * w + 1] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = 0.f;
        b = 0.f;
        for (int i = 0; i < NT / 64; ++i) {
            a += red[2 * i];
            b += red[2 * i + 1];
        }
    }
}


// Sum the `chunks` (s1, s2) partial pairs one plane's blocks left in ws; every thread gets the totals.
// Lets the consumer kernel finish the reduction itself instead of waiting for a tiny finalize launch.
__device__ __forceinline__ void plane_totals(const float* __restrict__ ws, long long plane, int chunks, double& t1,
                                             double& t2, double* red /*[2*NT/64]*/) {
    double a = 0.0, b = 0.0;
    for (int c = threadIdx.x; c < chunks; c += NT) {
        a += (double)ws[2 * (plane * chunks + c)];
        b += (double)ws[2 * (plane * chunks + c) + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_down(a, o, 64);
        b += __shfl_down(b, o, 64);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        red[2 * w] = a;
        red[2 * w + 1] = b;
    }
    __syncthreads();
    t1 = 0.0, t2 = 0.0;
    for (int i = 0; i < NT / 64; ++i) {
        t1 += red[2 * i];
        t2 += red[2 * i + 1];
    }
}

// value the norm sees when a per-channel constant `pb` (the bias of the convolution in front of it) is folded in:
// exactly what the reference materialises, i.e. (x + pb) rounded to the tensor dtype (idempotent for pb = 0)
template <typename T>
__device__ __forceinline__ float pre(float v, float pb, bool on);
template <>
__device__ __forceinline__ float pre<float>(float v, float pb, bool on) { return on ? v + pb : v; }
template <>
__device__ __forceinline__ float pre<u16>(float v, float pb, bool on) {
    if (!on) return v;                        // block-uniform: no rounding work when nothing is folded in
    return bf16_to_f32(f32_to_bf16(v + pb));
}

// ---- instance norm: statistics ---------------------------------------------------
// partial sums relative to a per-plane pivot (first element) to avoid cancellation in fp32
template <typename T>
__global__ __launch_bounds__(NT) void in_stats_partial(const T* __restrict__ x, float* __restrict__ ws, long long hw,
                                                       int chunks, const float* __restrict__ pre_bias, int channels,
                                                       const float* __restrict__ q) {
    __shared__ float red[2 * NT / 64];
    const long long plane = blockIdx.x / chunks;
    const int chunk = blockIdx.x % chunks;
    const T* xp = x + plane * hw;
    const float pb = pre_bias ? pre_bias[plane % channels] : 0.f;
    const float pivot = pre<T>(VecIO<T>::load1(xp), pb, pre_bias != nullptr);
    float s1 = 0.f, s2 = 0.f;
    for_chunk<T, 1>(hw, chunks, chunk, xp, xp, [&](long long e, auto cnt, const float* v, const float*) {
#pragma unroll
        for (int i = 0; i < cnt(); ++i) {
            const float d = pre<T>(v[i], pb, pre_bias != nullptr) - pivot;
            const float w = q ? q[e + i] : 1.f;           // quadrature weights (sum 1): geometric norm on the sphere
            s1 += w * d;
            s2 += w * d * d;
        }
    });
    block_reduce2(s1, s2, red);
    if (threadIdx.x == 0) {
        ws[2 * (long long)blockIdx.x] = s1;
        ws[2 * (long long)blockIdx.x + 1] = s2;
    }
}

// ---- plane sums (bias gradients of the 1x1 convolutions: sum of the output gradient over batch-plane pixels) -------
template <typename T>
__global__ __launch_bounds__(NT) void plane_sum_partial(const T* __restrict__ x, float* __restrict__ ws, long long hw,
                                                        int chunks) {
    __shared__ float red[2 * NT / 64];
    const long long plane = blockIdx.x / chunks;
    const int chunk = blockIdx.x % chunks;
    const T* xp = x + plane * hw;
    float s1 = 0.f, s2 = 0.f;
    for_chunk<T, 1>(hw, chunks, chunk, xp, xp, [&](long long e, auto cnt, const float* v, const float*) {
        if constexpr (cnt() > 1) {
#pragma unroll
            for (int i = 0; i < cnt(); i += 2) {
                s1 += v[i];
                s2 += v[i + 1];
            }
        } else {
            s1 += v[0];
        }
    });
    s1 += s2;
    s2 = 0.f;
    block_reduce2(s1, s2, red);
    if (threadIdx.x == 0) {
        ws[2 * (long long)blockIdx.x] = s1;
        ws[2 * (long long)blockIdx.x + 1] = 0.f;
    }
}

template <typename T>
__global__ void in_stats_final(const T* __restrict__ x, const float* __restrict__ ws, float* __restrict__ stats,
                               long long planes, long long hw, int chunks, float eps, float qsum) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= planes) return;
    const double pivot = (double)VecIO<T>::load1(x + p * hw);
    double s1 = 0.0, s2 = 0.0;
    for (int c = 0; c < chunks; ++c) {
        s1 += (double)ws[2 * (p * chunks + c)];
        s2 += (double)ws[2 * (p * chunks + c) + 1];
    }
    // plain: count = hw; quadrature-weighted (qsum = sum of this shard's weights): count = qsum — the local moments the
    // distributed geometric norm merges (makani/mpu/layer_norm.py:207-222)
    const double cnt = qsum > 0.f ? (double)qsum : (double)hw;
    const double m = s1 / cnt;
    double var = s2 / cnt - m * m;   // biased variance, as nn.InstanceNorm2d
    if (var < 0.0) var = 0.0;
    stats[2 * p] = (float)(pivot + m);
    stats[2 * p + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// ---- instance norm: apply (+ GELU) --------------------------------------------------
template <typename T, bool GELU>
__global__ __launch_bounds__(NT) void in_apply(const T* __restrict__ x, T* __restrict__ y, float* __restrict__ stats,
                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                               const float* __restrict__ ws, float eps, int channels, long long hw,
                                               int chunks, const float* __restrict__ pre_bias, float qsum) {
    __shared__ double redd[2 * NT / 64];
    const long long plane = blockIdx.x / chunks;
    const int chunk = blockIdx.x % chunks;
    const int c = (int)(plane % channels);
    const float pb = pre_bias ? pre_bias[c] : 0.f;
    float mean, rstd;
    if (ws) {                    // statistics straight from the partial sums (same arithmetic as in_stats_final)
        double s1, s2;
        plane_totals(ws, plane, chunks, s1, s2, redd);
        const double pivot = (double)pre<T>(VecIO<T>::load1(x + plane * hw), pb, pre_bias != nullptr);
        double m, var;
        if (qsum > 0.f) {        // quadrature weights, Q = sum q (< 1 on a cropped grid): mean = sum q x, var = sum q (x - mean)^2
            const double Q = (double)qsum;
            const double mu = s1 + pivot * Q;
            var = s2 + 2.0 * (pivot - mu) * s1 + (pivot - mu) * (pivot - mu) * Q;
            m = mu - pivot;
        } else {
            m = s1 / (double)hw;
            var = s2 / (double)hw - m * m;
        }
        if (var < 0.0) var = 0.0;
        mean = (float)(pivot + m);
        rstd = (float)(1.0 / sqrt(var + (double)eps));
        if (chunk == 0 && threadIdx.x == 0) {
            stats[2 * plane] = mean;
            stats[2 * plane + 1] = rstd;
        }
    } else {
        mean = stats[2 * plane], rstd = stats[2 * plane + 1];
    }
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = rstd * g, sh = b - mean * rstd * g;
    const T* xp = x + plane * hw;
    T* yp = y + plane * hw;
    for_chunk<T, 1, true>(hw, chunks, chunk, xp, xp, [&](long long e, auto cnt, const float* v, const float*) {
        float o[cnt()];
#pragma unroll
        for (int i = 0; i < cnt(); ++i) o[i] = pre<T>(v[i], pb, pre_bias != nullptr) * sc + sh;
        if (GELU) gelu_n<T, cnt()>(o);
        store_n(yp + e, o, cnt);
    });
}

// ---- instance norm backward ------------------------------------------------------------
// n = (x - mean) rstd ; a = n gamma + beta ; y = GELU ? gelu(a) : a ; ga = gy * (GELU ? gelu'(a) : 1)
// pass 1: S1 = sum ga, S2 = sum ga*n  (per plane)
template <typename T, bool GELU>
__global__ __launch_bounds__(NT) void in_bwd_partial(const T* __restrict__ x, const T* __restrict__ gy,
                                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ ws,
                                                     int channels, long long hw, int chunks,
                                                     const float* __restrict__ pre_bias) {
    __shared__ float red[2 * NT / 64];
    const long long plane = blockIdx.x / chunks;
    const int chunk = blockIdx.x % chunks;
    const int c = (int)(plane % channels);
    const float pb = pre_bias ? pre_bias[c] : 0.f;
#if MK_PW_DIAG_VGPR
    // diagnostic build (tools/two_stream_micro.py): the per-plane scalars live in VECTOR registers
    float mean, rstd;
    asm volatile("v_mov_b32 %0, %1" : "=v"(mean) : "s"(stats[2 * plane]));
    asm volatile("v_mov_b32 %0, %1" : "=v"(rstd) : "s"(stats[2 * plane + 1]));
#else
    const float mean = stats[2 * plane], rstd = stats[2 * plane + 1];
#endif
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const T* xp = x + plane * hw;
    const T* gp = gy + plane * hw;
    float s1 = 0.f, s2 = 0.f;
    for_chunk<T, 2>(hw, chunks, chunk, xp, gp, [&](long long e, auto cnt, const float* v, const float* d) {
        float nn[cnt()], dg[cnt()];
#pragma unroll
        for (int i = 0; i < cnt(); ++i) nn[i] = (pre<T>(v[i], pb, pre_bias != nullptr) - mean) * rstd;
        if (GELU) gelu_grad_affine_n<T, cnt()>(nn, g, b, dg);
#pragma unroll
        for (int i = 0; i < cnt(); ++i) {
            const float n = nn[i];
            float ga = d[i];
            if (GELU) ga *= dg[i];
            s1 += ga;
            s2 += ga * n;
        }
    });
    block_reduce2(s1, s2, red);
    if (threadIdx.x == 0) {
        ws[2 * (long long)blockIdx.x] = s1;
        ws[2 * (long long)blockIdx.x + 1] = s2;
    }
}

__global__ void sum_chunks_final(const float* __restrict__ ws, float* __restrict__ sums, long long planes, int chunks) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= planes) return;
    double s1 = 0.0, s2 = 0.0;
    for (int c = 0; c < chunks; ++c) {
        s1 += (double)ws[2 * (p * chunks + c)];
        s2 += (double)ws[2 * (p * chunks + c) + 1];
    }
    sums[p] = (float)s1;                 // planar (2, planes): each row is a contiguous per-plane vector
    sums[planes + p] = (float)s2;
}

// pass 2: gx = rstd * gamma * (ga - S1/hw - n * S2/hw)
template <typename T, bool GELU>
__global__ __launch_bounds__(NT) void in_bwd_apply(const T* __restrict__ x, const T* __restrict__ gy, T* __restrict__ gx,
                                                   const float* __restrict__ stats, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, float* __restrict__ sums,
                                                   const float* __restrict__ ws, int channels, long long hw, int chunks,
                                                   float inv_total, const float* __restrict__ pre_bias,
                                                   const float* __restrict__ q, float qsum) {
    __shared__ double redd[2 * NT / 64];
    const long long plane = blockIdx.x / chunks;
    const int chunk = blockIdx.x % chunks;
    const long long planes = gridDim.x / chunks;
    const int c = (int)(plane % channels);
    const float pb = pre_bias ? pre_bias[c] : 0.f;
    const float mean = stats[2 * plane], rstd = stats[2 * plane + 1];
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    float t1, t2;
    if (ws) {                    // finish the reduction of pass 1 here (and publish it for dgamma / dbeta)
        double d1, d2;
        plane_totals(ws, plane, chunks, d1, d2, redd);
        t1 = (float)d1, t2 = (float)d2;
        if (chunk == 0 && threadIdx.x == 0) {
            sums[plane] = t1;
            sums[planes + plane] = t2;
        }
    } else {
        t1 = sums[plane], t2 = sums[planes + plane];
    }
    const float m1 = q ? t1 : t1 * inv_total, m2 = q ? t2 : t2 * inv_total;     // weighted: the factor is q[i], applied per element
    const float cq = q ? mean * rstd * (1.f - qsum) : 0.f;                      // 0 when the weights sum to 1
    const float k = rstd * g;
    const T* xp = x + plane * hw;
    const T* gp = gy + plane * hw;
    T* op = gx + plane * hw;
    for_chunk<T, 2, true>(hw, chunks, chunk, xp, gp, [&](long long e, auto cnt, const float* v, const float* d) {
        float o[cnt()], nn[cnt()], dg[cnt()];
#pragma unroll
        for (int i = 0; i < cnt(); ++i) nn[i] = (pre<T>(v[i], pb, pre_bias != nullptr) - mean) * rstd;
        if (GELU) gelu_grad_affine_n<T, cnt()>(nn, g, b, dg);
#pragma unroll
        for (int i = 0; i < cnt(); ++i) {
            const float n = nn[i];
            float ga = d[i];
            if (GELU) ga *= dg[i];
            const float w = q ? q[e + i] : 1.f;
            o[i] = k * (ga - w * (m1 + (n - cq) * m2));
        }
        store_n(op + e, o, cnt);
    });
}

// ---- instance norm in ONE pass over the plane (round 6) -------------------------------------------------------------
// The two-kernel forms above read the plane twice (statistics, then apply: 3 plane-sized transfers forward, 5 backward).
// Here a block keeps its chunk of the plane IN REGISTERS between the two phases: load (every 16-byte vector in flight at
// once) -> partial sums -> publish -> wait for the partial sums of the plane's other blocks -> normalise from registers
// -> store: 2 transfers forward, 3 backward.
//   * The blocks of one plane have consecutive block indices and a bounded register footprint, so they are co-resident
//     or next in the dispatcher's in-order queue: the wait cannot deadlock (the lowest unfinished plane's blocks were all
//     dispatched before any block of a later plane on their XCD).
//   * Hand-over without fences: a block publishes its pair (s1, s2) as ONE 64-bit atomic store at agent scope into the
//     slot (plane, chunk); a slot holds the sentinel (all ones: a NaN pair no sum produces) until then.  Readers spin on
//     the slots with agent-scope atomic loads.  The last block of a plane to finish reading re-arms the plane's slots and
//     its departure counter, so the buffer is ready for the next launch on the same stream (also under hipGraph replay).
//   * Every block sums the plane's partial pairs in the same fixed order in fp64: all of them normalise with bit-identical
//     statistics, and the result does not depend on timing.
// 16-byte vectors a lane keeps per tensor (4 registers each), per kind of kernel (0 forward, 1 backward, 2 backward through the
// fused GELU).  Fewer slots = more, smaller blocks per plane = more blocks in different phases at once; measured on rotating
// 88 MB planes (profiles/r06_norm_one_pass.md): 15 slots 44.8 / 92.0 / 145 us, 8 slots 39.9 / 92.4 / 108, 5 slots 40.0 / 92.4 /
// 97.5 (forward / backward / backward + GELU at 240 x 480 x 384; two-kernel path 52.0 / 97.3 / 107), and at 721 x 1440 8 slots
// win the forward (0.375 ms) and the plain backward (0.520 vs 0.766 two-kernel), 5 the GELU backward (0.816 vs 0.910)
#ifndef MK_FUSED_SLOTS_FWD
#define MK_FUSED_SLOTS_FWD 8
#endif
#ifndef MK_FUSED_SLOTS_BWD
#define MK_FUSED_SLOTS_BWD 8
#endif
#ifndef MK_FUSED_SLOTS_BWD_GELU
#define MK_FUSED_SLOTS_BWD_GELU 5
#endif
__host__ __device__ constexpr int fused_slots(int kind) { return kind == 0 ? MK_FUSED_SLOTS_FWD : (kind == 1 ? MK_FUSED_SLOTS_BWD : MK_FUSED_SLOTS_BWD_GELU); }
constexpr unsigned long long FUSED_SENTINEL = ~0ull;

template <typename T>
__host__ __device__ constexpr long long fused_chunk_cap(int kind) {
    return (long long)NT * VecIO<T>::N * fused_slots(kind);
}

__device__ __forceinline__ void fused_publish(unsigned long long* slot, float s1, float s2) {
    unsigned long long v = ((unsigned long long)__float_as_uint(s2) << 32) | (unsigned long long)__float_as_uint(s1);
    if (v == FUSED_SENTINEL) v ^= 1ull;                  // (a NaN pair with this exact payload: keep it a NaN, not the sentinel)
    __hip_atomic_store(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// totals of the plane's `chunks` (<= NT) published pairs, identical in every block of the plane; then the departure
// protocol (the last block to leave re-arms the plane's slots).  All threads must call it.
__device__ __forceinline__ void fused_gather(unsigned long long* slots, unsigned* depart, long long plane, int chunks, double& t1,
                                             double& t2, double* redd /*[2*NT/64]*/) {
    double a = 0.0, b = 0.0;
    if ((int)threadIdx.x < chunks) {
        unsigned long long* sp = slots + plane * chunks + threadIdx.x;
        unsigned long long v;
        while ((v = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == FUSED_SENTINEL) __builtin_amdgcn_s_sleep(4);
        a = (double)__uint_as_float((unsigned)(v & 0xffffffffull));
        b = (double)__uint_as_float((unsigned)(v >> 32));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_down(a, o, 64);
        b += __shfl_down(b, o, 64);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        redd[2 * w] = a;
        redd[2 * w + 1] = b;
    }
    __syncthreads();                                     // (every thread of this block has read its slot)
    t1 = 0.0, t2 = 0.0;
    for (int i = 0; i < NT / 64; ++i) {
        t1 += redd[2 * i];
        t2 += redd[2 * i + 1];
    }
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(depart + plane, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)chunks - 1u) {              // everybody else has read: re-arm for the next launch
            for (int c = 0; c < chunks; ++c)
                __hip_atomic_store(slots + plane * chunks + c, FUSED_SENTINEL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(depart + plane, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <typename T, bool GELU>
__global__ __launch_bounds__(NT) void in_fwd_fused(const T* __restrict__ x, T* __restrict__ y, float* __restrict__ stats,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   unsigned long long* __restrict__ slots, unsigned* __restrict__ depart, float eps,
                                                   int channels, long long hw, int chunks, const float* __restrict__ pre_bias) {
    constexpr int VEC = VecIO<T>::N;
    constexpr int FUSED_SLOTS = fused_slots(0);
    typedef typename VecIO<T>::Raw Raw;
    __shared__ float red[2 * NT / 64];
    __shared__ double redd[2 * NT / 64];
    const long long plane = blockIdx.x / chunks;
    const int chunk = blockIdx.x % chunks;
    const int c = (int)(plane % channels);
    const bool has_pb = pre_bias != nullptr;
    const float pb = has_pb ? pre_bias[c] : 0.f;
    const T* xp = x + plane * hw;
    T* yp = y + plane * hw;
    const long long len = chunk_len(hw, chunks, VEC);
    const long long c0 = (long long)chunk * len;
    const long long c1 = min(hw, c0 + len);
    Raw raw[FUSED_SLOTS];
#pragma unroll
    for (int s = 0; s < FUSED_SLOTS; ++s) {
        const long long e = c0 + ((long long)s * NT + threadIdx.x) * VEC;
        raw[s] = VecIO<T>::load_raw_nt(xp + (e < c1 ? e : c0));             // (lanes past the end re-read the chunk's first vector)
    }
    const float pivot = pre<T>(VecIO<T>::load1(xp), pb, has_pb);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int s = 0; s < FUSED_SLOTS; ++s) {
        const long long e = c0 + ((long long)s * NT + threadIdx.x) * VEC;
        if (e < c1) {
            float v[VEC];
            VecIO<T>::unpack(raw[s], v);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float d = pre<T>(v[i], pb, has_pb) - pivot;
                s1 += d;
                s2 += d * d;
            }
        }
    }
    block_reduce2(s1, s2, red);
    if (threadIdx.x == 0) fused_publish(slots + plane * chunks + chunk, s1, s2);
    double t1, t2;
    fused_gather(slots, depart, plane, chunks, t1, t2, redd);
    const double m = t1 / (double)hw;
    double var = t2 / (double)hw - m * m;        // biased variance, as nn.InstanceNorm2d (same arithmetic as in_apply)
    if (var < 0.0) var = 0.0;
    const float mean = (float)((double)pivot + m);
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (chunk == 0 && threadIdx.x == 0) {
        stats[2 * plane] = mean;
        stats[2 * plane + 1] = rstd;
    }
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = rstd * g, sh = b - mean * rstd * g;
#pragma unroll
    for (int s = 0; s < FUSED_SLOTS; ++s) {
        const long long e = c0 + ((long long)s * NT + threadIdx.x) * VEC;
        if (e < c1) {
            float v[VEC], o[VEC];
            VecIO<T>::unpack(raw[s], v);
#pragma unroll
            for (int i = 0; i < VEC; ++i) o[i] = pre<T>(v[i], pb, has_pb) * sc + sh;
            if (GELU) gelu_n<T, VEC>(o);
            VecIO<T>::store(yp + e, o);
        }
    }
}

// backward in one pass: gx = rstd gamma (ga - S1 / hw - n S2 / hw) with S1 = sum ga, S2 = sum ga n over the plane
template <typename T, bool GELU>
__global__ __launch_bounds__(NT) void in_bwd_fused(const T* __restrict__ x, const T* __restrict__ gy, T* __restrict__ gx,
                                                   const float* __restrict__ stats, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, float* __restrict__ sums,
                                                   unsigned long long* __restrict__ slots, unsigned* __restrict__ depart, int channels,
                                                   long long hw, int chunks, float inv_total, const float* __restrict__ pre_bias) {
    constexpr int VEC = VecIO<T>::N;
    constexpr int FUSED_SLOTS = fused_slots(GELU ? 2 : 1);
    typedef typename VecIO<T>::Raw Raw;
    __shared__ float red[2 * NT / 64];
    __shared__ double redd[2 * NT / 64];
    const long long plane = blockIdx.x / chunks;
    const int chunk = blockIdx.x % chunks;
    const long long planes = gridDim.x / chunks;
    const int c = (int)(plane % channels);
    const bool has_pb = pre_bias != nullptr;
    const float pb = has_pb ? pre_bias[c] : 0.f;
    const float mean = stats[2 * plane], rstd = stats[2 * plane + 1];
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const T* xp = x + plane * hw;
    const T* gp = gy + plane * hw;
    T* op = gx + plane * hw;
    const long long len = chunk_len(hw, chunks, VEC);
    const long long c0 = (long long)chunk * len;
    const long long c1 = min(hw, c0 + len);
    Raw rx[FUSED_SLOTS], rg[FUSED_SLOTS];
#pragma unroll
    for (int s = 0; s < FUSED_SLOTS; ++s) {
        const long long e = c0 + ((long long)s * NT + threadIdx.x) * VEC;
        const long long a = e < c1 ? e : c0;
        rx[s] = VecIO<T>::load_raw_nt(xp + a);
        rg[s] = VecIO<T>::load_raw_nt(gp + a);
    }
    // through the fused GELU the activation gradient ga = gy gelu'(a) costs ~25 vector instructions per element: it is kept in
    // registers (fp32, the value the second phase would recompute bit for bit) instead of being evaluated twice
    constexpr int NGA = GELU ? FUSED_SLOTS * VEC : 1;
    float gak[NGA];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int s = 0; s < FUSED_SLOTS; ++s) {
        const long long e = c0 + ((long long)s * NT + threadIdx.x) * VEC;
        if (e < c1) {
            float v[VEC], d[VEC], nn[VEC], dg[VEC];
            VecIO<T>::unpack(rx[s], v);
            VecIO<T>::unpack(rg[s], d);
#pragma unroll
            for (int i = 0; i < VEC; ++i) nn[i] = (pre<T>(v[i], pb, has_pb) - mean) * rstd;
            if (GELU) gelu_grad_affine_n<T, VEC>(nn, g, b, dg);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float n = nn[i];
                float ga = d[i];
                if (GELU) {
                    ga *= dg[i];
                    gak[s * VEC + i] = ga;
                }
                s1 += ga;
                s2 += ga * n;
            }
        }
    }
    block_reduce2(s1, s2, red);
    if (threadIdx.x == 0) fused_publish(slots + plane * chunks + chunk, s1, s2);
    double d1, d2;
    fused_gather(slots, depart, plane, chunks, d1, d2, redd);
    const float t1 = (float)d1, t2 = (float)d2;
    if (chunk == 0 && threadIdx.x == 0) {
        sums[plane] = t1;
        sums[planes + plane] = t2;
    }
    const float m1 = t1 * inv_total, m2 = t2 * inv_total;
    const float k = rstd * g;
#pragma unroll
    for (int s = 0; s < FUSED_SLOTS; ++s) {
        const long long e = c0 + ((long long)s * NT + threadIdx.x) * VEC;
        if (e < c1) {
            float v[VEC], d[VEC], o[VEC];
            VecIO<T>::unpack(rx[s], v);
            VecIO<T>::unpack(rg[s], d);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float n = (pre<T>(v[i], pb, has_pb) - mean) * rstd;
                const float ga = GELU ? gak[s * VEC + i] : d[i];
                o[i] = k * (ga - (m1 + n * m2));
            }
            VecIO<T>::store(op + e, o);
        }
    }
}

// ---- bias + GELU -------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NT) void bias_gelu_fwd(const T* __restrict__ x, const float* __restrict__ bias,
                                                    T* __restrict__ y, int channels, long long hw, int chunks) {
    const long long plane = blockIdx.x / chunks;
    const int chunk = blockIdx.x % chunks;
    const float b = bias ? bias[plane % channels] : 0.f;
    const T* xp = x + plane * hw;
    T* yp = y + plane * hw;
    for_chunk<T, 1>(hw, chunks, chunk, xp, xp, [&](long long e, auto cnt, const float* v, const float*) {
        float o[cnt()];
#pragma unroll
        for (int i = 0; i < cnt(); ++i) o[i] = gelu_f(v[i] + b);
        store_n(yp + e, o, cnt);
    });
}

// gx = gy * gelu'(x + b); per-chunk partial sum of gx -> ws (for the bias gradient)
template <typename T>
__global__ __launch_bounds__(NT) void bias_gelu_bwd(const T* __restrict__ x, const float* __restrict__ bias,
                                                    const T* __restrict__ gy, T* __restrict__ gx,
                                                    float* __restrict__ ws, int channels, long long hw, int chunks) {
    __shared__ float red[2 * NT / 64];
    const long long plane = blockIdx.x / chunks;
    const int chunk = blockIdx.x % chunks;
    const float b = bias ? bias[plane % channels] : 0.f;
    const T* xp = x + plane * hw;
    const T* gp = gy + plane * hw;
    T* op = gx + plane * hw;
    float s1 = 0.f, s2 = 0.f;
    for_chunk<T, 2>(hw, chunks, chunk, xp, gp, [&](long long e, auto cnt, const float* v, const float* d) {
        float o[cnt()];
#pragma unroll
        for (int i = 0; i < cnt(); ++i) {
            o[i] = d[i] * gelu_grad_f(v[i] + b);
            s1 += o[i];
        }
        store_n(op + e, o, cnt);
    });
    if (ws) {
        block_reduce2(s1, s2, red);
        if (threadIdx.x == 0) {
            ws[2 * (long long)blockIdx.x] = s1;
            ws[2 * (long long)blockIdx.x + 1] = 0.f;
        }
    }
}


// ---- quadrature-weighted L^p plane sums (geometric losses) -----------------------------------
//   mode 0:  sum_i q[i] * a[i] (* w[i])                    (GridQuadrature.forward, makani/utils/grids.py:185-191)
//   mode 1:  sum_i q[i] * |a[i] - b[i]|^p (* w[i])         (GeometricLpLoss, makani/utils/losses/lp_loss.py:61-75)
// a and b may have different dtypes (prediction bf16 / target f32); 4 elements per lane and step.
constexpr int LP_UNROLL = 8;
constexpr long long LP_CHUNK = (long long)NT * 4 * LP_UNROLL;

__device__ __forceinline__ void load4(const float* p, float* v) {
    const f32x4 r = *reinterpret_cast<const f32x4*>(p);
    v[0] = r[0], v[1] = r[1], v[2] = r[2], v[3] = r[3];
}
__device__ __forceinline__ void load4(const u16* p, float* v) {
    const uint2 r = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(r.x << 16), v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16), v[3] = __uint_as_float(r.y & 0xffff0000u);
}
__device__ __forceinline__ void store4(float* p, const float* v) {
    f32x4 r = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = r;
}
__device__ __forceinline__ void store4(u16* p, const float* v) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v[0], v[1]),
                                              pack_bf16x2(v[2], v[3]));
}
__device__ __forceinline__ float lp_pow(float d, float p) {
    const float ad = fabsf(d);
    return p == 2.f ? d * d : (p == 1.f ? ad : powf(ad, p));
}
__device__ __forceinline__ float lp_pow_grad(float d, float p) {      // d/dd |d|^p  (0 at d = 0 like torch.abs)
    if (p == 2.f) return 2.f * d;
    const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    return p == 1.f ? sg : (d == 0.f ? 0.f : p * powf(fabsf(d), p - 1.f) * sg);
}

template <typename TA, typename TB, typename F>
__device__ __forceinline__ void lp_for_chunk(long long hw, int chunk, F&& f) {
    const long long c0 = (long long)chunk * LP_CHUNK;
    const long long c1 = min(hw, c0 + LP_CHUNK);
    if ((hw & 3) == 0) {
#pragma unroll
        for (int u = 0; u < LP_UNROLL; ++u) {
            const long long e = c0 + ((long long)u * NT + threadIdx.x) * 4;
            if (e < c1) f(e, true);
        }
    } else {
        for (long long e = c0 + threadIdx.x; e < c1; e += NT) f(e, false);
    }
}

template <typename TA, typename TB>
__global__ __launch_bounds__(NT) void quad_lp_fwd(const TA* __restrict__ a, const TB* __restrict__ b,
                                                  const float* __restrict__ wgt, const float* __restrict__ q,
                                                  float* __restrict__ ws, long long hw, int chunks, int mode, float p) {
    __shared__ float red[2 * NT / 64];
    const long long plane = blockIdx.x / chunks;
    const int chunk = blockIdx.x % chunks;
    const TA* ap = a + plane * hw;
    const TB* bp = b ? b + plane * hw : nullptr;
    const float* wp = wgt ? wgt + plane * hw : nullptr;
    float s1 = 0.f, s2 = 0.f;
    lp_for_chunk<TA, TB>(hw, chunk, [&](long long e, bool vec) {
        float va[4], vb[4] = {0.f, 0.f, 0.f, 0.f}, vq[4], vw[4] = {1.f, 1.f, 1.f, 1.f};
        const int cnt = vec ? 4 : 1;
        if (vec) {
            load4(ap + e, va);
            load4(q + e, vq);
            if (bp) load4(bp + e, vb);
            if (wp) load4(wp + e, vw);
        } else {
            va[0] = VecIO<TA>::load1(ap + e);
            vq[0] = q[e];
            if (bp) vb[0] = VecIO<TB>::load1(bp + e);
            if (wp) vw[0] = wp[e];
        }
        for (int i = 0; i < cnt; ++i) {
            const float t = mode ? lp_pow(va[i] - vb[i], p) : va[i];
            s1 += vq[i] * (t * vw[i]);
        }
    });
    block_reduce2(s1, s2, red);
    if (threadIdx.x == 0) {
        ws[2 * (long long)blockIdx.x] = s1;
        ws[2 * (long long)blockIdx.x + 1] = 0.f;
    }
}

// da[i] = g[plane] * q[i] * w[i] * (mode ? d|d|^p/dd : 1),  db = -da
template <typename TA, typename TB>
__global__ __launch_bounds__(NT) void quad_lp_bwd(const TA* __restrict__ a, const TB* __restrict__ b,
                                                  const float* __restrict__ wgt, const float* __restrict__ q,
                                                  const float* __restrict__ g, TA* __restrict__ da, TB* __restrict__ db,
                                                  long long hw, int chunks, int mode, float p) {
    const long long plane = blockIdx.x / chunks;
    const int chunk = blockIdx.x % chunks;
    const TA* ap = a + plane * hw;
    const TB* bp = b ? b + plane * hw : nullptr;
    const float* wp = wgt ? wgt + plane * hw : nullptr;
    TA* dap = da ? da + plane * hw : nullptr;
    TB* dbp = db ? db + plane * hw : nullptr;
    const float gp = g[plane];
    lp_for_chunk<TA, TB>(hw, chunk, [&](long long e, bool vec) {
        float va[4] = {0.f, 0.f, 0.f, 0.f}, vb[4] = {0.f, 0.f, 0.f, 0.f}, vq[4], vw[4] = {1.f, 1.f, 1.f, 1.f}, r[4], rn[4];
        const int cnt = vec ? 4 : 1;
        if (vec) {
            if (mode) load4(ap + e, va);
            load4(q + e, vq);
            if (bp) load4(bp + e, vb);
            if (wp) load4(wp + e, vw);
        } else {
            if (mode) va[0] = VecIO<TA>::load1(ap + e);
            vq[0] = q[e];
            if (bp) vb[0] = VecIO<TB>::load1(bp + e);
            if (wp) vw[0] = wp[e];
        }
        for (int i = 0; i < cnt; ++i) {
            r[i] = gp * vq[i] * vw[i] * (mode ? lp_pow_grad(va[i] - vb[i], p) : 1.f);
            rn[i] = -r[i];
        }
        if (vec) {
            if (dap) store4(dap + e, r);
            if (dbp) store4(dbp + e, rn);
        } else {
            if (dap) VecIO<TA>::store1(dap + e, r[0]);
            if (dbp) VecIO<TB>::store1(dbp + e, rn[0]);
        }
    });
}

// chunks per plane: as many as keep every block resident (8 blocks of 256 lanes per CU, 256 CUs), never less than one
// pass of work per block
template <typename T>
int chunks_for(long long hw, long long planes) {
    constexpr long long target = 2048;
    const long long maxc = (hw + pass_elems<T>() - 1) / pass_elems<T>();
    const long long want = std::max(1ll, (target + planes / 2) / planes);
    return (int)std::min(maxc, want);
}

// chunks per plane of the fused kernels: the two-kernel choice (chunks_for) when its chunks fit the registers, else as few as fit
template <typename T>
int fused_chunks_for(long long hw, long long planes, int kind) {
    const int c = chunks_for<T>(hw, planes);
    if (chunk_len(hw, c, VecIO<T>::N) <= fused_chunk_cap<T>(kind)) return c;
    return (int)((hw + fused_chunk_cap<T>(kind) - 1) / fused_chunk_cap<T>(kind));
}

int check_common(const void* a, long long planes, long long hw, int dtype, const char* what) {
    MK_REQUIRE(a != nullptr, "%s: null pointer", what);
    MK_REQUIRE(planes > 0 && hw > 0, "%s: bad shape planes=%lld hw=%lld", what, planes, hw);
    MK_REQUIRE(dtype == MK_F32 || dtype == MK_BF16, "%s: bad dtype %d", what, dtype);
    MK_REQUIRE(planes * (long long)((hw + 1023) / 1024) < (1ll << 31), "%s: grid too large", what);
    return 0;
}

}  // namespace

extern "C" int mk_pointwise_chunks(long long hw, int dtype, long long planes) {
    if (hw <= 0 || planes <= 0) return 1;
    return dtype == MK_BF16 ? chunks_for<u16>(hw, planes) : chunks_for<float>(hw, planes);
}

#define DISPATCH_DTYPE(dtype, CALL_F32, CALL_BF16) \
    do {                                           \
        if ((dtype) == MK_F32) {                   \
            CALL_F32;                              \
        } else {                                   \
            CALL_BF16;                             \
        }                                          \
    } while (0)

extern "C" int mk_instnorm_stats(const void* x, int dtype, float* stats, float* ws, long long planes, long long hw,
                                 float eps, const float* quad, float quad_sum, void* stream) {
    int rc = check_common(x, planes, hw, dtype, "instnorm_stats");
    if (rc) return rc;
    MK_REQUIRE(stats && ws, "instnorm_stats: null stats/ws");
    MK_REQUIRE(quad == nullptr || quad_sum > 0.f, "instnorm_stats: quadrature weights need their (positive) sum");
    hipStream_t s = (hipStream_t)stream;
    const int fb = (int)((planes + 255) / 256);
    const float qs = quad ? quad_sum : 0.f;
    if (dtype == MK_F32) {
        const int ch = chunks_for<float>(hw, planes);
        hipLaunchKernelGGL(in_stats_partial<float>, dim3((unsigned)(planes * ch)), dim3(NT), 0, s, (const float*)x, ws, hw, ch, nullptr, 1, quad);
        hipLaunchKernelGGL(in_stats_final<float>, dim3(fb), dim3(256), 0, s, (const float*)x, ws, stats, planes, hw, ch, eps, qs);
    } else {
        const int ch = chunks_for<u16>(hw, planes);
        hipLaunchKernelGGL(in_stats_partial<u16>, dim3((unsigned)(planes * ch)), dim3(NT), 0, s, (const u16*)x, ws, hw, ch, nullptr, 1, quad);
        hipLaunchKernelGGL(in_stats_final<u16>, dim3(fb), dim3(256), 0, s, (const u16*)x, ws, stats, planes, hw, ch, eps, qs);
    }
    return mk_check_launch("mk_instnorm_stats");
}

static int instnorm_apply_impl(const void* x, void* y, int dtype, float* stats, const float* gamma, const float* beta,
                               const float* ws, float eps, long long planes, int channels, long long hw, int fuse_gelu,
                               const float* pre_bias, float qsum, hipStream_t s);

// merged (mean, rstd) of planes sharded over P ranks from the ranks' local (mean, rstd) and pixel counts: the pairwise update
// of makani/mpu/layer_norm.py:31-81 (Chan et al.) carried in fp64, one thread per plane
__global__ void in_merge_stats(const float* __restrict__ all, const float* __restrict__ counts, float* __restrict__ out,
                               long long planes, int P, float eps) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= planes) return;
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (int r = 0; r < P; ++r) {
        const double c = (double)counts[r];
        if (!(c > 0.0)) continue;
        const double mu = (double)all[((long long)r * planes + p) * 2];
        const double rs = (double)all[((long long)r * planes + p) * 2 + 1];
        double var = 1.0 / (rs * rs) - (double)eps;
        if (var < 0.0) var = 0.0;
        const double nn = n + c, delta = mu - mean;
        mean += delta * (c / nn);
        m2 += var * c + delta * delta * (n * c / nn);
        n = nn;
    }
    out[2 * p] = (float)mean;
    out[2 * p + 1] = (float)(1.0 / sqrt((n > 0.0 ? m2 / n : 0.0) + (double)eps));
}

extern "C" int mk_instnorm_merge(const float* all_stats, const float* counts, float* stats, long long planes, int nranks,
                                 float eps, void* stream) {
    MK_REQUIRE(all_stats && counts && stats && planes > 0 && nranks > 0, "instnorm_merge: bad args");
    hipLaunchKernelGGL(in_merge_stats, dim3((unsigned)((planes + 255) / 256)), dim3(256), 0, (hipStream_t)stream, all_stats, counts,
                       stats, planes, nranks, eps);
    return mk_check_launch("mk_instnorm_merge");
}

extern "C" int mk_instnorm_apply(const void* x, void* y, int dtype, const float* stats, const float* gamma,
                                 const float* beta, long long planes, int channels, long long hw, int fuse_gelu,
                                 void* stream) {
    int rc = check_common(x, planes, hw, dtype, "instnorm_apply");
    if (rc) return rc;
    MK_REQUIRE(y && stats && channels > 0, "instnorm_apply: bad args");
    return instnorm_apply_impl(x, y, dtype, const_cast<float*>(stats), gamma, beta, nullptr, 0.f, planes, channels, hw,
                               fuse_gelu, nullptr, 0.f, (hipStream_t)stream);
}

extern "C" int mk_instnorm_fwd(const void* x, void* y, int dtype, float* stats, float* ws, const float* gamma,
                               const float* beta, const float* pre_bias, const float* quad, float quad_sum,
                               long long planes, int channels, long long hw, float eps, int fuse_gelu, void* stream) {
    int rc = check_common(x, planes, hw, dtype, "instnorm_fwd");
    if (rc) return rc;
    MK_REQUIRE(y && stats && ws && channels > 0, "instnorm_fwd: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MK_F32)
        hipLaunchKernelGGL(in_stats_partial<float>, dim3((unsigned)(planes * chunks_for<float>(hw, planes))), dim3(NT), 0, s, (const float*)x, ws, hw, chunks_for<float>(hw, planes), pre_bias, channels, quad);
    else
        hipLaunchKernelGGL(in_stats_partial<u16>, dim3((unsigned)(planes * chunks_for<u16>(hw, planes))), dim3(NT), 0, s, (const u16*)x, ws, hw, chunks_for<u16>(hw, planes), pre_bias, channels, quad);
    MK_REQUIRE(quad == nullptr || quad_sum > 0.f, "instnorm_fwd: quadrature weights need their (positive) sum");
    return instnorm_apply_impl(x, y, dtype, stats, gamma, beta, ws, eps, planes, channels, hw, fuse_gelu, pre_bias,
                               quad ? quad_sum : 0.f, s);
}

static int instnorm_apply_impl(const void* x, void* y, int dtype, float* stats, const float* gamma, const float* beta,
                               const float* ws, float eps, long long planes, int channels, long long hw, int fuse_gelu,
                               const float* pre_bias, float qsum, hipStream_t s) {
    if (dtype == MK_F32) {
        const int ch = chunks_for<float>(hw, planes);
        dim3 g((unsigned)(planes * ch));
        if (fuse_gelu)
            hipLaunchKernelGGL((in_apply<float, true>), g, dim3(NT), 0, s, (const float*)x, (float*)y, stats, gamma, beta, ws, eps, channels, hw, ch, pre_bias, qsum);
        else
            hipLaunchKernelGGL((in_apply<float, false>), g, dim3(NT), 0, s, (const float*)x, (float*)y, stats, gamma, beta, ws, eps, channels, hw, ch, pre_bias, qsum);
    } else {
        const int ch = chunks_for<u16>(hw, planes);
        dim3 g((unsigned)(planes * ch));
        if (fuse_gelu)
            hipLaunchKernelGGL((in_apply<u16, true>), g, dim3(NT), 0, s, (const u16*)x, (u16*)y, stats, gamma, beta, ws, eps, channels, hw, ch, pre_bias, qsum);
        else
            hipLaunchKernelGGL((in_apply<u16, false>), g, dim3(NT), 0, s, (const u16*)x, (u16*)y, stats, gamma, beta, ws, eps, channels, hw, ch, pre_bias, qsum);
    }
    return mk_check_launch("mk_instnorm_apply");
}

extern "C" int mk_instnorm_bwd(const void* x, const void* gy, void* gx, int dtype, const float* stats,
                               const float* gamma, const float* beta, const float* pre_bias, const float* quad,
                               float quad_sum, float* sums, float* ws, long long planes, int channels, long long hw,
                               long long hw_total, int phase, int fuse_gelu, void* stream) {
    int rc = check_common(x, planes, hw, dtype, "instnorm_bwd");
    if (rc) return rc;
    MK_REQUIRE(gy && gx && stats && sums && ws && channels > 0, "instnorm_bwd: bad args");
    MK_REQUIRE(phase >= 0 && phase <= 2 && hw_total >= hw, "instnorm_bwd: bad phase / hw_total");
    hipStream_t s = (hipStream_t)stream;
    const int fb = (int)((planes + 255) / 256);
    const float inv_total = 1.0f / (float)hw_total;
#define IN_BWD(T, G)                                                                                                   \
    do {                                                                                                               \
        const int ch = chunks_for<T>(hw, planes);                                                                              \
        dim3 g((unsigned)(planes * ch));                                                                               \
        if (phase != 2)                                                                                                \
            hipLaunchKernelGGL((in_bwd_partial<T, G>), g, dim3(NT), 0, s, (const T*)x, (const T*)gy, stats, gamma,    \
                               beta, ws, channels, hw, ch, pre_bias);                                                  \
        if (phase == 1)                                                                                                \
            hipLaunchKernelGGL(sum_chunks_final, dim3(fb), dim3(256), 0, s, ws, sums, planes, ch);                     \
        if (phase != 1)   /* phase 0: the apply kernel finishes the reduction itself and publishes `sums` */           \
            hipLaunchKernelGGL((in_bwd_apply<T, G>), g, dim3(NT), 0, s, (const T*)x, (const T*)gy, (T*)gx, stats,     \
                               gamma, beta, sums, phase == 0 ? ws : nullptr, channels, hw, ch, inv_total, pre_bias,    \
                               quad, quad_sum);                                                                        \
    } while (0)
    if (dtype == MK_F32) {
        if (fuse_gelu) IN_BWD(float, true); else IN_BWD(float, false);
    } else {
        if (fuse_gelu) IN_BWD(u16, true); else IN_BWD(u16, false);
    }
#undef IN_BWD
    return mk_check_launch("mk_instnorm_bwd");
}

// ---- one-pass instance norm (in_fwd_fused / in_bwd_fused) ----
// mk_instnorm_fused_chunks: chunks per plane of the one-pass kernels of `kind` (0 forward, 1 backward, 2 backward through the
// fused GELU), 0 if the plane shape is not served (hw not a multiple of
// the 16-byte vector, or more chunks than a block has threads).  `slots`: planes * chunks 64-bit words, ALL ONES before the
// first launch; `depart`: planes 32-bit words, ZERO before the first launch.  The kernels leave both as they found them, so one
// buffer pair serves every later launch on the same stream (not two launches that may run concurrently).
extern "C" int mk_instnorm_fused_chunks(long long hw, int dtype, long long planes, int kind) {
    if (hw <= 0 || planes <= 0 || kind < 0 || kind > 2) return 0;
    const int vec = dtype == MK_BF16 ? VecIO<u16>::N : VecIO<float>::N;
    if (hw % vec) return 0;
    const int c = dtype == MK_BF16 ? fused_chunks_for<u16>(hw, planes, kind) : fused_chunks_for<float>(hw, planes, kind);
    if (c > NT || planes * (long long)c >= (1ll << 31)) return 0;
    return c;
}

extern "C" int mk_instnorm_fwd_fused(const void* x, void* y, int dtype, float* stats, const float* gamma, const float* beta,
                                     const float* pre_bias, void* slots, void* depart, long long planes, int channels, long long hw,
                                     float eps, int fuse_gelu, void* stream) {
    int rc = check_common(x, planes, hw, dtype, "instnorm_fwd_fused");
    if (rc) return rc;
    const int ch = mk_instnorm_fused_chunks(hw, dtype, planes, 0);
    MK_REQUIRE(ch > 0, "instnorm_fwd_fused: plane shape not served (hw = %lld)", hw);
    MK_REQUIRE(y && stats && slots && depart && channels > 0, "instnorm_fwd_fused: bad args");
    hipStream_t s = (hipStream_t)stream;
    const dim3 g((unsigned)(planes * ch));
    unsigned long long* sl = (unsigned long long*)slots;
    unsigned* dp = (unsigned*)depart;
#define IN_FF(T, G) hipLaunchKernelGGL((in_fwd_fused<T, G>), g, dim3(NT), 0, s, (const T*)x, (T*)y, stats, gamma, beta, sl, dp, eps, channels, hw, ch, pre_bias)
    if (dtype == MK_F32) {
        if (fuse_gelu) IN_FF(float, true); else IN_FF(float, false);
    } else {
        if (fuse_gelu) IN_FF(u16, true); else IN_FF(u16, false);
    }
#undef IN_FF
    return mk_check_launch("mk_instnorm_fwd_fused");
}

extern "C" int mk_instnorm_bwd_fused(const void* x, const void* gy, void* gx, int dtype, const float* stats, const float* gamma,
                                     const float* beta, const float* pre_bias, float* sums, void* slots, void* depart, long long planes,
                                     int channels, long long hw, int fuse_gelu, void* stream) {
    int rc = check_common(x, planes, hw, dtype, "instnorm_bwd_fused");
    if (rc) return rc;
    const int ch = mk_instnorm_fused_chunks(hw, dtype, planes, fuse_gelu ? 2 : 1);
    MK_REQUIRE(ch > 0, "instnorm_bwd_fused: plane shape not served (hw = %lld)", hw);
    MK_REQUIRE(gy && gx && stats && sums && slots && depart && channels > 0, "instnorm_bwd_fused: bad args");
    hipStream_t s = (hipStream_t)stream;
    const dim3 g((unsigned)(planes * ch));
    const float inv_total = 1.0f / (float)hw;
    unsigned long long* sl = (unsigned long long*)slots;
    unsigned* dp = (unsigned*)depart;
#define IN_BF(T, G) hipLaunchKernelGGL((in_bwd_fused<T, G>), g, dim3(NT), 0, s, (const T*)x, (const T*)gy, (T*)gx, stats, gamma, beta, sums, sl, dp, channels, hw, ch, inv_total, pre_bias)
    if (dtype == MK_F32) {
        if (fuse_gelu) IN_BF(float, true); else IN_BF(float, false);
    } else {
        if (fuse_gelu) IN_BF(u16, true); else IN_BF(u16, false);
    }
#undef IN_BF
    return mk_check_launch("mk_instnorm_bwd_fused");
}

extern "C" int mk_plane_sums(const void* x, int dtype, float* sums, float* ws, long long planes, long long hw, void* stream) {
    int rc = check_common(x, planes, hw, dtype, "plane_sums");
    if (rc) return rc;
    MK_REQUIRE(sums && ws, "plane_sums: null sums/ws");
    hipStream_t s = (hipStream_t)stream;
    const int fb = (int)((planes + 255) / 256);
    if (dtype == MK_F32) {
        const int ch = chunks_for<float>(hw, planes);
        hipLaunchKernelGGL(plane_sum_partial<float>, dim3((unsigned)(planes * ch)), dim3(NT), 0, s, (const float*)x, ws, hw, ch);
        hipLaunchKernelGGL(sum_chunks_final, dim3(fb), dim3(256), 0, s, ws, sums, planes, ch);
    } else {
        const int ch = chunks_for<u16>(hw, planes);
        hipLaunchKernelGGL(plane_sum_partial<u16>, dim3((unsigned)(planes * ch)), dim3(NT), 0, s, (const u16*)x, ws, hw, ch);
        hipLaunchKernelGGL(sum_chunks_final, dim3(fb), dim3(256), 0, s, ws, sums, planes, ch);
    }
    return mk_check_launch("mk_plane_sums");
}

extern "C" int mk_bias_gelu_fwd(const void* x, const float* bias, void* y, int dtype, long long planes, int channels,
                                long long hw, void* stream) {
    int rc = check_common(x, planes, hw, dtype, "bias_gelu_fwd");
    if (rc) return rc;
    MK_REQUIRE(y && channels > 0, "bias_gelu_fwd: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MK_F32) {
        const int ch = chunks_for<float>(hw, planes);
        hipLaunchKernelGGL(bias_gelu_fwd<float>, dim3((unsigned)(planes * ch)), dim3(NT), 0, s, (const float*)x, bias, (float*)y, channels, hw, ch);
    } else {
        const int ch = chunks_for<u16>(hw, planes);
        hipLaunchKernelGGL(bias_gelu_fwd<u16>, dim3((unsigned)(planes * ch)), dim3(NT), 0, s, (const u16*)x, bias, (u16*)y, channels, hw, ch);
    }
    return mk_check_launch("mk_bias_gelu_fwd");
}

extern "C" int mk_bias_gelu_bwd(const void* x, const float* bias, const void* gy, void* gx, float* sums, float* ws,
                                int dtype, long long planes, int channels, long long hw, void* stream) {
    int rc = check_common(x, planes, hw, dtype, "bias_gelu_bwd");
    if (rc) return rc;
    MK_REQUIRE(gy && gx && channels > 0, "bias_gelu_bwd: bad args");
    MK_REQUIRE((sums == nullptr) == (ws == nullptr), "bias_gelu_bwd: sums and ws go together");
    hipStream_t s = (hipStream_t)stream;
    const int fb = (int)((planes + 255) / 256);
    if (dtype == MK_F32) {
        const int ch = chunks_for<float>(hw, planes);
        hipLaunchKernelGGL(bias_gelu_bwd<float>, dim3((unsigned)(planes * ch)), dim3(NT), 0, s, (const float*)x, bias, (const float*)gy, (float*)gx, ws, channels, hw, ch);
        if (sums) hipLaunchKernelGGL(sum_chunks_final, dim3(fb), dim3(256), 0, s, ws, sums, planes, ch);
    } else {
        const int ch = chunks_for<u16>(hw, planes);
        hipLaunchKernelGGL(bias_gelu_bwd<u16>, dim3((unsigned)(planes * ch)), dim3(NT), 0, s, (const u16*)x, bias, (const u16*)gy, (u16*)gx, ws, channels, hw, ch);
        if (sums) hipLaunchKernelGGL(sum_chunks_final, dim3(fb), dim3(256), 0, s, ws, sums, planes, ch);
    }
    return mk_check_launch("mk_bias_gelu_bwd");
}

extern "C" int mk_quad_lp_chunks(long long hw) { return (int)((hw + LP_CHUNK - 1) / LP_CHUNK); }

#define MK_LP_DISPATCH(KERN, ...)                                                                         \
    do {                                                                                                  \
        if (a_dtype == MK_F32 && b_dtype == MK_F32)                                                       \
            hipLaunchKernelGGL((KERN<float, float>), grid, dim3(NT), 0, s, (const float*)a, (const float*)b, __VA_ARGS__); \
        else if (a_dtype == MK_F32)                                                                       \
            hipLaunchKernelGGL((KERN<float, u16>), grid, dim3(NT), 0, s, (const float*)a, (const u16*)b, __VA_ARGS__);     \
        else if (b_dtype == MK_F32)                                                                       \
            hipLaunchKernelGGL((KERN<u16, float>), grid, dim3(NT), 0, s, (const u16*)a, (const float*)b, __VA_ARGS__);     \
        else                                                                                              \
            hipLaunchKernelGGL((KERN<u16, u16>), grid, dim3(NT), 0, s, (const u16*)a, (const u16*)b, __VA_ARGS__);         \
    } while (0)

extern "C" int mk_quad_lp_fwd(const void* a, int a_dtype, const void* b, int b_dtype, const float* wgt, const float* q,
                              float* sums, float* ws, long long planes, long long hw, int mode, float p, void* stream) {
    int rc = check_common(a, planes, hw, a_dtype, "quad_lp_fwd");
    if (rc) return rc;
    MK_REQUIRE(q && sums && ws, "quad_lp_fwd: null pointer");
    MK_REQUIRE(b_dtype == MK_F32 || b_dtype == MK_BF16, "quad_lp_fwd: bad dtype %d", b_dtype);
    MK_REQUIRE(mode == 0 || mode == 1, "quad_lp_fwd: mode %d", mode);
    MK_REQUIRE(mode == 0 || p > 0.f, "quad_lp_fwd: p = %f must be positive", p);
    if (mode == 0) b = nullptr;
    hipStream_t s = (hipStream_t)stream;
    const int ch = mk_quad_lp_chunks(hw);
    const dim3 grid((unsigned)(planes * ch));
    MK_LP_DISPATCH(quad_lp_fwd, wgt, q, ws, hw, ch, mode, p);
    hipLaunchKernelGGL(sum_chunks_final, dim3((unsigned)((planes + 255) / 256)), dim3(256), 0, s, ws, sums, planes, ch);
    return mk_check_launch("mk_quad_lp_fwd");
}

extern "C" int mk_quad_lp_bwd(const void* a, int a_dtype, const void* b, int b_dtype, const float* wgt, const float* q,
                              const float* g, void* da, void* db, long long planes, long long hw, int mode, float p,
                              void* stream) {
    int rc = check_common(a, planes, hw, a_dtype, "quad_lp_bwd");
    if (rc) return rc;
    MK_REQUIRE(q && g && (da || db), "quad_lp_bwd: null pointer");
    MK_REQUIRE(b_dtype == MK_F32 || b_dtype == MK_BF16, "quad_lp_bwd: bad dtype %d", b_dtype);
    MK_REQUIRE(mode == 0 || mode == 1, "quad_lp_bwd: mode %d", mode);
    MK_REQUIRE(db == nullptr || b != nullptr, "quad_lp_bwd: db without b");
    hipStream_t s = (hipStream_t)stream;
    const int ch = mk_quad_lp_chunks(hw);
    const dim3 grid((unsigned)(planes * ch));
    if (a_dtype == MK_F32 && b_dtype == MK_F32)
        hipLaunchKernelGGL((quad_lp_bwd<float, float>), grid, dim3(NT), 0, s, (const float*)a, (const float*)b, wgt, q, g, (float*)da, (float*)db, hw, ch, mode, p);
    else if (a_dtype == MK_F32)
        hipLaunchKernelGGL((quad_lp_bwd<float, u16>), grid, dim3(NT), 0, s, (const float*)a, (const u16*)b, wgt, q, g, (float*)da, (u16*)db, hw, ch, mode, p);
    else if (b_dtype == MK_F32)
        hipLaunchKernelGGL((quad_lp_bwd<u16, float>), grid, dim3(NT), 0, s, (const u16*)a, (const float*)b, wgt, q, g, (u16*)da, (float*)db, hw, ch, mode, p);
    else
        hipLaunchKernelGGL((quad_lp_bwd<u16, u16>), grid, dim3(NT), 0, s, (const u16*)a, (const u16*)b, wgt, q, g, (u16*)da, (u16*)db, hw, ch, mode, p);
    return mk_check_launch("mk_quad_lp_bwd");
}
