// HBM-streaming pointwise kernels on NCHW planes (gfx950):
//   instance norm (stats / apply / backward) with optional fused exact-erf GELU,
//   bias + GELU forward / backward.
// All kernels move 16 B per lane (4 x f32 or 8 x bf16), compute in fp32, and are laid out as
// (plane, chunk) work items: one block streams CHUNK contiguous elements of one (batch, channel)
// plane, so per-plane scalars (mean, rstd, gamma, beta, bias) are block-uniform.
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int UNROLL = 4;

template <typename T>
struct VecIO;
template <>
struct VecIO<float> {
    static constexpr int N = 4;
    __device__ static __forceinline__ void load(const float* p, float* v) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = r[i];
    }
    __device__ static __forceinline__ void store(float* p, const float* v) {
        f32x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = v[i];
        *reinterpret_cast<f32x4*>(p) = r;
    }
    __device__ static __forceinline__ float load1(const float* p) { return *p; }
    __device__ static __forceinline__ void store1(float* p, float v) { *p = v; }
};
template <>
struct VecIO<u16> {
    static constexpr int N = 8;
    __device__ static __forceinline__ void load(const u16* p, float* v) {
        const uint4 r = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    __device__ static __forceinline__ void store(u16* p, const float* v) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = (uint32_t)f32_to_bf16(v[2 * i]) | ((uint32_t)f32_to_bf16(v[2 * i + 1]) << 16);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __device__ static __forceinline__ float load1(const u16* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void store1(u16* p, float v) { *p = f32_to_bf16(v); }
};

template <typename T>
__host__ __device__ constexpr long long chunk_elems() {
    return (long long)NT * VecIO<T>::N * UNROLL;
}

// Visit every element of this block's chunk.  f(value_index_in_plane, float* vals, n) is called
// with up to VEC values; vector path needs 16-byte aligned plane bases (hw % VEC == 0).
template <typename T, typename F>
__device__ __forceinline__ void for_chunk(long long hw, int chunk, F&& f) {
    constexpr int VEC = VecIO<T>::N;
    const long long c0 = (long long)chunk * chunk_elems<T>();
    const long long c1 = min(hw, c0 + chunk_elems<T>());
    if ((hw % VEC) == 0) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long long e = c0 + ((long long)u * NT + threadIdx.x) * VEC;
            if (e < c1) f(e, VEC, true);
        }
    } else {
        for (long long e = c0 + threadIdx.x; e < c1; e += NT) f(e, 1, false);
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

// ---- instance norm: statistics ---------------------------------------------------
// partial sums relative to a per-plane pivot (first element) to avoid cancellation in fp32
template <typename T>
__global__ __launch_bounds__(NT) void in_stats_partial(const T* __restrict__ x, float* __restrict__ ws, long long hw,
                                                       int chunks) {
    __shared__ float red[2 * NT / 64];
    const long long plane = blockIdx.x / chunks;
    const int chunk = blockIdx.x % chunks;
    const T* xp = x + plane * hw;
    const float pivot = VecIO<T>::load1(xp);
    float s1 = 0.f, s2 = 0.f;
    for_chunk<T>(hw, chunk, [&](long long e, int n, bool vec) {
        float v[VecIO<T>::N];
        if (vec)
            VecIO<T>::load(xp + e, v);
        else
            v[0] = VecIO<T>::load1(xp + e);
        for (int i = 0; i < (vec ? VecIO<T>::N : 1); ++i) {
            const float d = v[i] - pivot;
            s1 += d;
            s2 += d * d;
        }
    });
    block_reduce2(s1, s2, red);
    if (threadIdx.x == 0) {
        ws[2 * (long long)blockIdx.x] = s1;
        ws[2 * (long long)blockIdx.x + 1] = s2;
    }
}

template <typename T>
__global__ void in_stats_final(const T* __restrict__ x, const float* __restrict__ ws, float* __restrict__ stats,
                               long long planes, long long hw, int chunks, float eps) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= planes) return;
    const double pivot = (double)VecIO<T>::load1(x + p * hw);
    double s1 = 0.0, s2 = 0.0;
    for (int c = 0; c < chunks; ++c) {
        s1 += (double)ws[2 * (p * chunks + c)];
        s2 += (double)ws[2 * (p * chunks + c) + 1];
    }
    const double m = s1 / (double)hw;
    double var = s2 / (double)hw - m * m;   // biased variance, as nn.InstanceNorm2d
    if (var < 0.0) var = 0.0;
    stats[2 * p] = (float)(pivot + m);
    stats[2 * p + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// ---- instance norm: apply (+ GELU) --------------------------------------------------
template <typename T, bool GELU>
__global__ __launch_bounds__(NT) void in_apply(const T* __restrict__ x, T* __restrict__ y,
                                               const float* __restrict__ stats, const float* __restrict__ gamma,
                                               const float* __restrict__ beta, int channels, long long hw, int chunks) {
    const long long plane = blockIdx.x / chunks;
    const int chunk = blockIdx.x % chunks;
    const int c = (int)(plane % channels);
    const float mean = stats[2 * plane], rstd = stats[2 * plane + 1];
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = rstd * g, sh = b - mean * rstd * g;
    const T* xp = x + plane * hw;
    T* yp = y + plane * hw;
    for_chunk<T>(hw, chunk, [&](long long e, int n, bool vec) {
        float v[VecIO<T>::N];
        if (vec) {
            VecIO<T>::load(xp + e, v);
#pragma unroll
            for (int i = 0; i < VecIO<T>::N; ++i) {
                const float a = v[i] * sc + sh;
                v[i] = GELU ? gelu_f(a) : a;
            }
            VecIO<T>::store(yp + e, v);
        } else {
            const float a = VecIO<T>::load1(xp + e) * sc + sh;
            VecIO<T>::store1(yp + e, GELU ? gelu_f(a) : a);
        }
    });
}

// ---- instance norm backward ------------------------------------------------------------
// n = (x - mean) rstd ; a = n gamma + beta ; y = GELU ? gelu(a) : a ; ga = gy * (GELU ? gelu'(a) : 1)
// pass 1: S1 = sum ga, S2 = sum ga*n  (per plane)
template <typename T, bool GELU>
__global__ __launch_bounds__(NT) void in_bwd_partial(const T* __restrict__ x, const T* __restrict__ gy,
                                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ ws,
                                                     int channels, long long hw, int chunks) {
    __shared__ float red[2 * NT / 64];
    const long long plane = blockIdx.x / chunks;
    const int chunk = blockIdx.x % chunks;
    const int c = (int)(plane % channels);
    const float mean = stats[2 * plane], rstd = stats[2 * plane + 1];
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const T* xp = x + plane * hw;
    const T* gp = gy + plane * hw;
    float s1 = 0.f, s2 = 0.f;
    for_chunk<T>(hw, chunk, [&](long long e, int n_, bool vec) {
        float v[VecIO<T>::N], d[VecIO<T>::N];
        const int cnt = vec ? VecIO<T>::N : 1;
        if (vec) {
            VecIO<T>::load(xp + e, v);
            VecIO<T>::load(gp + e, d);
        } else {
            v[0] = VecIO<T>::load1(xp + e);
            d[0] = VecIO<T>::load1(gp + e);
        }
        for (int i = 0; i < cnt; ++i) {
            const float n = (v[i] - mean) * rstd;
            float ga = d[i];
            if (GELU) ga *= gelu_grad_f(n * g + b);
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
    sums[2 * p] = (float)s1;
    sums[2 * p + 1] = (float)s2;
}

// pass 2: gx = rstd * gamma * (ga - S1/hw - n * S2/hw)
template <typename T, bool GELU>
__global__ __launch_bounds__(NT) void in_bwd_apply(const T* __restrict__ x, const T* __restrict__ gy, T* __restrict__ gx,
                                                   const float* __restrict__ stats, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, const float* __restrict__ sums,
                                                   int channels, long long hw, int chunks, float inv_total) {
    const long long plane = blockIdx.x / chunks;
    const int chunk = blockIdx.x % chunks;
    const int c = (int)(plane % channels);
    const float mean = stats[2 * plane], rstd = stats[2 * plane + 1];
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float m1 = sums[2 * plane] * inv_total, m2 = sums[2 * plane + 1] * inv_total;
    const float k = rstd * g;
    const T* xp = x + plane * hw;
    const T* gp = gy + plane * hw;
    T* op = gx + plane * hw;
    for_chunk<T>(hw, chunk, [&](long long e, int n_, bool vec) {
        float v[VecIO<T>::N], d[VecIO<T>::N];
        const int cnt = vec ? VecIO<T>::N : 1;
        if (vec) {
            VecIO<T>::load(xp + e, v);
            VecIO<T>::load(gp + e, d);
        } else {
            v[0] = VecIO<T>::load1(xp + e);
            d[0] = VecIO<T>::load1(gp + e);
        }
        for (int i = 0; i < cnt; ++i) {
            const float n = (v[i] - mean) * rstd;
            float ga = d[i];
            if (GELU) ga *= gelu_grad_f(n * g + b);
            v[i] = k * (ga - m1 - n * m2);
        }
        if (vec)
            VecIO<T>::store(op + e, v);
        else
            VecIO<T>::store1(op + e, v[0]);
    });
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
    for_chunk<T>(hw, chunk, [&](long long e, int n, bool vec) {
        float v[VecIO<T>::N];
        if (vec) {
            VecIO<T>::load(xp + e, v);
#pragma unroll
            for (int i = 0; i < VecIO<T>::N; ++i) v[i] = gelu_f(v[i] + b);
            VecIO<T>::store(yp + e, v);
        } else {
            VecIO<T>::store1(yp + e, gelu_f(VecIO<T>::load1(xp + e) + b));
        }
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
    for_chunk<T>(hw, chunk, [&](long long e, int n, bool vec) {
        float v[VecIO<T>::N], d[VecIO<T>::N];
        const int cnt = vec ? VecIO<T>::N : 1;
        if (vec) {
            VecIO<T>::load(xp + e, v);
            VecIO<T>::load(gp + e, d);
        } else {
            v[0] = VecIO<T>::load1(xp + e);
            d[0] = VecIO<T>::load1(gp + e);
        }
        for (int i = 0; i < cnt; ++i) {
            v[i] = d[i] * gelu_grad_f(v[i] + b);
            s1 += v[i];
        }
        if (vec)
            VecIO<T>::store(op + e, v);
        else
            VecIO<T>::store1(op + e, v[0]);
    });
    if (ws) {
        block_reduce2(s1, s2, red);
        if (threadIdx.x == 0) {
            ws[2 * (long long)blockIdx.x] = s1;
            ws[2 * (long long)blockIdx.x + 1] = 0.f;
        }
    }
}

template <typename T>
int chunks_for(long long hw) {
    return (int)((hw + chunk_elems<T>() - 1) / chunk_elems<T>());
}

int check_common(const void* a, long long planes, long long hw, int dtype, const char* what) {
    MK_REQUIRE(a != nullptr, "%s: null pointer", what);
    MK_REQUIRE(planes > 0 && hw > 0, "%s: bad shape planes=%lld hw=%lld", what, planes, hw);
    MK_REQUIRE(dtype == MK_F32 || dtype == MK_BF16, "%s: bad dtype %d", what, dtype);
    MK_REQUIRE(planes * (long long)((hw + 1023) / 1024) < (1ll << 31), "%s: grid too large", what);
    return 0;
}

}  // namespace

extern "C" int mk_pointwise_chunks(long long hw, int dtype) {
    return dtype == MK_BF16 ? chunks_for<u16>(hw) : chunks_for<float>(hw);
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
                                 float eps, void* stream) {
    int rc = check_common(x, planes, hw, dtype, "instnorm_stats");
    if (rc) return rc;
    MK_REQUIRE(stats && ws, "instnorm_stats: null stats/ws");
    hipStream_t s = (hipStream_t)stream;
    const int fb = (int)((planes + 255) / 256);
    if (dtype == MK_F32) {
        const int ch = chunks_for<float>(hw);
        hipLaunchKernelGGL(in_stats_partial<float>, dim3((unsigned)(planes * ch)), dim3(NT), 0, s, (const float*)x, ws, hw, ch);
        hipLaunchKernelGGL(in_stats_final<float>, dim3(fb), dim3(256), 0, s, (const float*)x, ws, stats, planes, hw, ch, eps);
    } else {
        const int ch = chunks_for<u16>(hw);
        hipLaunchKernelGGL(in_stats_partial<u16>, dim3((unsigned)(planes * ch)), dim3(NT), 0, s, (const u16*)x, ws, hw, ch);
        hipLaunchKernelGGL(in_stats_final<u16>, dim3(fb), dim3(256), 0, s, (const u16*)x, ws, stats, planes, hw, ch, eps);
    }
    return mk_check_launch("mk_instnorm_stats");
}

extern "C" int mk_instnorm_apply(const void* x, void* y, int dtype, const float* stats, const float* gamma,
                                 const float* beta, long long planes, int channels, long long hw, int fuse_gelu,
                                 void* stream) {
    int rc = check_common(x, planes, hw, dtype, "instnorm_apply");
    if (rc) return rc;
    MK_REQUIRE(y && stats && channels > 0, "instnorm_apply: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MK_F32) {
        const int ch = chunks_for<float>(hw);
        dim3 g((unsigned)(planes * ch));
        if (fuse_gelu)
            hipLaunchKernelGGL((in_apply<float, true>), g, dim3(NT), 0, s, (const float*)x, (float*)y, stats, gamma, beta, channels, hw, ch);
        else
            hipLaunchKernelGGL((in_apply<float, false>), g, dim3(NT), 0, s, (const float*)x, (float*)y, stats, gamma, beta, channels, hw, ch);
    } else {
        const int ch = chunks_for<u16>(hw);
        dim3 g((unsigned)(planes * ch));
        if (fuse_gelu)
            hipLaunchKernelGGL((in_apply<u16, true>), g, dim3(NT), 0, s, (const u16*)x, (u16*)y, stats, gamma, beta, channels, hw, ch);
        else
            hipLaunchKernelGGL((in_apply<u16, false>), g, dim3(NT), 0, s, (const u16*)x, (u16*)y, stats, gamma, beta, channels, hw, ch);
    }
    return mk_check_launch("mk_instnorm_apply");
}

extern "C" int mk_instnorm_bwd(const void* x, const void* gy, void* gx, int dtype, const float* stats,
                               const float* gamma, const float* beta, float* sums, float* ws, long long planes,
                               int channels, long long hw, long long hw_total, int phase, int fuse_gelu, void* stream) {
    int rc = check_common(x, planes, hw, dtype, "instnorm_bwd");
    if (rc) return rc;
    MK_REQUIRE(gy && gx && stats && sums && ws && channels > 0, "instnorm_bwd: bad args");
    MK_REQUIRE(phase >= 0 && phase <= 2 && hw_total >= hw, "instnorm_bwd: bad phase / hw_total");
    hipStream_t s = (hipStream_t)stream;
    const int fb = (int)((planes + 255) / 256);
    const float inv_total = 1.0f / (float)hw_total;
#define IN_BWD(T, G)                                                                                                   \
    do {                                                                                                               \
        const int ch = chunks_for<T>(hw);                                                                              \
        dim3 g((unsigned)(planes * ch));                                                                               \
        if (phase != 2) {                                                                                              \
            hipLaunchKernelGGL((in_bwd_partial<T, G>), g, dim3(NT), 0, s, (const T*)x, (const T*)gy, stats, gamma,    \
                               beta, ws, channels, hw, ch);                                                            \
            hipLaunchKernelGGL(sum_chunks_final, dim3(fb), dim3(256), 0, s, ws, sums, planes, ch);                     \
        }                                                                                                              \
        if (phase != 1)                                                                                                \
            hipLaunchKernelGGL((in_bwd_apply<T, G>), g, dim3(NT), 0, s, (const T*)x, (const T*)gy, (T*)gx, stats,     \
                               gamma, beta, sums, channels, hw, ch, inv_total);                                        \
    } while (0)
    if (dtype == MK_F32) {
        if (fuse_gelu) IN_BWD(float, true); else IN_BWD(float, false);
    } else {
        if (fuse_gelu) IN_BWD(u16, true); else IN_BWD(u16, false);
    }
#undef IN_BWD
    return mk_check_launch("mk_instnorm_bwd");
}

extern "C" int mk_bias_gelu_fwd(const void* x, const float* bias, void* y, int dtype, long long planes, int channels,
                                long long hw, void* stream) {
    int rc = check_common(x, planes, hw, dtype, "bias_gelu_fwd");
    if (rc) return rc;
    MK_REQUIRE(y && channels > 0, "bias_gelu_fwd: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MK_F32) {
        const int ch = chunks_for<float>(hw);
        hipLaunchKernelGGL(bias_gelu_fwd<float>, dim3((unsigned)(planes * ch)), dim3(NT), 0, s, (const float*)x, bias, (float*)y, channels, hw, ch);
    } else {
        const int ch = chunks_for<u16>(hw);
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
        const int ch = chunks_for<float>(hw);
        hipLaunchKernelGGL(bias_gelu_bwd<float>, dim3((unsigned)(planes * ch)), dim3(NT), 0, s, (const float*)x, bias, (const float*)gy, (float*)gx, ws, channels, hw, ch);
        if (sums) hipLaunchKernelGGL(sum_chunks_final, dim3(fb), dim3(256), 0, s, ws, sums, planes, ch);
    } else {
        const int ch = chunks_for<u16>(hw);
        hipLaunchKernelGGL(bias_gelu_bwd<u16>, dim3((unsigned)(planes * ch)), dim3(NT), 0, s, (const u16*)x, bias, (const u16*)gy, (u16*)gx, ws, channels, hw, ch);
        if (sums) hipLaunchKernelGGL(sum_chunks_final, dim3(fb), dim3(256), 0, s, ws, sums, planes, ch);
    }
    return mk_check_launch("mk_bias_gelu_bwd");
}
