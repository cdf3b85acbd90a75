// Layer norm over the CHANNELS of an NCHW tensor, one statistic per grid point (gfx950).
//
// Replaces makani's DistributedLayerNorm (makani/mpu/layer_norm.py:256-290: nn.LayerNorm(C) between two transposes of
// the NCHW activation, `normalization_layer="layer_norm"` of sfnonet.py:609-613 / fourcastnet3.py:95-96) and its autograd
// WITHOUT the transposes: the pixel index stays the contiguous one, a lane owns VEC consecutive pixels and walks the
// channel planes, so every access is a coalesced 16-byte (f32) / 8-byte (bf16) vector of one plane.
//
//   x, gy, gx, y : (B, C, P) planes, f32 | bf16 (y may be f32 for bf16 x: nn.LayerNorm runs in fp32 under autocast)
//   stats        : (B, 2, P) f32  [mean, rstd] per grid point, written by the forward, read by both backward kernels
//   forward   y[c][p]  = (x[c][p] - mean[p]) * rstd[p] * gamma[c] + beta[c]
//   backward  gx[c][p] = rstd[p] * (t[c][p] - mean_c t - xh[c][p] * mean_c(t xh)),  t = gamma[c] gy[c][p],  xh = (x - mean) rstd
//   wgrad     dgamma[c] = sum_{b,p} gy xh,  dbeta[c] = sum_{b,p} gy      (per-chunk partial sums, added up by the caller)
//
// HBM-bound: forward 2 reads + 1 write of the tensor (the moments need the whole channel column before the first output;
// C x VEC values do not fit a lane's registers at C = 384), backward 2 x (x, gy) reads + 1 write, wgrad 1 x (x, gy).
// Moments: shifted sums (shift = the first channel's value) in fp32 — no cancellation for columns with a large mean.
#include "common.h"

namespace {

constexpr int LNT = 256;
constexpr int CU_ = 8;       // channel planes whose loads are issued together

template <typename T, int VEC>
__device__ __forceinline__ void ldpx(const T* p, float (&v)[VEC]);
template <>
__device__ __forceinline__ void ldpx<float, 4>(const float* p, float (&v)[4]) {
    const f32x4 r = *reinterpret_cast<const f32x4*>(p);
    v[0] = r[0], v[1] = r[1], v[2] = r[2], v[3] = r[3];
}
template <>
__device__ __forceinline__ void ldpx<float, 1>(const float* p, float (&v)[1]) { v[0] = *p; }
template <>
__device__ __forceinline__ void ldpx<u16, 4>(const u16* p, float (&v)[4]) {
    const uint2 r = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(r.x << 16), v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16), v[3] = __uint_as_float(r.y & 0xffff0000u);
}
template <>
__device__ __forceinline__ void ldpx<u16, 1>(const u16* p, float (&v)[1]) { v[0] = bf16_to_f32(*p); }

template <typename T, int VEC>
__device__ __forceinline__ void stpx(T* p, const float (&v)[VEC]);
template <>
__device__ __forceinline__ void stpx<float, 4>(float* p, const float (&v)[4]) {
    f32x4 r;
    r[0] = v[0], r[1] = v[1], r[2] = v[2], r[3] = v[3];
    *reinterpret_cast<f32x4*>(p) = r;
}
template <>
__device__ __forceinline__ void stpx<float, 1>(float* p, const float (&v)[1]) { *p = v[0]; }
template <>
__device__ __forceinline__ void stpx<u16, 4>(u16* p, const float (&v)[4]) {
    uint2 r;
    r.x = pack_bf16x2(v[0], v[1]);
    r.y = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = r;
}
template <>
__device__ __forceinline__ void stpx<u16, 1>(u16* p, const float (&v)[1]) { *p = f32_to_bf16(v[0]); }

// grid: (pixel tiles, B).  A lane owns VEC consecutive grid points of one sample.
template <typename TI, typename TO, int VEC>
__global__ __launch_bounds__(LNT) void chan_ln_fwd_kernel(const TI* __restrict__ x, TO* __restrict__ y, float* __restrict__ stats,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, int C,
                                                          long long P, float eps) {
    const long long p0 = ((long long)blockIdx.x * LNT + threadIdx.x) * VEC;
    if (p0 >= P) return;
    const int b = blockIdx.y;
    const TI* xb = x + (long long)b * C * P + p0;
    float x0[VEC], s[VEC], q[VEC];
    ldpx<TI, VEC>(xb, x0);
#pragma unroll
    for (int i = 0; i < VEC; ++i) s[i] = 0.f, q[i] = 0.f;
    for (int c0 = 0; c0 < C; c0 += CU_) {
        float v[CU_][VEC];
#pragma unroll
        for (int u = 0; u < CU_; ++u) ldpx<TI, VEC>(xb + (long long)min(c0 + u, C - 1) * P, v[u]);      // all loads first
#pragma unroll
        for (int u = 0; u < CU_; ++u)
            if (c0 + u < C) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float d = v[u][i] - x0[i];
                    s[i] += d;
                    q[i] = fmaf(d, d, q[i]);
                }
            }
    }
    const float inv_c = 1.f / (float)C;
    float mean[VEC], rstd[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        const float ms = s[i] * inv_c;
        mean[i] = x0[i] + ms;
        rstd[i] = rsqrtf(fmaxf(q[i] * inv_c - ms * ms, 0.f) + eps);
    }
    stpx<float, VEC>(stats + ((long long)b * 2 + 0) * P + p0, mean);
    stpx<float, VEC>(stats + ((long long)b * 2 + 1) * P + p0, rstd);
    TO* yb = y + (long long)b * C * P + p0;
    for (int c0 = 0; c0 < C; c0 += CU_) {
        float v[CU_][VEC];
#pragma unroll
        for (int u = 0; u < CU_; ++u) ldpx<TI, VEC>(xb + (long long)min(c0 + u, C - 1) * P, v[u]);
#pragma unroll
        for (int u = 0; u < CU_; ++u)
            if (c0 + u < C) {
                const float g = gamma ? gamma[c0 + u] : 1.f, bt = beta ? beta[c0 + u] : 0.f;
                float o[VEC];
#pragma unroll
                for (int i = 0; i < VEC; ++i) o[i] = fmaf((v[u][i] - mean[i]) * rstd[i], g, bt);
                stpx<TO, VEC>(yb + (long long)(c0 + u) * P, o);
            }
    }
}

template <typename TI, typename TG, int VEC>
__global__ __launch_bounds__(LNT) void chan_ln_bwd_kernel(const TI* __restrict__ x, const TG* __restrict__ gy, TI* __restrict__ gx,
                                                          const float* __restrict__ stats, const float* __restrict__ gamma, int C,
                                                          long long P) {
    const long long p0 = ((long long)blockIdx.x * LNT + threadIdx.x) * VEC;
    if (p0 >= P) return;
    const int b = blockIdx.y;
    const TI* xb = x + (long long)b * C * P + p0;
    const TG* gb = gy + (long long)b * C * P + p0;
    float mean[VEC], rstd[VEC], s1[VEC], s2[VEC];
    ldpx<float, VEC>(stats + ((long long)b * 2 + 0) * P + p0, mean);
    ldpx<float, VEC>(stats + ((long long)b * 2 + 1) * P + p0, rstd);
#pragma unroll
    for (int i = 0; i < VEC; ++i) s1[i] = 0.f, s2[i] = 0.f;
    constexpr int U2 = CU_ / 2;
    for (int c0 = 0; c0 < C; c0 += U2) {
        float v[U2][VEC], g[U2][VEC];
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            ldpx<TI, VEC>(xb + (long long)min(c0 + u, C - 1) * P, v[u]);
            ldpx<TG, VEC>(gb + (long long)min(c0 + u, C - 1) * P, g[u]);
        }
#pragma unroll
        for (int u = 0; u < U2; ++u)
            if (c0 + u < C) {
                const float gm = gamma ? gamma[c0 + u] : 1.f;
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float t = gm * g[u][i];
                    s1[i] += t;
                    s2[i] = fmaf(t, (v[u][i] - mean[i]) * rstd[i], s2[i]);
                }
            }
    }
    const float inv_c = 1.f / (float)C;
#pragma unroll
    for (int i = 0; i < VEC; ++i) s1[i] *= inv_c, s2[i] *= inv_c;
    TI* ob = gx + (long long)b * C * P + p0;
    for (int c0 = 0; c0 < C; c0 += U2) {
        float v[U2][VEC], g[U2][VEC];
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            ldpx<TI, VEC>(xb + (long long)min(c0 + u, C - 1) * P, v[u]);
            ldpx<TG, VEC>(gb + (long long)min(c0 + u, C - 1) * P, g[u]);
        }
#pragma unroll
        for (int u = 0; u < U2; ++u)
            if (c0 + u < C) {
                const float gm = gamma ? gamma[c0 + u] : 1.f;
                float o[VEC];
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float xh = (v[u][i] - mean[i]) * rstd[i];
                    o[i] = rstd[i] * (gm * g[u][i] - s1[i] - xh * s2[i]);
                }
                stpx<TI, VEC>(ob + (long long)(c0 + u) * P, o);
            }
    }
}

// grid: (chunks, C).  partial[0][c][chunk] = sum gy xh, partial[1][c][chunk] = sum gy over this chunk of the B * P grid points
template <typename TI, typename TG>
__global__ __launch_bounds__(LNT) void chan_ln_wgrad_kernel(const TI* __restrict__ x, const TG* __restrict__ gy,
                                                            const float* __restrict__ stats, float* __restrict__ partial, int B,
                                                            int C, long long P) {
    __shared__ float red[2][LNT / 64];
    const int c = blockIdx.y, chunks = gridDim.x;
    const long long per = (P + chunks - 1) / chunks;
    const long long a0 = (long long)blockIdx.x * per, a1 = min(P, a0 + per);
    float dg = 0.f, db = 0.f;
    for (int b = 0; b < B; ++b) {
        const TI* xp = x + ((long long)b * C + c) * P;
        const TG* gp = gy + ((long long)b * C + c) * P;
        const float* mp = stats + (long long)b * 2 * P;
        for (long long p = a0 + threadIdx.x; p < a1; p += LNT) {
            float xv[1], gv[1];
            ldpx<TI, 1>(xp + p, xv);
            ldpx<TG, 1>(gp + p, gv);
            dg = fmaf(gv[0], (xv[0] - mp[p]) * mp[P + p], dg);
            db += gv[0];
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        dg += __shfl_down(dg, o, 64);
        db += __shfl_down(db, o, 64);
    }
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = dg, red[1][threadIdx.x >> 6] = db;
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, bsum = 0.f;
        for (int i = 0; i < LNT / 64; ++i) a += red[0][i], bsum += red[1][i];
        partial[((long long)0 * C + c) * chunks + blockIdx.x] = a;
        partial[((long long)1 * C + c) * chunks + blockIdx.x] = bsum;
    }
}

template <typename TI, typename TO>
int launch_fwd(const void* x, void* y, float* stats, const float* gamma, const float* beta, int B, int C, long long P, float eps,
               hipStream_t s) {
    if (P % 4 == 0) {
        const dim3 grid((unsigned)((P / 4 + LNT - 1) / LNT), (unsigned)B);
        hipLaunchKernelGGL((chan_ln_fwd_kernel<TI, TO, 4>), grid, dim3(LNT), 0, s, (const TI*)x, (TO*)y, stats, gamma, beta, C, P, eps);
    } else {
        const dim3 grid((unsigned)((P + LNT - 1) / LNT), (unsigned)B);
        hipLaunchKernelGGL((chan_ln_fwd_kernel<TI, TO, 1>), grid, dim3(LNT), 0, s, (const TI*)x, (TO*)y, stats, gamma, beta, C, P, eps);
    }
    return mk_check_launch("mk_chan_layernorm_fwd");
}

template <typename TI, typename TG>
int launch_bwd(const void* x, const void* gy, void* gx, const float* stats, const float* gamma, int B, int C, long long P,
               hipStream_t s) {
    if (P % 4 == 0) {
        const dim3 grid((unsigned)((P / 4 + LNT - 1) / LNT), (unsigned)B);
        hipLaunchKernelGGL((chan_ln_bwd_kernel<TI, TG, 4>), grid, dim3(LNT), 0, s, (const TI*)x, (const TG*)gy, (TI*)gx, stats, gamma, C, P);
    } else {
        const dim3 grid((unsigned)((P + LNT - 1) / LNT), (unsigned)B);
        hipLaunchKernelGGL((chan_ln_bwd_kernel<TI, TG, 1>), grid, dim3(LNT), 0, s, (const TI*)x, (const TG*)gy, (TI*)gx, stats, gamma, C, P);
    }
    return mk_check_launch("mk_chan_layernorm_bwd");
}

}  // namespace

extern "C" int mk_chan_layernorm_chunks(int C, long long P) {
    long long ch = (256ll * 8 + C - 1) / C;                  // ~8 blocks per CU over all channels
    const long long most = (P + 4 * LNT - 1) / (4 * LNT);     // at least four elements per lane and batch entry
    if (ch > most) ch = most;
    return (int)(ch < 1 ? 1 : ch);
}

extern "C" int mk_chan_layernorm_fwd(const void* x, int x_dtype, void* y, int y_dtype, float* stats, const float* gamma,
                                     const float* beta, int B, int C, long long P, float eps, void* stream) {
    MK_REQUIRE(x && y && stats && B > 0 && C > 0 && P > 0 && B <= 65535, "chan_layernorm_fwd: bad arguments");
    MK_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)stats) & 15) == 0, "chan_layernorm_fwd: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == MK_F32 && y_dtype == MK_F32) return launch_fwd<float, float>(x, y, stats, gamma, beta, B, C, P, eps, s);
    if (x_dtype == MK_BF16 && y_dtype == MK_F32) return launch_fwd<u16, float>(x, y, stats, gamma, beta, B, C, P, eps, s);
    if (x_dtype == MK_BF16 && y_dtype == MK_BF16) return launch_fwd<u16, u16>(x, y, stats, gamma, beta, B, C, P, eps, s);
    mk_set_error("chan_layernorm_fwd: unsupported dtype combination (x f32 -> y f32, x bf16 -> y f32 | bf16)");
    return MK_EINVAL;
}

extern "C" int mk_chan_layernorm_bwd(const void* x, int x_dtype, const void* gy, int g_dtype, void* gx, const float* stats,
                                     const float* gamma, int B, int C, long long P, void* stream) {
    MK_REQUIRE(x && gy && gx && stats && B > 0 && C > 0 && P > 0 && B <= 65535, "chan_layernorm_bwd: bad arguments");
    MK_REQUIRE((((uintptr_t)x | (uintptr_t)gy | (uintptr_t)gx | (uintptr_t)stats) & 15) == 0, "chan_layernorm_bwd: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == MK_F32 && g_dtype == MK_F32) return launch_bwd<float, float>(x, gy, gx, stats, gamma, B, C, P, s);
    if (x_dtype == MK_BF16 && g_dtype == MK_F32) return launch_bwd<u16, float>(x, gy, gx, stats, gamma, B, C, P, s);
    if (x_dtype == MK_BF16 && g_dtype == MK_BF16) return launch_bwd<u16, u16>(x, gy, gx, stats, gamma, B, C, P, s);
    mk_set_error("chan_layernorm_bwd: unsupported dtype combination");
    return MK_EINVAL;
}

extern "C" int mk_chan_layernorm_wgrad(const void* x, int x_dtype, const void* gy, int g_dtype, const float* stats, float* partial,
                                       int B, int C, long long P, void* stream) {
    MK_REQUIRE(x && gy && stats && partial && B > 0 && C > 0 && P > 0 && C <= 65535, "chan_layernorm_wgrad: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)mk_chan_layernorm_chunks(C, P), (unsigned)C);
#define MK_LN_WG(TI, TG) \
    hipLaunchKernelGGL((chan_ln_wgrad_kernel<TI, TG>), grid, dim3(LNT), 0, s, (const TI*)x, (const TG*)gy, stats, partial, B, C, P)
    if (x_dtype == MK_F32 && g_dtype == MK_F32) MK_LN_WG(float, float);
    else if (x_dtype == MK_BF16 && g_dtype == MK_F32) MK_LN_WG(u16, float);
    else if (x_dtype == MK_BF16 && g_dtype == MK_BF16) MK_LN_WG(u16, u16);
    else {
        mk_set_error("chan_layernorm_wgrad: unsupported dtype combination");
        return MK_EINVAL;
    }
#undef MK_LN_WG
    return mk_check_launch("mk_chan_layernorm_wgrad");
}
