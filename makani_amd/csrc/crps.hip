// Ensemble CRPS on the sphere: pointwise score over the ensemble dimension + quadrature over the plane, fused (gfx950).
//
// Replaces the kernels of makani/utils/losses/crps_loss.py ("skillspread" :124-162 — the default of CRPSLoss :277-452 —,
// "probability weighted moment" :164-203, "naive skillspread" :205-243, "gauss" :245-275) together with the quadrature sum
// `torch.sum(crps * quad_weight * spatial_weights, dim=-1)` (:435-438) and their autograd.
//
//   forecasts f[b][e][c][p]  (B, E, C, HW) f32 | bf16      observations o[b][c][p] (f32 | bf16)
//   out[b * C + c] = sum_p q[p] * w[b][c][p] * crps(o, f[:, p])                      (w optional)
//   gf[b][e][c][p] = gout[b * C + c] * q[p] * w * d crps / d f_e
//
// One thread owns one point: the E <= 32 members live in registers, ranks come from E^2 comparisons (ordinal ranks with the
// stable tie order of torch.argsort, as rankdata() of the reference), everything else is a handful of flops: the kernel reads
// (E + 1) values and, in backward, writes E — HBM-bound.  A NaN observation scores 0 and yields zero gradients, as the
// reference's masking does.  "cdf" (crps_loss.py:55-122, the form FourCastNet3's first pre-training stage uses,
// config/fourcastnet3.yaml:151-163): the members are put in rank order and the piecewise integral of (F - H)^2 is accumulated
// exactly as the reference's loop does (same branch conditions), optionally with per-member ensemble weights; its gradient
// is the derivative of that loop with respect to the sorted members, scattered back through the ranks.
// Ensemble sizes: every E in 2..32 — the kernels are instantiated for EM in {2, 4, 8, 16, 32} and run any E <= EM with the
// surplus members predicated off.
#include "common.h"

namespace {

constexpr int CNT = 256;
constexpr int MAXE = 32;

template <typename T>
__device__ __forceinline__ float ldv(const T* p);
template <>
__device__ __forceinline__ float ldv<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ldv<u16>(const u16* p) { return bf16_to_f32(*p); }
__device__ __forceinline__ void stv(float* p, float v) { *p = v; }
__device__ __forceinline__ void stv(u16* p, float v) { *p = f32_to_bf16(v); }

enum { CRPS_SKILLSPREAD = 0, CRPS_PWM = 1, CRPS_NAIVE = 2, CRPS_GAUSS = 3, CRPS_CDF = 4 };

__device__ __forceinline__ float sgn(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

// score of one point and (GRAD) its derivative with respect to every member, written back into f[].
// EM = compiled capacity, E <= EM = members present (members e >= E are ignored; callers load them as zeros).
// ew: per-member ensemble weights of the "cdf" form (nullptr = uniform)
template <int EM, bool GRAD>
__device__ __forceinline__ float crps_point(float (&f)[EM], int E, float obs, int type, float alpha, float eps,
                                            const float* __restrict__ ew) {
    const bool masked = type != CRPS_GAUSS && type != CRPS_CDF && obs != obs;      // NaN observation (gauss / cdf of the reference do not mask)
    const float o = masked ? 0.f : obs;
    const float inv_e = 1.f / (float)E;
    float skill = 0.f;
#pragma unroll
    for (int e = 0; e < EM; ++e)
        if (e < E) skill += fabsf(o - f[e]);
    skill *= inv_e;
    float score;
    float g[EM];
    if (type == CRPS_GAUSS) {
        float mu = 0.f;
#pragma unroll
        for (int e = 0; e < EM; ++e)
            if (e < E) mu += f[e];
        mu *= inv_e;
        float var = 0.f;
#pragma unroll
        for (int e = 0; e < EM; ++e)
            if (e < E) var += (f[e] - mu) * (f[e] - mu);
        var *= inv_e;
        const float sraw = sqrtf(var);
        const float sigma = fmaxf(sraw, eps);
        const float z = (obs - mu) / sigma;
        const float pdf = 0.3989422804014327f * __expf(-0.5f * z * z);
        const float cdf2m1 = erff(z * 0.7071067811865476f);
        score = sigma * (z * cdf2m1 + 2.f * pdf - 0.5641895835477563f);
        if (GRAD) {
            const float dmu = -cdf2m1, dsig = 2.f * pdf - 0.5641895835477563f;
#pragma unroll
            for (int e = 0; e < EM; ++e)
                g[e] = dmu * inv_e + ((sraw > eps) ? dsig * (f[e] - mu) * inv_e / sigma : 0.f);
        }
    } else {
        // ordinal ranks 1..E: members smaller than f_e, plus equal members that come before e
        float acc = 0.f;               // sum_e (2 r_e - E - 1) f_e   (skillspread / naive)   or   sum_e (r_e - 1) f_e   (pwm)
        float mean = 0.f;
        float coef[EM];
        int rank[EM];
#pragma unroll
        for (int e = 0; e < EM; ++e) {
            int r = 1;
            float ssum = 0.f;          // sum_j sign(f_e - f_j)   (naive form: zero contribution from ties)
#pragma unroll
            for (int j = 0; j < EM; ++j) {
                if (j < E) {
                    r += (f[j] < f[e] || (f[j] == f[e] && j < e)) ? 1 : 0;
                    ssum += sgn(f[e] - f[j]);
                }
            }
            rank[e] = r;
            coef[e] = (type == CRPS_PWM) ? (float)(r - 1) : ((type == CRPS_NAIVE) ? ssum : (float)(2 * r - E - 1));
            if (e < E) {
                acc += coef[e] * f[e];
                mean += f[e];
            }
        }
        mean *= inv_e;
        if (type == CRPS_CDF) {
            // members (and their weights) in rank order; then the reference's loop (crps_loss.py:84-117), prev_forecast of the
            // first step is the constant 0, forecast_cdf advances by w_n / sum(w)
            float total = 0.f;
#pragma unroll
            for (int e = 0; e < EM; ++e)
                if (e < E) total += ew ? ew[e] : 1.f;
            float obs_cdf = 0.f, fc = 0.f, prev = 0.f, integ = 0.f, last = 0.f;
            float gs[EM];              // gradient with respect to the n-th sorted member
#pragma unroll
            for (int n = 0; n < EM; ++n) {
                gs[n] = 0.f;
                if (n < E) {
                    float fo = 0.f, wn = 0.f;
#pragma unroll
                    for (int e = 0; e < EM; ++e) {
                        const bool hit = e < E && rank[e] == n + 1;
                        fo = hit ? f[e] : fo;
                        wn = hit ? (ew ? ew[e] : 1.f) : wn;
                    }
                    const bool cond = (obs < fo) && fabsf(obs_cdf) < 1.0e-7f;
                    const float d = fc - obs_cdf;
                    integ += cond ? ((obs - prev) * fc * fc + (fo - obs) * (fc - 1.f) * (fc - 1.f)) : ((fo - prev) * d * d);
                    if (GRAD) {
                        gs[n] = cond ? (fc - 1.f) * (fc - 1.f) : d * d;
                        const float gp = cond ? -fc * fc : -d * d;
                        if (n > 0) gs[n - 1] += gp;
                    }
                    obs_cdf = cond ? 1.f : obs_cdf;
                    fc += wn / total;
                    prev = fo;
                    last = fo;
                }
            }
            // a NaN observation propagates (the reference's torch.clamp(observation - forecast, min=0) does; fmaxf would
            // drop it and report a finite, meaningless score): NaN score, zero gradients.  The zero gradients DEPART from the
            // reference's backward on purpose: there only the clamp term's gradient vanishes and the piecewise-integral terms still
            // hand finite, non-zero gradients to the members of a point whose score is NaN; a training step that masks NaN
            // scores would then still be pulled by those points.  Here a point without an observation moves nothing.
            const bool obs_nan = obs != obs;
            score = obs_nan ? obs : integ + fmaxf(obs - last, 0.f);
            if (GRAD) {
#pragma unroll
                for (int e = 0; e < EM; ++e) {
                    float ge = 0.f;
#pragma unroll
                    for (int n = 0; n < EM; ++n) ge = (n < E && rank[e] == n + 1) ? gs[n] : ge;
                    g[e] = obs_nan ? 0.f : ge - ((rank[e] == E && obs > last) ? 1.f : 0.f);
                }
            }
        } else if (type == CRPS_PWM) {
            const float c1 = 1.f / (float)(E * (E - 1));
            score = skill + mean - 2.f * acc * c1;
            if (GRAD) {
#pragma unroll
                for (int e = 0; e < EM; ++e) g[e] = sgn(f[e] - o) * inv_e + inv_e - 2.f * coef[e] * c1;
            }
        } else {
            // espread = 2 mean((2r - E - 1) f) (E - 1 + alpha) / (E (E - 1));  naive: sum_ij |f_i - f_j| (E - 1 + alpha) / (E^2 (E - 1))
            const float c = ((float)E - 1.f + alpha) / (float)(E * (E - 1));
            score = skill - acc * inv_e * c;
            if (GRAD) {
#pragma unroll
                for (int e = 0; e < EM; ++e) g[e] = sgn(f[e] - o) * inv_e - coef[e] * inv_e * c;
            }
        }
    }
    if (GRAD) {
#pragma unroll
        for (int e = 0; e < EM; ++e) f[e] = masked ? 0.f : g[e];
    }
    return masked ? 0.f : score;
}

// grid: (chunks, planes = B * C).  Forward: partial[plane][chunk]; backward: gf written in place of the loop.
template <typename TF, typename TO, int EM, bool GRAD>
__global__ __launch_bounds__(CNT) void crps_kernel(const TF* __restrict__ f, const TO* __restrict__ obs, const float* __restrict__ q,
                                                   const float* __restrict__ w, const float* __restrict__ gout, float* __restrict__ partial,
                                                   TF* __restrict__ gf, int E, int C, long long hw, int type, float alpha, float eps,
                                                   const float* __restrict__ ew) {
    __shared__ float red[CNT / 64];
    const int plane = blockIdx.y, b = plane / C, c = plane % C;
    const long long estride = (long long)C * hw;
    const TF* fp = f + ((long long)b * E * C + c) * hw;
    const TO* op = obs + (long long)plane * hw;
    const float* wp = w ? w + (long long)plane * hw : nullptr;
    const float go = GRAD ? gout[plane] : 0.f;
    float sum = 0.f;
    for (long long p = (long long)blockIdx.x * CNT + threadIdx.x; p < hw; p += (long long)gridDim.x * CNT) {
        float v[EM];
#pragma unroll
        for (int e = 0; e < EM; ++e) v[e] = (e < E) ? ldv(fp + e * estride + p) : 0.f;
        const float s = crps_point<EM, GRAD>(v, E, ldv(op + p), type, alpha, eps, ew);
        const float wt = q[p] * (wp ? wp[p] : 1.f);
        if (GRAD) {
            TF* gp = gf + ((long long)b * E * C + c) * hw;
#pragma unroll
            for (int e = 0; e < EM; ++e)
                if (e < E) stv(gp + e * estride + p, go * wt * v[e]);
        } else {
            sum += wt * s;
        }
    }
    if (!GRAD) {
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
            for (int i = 0; i < CNT / 64; ++i) t += red[i];
            partial[(long long)plane * gridDim.x + blockIdx.x] = t;
        }
    }
}

// "naive skillspread" on COMPLEX members (SpectralCRPSLoss(absolute=False), crps_loss.py:205-243,536-545: the score of the
// spherical-harmonic coefficients themselves instead of their absolute values): |.| is the complex modulus,
//   crps = mean_e |o - f_e| - (E - 1 + alpha) / (2 E^2 (E - 1)) sum_{i,j} |f_i - f_j|,
// gradient in torch's convention (d / d re + i d / d im; the derivative of |z| is z / |z|, 0 at z = 0).
// f: (B, E, C, hw) complex64 (re, im interleaved), obs: (B, C, hw) complex64; gf like f.
template <int EM, bool GRAD>
__global__ __launch_bounds__(CNT) void crps_cplx_kernel(const float2* __restrict__ f, const float2* __restrict__ obs,
                                                        const float* __restrict__ q, const float* __restrict__ w,
                                                        const float* __restrict__ gout, float* __restrict__ partial,
                                                        float2* __restrict__ gf, int E, int C, long long hw, float alpha) {
    __shared__ float red[CNT / 64];
    const int plane = blockIdx.y, b = plane / C, c = plane % C;
    const long long estride = (long long)C * hw;
    const float2* fp = f + ((long long)b * E * C + c) * hw;
    const float2* op = obs + (long long)plane * hw;
    const float* wp = w ? w + (long long)plane * hw : nullptr;
    const float go = GRAD ? gout[plane] : 0.f;
    const float inv_e = 1.f / (float)E;
    const float coef = ((float)E - 1.f + alpha) / (float)((long long)E * E * (E - 1));
    float sum = 0.f;
    for (long long p = (long long)blockIdx.x * CNT + threadIdx.x; p < hw; p += (long long)gridDim.x * CNT) {
        float2 v[EM];
#pragma unroll
        for (int e = 0; e < EM; ++e) v[e] = (e < E) ? fp[e * estride + p] : make_float2(0.f, 0.f);
        float2 o = op[p];
        const bool masked = (o.x != o.x) || (o.y != o.y);
        if (masked) o = make_float2(0.f, 0.f);
        // Branch-free and every unordered pair once (the two ordered pairs contribute the same distance and opposite
        // gradients): members e >= E carry the mask 0.  The first version walked the E^2 ordered pairs under `j < E && j != e`
        // and `rr > 0` branches; fully unrolled at 32 members that was 1 024 conditional blocks, and hipcc spent five of the
        // library's seven minutes of build time on this one kernel.
        float skill = 0.f, spread = 0.f;
        float2 g[EM];
        float mk[EM];
#pragma unroll
        for (int e = 0; e < EM; ++e) {
            mk[e] = (e < E) ? 1.f : 0.f;
            const float dx = v[e].x - o.x, dy = v[e].y - o.y;
            const float r = sqrtf(dx * dx + dy * dy);
            skill += mk[e] * r;
            const float ir = (GRAD && r > 0.f) ? mk[e] * inv_e / r : 0.f;
            g[e] = make_float2(dx * ir, dy * ir);
        }
#pragma unroll
        for (int e = 0; e < EM; ++e) {
#pragma unroll
            for (int j = e + 1; j < EM; ++j) {
                const float ex = v[e].x - v[j].x, ey = v[e].y - v[j].y;
                const float rr = sqrtf(ex * ex + ey * ey);
                const float m = mk[e] * mk[j];
                spread += 2.f * m * rr;
                if (GRAD) {
                    const float cr = (rr > 0.f) ? coef * m / rr : 0.f;
                    g[e].x -= cr * ex;
                    g[e].y -= cr * ey;
                    g[j].x += cr * ex;
                    g[j].y += cr * ey;
                }
            }
        }
        const float wt = q[p] * (wp ? wp[p] : 1.f);
        if (GRAD) {
            float2* gp = gf + ((long long)b * E * C + c) * hw;
#pragma unroll
            for (int e = 0; e < EM; ++e)
                if (e < E) gp[e * estride + p] = masked ? make_float2(0.f, 0.f) : make_float2(go * wt * g[e].x, go * wt * g[e].y);
        } else {
            sum += masked ? 0.f : wt * (skill * inv_e - 0.5f * coef * spread);
        }
    }
    if (!GRAD) {
        for (int o2 = 32; o2 > 0; o2 >>= 1) sum += __shfl_down(sum, o2, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
            for (int i = 0; i < CNT / 64; ++i) t += red[i];
            partial[(long long)plane * gridDim.x + blockIdx.x] = t;
        }
    }
}

template <typename TF, typename TO, bool GRAD>
int launch_e(int E, dim3 grid, hipStream_t s, const TF* f, const TO* obs, const float* q, const float* w, const float* gout,
             float* partial, TF* gf, int C, long long hw, int type, float alpha, float eps, const float* ew) {
#define MK_CRPS_E(N)                                                                                                              \
    if (E <= N) {                                                                                                                 \
        hipLaunchKernelGGL((crps_kernel<TF, TO, N, GRAD>), grid, dim3(CNT), 0, s, f, obs, q, w, gout, partial, gf, E, C, hw, type, \
                           alpha, eps, ew);                                                                                       \
        return mk_check_launch("mk_crps");                                                                                        \
    }
    // the smallest instantiated capacity that holds E members
    MK_CRPS_E(2) MK_CRPS_E(4) MK_CRPS_E(8) MK_CRPS_E(16) MK_CRPS_E(32)
#undef MK_CRPS_E
    mk_set_error("crps: ensemble size %d exceeds the register-resident limit of 32 members", E);
    return MK_EUNSUP;
}

}  // namespace

extern "C" int mk_crps_chunks(long long hw) {
    long long c = (hw + 4 * CNT - 1) / (4 * CNT);
    return (int)(c < 1 ? 1 : (c > 64 ? 64 : c));
}

// grad == 0: partial (planes * mk_crps_chunks(hw)) f32 receives the chunk sums (the caller adds them up);
// grad == 1: gf (same shape and dtype as f) receives gout[plane] * q * w * dcrps/df
// ens_w: optional (E,) f32 per-member weights, used by type 4 ("cdf") only
extern "C" int mk_crps(const void* f, int f_dtype, const void* obs, int o_dtype, const float* q, const float* w, const float* gout,
                       float* partial, void* gf, int B, int E, int C, long long hw, int type, float alpha, float eps, int grad,
                       const float* ens_w, void* stream) {
    MK_REQUIRE(f && obs && q && B > 0 && E >= 2 && E <= MAXE && C > 0 && hw > 0, "crps: bad arguments (2 <= E <= 32)");
    MK_REQUIRE(type >= 0 && type <= 4, "crps: unknown score type %d", type);
    MK_REQUIRE(!ens_w || type == CRPS_CDF, "crps: ensemble weights are defined for the cdf form only");
    MK_REQUIRE(grad ? (gout && gf) : (partial != nullptr), "crps: missing output");
    MK_REQUIRE((long long)B * C <= 65535, "crps: too many planes");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)mk_crps_chunks(hw), (unsigned)(B * C));
#define MK_CRPS_GO(TF, TO)                                                                                                          \
    return grad ? launch_e<TF, TO, true>(E, grid, s, (const TF*)f, (const TO*)obs, q, w, gout, partial, (TF*)gf, C, hw, type, alpha, eps, ens_w) \
                : launch_e<TF, TO, false>(E, grid, s, (const TF*)f, (const TO*)obs, q, w, gout, partial, (TF*)gf, C, hw, type, alpha, eps, ens_w)
    if (f_dtype == MK_F32 && o_dtype == MK_F32) MK_CRPS_GO(float, float);
    if (f_dtype == MK_BF16 && o_dtype == MK_F32) MK_CRPS_GO(u16, float);
    if (f_dtype == MK_BF16 && o_dtype == MK_BF16) MK_CRPS_GO(u16, u16);
    if (f_dtype == MK_F32 && o_dtype == MK_BF16) MK_CRPS_GO(float, u16);
#undef MK_CRPS_GO
    mk_set_error("crps: unsupported dtype combination");
    return MK_EINVAL;
}

// The complex "naive skillspread" score (see crps_cplx_kernel): f (B, E, C, hw) and obs (B, C, hw) complex64, q / w / gout /
// partial / gf as mk_crps (gf complex64).  2 <= E <= 32.
extern "C" int mk_crps_complex(const void* f, const void* obs, const float* q, const float* w, const float* gout, float* partial,
                               void* gf, int B, int E, int C, long long hw, float alpha, int grad, void* stream) {
    MK_REQUIRE(f && obs && q && B > 0 && E >= 2 && E <= MAXE && C > 0 && hw > 0, "crps_complex: bad arguments (2 <= E <= 32)");
    MK_REQUIRE(grad ? (gout && gf) : (partial != nullptr), "crps_complex: missing output");
    MK_REQUIRE((long long)B * C <= 65535, "crps_complex: too many planes");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)mk_crps_chunks(hw), (unsigned)(B * C));
#define MK_CX(N)                                                                                                                     \
    if (E <= N) {                                                                                                                    \
        if (grad)                                                                                                                    \
            hipLaunchKernelGGL((crps_cplx_kernel<N, true>), grid, dim3(CNT), 0, s, (const float2*)f, (const float2*)obs, q, w, gout,  \
                               partial, (float2*)gf, E, C, hw, alpha);                                                               \
        else                                                                                                                         \
            hipLaunchKernelGGL((crps_cplx_kernel<N, false>), grid, dim3(CNT), 0, s, (const float2*)f, (const float2*)obs, q, w, gout, \
                               partial, (float2*)gf, E, C, hw, alpha);                                                               \
        return mk_check_launch("mk_crps_complex");                                                                                   \
    }
    MK_CX(2) MK_CX(4) MK_CX(8) MK_CX(16) MK_CX(32)
#undef MK_CX
    return MK_EUNSUP;
}
