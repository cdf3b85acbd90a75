// Discrete-continuous (DISCO) convolution contraction on the sphere and bilinear S2 resampling (gfx950).
//
// Replaces th.DiscreteContinuousConvS2's sparse contraction (`_disco_s2_contraction_*`) and th.ResampleS2
// [torch-harmonics, un-vendored; call sites makani/models/networks/fourcastnet3.py:189-205,356-381,518-534].
//
//   forward   y[pl][k][t][p] = sum_{n in list(t, k)} val[n] * x[pl][lat_lo[t] + row[n]][(lon[n] + p * s) mod nlon_in]
//   adjoint   gx[pl][i][q]   = sum_{n in tlist(i), (q - lon[n]) mod nlon_in = p * s} val[n] * gy[pl][k[n]][t[n]][p]
//
// s = nlon_in / nlon_out.  The convolution tensor has the same sparsity pattern at every output longitude, so it is kept
// as short lists per (output latitude, basis function); the lists are walked with wave-uniform (scalar) loads.
// Forward: one workgroup per (output latitude, group of PB planes) copies the <= max_rows input latitude rows it needs
// into LDS (each HBM byte of x is read once per output latitude that touches it: 2 cutoff / dlat + 1 times, from L2 after
// the first), lanes own output longitudes: LDS reads of consecutive lanes are s floats apart.  Output: the NCHW tensor
// (planes * K, nlat_out, nlon_out) the channel GEMM kernels consume in place.  HBM-bound: bytes = x + K * y.
// Adjoint: deterministic gather, one workgroup per (input latitude, plane), gy read through L2.
#include "common.h"

namespace {

template <typename T>
__device__ __forceinline__ float ldf(const T* p);
template <>
__device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ldf<u16>(const u16* p) { return bf16_to_f32(*p); }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(u16* p, float v) { *p = f32_to_bf16(v); }

constexpr int DNT = 256;

template <typename T, int PB>
__global__ __launch_bounds__(DNT) void disco_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, const int* __restrict__ off,
                                                        const int* __restrict__ nrow, const int* __restrict__ nlon,
                                                        const float* __restrict__ nval, const int* __restrict__ lat_lo,
                                                        const int* __restrict__ lat_n, int max_rows, int planes, int K,
                                                        int nlat_in, int nlon_in, int nlat_out, int nlon_out) {
    extern __shared__ __attribute__((aligned(16))) float xs[];      // [PB][rows][nlon_in]
    const int t = blockIdx.x, p0 = blockIdx.y * PB, tid = threadIdx.x;
    const int npl = min(PB, planes - p0);
    const int lo = lat_lo[t], nr = lat_n[t];
    const int s = nlon_in / nlon_out;
    const long long plane_in = (long long)nlat_in * nlon_in, plane_out = (long long)nlat_out * nlon_out;
    for (int b = 0; b < npl; ++b) {
        const T* src = x + (p0 + b) * plane_in + (long long)lo * nlon_in;
        float* dst = xs + (long long)b * max_rows * nlon_in;
        for (int e = tid; e < nr * nlon_in; e += DNT) dst[e] = ldf(src + e);
    }
    __syncthreads();
    for (int p = tid; p < nlon_out; p += DNT) {
        const int sh = p * s;
        for (int k = 0; k < K; ++k) {
            const int n0 = off[t * K + k], n1 = off[t * K + k + 1];
            float acc[PB];
#pragma unroll
            for (int b = 0; b < PB; ++b) acc[b] = 0.f;
            for (int n = n0; n < n1; ++n) {
                const float v = nval[n];
                int c = nlon[n] + sh;
                c -= (c >= nlon_in) ? nlon_in : 0;
                const float* r = xs + nrow[n] * nlon_in + c;
#pragma unroll
                for (int b = 0; b < PB; ++b) acc[b] = fmaf(v, r[(long long)b * max_rows * nlon_in], acc[b]);
            }
#pragma unroll
            for (int b = 0; b < PB; ++b)
                if (b < npl) stf(y + ((p0 + b) * (long long)K + k) * plane_out + (long long)t * nlon_out + p, acc[b]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(DNT) void disco_bwd_kernel(const T* __restrict__ gy, T* __restrict__ gx, const int* __restrict__ off,
                                                        const int* __restrict__ nk, const int* __restrict__ nt,
                                                        const int* __restrict__ nlon, const float* __restrict__ nval, int K,
                                                        int nlat_in, int nlon_in, int nlat_out, int nlon_out) {
    const int i = blockIdx.x, pl = blockIdx.y, tid = threadIdx.x;
    const int s = nlon_in / nlon_out;
    const long long plane_out = (long long)nlat_out * nlon_out;
    const T* g = gy + (long long)pl * K * plane_out;
    const int n0 = off[i], n1 = off[i + 1];
    for (int q = tid; q < nlon_in; q += DNT) {
        float acc = 0.f;
        for (int n = n0; n < n1; ++n) {
            int d = q - nlon[n];
            d += (d < 0) ? nlon_in : 0;
            const int p = d / s;
            if (p * s == d) acc = fmaf(nval[n], ldf(g + nk[n] * plane_out + (long long)nt[n] * nlon_out + p), acc);
        }
        stf(gx + ((long long)pl * nlat_in + i) * nlon_in + q, acc);
    }
}

// ---- bilinear resampling --------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float tot = 0.f;
    for (int q = 0; q < DNT / 64; ++q) tot += red[q];
    return tot;
}

// row source: r >= 0 input row r; -1 / -2: the mean over longitude of the first / last input row (pole extension)
template <typename T>
__global__ __launch_bounds__(DNT) void resample_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, const int* __restrict__ lat_a,
                                                           const int* __restrict__ lat_b, const float* __restrict__ lat_w,
                                                           const int* __restrict__ lon_l, const int* __restrict__ lon_r,
                                                           const float* __restrict__ lon_w, int nlat_in, int nlon_in,
                                                           int nlat_out, int nlon_out) {
    __shared__ float red[DNT / 64];
    const int t = blockIdx.x, pl = blockIdx.y, tid = threadIdx.x;
    const T* xp = x + (long long)pl * nlat_in * nlon_in;
    const int a = lat_a[t], b = lat_b[t];
    const float w = lat_w[t];
    float ma = 0.f, mb = 0.f;
    if (a < 0) {
        const T* row = xp + (long long)(a == -1 ? 0 : nlat_in - 1) * nlon_in;
        float sacc = 0.f;
        for (int j = tid; j < nlon_in; j += DNT) sacc += ldf(row + j);
        ma = block_sum(sacc, red) / (float)nlon_in;
    }
    if (b < 0) {
        const T* row = xp + (long long)(b == -1 ? 0 : nlat_in - 1) * nlon_in;
        float sacc = 0.f;
        for (int j = tid; j < nlon_in; j += DNT) sacc += ldf(row + j);
        mb = block_sum(sacc, red) / (float)nlon_in;
    }
    const T* ra = xp + (long long)max(a, 0) * nlon_in;
    const T* rb = xp + (long long)max(b, 0) * nlon_in;
    for (int p = tid; p < nlon_out; p += DNT) {
        const int l = lon_l[p], r = lon_r[p];
        const float al = a < 0 ? ma : ldf(ra + l), ar = a < 0 ? ma : ldf(ra + r);
        const float bl = b < 0 ? mb : ldf(rb + l), br = b < 0 ? mb : ldf(rb + r);
        const float yl = al + w * (bl - al), yr = ar + w * (br - ar);           // latitude first, as the reference
        stf(y + ((long long)pl * nlat_out + t) * nlon_out + p, yl + lon_w[p] * (yr - yl));
    }
}

// adjoint: gx[i][j] = sum_{(t, wl) in lat_inv(i)} wl * sum_{(p, wp) in lon_inv(j)} wp * gy[t][p]
//                     + [i is a polar row] (1 / nlon_in) * sum_{(t, wl) in pole_inv} wl * sum_p gy[t][p]
template <typename T>
__global__ __launch_bounds__(DNT) void resample_bwd_kernel(const T* __restrict__ gy, T* __restrict__ gx, const int* __restrict__ lat_off,
                                                           const int* __restrict__ lat_t, const float* __restrict__ lat_wt,
                                                           const int* __restrict__ lon_off, const int* __restrict__ lon_p,
                                                           const float* __restrict__ lon_wt, const int* __restrict__ pole_off,
                                                           const int* __restrict__ pole_t, const float* __restrict__ pole_wt,
                                                           int nlat_in, int nlon_in, int nlat_out, int nlon_out) {
    __shared__ float red[DNT / 64];
    const int i = blockIdx.x, pl = blockIdx.y, tid = threadIdx.x;
    const T* g = gy + (long long)pl * nlat_out * nlon_out;
    float pole = 0.f;
    const int which = (i == 0) ? 0 : ((i == nlat_in - 1) ? 1 : -1);
    if (which >= 0) {
        for (int n = pole_off[which]; n < pole_off[which + 1]; ++n) {
            const T* row = g + (long long)pole_t[n] * nlon_out;
            float sacc = 0.f;
            for (int p = tid; p < nlon_out; p += DNT) sacc += ldf(row + p);
            pole += pole_wt[n] * block_sum(sacc, red);
        }
        pole /= (float)nlon_in;
    }
    const int a0 = lat_off[i], a1 = lat_off[i + 1];
    for (int j = tid; j < nlon_in; j += DNT) {
        float acc = pole;
        const int b0 = lon_off[j], b1 = lon_off[j + 1];
        for (int n = a0; n < a1; ++n) {
            const T* row = g + (long long)lat_t[n] * nlon_out;
            float h = 0.f;
            for (int m = b0; m < b1; ++m) h = fmaf(lon_wt[m], ldf(row + lon_p[m]), h);
            acc = fmaf(lat_wt[n], h, acc);
        }
        stf(gx + ((long long)pl * nlat_in + i) * nlon_in + j, acc);
    }
}

template <typename T>
int launch_disco_fwd(const T* x, T* y, const int* off, const int* nrow, const int* nlon, const float* nval, const int* lat_lo,
                     const int* lat_n, int max_rows, int planes, int K, int nlat_in, int nlon_in, int nlat_out, int nlon_out,
                     hipStream_t s) {
    const size_t per_plane = (size_t)max_rows * nlon_in * sizeof(float);
    constexpr size_t LDS_CAP = 144 * 1024;
    MK_REQUIRE(per_plane <= LDS_CAP, "disco: %d input rows of %d longitudes do not fit the LDS", max_rows, nlon_in);
#define MK_DISCO_GO(PB)                                                                                                          \
    do {                                                                                                                         \
        auto kern = disco_fwd_kernel<T, PB>;                                                                                     \
        const size_t lds = per_plane * PB;                                                                                       \
        if (lds > 64 * 1024)                                                                                                     \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kern, dim3(nlat_out, (planes + PB - 1) / PB), dim3(DNT), lds, s, x, y, off, nrow, nlon, nval,        \
                           lat_lo, lat_n, max_rows, planes, K, nlat_in, nlon_in, nlat_out, nlon_out);                           \
    } while (0)
    if (per_plane * 4 <= LDS_CAP / 2 && planes >= 4)
        MK_DISCO_GO(4);
    else if (per_plane * 2 <= LDS_CAP / 2 && planes >= 2)
        MK_DISCO_GO(2);
    else
        MK_DISCO_GO(1);
#undef MK_DISCO_GO
    return mk_check_launch("mk_disco_fwd");
}

}  // namespace

extern "C" int mk_disco_fwd(const void* x, void* y, int dtype, const int* off, const int* nrow, const int* nlon, const float* nval,
                            const int* lat_lo, const int* lat_n, int max_rows, int planes, int K, int nlat_in, int nlon_in,
                            int nlat_out, int nlon_out, void* stream) {
    MK_REQUIRE(x && y && off && nrow && nlon && nval && lat_lo && lat_n, "disco_fwd: null pointer");
    MK_REQUIRE(planes > 0 && K > 0 && nlon_out > 0 && nlon_in % nlon_out == 0, "disco_fwd: nlon_in must be a multiple of nlon_out");
    MK_REQUIRE(planes <= 65535 * 4, "disco_fwd: too many planes");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MK_F32)
        return launch_disco_fwd<float>((const float*)x, (float*)y, off, nrow, nlon, nval, lat_lo, lat_n, max_rows, planes, K,
                                       nlat_in, nlon_in, nlat_out, nlon_out, s);
    return launch_disco_fwd<u16>((const u16*)x, (u16*)y, off, nrow, nlon, nval, lat_lo, lat_n, max_rows, planes, K, nlat_in,
                                 nlon_in, nlat_out, nlon_out, s);
}

extern "C" int mk_disco_bwd(const void* gy, void* gx, int dtype, const int* off, const int* nk, const int* nt, const int* nlon,
                            const float* nval, int planes, int K, int nlat_in, int nlon_in, int nlat_out, int nlon_out,
                            void* stream) {
    MK_REQUIRE(gy && gx && off && nk && nt && nlon && nval, "disco_bwd: null pointer");
    MK_REQUIRE(planes > 0 && planes <= 65535 && nlon_out > 0 && nlon_in % nlon_out == 0, "disco_bwd: bad extents");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(nlat_in, planes), block(DNT);
    if (dtype == MK_F32)
        hipLaunchKernelGGL(disco_bwd_kernel<float>, grid, block, 0, s, (const float*)gy, (float*)gx, off, nk, nt, nlon, nval, K,
                           nlat_in, nlon_in, nlat_out, nlon_out);
    else
        hipLaunchKernelGGL(disco_bwd_kernel<u16>, grid, block, 0, s, (const u16*)gy, (u16*)gx, off, nk, nt, nlon, nval, K,
                           nlat_in, nlon_in, nlat_out, nlon_out);
    return mk_check_launch("mk_disco_bwd");
}

extern "C" int mk_resample_fwd(const void* x, void* y, int dtype, const int* lat_a, const int* lat_b, const float* lat_w,
                               const int* lon_l, const int* lon_r, const float* lon_w, int planes, int nlat_in, int nlon_in,
                               int nlat_out, int nlon_out, void* stream) {
    MK_REQUIRE(x && y && lat_a && lat_b && lat_w && lon_l && lon_r && lon_w, "resample_fwd: null pointer");
    MK_REQUIRE(planes > 0 && planes <= 65535, "resample_fwd: bad plane count");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(nlat_out, planes), block(DNT);
    if (dtype == MK_F32)
        hipLaunchKernelGGL(resample_fwd_kernel<float>, grid, block, 0, s, (const float*)x, (float*)y, lat_a, lat_b, lat_w, lon_l,
                           lon_r, lon_w, nlat_in, nlon_in, nlat_out, nlon_out);
    else
        hipLaunchKernelGGL(resample_fwd_kernel<u16>, grid, block, 0, s, (const u16*)x, (u16*)y, lat_a, lat_b, lat_w, lon_l, lon_r,
                           lon_w, nlat_in, nlon_in, nlat_out, nlon_out);
    return mk_check_launch("mk_resample_fwd");
}

extern "C" int mk_resample_bwd(const void* gy, void* gx, int dtype, const int* lat_off, const int* lat_t, const float* lat_wt,
                               const int* lon_off, const int* lon_p, const float* lon_wt, const int* pole_off, const int* pole_t,
                               const float* pole_wt, int planes, int nlat_in, int nlon_in, int nlat_out, int nlon_out,
                               void* stream) {
    MK_REQUIRE(gy && gx && lat_off && lon_off && pole_off, "resample_bwd: null pointer");
    MK_REQUIRE(planes > 0 && planes <= 65535, "resample_bwd: bad plane count");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(nlat_in, planes), block(DNT);
    if (dtype == MK_F32)
        hipLaunchKernelGGL(resample_bwd_kernel<float>, grid, block, 0, s, (const float*)gy, (float*)gx, lat_off, lat_t, lat_wt,
                           lon_off, lon_p, lon_wt, pole_off, pole_t, pole_wt, nlat_in, nlon_in, nlat_out, nlon_out);
    else
        hipLaunchKernelGGL(resample_bwd_kernel<u16>, grid, block, 0, s, (const u16*)gy, (u16*)gx, lat_off, lat_t, lat_wt, lon_off,
                           lon_p, lon_wt, pole_off, pole_t, pole_wt, nlat_in, nlon_in, nlat_out, nlon_out);
    return mk_check_launch("mk_resample_bwd");
}
