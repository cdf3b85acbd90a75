// Discrete-continuous (DISCO) convolution contraction on the sphere and bilinear S2 resampling (gfx950).
//
// Replaces th.DiscreteContinuousConvS2's sparse contraction (`_disco_s2_contraction_*`) and th.ResampleS2
// [torch-harmonics, un-vendored; call sites makani/models/networks/fourcastnet3.py:189-205,356-381,518-534].
//
//   forward   y[pl][k][t][p] = sum_{n in list(t, k)} val[n] * x[pl][lat_lo[t] + row[n]][(lon[n] + p * s) mod nlon_in]
//   adjoint   gx[pl][i][q]   = sum_{n in tlist(i), (q - lon[n]) mod nlon_in = p * s} val[n] * gy[pl][k[n]][t[n]][p]
//
// s = nlon_in / nlon_out.  The convolution tensor has the same sparsity pattern at every output longitude, so it is kept
// as short lists per (output latitude, basis function).
// Forward: one workgroup per (output latitude, group of PB <= 4 planes) copies the <= max_rows input latitude rows it needs
// into LDS as fp32 (each HBM byte of x is read once per output latitude that touches it, from L2 after the first); a lane
// owns RR output longitudes (tid + 256 r) of all PB planes, i.e. RR * PB accumulators, and walks the list of one basis
// function, staged through LDS in chunks of 512 entries and read as a broadcast: one 16-byte LDS read and 3 RR index
// instructions per RR * PB multiply-adds.  Output: the NCHW tensor (planes * K, nlat_out, nlon_out) the channel GEMM kernels
// consume in place.  The contraction is fp32-VALU bound, not HBM bound (FourCastNet3's local block: 570 GFLOP per launch for
// 3.5 GB of traffic); a first version that walked the lists with dependent scalar loads per entry ran 10 x slower.
// Adjoint: on equal longitude counts the same correlation over transposed lists (gradient rows of one basis function at a
// time in LDS); otherwise a deterministic gather per input point.
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
constexpr int DCH = 512;          // list entries staged in LDS per chunk

struct __attribute__((aligned(16))) DEntry {
    int base, lon;              // row * nlon_in * PB * 4 and lon * PB * 4: BYTE offsets into the plane-interleaved row image
    float val;
    int pad;
};

template <int PB>
struct PlaneVec;
template <>
struct PlaneVec<1> { typedef float type; };
template <>
struct PlaneVec<2> { typedef f32x2 type; };
template <>
struct PlaneVec<4> { typedef f32x4 type; };

// acc[r][b] += sum over the staged list chunk of val * xs[row][(lon + shift[r]) mod nlon_in][b]
// The row image keeps the PB planes interleaved, so one LDS read (4 / 8 / 16 bytes) fetches the operands of all planes;
// every lane reads the same list entry (LDS broadcast).  All offsets are byte offsets (premultiplied by 4 PB): 4 integer
// instructions (two adds, a subtract, an unsigned min for the longitude wrap) and one LDS read per PB multiply-adds.  Unrolled by 4: the list reads and the x reads of four entries are in flight together
// (the loop is bound by LDS latency at the 1-2 workgroups per CU that the row images allow).
template <int PB, int RR>
__device__ __forceinline__ void disco_accumulate(float (&acc)[RR][PB], const DEntry* __restrict__ lst, int cnt,
                                                 const float* __restrict__ xs, const int (&shift)[RR], int wrap) {
    typedef typename PlaneVec<PB>::type vec;
    typedef __attribute__((address_space(3))) const unsigned char lds_b;    // 32-bit LDS addressing, byte offsets
    typedef __attribute__((address_space(3))) const vec lds_vec;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const i32x4 lds_entry;        // a DEntry read as one 16-byte vector
    lds_b* xl = (lds_b*)xs;
    lds_entry* ll = (lds_entry*)lst;
#pragma unroll 4
    for (int n = 0; n < cnt; ++n) {
        const i32x4 ev = ll[n];
        const float val = __int_as_float(ev[2]);
#pragma unroll
        for (int q = 0; q < RR; ++q) {
            // (lon + shift) mod wrap without a compare / select: the smaller of c and c - wrap as UNSIGNED numbers
            const unsigned c1 = (unsigned)(ev[1] + shift[q]);
            const unsigned c = min(c1, c1 - (unsigned)wrap);
            const vec v = *(lds_vec*)(xl + (unsigned)ev[0] + c);
#pragma unroll
            for (int b = 0; b < PB; ++b) {
                float xv;
                if constexpr (PB == 1) xv = v; else xv = v[b];
                acc[q][b] = fmaf(val, xv, acc[q][b]);
            }
        }
    }
}

template <int PB>
__device__ __forceinline__ void stage_list(DEntry* lst, const int* __restrict__ nrow, const int* __restrict__ nlon,
                                           const float* __restrict__ nval, int n0, int cnt, int tid, int row_elems) {
    for (int e = tid; e < cnt; e += DNT) {
        DEntry d;
        d.base = nrow[n0 + e] * row_elems * PB * 4;          // byte offsets into the fp32 image
        d.lon = nlon[n0 + e] * PB * 4;
        d.val = nval[n0 + e];
        d.pad = 0;
        lst[e] = d;
    }
}

// rows [0, nr) of PB planes -> the interleaved LDS image xs[row][lon][plane]
template <typename T, int PB>
__device__ __forceinline__ void stage_rows(float* xs, const T* __restrict__ src0, long long plane_step, int first, int planes,
                                           int nelem, int tid) {
    for (int b = 0; b < PB; ++b) {
        const T* src = src0 + (long long)min(first + b, planes - 1) * plane_step;
        for (int e = tid; e < nelem; e += DNT) xs[e * PB + b] = ldf(src + e);
    }
}

// forward: workgroup = (output latitude t, PB planes); lane owns output longitudes tid + 256 r, r < RR
template <typename T, int PB, int RR>
__global__ __launch_bounds__(DNT) void disco_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, const int* __restrict__ off,
                                                        const int* __restrict__ nrow, const int* __restrict__ nlon,
                                                        const float* __restrict__ nval, const int* __restrict__ lat_lo,
                                                        const int* __restrict__ lat_n, int max_rows, int planes, int K,
                                                        int nlat_in, int nlon_in, int nlat_out, int nlon_out) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    DEntry* lst = reinterpret_cast<DEntry*>(smem_f);
    float* xs = smem_f + DCH * (sizeof(DEntry) / sizeof(float));      // [PB][max_rows][nlon_in]
    const int t = blockIdx.x, p0 = blockIdx.y * PB, tid = threadIdx.x;
    const int npl = min(PB, planes - p0);
    const int lo = lat_lo[t], nr = lat_n[t];
    const int s = nlon_in / nlon_out;
    const long long plane_in = (long long)nlat_in * nlon_in, plane_out = (long long)nlat_out * nlon_out;
    stage_rows<T, PB>(xs, x + (long long)lo * nlon_in, plane_in, p0, planes, nr * nlon_in, tid);
    int shift[RR];
#pragma unroll
    for (int q = 0; q < RR; ++q) shift[q] = ((min(tid + q * DNT, nlon_out - 1) * s) % nlon_in) * PB * 4;
    for (int k = 0; k < K; ++k) {
        float acc[RR][PB];
#pragma unroll
        for (int q = 0; q < RR; ++q)
#pragma unroll
            for (int b = 0; b < PB; ++b) acc[q][b] = 0.f;
        const int n0 = off[t * K + k], n1 = off[t * K + k + 1];
        for (int c0 = n0; c0 < n1; c0 += DCH) {
            const int cnt = min(DCH, n1 - c0);
            __syncthreads();                                     // previous chunk consumed (and, first time, x staged)
            stage_list<PB>(lst, nrow, nlon, nval, c0, cnt, tid, nlon_in);
            __syncthreads();
            disco_accumulate<PB, RR>(acc, lst, cnt, xs, shift, nlon_in * PB * 4);
        }
#pragma unroll
        for (int q = 0; q < RR; ++q) {
            const int p = tid + q * DNT;
            if (p < nlon_out) {
#pragma unroll
                for (int b = 0; b < PB; ++b)
                    if (b < npl) stf(y + ((p0 + b) * (long long)K + k) * plane_out + (long long)t * nlon_out + p, acc[q][b]);
            }
        }
    }
}

// adjoint for nlon_in == nlon_out: gx[pl][i][q] = sum_k sum_{n in tlist(i, k)} val * gy[pl][k][t_lo + row][(lon' + q) mod nlon],
// lon' = (-lon) mod nlon: the same correlation as the forward kernel, once per basis function with that function's
// gradient rows staged in LDS
template <typename T, int PB, int RR>
__global__ __launch_bounds__(DNT) void disco_bwd_same_kernel(const T* __restrict__ gy, T* __restrict__ gx, const int* __restrict__ off,
                                                             const int* __restrict__ nrow, const int* __restrict__ nlst,
                                                             const float* __restrict__ nval, const int* __restrict__ t_lo,
                                                             const int* __restrict__ t_n, int max_rows, int planes, int K,
                                                             int nlat_in, int nlon, int nlat_out) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    DEntry* lst = reinterpret_cast<DEntry*>(smem_f);
    float* xs = smem_f + DCH * (sizeof(DEntry) / sizeof(float));
    const int i = blockIdx.x, p0 = blockIdx.y * PB, tid = threadIdx.x;
    const int npl = min(PB, planes - p0);
    const long long plane_out = (long long)nlat_out * nlon;
    int shift[RR];
#pragma unroll
    for (int q = 0; q < RR; ++q) shift[q] = min(tid + q * DNT, nlon - 1) * PB * 4;
    float acc[RR][PB];
#pragma unroll
    for (int q = 0; q < RR; ++q)
#pragma unroll
        for (int b = 0; b < PB; ++b) acc[q][b] = 0.f;
    for (int k = 0; k < K; ++k) {
        const int lo = t_lo[i * K + k], nr = t_n[i * K + k];
        const int n0 = off[i * K + k], n1 = off[i * K + k + 1];
        if (n0 == n1) continue;                                   // uniform
        __syncthreads();                                          // previous k's rows consumed
        stage_rows<T, PB>(xs, gy + (long long)k * plane_out + (long long)lo * nlon, (long long)K * plane_out, p0, planes,
                          nr * nlon, tid);
        for (int c0 = n0; c0 < n1; c0 += DCH) {
            const int cnt = min(DCH, n1 - c0);
            if (c0 != n0) __syncthreads();
            stage_list<PB>(lst, nrow, nlst, nval, c0, cnt, tid, nlon);
            __syncthreads();
            disco_accumulate<PB, RR>(acc, lst, cnt, xs, shift, nlon * PB * 4);
        }
    }
#pragma unroll
    for (int q = 0; q < RR; ++q) {
        const int p = tid + q * DNT;
        if (p < nlon) {
#pragma unroll
            for (int b = 0; b < PB; ++b)
                if (b < npl) stf(gx + ((long long)(p0 + b) * nlat_in + i) * nlon + p, acc[q][b]);
        }
    }
}

// general adjoint (nlon_in = s * nlon_out, s > 1: the encoder; its input usually needs no gradient): gather per input point
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
// One workgroup per (input row, plane).  Every gradient row the input row receives from (2-4 of them when upsampling by 2) is
// staged ONCE in LDS as fp32 with coalesced loads; the longitude stencils then gather from LDS.  (The first version gathered
// from global memory, lat entries x lon entries = 16 scattered reads per input point: 6-11 ms per FourCastNet3 decoder call
// depending on what the caches held; this form reads each gradient row twice in total.)
template <typename T>
__global__ __launch_bounds__(DNT) void resample_bwd_kernel(const T* __restrict__ gy, T* __restrict__ gx, const int* __restrict__ lat_off,
                                                           const int* __restrict__ lat_t, const float* __restrict__ lat_wt,
                                                           const int* __restrict__ lon_off, const int* __restrict__ lon_p,
                                                           const float* __restrict__ lon_wt, const int* __restrict__ pole_off,
                                                           const int* __restrict__ pole_t, const float* __restrict__ pole_wt,
                                                           int nlat_in, int nlon_in, int nlat_out, int nlon_out) {
    extern __shared__ __attribute__((aligned(16))) float rowbuf[];       // nlon_out floats
    __shared__ float red[DNT / 64];
    const int i = blockIdx.x, pl = blockIdx.y, tid = threadIdx.x;
    const T* g = gy + (long long)pl * nlat_out * nlon_out;
    constexpr int MAXQ = 8;                                              // input longitudes per thread: nlon_in <= 8 * 256
    float acc[MAXQ];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) acc[q] = 0.f;
    auto stage = [&](int t) {                                            // gradient row t -> LDS; returns this thread's partial row sum
        const T* row = g + (long long)t * nlon_out;
        float sacc = 0.f;
        for (int p = tid; p < nlon_out; p += DNT) {
            const float v = ldf(row + p);
            rowbuf[p] = v;
            sacc += v;
        }
        return sacc;
    };
    float pole = 0.f;
    const int which = (i == 0) ? 0 : ((i == nlat_in - 1) ? 1 : -1);
    if (which >= 0) {
        for (int n = pole_off[which]; n < pole_off[which + 1]; ++n) {
            float sacc = 0.f;
            const T* row = g + (long long)pole_t[n] * nlon_out;
            for (int p = tid; p < nlon_out; p += DNT) sacc += ldf(row + p);
            pole += pole_wt[n] * block_sum(sacc, red);
        }
        pole /= (float)nlon_in;
    }
    const int a0 = lat_off[i], a1 = lat_off[i + 1];
    for (int n = a0; n < a1; ++n) {
        __syncthreads();                                                 // the previous row is consumed
        (void)stage(lat_t[n]);
        __syncthreads();
        const float wl = lat_wt[n];
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int j = tid + q * DNT;
            if (j < nlon_in) {
                float h = 0.f;
                for (int m = lon_off[j]; m < lon_off[j + 1]; ++m) h = fmaf(lon_wt[m], rowbuf[lon_p[m]], h);
                acc[q] = fmaf(wl, h, acc[q]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const int j = tid + q * DNT;
        if (j < nlon_in) stf(gx + ((long long)pl * nlat_in + i) * nlon_in + j, acc[q] + pole);
    }
}

constexpr size_t DISCO_LDS_CAP = 150 * 1024;
constexpr size_t DISCO_LIST_BYTES = DCH * sizeof(DEntry);

// planes per workgroup: as many as fit (<= 4): one 16-byte LDS read then feeds four multiply-adds
inline int disco_planes_per_wg(size_t per_plane, int planes) {
    for (int pb = 4; pb > 1; pb >>= 1)
        if (planes >= pb && per_plane * pb + DISCO_LIST_BYTES <= DISCO_LDS_CAP) return pb;
    return 1;
}

template <typename KERN, typename... Args>
int disco_launch(KERN kern, dim3 grid, size_t lds, hipStream_t s, const char* what, Args... args) {
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(DNT), lds, s, args...);
    return mk_check_launch(what);
}

template <typename T>
int launch_disco_fwd(const T* x, T* y, const int* off, const int* nrow, const int* nlon, const float* nval, const int* lat_lo,
                     const int* lat_n, int max_rows, int planes, int K, int nlat_in, int nlon_in, int nlat_out, int nlon_out,
                     hipStream_t s) {
    const size_t per_plane = (size_t)max_rows * nlon_in * sizeof(float);
    MK_REQUIRE(per_plane + DISCO_LIST_BYTES <= DISCO_LDS_CAP, "disco: %d input rows of %d longitudes do not fit the LDS", max_rows, nlon_in);
    MK_REQUIRE(nlon_out <= 6 * DNT, "disco: at most %d output longitudes", 6 * DNT);
    const int pb = disco_planes_per_wg(per_plane, planes);
    const size_t lds = per_plane * pb + DISCO_LIST_BYTES;
    const dim3 grid(nlat_out, (planes + pb - 1) / pb);
#define MK_DISCO_GO(PB, RR)                                                                                                      \
    return disco_launch(disco_fwd_kernel<T, PB, RR>, grid, lds, s, "mk_disco_fwd", x, y, off, nrow, nlon, nval, lat_lo, lat_n,  \
                        max_rows, planes, K, nlat_in, nlon_in, nlat_out, nlon_out)
    if (nlon_out <= 3 * DNT) {
        if (pb == 4) MK_DISCO_GO(4, 3);
        if (pb == 2) MK_DISCO_GO(2, 3);
        MK_DISCO_GO(1, 3);
    }
    if (pb == 4) MK_DISCO_GO(4, 6);
    if (pb == 2) MK_DISCO_GO(2, 6);
    MK_DISCO_GO(1, 6);
#undef MK_DISCO_GO
}

template <typename T>
int launch_disco_bwd_same(const T* gy, T* gx, const int* off, const int* nrow, const int* nlon_l, const float* nval,
                          const int* t_lo, const int* t_n, int max_rows, int planes, int K, int nlat_in, int nlon, int nlat_out,
                          hipStream_t s) {
    const size_t per_plane = (size_t)max_rows * nlon * sizeof(float);
    MK_REQUIRE(per_plane + DISCO_LIST_BYTES <= DISCO_LDS_CAP, "disco: %d gradient rows of %d longitudes do not fit the LDS", max_rows, nlon);
    MK_REQUIRE(nlon <= 6 * DNT, "disco: at most %d longitudes", 6 * DNT);
    const int pb = disco_planes_per_wg(per_plane, planes);
    const size_t lds = per_plane * pb + DISCO_LIST_BYTES;
    const dim3 grid(nlat_in, (planes + pb - 1) / pb);
#define MK_DISCO_GO(PB, RR)                                                                                                      \
    return disco_launch(disco_bwd_same_kernel<T, PB, RR>, grid, lds, s, "mk_disco_bwd", gy, gx, off, nrow, nlon_l, nval, t_lo,  \
                        t_n, max_rows, planes, K, nlat_in, nlon, nlat_out)
    if (nlon <= 3 * DNT) {
        if (pb == 4) MK_DISCO_GO(4, 3);
        if (pb == 2) MK_DISCO_GO(2, 3);
        MK_DISCO_GO(1, 3);
    }
    if (pb == 4) MK_DISCO_GO(4, 6);
    if (pb == 2) MK_DISCO_GO(2, 6);
    MK_DISCO_GO(1, 6);
#undef MK_DISCO_GO
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

extern "C" int mk_disco_bwd_same(const void* gy, void* gx, int dtype, const int* off, const int* nrow, const int* nlon_l,
                                 const float* nval, const int* t_lo, const int* t_n, int max_rows, int planes, int K,
                                 int nlat_in, int nlon, int nlat_out, void* stream) {
    MK_REQUIRE(gy && gx && off && nrow && nlon_l && nval && t_lo && t_n, "disco_bwd_same: null pointer");
    MK_REQUIRE(planes > 0 && K > 0 && planes <= 65535 * 4, "disco_bwd_same: bad extents");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MK_F32)
        return launch_disco_bwd_same<float>((const float*)gy, (float*)gx, off, nrow, nlon_l, nval, t_lo, t_n, max_rows, planes, K,
                                            nlat_in, nlon, nlat_out, s);
    return launch_disco_bwd_same<u16>((const u16*)gy, (u16*)gx, off, nrow, nlon_l, nval, t_lo, t_n, max_rows, planes, K, nlat_in,
                                      nlon, nlat_out, s);
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
    MK_REQUIRE(nlon_in <= 8 * DNT && nlon_out <= 15360, "resample_bwd: at most %d input / 15360 output longitudes", 8 * DNT);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(nlat_in, planes), block(DNT);
    const size_t lds = (size_t)nlon_out * sizeof(float);
    if (dtype == MK_F32)
        hipLaunchKernelGGL(resample_bwd_kernel<float>, grid, block, lds, s, (const float*)gy, (float*)gx, lat_off, lat_t, lat_wt,
                           lon_off, lon_p, lon_wt, pole_off, pole_t, pole_wt, nlat_in, nlon_in, nlat_out, nlon_out);
    else
        hipLaunchKernelGGL(resample_bwd_kernel<u16>, grid, block, lds, s, (const u16*)gy, (u16*)gx, lat_off, lat_t, lat_wt, lon_off,
                           lon_p, lon_wt, pole_off, pole_t, pole_wt, nlat_in, nlon_in, nlat_out, nlon_out);
    return mk_check_launch("mk_resample_bwd");
}
