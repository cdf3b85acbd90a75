// MK_HIPCC_FLAGS: -fno-slp-vectorize
// (gfx950: v_pk_mul_f32 / v_pk_add_f32 whose src1 is a VGPR pair read through op_sel return wrong results while certain
//  matrix-core kernels run on the same compute unit — tools/pk_hazard_probe.py, docs/LAB_NOTEBOOK.md round 6.  The SLP vectoriser
//  emits exactly those forms from plain scalar code, so this file is compiled without it; tools/pk_opsel_scan.py checks the ISA.)
// Longitude real FFTs of the spherical harmonic transform (gfx950).
//
//   mk_rfft_rows : x[row][lat][lon] (f32|bf16) -> F[m][ri][row][lat]  (truncated to mmax modes, weighted)
//   mk_irfft_rows: F[m][ri][row][lat]          -> x[row][lat][lon]    (zero-padded beyond mmax, weighted)
//
// HBM-bound kernels.  One workgroup transforms RB consecutive latitudes of one (batch, channel)
// plane: rows are read/written as full contiguous lines (16 B / lane), the spectrum side is
// written in lat-major F-layout so the Legendre GEMM (batched over m, contracting lat) streams it
// with unit stride.  A real length-N transform runs as a complex length-N/2 Stockham autosort FFT
// in LDS (mixed radix 2/3/4/5 + generic small primes, twiddles from a host-fp64 table), followed
// by the Hermitian untangling step; truncation to mmax modes / zero padding happen in that step,
// so only the modes the SHT keeps ever touch HBM.  Work items are handed to XCDs in contiguous
// ranges (xcd_remap) so that lat-adjacent workgroups share an L2 and their RB-float runs of the
// F-layout merge into full lines before write-back.
#include <stdlib.h>

#include "fft_common.h"

int mk_fft_fast_dispatch(bool inverse, const void* in, void* out, int dtype, const float* twiddle, int B, int C, int Cp,
                         int nlat, int nlon, int mmax, float w_dc, float w_pos, float w_nyq, const MkFftSeg* seg, void* stream);

static bool use_fast_fft() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("MAKANI_AMD_FFT_GENERIC");
        v = (e && e[0] == '1') ? 0 : 1;
    }
    return v == 1;
}

namespace {

constexpr int NT = 512;
constexpr int MAX_RADIX_PASSES = 16;
constexpr int MAX_GENERIC_RADIX = 32;

struct RadixList {
    int n;
    int r[MAX_RADIX_PASSES];
};

// One Stockham pass (decimation in time) of radix R over RB rows of length n2 held in LDS.
//   src/dst: [RB][LS] float2;  tw: exp(-2 pi i q / N), N = 2*n2;  Ns = product of earlier radices.
template <int R>
__device__ __forceinline__ void stockham_pass(const float2* __restrict__ src, float2* __restrict__ dst,
                                              const float2* __restrict__ tw, int RB, int LS, int n2, int N, int Ns,
                                              int tid) {
    const int nb = n2 / R;               // butterflies per row
    const int tstep = N / (Ns * R);      // twiddle index step
    for (int idx = tid; idx < RB * nb; idx += NT) {
        const int row = idx / nb, j = idx - row * nb;
        const int k = j % Ns;
        const float2* s = src + row * LS;
        float2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = s[j + r * nb];
        if (Ns > 1) {
#pragma unroll
            for (int r = 1; r < R; ++r) v[r] = cmul(v[r], tw[k * r * tstep]);
        }
        dft_small<R>(v);
        const int j0 = (j - k) * R + k;
        float2* d = dst + row * LS + j0;
#pragma unroll
        for (int r = 0; r < R; ++r) d[r * Ns] = v[r];
    }
}

// generic radix (small odd primes > 5): O(R^2) with table twiddles
__device__ __noinline__ void stockham_pass_generic(const float2* __restrict__ src, float2* __restrict__ dst,
                                                   const float2* __restrict__ tw, int R, int RB, int LS, int n2,
                                                   int N, int Ns, int tid) {
    const int nb = n2 / R;
    const int tstep = N / (Ns * R);
    const int rstep = N / R;
    for (int idx = tid; idx < RB * nb; idx += NT) {
        const int row = idx / nb, j = idx - row * nb;
        const int k = j % Ns;
        const float2* s = src + row * LS;
        const int j0 = (j - k) * R + k;
        float2* d = dst + row * LS + j0;
        for (int o = 0; o < R; ++o) {
            float2 acc = make_float2(0.f, 0.f);
            for (int r = 0; r < R; ++r) {
                float2 x = s[j + r * nb];
                x = cmul(x, tw[k * r * tstep]);
                x = cmul(x, tw[((r * o) % R) * rstep]);
                acc = cadd(acc, x);
            }
            d[o * Ns] = acc;
        }
    }
}

// runs all passes; returns pointer to the buffer holding the result
__device__ __forceinline__ float2* run_passes(float2* a, float2* b, const float2* tw, const RadixList& rl, int RB,
                                              int LS, int n2, int N, int tid) {
    float2* src = a;
    float2* dst = b;
    int Ns = 1;
    for (int p = 0; p < rl.n; ++p) {
        const int R = rl.r[p];
        switch (R) {
            case 2: stockham_pass<2>(src, dst, tw, RB, LS, n2, N, Ns, tid); break;
            case 3: stockham_pass<3>(src, dst, tw, RB, LS, n2, N, Ns, tid); break;
            case 4: stockham_pass<4>(src, dst, tw, RB, LS, n2, N, Ns, tid); break;
            case 5: stockham_pass<5>(src, dst, tw, RB, LS, n2, N, Ns, tid); break;
            default: stockham_pass_generic(src, dst, tw, R, RB, LS, n2, N, Ns, tid); break;
        }
        Ns *= R;
        __syncthreads();
        float2* t = src;
        src = dst;
        dst = t;
    }
    return src;
}

// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NT) void rfft_kernel(const T* __restrict__ x, float* __restrict__ F,
                                                  const float2* __restrict__ tw_g, const RadixList rl, int C, int Cp,
                                                  long long rows, long long planes, int nlat, int nlon, int mmax, int RB, int ngr, float w_dc,
                                                  float w_pos, float w_nyq) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int N = nlon, n2 = nlon / 2, LS = n2 + 1;
    float2* bufA = reinterpret_cast<float2*>(smem_raw);
    float2* bufB = bufA + RB * LS;
    float2* tw = bufB + RB * LS;
    const int tid = threadIdx.x;

    const long long item = xcd_remap(blockIdx.x, gridDim.x);
    // work item = one latitude x RB consecutive (batch, channel) planes: the spectrum side is written as
    // runs of RB consecutive rows of the k-major F^T layout  F[m][lat][ri][row]
    const int klat = (int)(item / ngr);
    const long long p0 = (item % ngr) * RB;
    const int nr = (int)min((long long)RB, planes - p0);

    for (int q = tid; q < N; q += NT) tw[q] = tw_g[q];

    // load rows: z[j] = x[2j] + i x[2j+1]
    const T* xr = x + (p0 * nlat + klat) * (long long)nlon;     // row r of the item: xr + r * nlat * nlon
    for (int idx = tid; idx < RB * n2; idx += NT) {
        const int row = idx / n2, j = idx - row * n2;
        float2 z = make_float2(0.f, 0.f);
        if (row < nr) z = load_pair<T>(xr + (long long)row * nlat * nlon + 2 * j);
        bufA[row * LS + j] = z;
    }
    __syncthreads();

    const float2* Z = run_passes(bufA, bufB, tw, rl, RB, LS, n2, N, tid);

    // Hermitian untangle + truncate + weight, write lat-major
    for (int idx = tid; idx < mmax * RB; idx += NT) {
        const int r = idx % RB, m = idx / RB;
        if (r >= nr) continue;
        const int ma = (m == n2) ? 0 : m;
        const int mb = (m == 0 || m == n2) ? 0 : n2 - m;
        const float2 A = Z[r * LS + ma];
        const float2 Bc = cconj(Z[r * LS + mb]);
        const float2 u = cadd(A, Bc), t = csub(A, Bc);
        const float2 wt = cmul(tw[m], t);
        float w = w_pos;
        float2 X = make_float2(0.5f * (u.x + wt.y), 0.5f * (u.y - wt.x));
        if (m == 0) {
            w = w_dc;
            X.y = 0.f;
        } else if (m == n2) {
            w = w_nyq;
            X.y = 0.f;
        }
        const long long pr = p0 + r;
        float* o = F + ((long long)m * nlat + klat) * 2 * rows + (pr / C) * Cp + (pr % C);
        o[0] = w * X.x;
        o[rows] = w * X.y;
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void irfft_kernel(const float* __restrict__ F, T* __restrict__ x,
                                                   const float2* __restrict__ tw_g, const RadixList rl, int C, int Cp,
                                                   long long rows, long long planes, int nlat, int nlon, int mmax, int RB, int ngr, float w_dc,
                                                   float w_pos, float w_nyq) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int N = nlon, n2 = nlon / 2, LS = n2 + 1;
    float2* bufA = reinterpret_cast<float2*>(smem_raw);
    float2* bufB = bufA + RB * LS;
    float2* tw = bufB + RB * LS;
    const int tid = threadIdx.x;

    const long long item = xcd_remap(blockIdx.x, gridDim.x);
    // work item = one latitude x RB consecutive (batch, channel) planes: the spectrum side is written as
    // runs of RB consecutive rows of the k-major F^T layout  F[m][lat][ri][row]
    const int klat = (int)(item / ngr);
    const long long p0 = (item % ngr) * RB;
    const int nr = (int)min((long long)RB, planes - p0);

    for (int q = tid; q < N; q += NT) tw[q] = tw_g[q];

    // stage the weighted half spectrum X'[m], m = 0..n2 (zero beyond mmax) into bufB[r][m]
    for (int idx = tid; idx < (n2 + 1) * RB; idx += NT) {
        const int r = idx % RB, m = idx / RB;
        float2 X = make_float2(0.f, 0.f);
        if (m < mmax && r < nr) {
            const long long pr = p0 + r;
            const float* s = F + ((long long)m * nlat + klat) * 2 * rows + (pr / C) * Cp + (pr % C);
            X = make_float2(s[0], s[rows]);
            if (m == 0) {
                X = make_float2(w_dc * X.x, 0.f);
            } else if (m == n2) {
                X = make_float2(w_nyq * X.x, 0.f);
            } else {
                X = make_float2(0.5f * w_pos * X.x, 0.5f * w_pos * X.y);
            }
        }
        bufB[r * LS + m] = X;
    }
    __syncthreads();

    // Zs[j] = (A + Bc) + i conj(W^j) (A - Bc),  A = X'[j], Bc = conj(X'[n2-j]); store conj(Zs)
    for (int idx = tid; idx < RB * n2; idx += NT) {
        const int row = idx / n2, j = idx - row * n2;
        const float2 A = bufB[row * LS + j];
        const float2 Bc = cconj(bufB[row * LS + n2 - j]);
        const float2 u = cadd(A, Bc), t = csub(A, Bc);
        const float2 wt = cmul(cconj(tw[j]), t);           // W^{-j} (A - Bc)
        const float2 Zs = make_float2(u.x - wt.y, u.y + wt.x);   // u + i*wt
        bufA[row * LS + j] = cconj(Zs);
    }
    __syncthreads();

    const float2* Z = run_passes(bufA, bufB, tw, rl, RB, LS, n2, N, tid);

    // x[2j] + i x[2j+1] = conj(FFT(conj(Zs)))
    T* xr = x + (p0 * nlat + klat) * (long long)nlon;
    for (int idx = tid; idx < RB * n2; idx += NT) {
        const int row = idx / n2, j = idx - row * n2;
        if (row >= nr) continue;
        const float2 z = Z[row * LS + j];
        store_pair<T>(xr + (long long)row * nlat * nlon + 2 * j, z.x, -z.y);
    }
}

int plan(int nlon, const int* radix, int nradix, RadixList* rl, int* RB, size_t* lds) {
    MK_REQUIRE(nlon >= 4 && (nlon % 2) == 0, "fft: nlon=%d must be even and >= 4", nlon);
    MK_REQUIRE(radix && nradix >= 1 && nradix <= MAX_RADIX_PASSES, "fft: bad radix list");
    const int n2 = nlon / 2;
    long long prod = 1;
    rl->n = nradix;
    for (int i = 0; i < nradix; ++i) {
        MK_REQUIRE(radix[i] >= 2 && radix[i] < MAX_GENERIC_RADIX, "fft: unsupported radix %d", radix[i]);
        rl->r[i] = radix[i];
        prod *= radix[i];
    }
    MK_REQUIRE(prod == n2, "fft: radix product %lld != nlon/2 = %d", prod, n2);
    const size_t LS = n2 + 1;
    const size_t budget = 104 * 1024;
    const size_t twb = (size_t)nlon * 8;
    if (twb + 2 * LS * 8 > budget) {
        mk_set_error("fft: nlon=%d does not fit the LDS plan", nlon);
        return MK_EUNSUP;
    }
    int rb = 16;
    while (rb > 1 && twb + 2 * (size_t)rb * LS * 8 > budget) rb >>= 1;
    *RB = rb;
    *lds = twb + 2 * (size_t)rb * LS * 8;
    return 0;
}

template <typename K>
int set_lds(K kernel, size_t lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) {
        mk_set_error("fft: hipFuncSetAttribute(%zu) failed: %s", lds, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

}  // namespace

extern "C" int mk_rfft_rows(const void* x, int x_dtype, float* F, const float* twiddle, const int* radix, int nradix,
                            int B, int C, int Cp, int nlat, int nlon, int mmax, float w_dc, float w_pos,
                            float w_nyq, void* stream) {
    MK_REQUIRE(x && F && twiddle, "rfft: null pointer");
    MK_REQUIRE(B > 0 && C > 0 && Cp >= C && nlat > 0, "rfft: bad shape B=%d C=%d Cp=%d nlat=%d", B, C, Cp, nlat);
    const long long rows = (long long)B * Cp;      // rows of the F-layout
    const long long planes = (long long)B * C;     // planes of x
    MK_REQUIRE(mmax >= 1 && mmax <= nlon / 2 + 1, "rfft: mmax=%d out of range for nlon=%d", mmax, nlon);
    MK_REQUIRE(x_dtype == MK_F32 || x_dtype == MK_BF16, "rfft: bad dtype %d", x_dtype);
    if (use_fast_fft()) {
        const int frc = mk_fft_fast_dispatch(false, x, F, x_dtype, twiddle, B, C, Cp, nlat, nlon, mmax, w_dc, w_pos, w_nyq, nullptr, stream);
        if (frc != -1000) return frc;
    }
    RadixList rl;
    int RB;
    size_t lds;
    int rc = plan(nlon, radix, nradix, &rl, &RB, &lds);
    if (rc) return rc;
    const int ngr = (int)((planes + RB - 1) / RB);
    const long long nblk = (long long)nlat * ngr;
    MK_REQUIRE(nblk < (1ll << 31), "rfft: grid too large");
    hipStream_t s = (hipStream_t)stream;
    const float2* tw = reinterpret_cast<const float2*>(twiddle);
    if (x_dtype == MK_F32) {
        if ((rc = set_lds(rfft_kernel<float>, lds))) return rc;
        hipLaunchKernelGGL(rfft_kernel<float>, dim3((unsigned)nblk), dim3(NT), lds, s, (const float*)x, F, tw, rl, C, Cp, rows, planes,
                           nlat, nlon, mmax, RB, ngr, w_dc, w_pos, w_nyq);
    } else if (x_dtype == MK_BF16) {
        if ((rc = set_lds(rfft_kernel<u16>, lds))) return rc;
        hipLaunchKernelGGL(rfft_kernel<u16>, dim3((unsigned)nblk), dim3(NT), lds, s, (const u16*)x, F, tw, rl, C, Cp, rows, planes,
                           nlat, nlon, mmax, RB, ngr, w_dc, w_pos, w_nyq);
    } else {
        MK_REQUIRE(false, "rfft: bad dtype %d", x_dtype);
    }
    return mk_check_launch("mk_rfft_rows");
}

extern "C" int mk_irfft_rows(const float* F, void* x, int x_dtype, const float* twiddle, const int* radix, int nradix,
                             int B, int C, int Cp, int nlat, int nlon, int mmax, float w_dc, float w_pos,
                             float w_nyq, void* stream) {
    MK_REQUIRE(x && F && twiddle, "irfft: null pointer");
    MK_REQUIRE(B > 0 && C > 0 && Cp >= C && nlat > 0, "irfft: bad shape B=%d C=%d Cp=%d nlat=%d", B, C, Cp, nlat);
    const long long rows = (long long)B * Cp;
    const long long planes = (long long)B * C;
    MK_REQUIRE(mmax >= 1 && mmax <= nlon / 2 + 1, "irfft: mmax=%d out of range for nlon=%d", mmax, nlon);
    MK_REQUIRE(x_dtype == MK_F32 || x_dtype == MK_BF16, "irfft: bad dtype %d", x_dtype);
    if (use_fast_fft()) {
        const int frc = mk_fft_fast_dispatch(true, F, x, x_dtype, twiddle, B, C, Cp, nlat, nlon, mmax, w_dc, w_pos, w_nyq, nullptr, stream);
        if (frc != -1000) return frc;
    }
    RadixList rl;
    int RB;
    size_t lds;
    int rc = plan(nlon, radix, nradix, &rl, &RB, &lds);
    if (rc) return rc;
    const int ngr = (int)((planes + RB - 1) / RB);
    const long long nblk = (long long)nlat * ngr;
    MK_REQUIRE(nblk < (1ll << 31), "irfft: grid too large");
    hipStream_t s = (hipStream_t)stream;
    const float2* tw = reinterpret_cast<const float2*>(twiddle);
    if (x_dtype == MK_F32) {
        if ((rc = set_lds(irfft_kernel<float>, lds))) return rc;
        hipLaunchKernelGGL(irfft_kernel<float>, dim3((unsigned)nblk), dim3(NT), lds, s, F, (float*)x, tw, rl, C, Cp, rows, planes, nlat,
                           nlon, mmax, RB, ngr, w_dc, w_pos, w_nyq);
    } else if (x_dtype == MK_BF16) {
        if ((rc = set_lds(irfft_kernel<u16>, lds))) return rc;
        hipLaunchKernelGGL(irfft_kernel<u16>, dim3((unsigned)nblk), dim3(NT), lds, s, F, (u16*)x, tw, rl, C, Cp, rows, planes, nlat,
                           nlon, mmax, RB, ngr, w_dc, w_pos, w_nyq);
    } else {
        MK_REQUIRE(false, "irfft: bad dtype %d", x_dtype);
    }
    return mk_check_launch("mk_irfft_rows");
}


// ---- segmented variants (distributed transforms): only the specialised kernels of fft_fast.hip implement them ----------
extern "C" int mk_fft_seg_supported(int nlon) {
    return nlon == 1440 || nlon == 720 || nlon == 480 || nlon == 360 || nlon == 128 || nlon == 72;
}

extern "C" int mk_rfft_rows_seg(const void* x, int x_dtype, float* F, const float* twiddle, int C, int nlat, int nlon, int mmax,
                                float w_dc, float w_pos, float w_nyq, const MkFftSeg* seg, void* stream) {
    MK_REQUIRE(x && F && twiddle && seg, "rfft_seg: null pointer");
    MK_REQUIRE(mmax >= 1 && mmax <= nlon / 2 + 1, "rfft_seg: mmax=%d out of range for nlon=%d", mmax, nlon);
    MK_REQUIRE(x_dtype == MK_F32 || x_dtype == MK_BF16, "rfft_seg: bad dtype %d", x_dtype);
    const int rc = mk_fft_fast_dispatch(false, x, F, x_dtype, twiddle, 1, C, (C + 3) / 4 * 4, nlat, nlon, mmax, w_dc, w_pos, w_nyq, seg, stream);
    if (rc == -1000) {
        mk_set_error("rfft_seg: nlon=%d has no specialised kernel (mk_fft_seg_supported)", nlon);
        return MK_EUNSUP;
    }
    return rc;
}

extern "C" int mk_irfft_rows_seg(const float* F, void* x, int x_dtype, const float* twiddle, int C, int nlat, int nlon, int mmax,
                                 float w_dc, float w_pos, float w_nyq, const MkFftSeg* seg, void* stream) {
    MK_REQUIRE(x && F && twiddle && seg, "irfft_seg: null pointer");
    MK_REQUIRE(mmax >= 1 && mmax <= nlon / 2 + 1, "irfft_seg: mmax=%d out of range for nlon=%d", mmax, nlon);
    MK_REQUIRE(x_dtype == MK_F32 || x_dtype == MK_BF16, "irfft_seg: bad dtype %d", x_dtype);
    const int rc = mk_fft_fast_dispatch(true, F, x, x_dtype, twiddle, 1, C, (C + 3) / 4 * 4, nlat, nlon, mmax, w_dc, w_pos, w_nyq, seg, stream);
    if (rc == -1000) {
        mk_set_error("irfft_seg: nlon=%d has no specialised kernel (mk_fft_seg_supported)", nlon);
        return MK_EUNSUP;
    }
    return rc;
}
