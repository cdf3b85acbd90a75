// Specialised longitude FFTs for the row lengths of the BASELINE grids (nlon = 1440, 480, ...).
//
// Same contract as the generic kernels in fft.hip, but everything that decides speed is a
// compile-time constant:
//   * the complex length N2 = nlon/2 is factored into 2-3 LARGE radices (720 = 10*9*8,
//     240 = 10*6*4) whose butterflies run entirely in registers (Dft<R>, fft_common.h), so a row
//     crosses LDS only between passes (in place: read -> barrier -> write);
//   * the first pass of the forward transform reads its operands straight from global memory and
//     the last pass of the inverse transform writes rows straight to global memory;
//   * index arithmetic is by constants, loops are fully unrolled;
//   * persistent workgroups own a CONTIGUOUS range of (plane, latitude-group) items, so the
//     RB-float runs they write to / read from the lat-major F-layout are adjacent in time and
//     address (they merge in the XCD's L2), and the twiddle table is staged into LDS once.
#include "fft_common.h"

namespace {

constexpr int NT = 256;

template <int ITEMS>
struct Rounds {
    static constexpr int value = (ITEMS + NT - 1) / NT;
};

// One in-place Stockham pass of radix R (Ns = product of earlier radices) over RB rows, split into
// its load half and its twiddle + butterfly + store half so that the loads of the NEXT work item can
// be issued early (software prefetch) and so that load and store may alias the same LDS buffer
// (all loads of a thread happen before the barrier, all stores after it).
template <int N2, int R, int RB>
struct PassShape {
    static constexpr int NB = N2 / R;
    static constexpr int ITEMS = RB * NB;
    static constexpr int NR = Rounds<ITEMS>::value;
};

template <int N2, int R, int RB>
using PassRegs = float2[PassShape<N2, R, RB>::NR][R];

template <int N2, int R, int RB, typename LoadFn>
__device__ __forceinline__ void pass_load(PassRegs<N2, R, RB>& v, LoadFn load, int tid) {
    using S = PassShape<N2, R, RB>;
#pragma unroll
    for (int q = 0; q < S::NR; ++q) {
        const int idx = tid + q * NT;
        if (idx < S::ITEMS) {
            const int row = idx / S::NB, j = idx % S::NB;
#pragma unroll
            for (int r = 0; r < R; ++r) v[q][r] = load(row, j + r * S::NB);
        }
    }
}

template <int N2, int R, int NS, int RB, typename StoreFn>
__device__ __forceinline__ void pass_compute_store(PassRegs<N2, R, RB>& v, const float2* __restrict__ tw,
                                                   StoreFn store, int tid) {
    using S = PassShape<N2, R, RB>;
    constexpr int TSTEP = 2 * N2 / (NS * R);
#pragma unroll
    for (int q = 0; q < S::NR; ++q) {
        const int idx = tid + q * NT;
        if (idx < S::ITEMS) {
            const int row = idx / S::NB, j = idx % S::NB;
            const int k = j % NS;
            if (NS > 1) {
#pragma unroll
                for (int r = 1; r < R; ++r) v[q][r] = cmul(v[q][r], tw[k * r * TSTEP]);
            }
            Dft<R>::run(v[q]);
            const int j0 = (j - k) * R + k;
#pragma unroll
            for (int o = 0; o < R; ++o) store(row, j0 + o * NS, v[q][Dft<R>::loc(o)]);
        }
    }
}

template <int N2, int R, int NS, int RB, bool SYNC_BETWEEN, typename LoadFn, typename StoreFn>
__device__ __forceinline__ void fft_pass(const float2* __restrict__ tw, LoadFn load, StoreFn store, int tid) {
    float2 v[PassShape<N2, R, RB>::NR][R];
    pass_load<N2, R, RB>(v, load, tid);
    if (SYNC_BETWEEN) __syncthreads();
    pass_compute_store<N2, R, NS, RB>(v, tw, store, tid);
}

struct ItemRange {
    long long begin, end;
};
__device__ __forceinline__ ItemRange my_items(long long nitems) {
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const long long per = (nitems + gridDim.x - 1) / gridDim.x;
    ItemRange r;
    r.begin = (long long)vb * per;
    r.end = min(nitems, r.begin + per);
    return r;
}

// ------------------------------------------------------------------------------------------
template <int N2, int R1, int R2, int R3, int RB, typename T>
__global__ __launch_bounds__(NT) void rfft_fast_kernel(const T* __restrict__ x, float* __restrict__ F,
                                                       const float2* __restrict__ tw_g, int C, int Cp, long long rows,
                                                       long long planes, int nlat, int mmax, int ngr, long long nitems,
                                                       float w_dc, float w_pos, float w_nyq) {
    static_assert(R1 * R2 * R3 == N2, "radix product");
    constexpr int N = 2 * N2, LS = N2 + 1;
    __shared__ __attribute__((aligned(16))) float2 smem[RB * LS + N];
    float2* buf = smem;
    float2* tw = smem + RB * LS;
    const int tid = threadIdx.x;
    for (int q = tid; q < N; q += NT) tw[q] = tw_g[q];
    __syncthreads();

    const ItemRange it = my_items(nitems);
    auto st_lds = [&](int row, int pos, float2 val) { buf[row * LS + pos] = val; };
    auto ld_lds = [&](int row, int pos) -> float2 { return buf[row * LS + pos]; };
    // rows of work item `itm` straight from global memory (first pass operands)
    float2 v1[PassShape<N2, R1, RB>::NR][R1];
    // work item = one latitude x RB consecutive (batch, channel) planes (k-major F^T layout, see fft.hip)
    const long long rstride = (long long)nlat * N;              // distance between the rows of an item
    auto prefetch = [&](long long itm) {
        const long long kl_ = itm / ngr;
        const long long p0_ = (itm - kl_ * ngr) * RB;
        const int nr_ = (int)min((long long)RB, planes - p0_);
        const T* xr_ = x + (p0_ * nlat + kl_) * (long long)N;
        pass_load<N2, R1, RB>(v1, [&](int row, int pos) -> float2 {
            return row < nr_ ? load_pair<T>(xr_ + (long long)row * rstride + 2 * pos) : make_float2(0.f, 0.f);
        }, tid);
    };
    if (it.begin < it.end) prefetch(it.begin);
    for (long long item = it.begin; item < it.end; ++item) {
        const long long klat = item / ngr;
        const long long p0 = (item - klat * ngr) * RB;
        const int nr = (int)min((long long)RB, planes - p0);

        pass_compute_store<N2, R1, 1, RB>(v1, tw, st_lds, tid);
        __syncthreads();
        if (item + 1 < it.end) prefetch(item + 1);       // in flight during passes 2, 3 and the untangle step
        fft_pass<N2, R2, R1, RB, true>(tw, ld_lds, st_lds, tid);
        __syncthreads();
        if constexpr (R3 > 1) {
            fft_pass<N2, R3, R1 * R2, RB, true>(tw, ld_lds, st_lds, tid);
            __syncthreads();
        }

        // Hermitian untangle + truncation + weights; lat-major stores (RB consecutive floats)
        for (int idx = tid; idx < mmax * RB; idx += NT) {
            const int r = idx % RB, m = idx / RB;
            if (r >= nr) continue;
            const int ma = (m == N2) ? 0 : m;
            const int mb = (m == 0 || m == N2) ? 0 : N2 - m;
            const float2 A = buf[r * LS + ma];
            const float2 Bc = cconj(buf[r * LS + mb]);
            const float2 u = cadd(A, Bc), t = csub(A, Bc);
            const float2 wt = cmul(tw[m], t);
            float w = w_pos;
            float2 X = make_float2(0.5f * (u.x + wt.y), 0.5f * (u.y - wt.x));
            if (m == 0) {
                w = w_dc;
                X.y = 0.f;
            } else if (m == N2) {
                w = w_nyq;
                X.y = 0.f;
            }
            const long long pr = p0 + r;
            float* o = F + ((long long)m * nlat + klat) * 2 * rows + (pr / C) * Cp + (pr % C);
            o[0] = w * X.x;
            o[rows] = w * X.y;
        }
        __syncthreads();
    }
}

template <int N2, int R1, int R2, int R3, int RB, typename T>
__global__ __launch_bounds__(NT) void irfft_fast_kernel(const float* __restrict__ F, T* __restrict__ x,
                                                        const float2* __restrict__ tw_g, int C, int Cp, long long rows,
                                                        long long planes, int nlat, int mmax, int ngr, long long nitems,
                                                        float w_dc, float w_pos, float w_nyq) {
    static_assert(R1 * R2 * R3 == N2, "radix product");
    constexpr int N = 2 * N2, LS = N2 + 1;
    __shared__ __attribute__((aligned(16))) float2 smem[RB * LS + N];
    float2* buf = smem;
    float2* tw = smem + RB * LS;
    const int tid = threadIdx.x;
    for (int q = tid; q < N; q += NT) tw[q] = tw_g[q];
    __syncthreads();

    const ItemRange it = my_items(nitems);
    const long long rstride = (long long)nlat * N;
    // weighted half spectrum X'[m], m = 0..N2 (zero beyond mmax) of work item `itm`, prefetched into registers
    constexpr int NSPEC = ((N2 + 1) * RB + NT - 1) / NT;
    // register prefetch of the next item's spectrum only where it fits without squeezing the pass
    // registers (it costs 2*NSPEC VGPRs that stay live across the passes); measured: 480 +36 %, 1440 -14 %
    constexpr bool PF = NSPEC <= 16;
    float2 spec[NSPEC];
    auto prefetch = [&](long long itm) {
        const long long kl_ = itm / ngr;
        const long long p0_ = (itm - kl_ * ngr) * RB;
        const int nr_ = (int)min((long long)RB, planes - p0_);
#pragma unroll
        for (int q = 0; q < NSPEC; ++q) {
            const int idx = tid + q * NT;
            const int r = idx % RB, m = idx / RB;
            float2 X = make_float2(0.f, 0.f);
            if (m < mmax && r < nr_) {
                const long long pr = p0_ + r;
                const float* sp = F + ((long long)m * nlat + kl_) * 2 * rows + (pr / C) * Cp + (pr % C);
                X = make_float2(sp[0], sp[rows]);
            }
            spec[q] = X;
        }
    };
    if (PF && it.begin < it.end) prefetch(it.begin);
    for (long long item = it.begin; item < it.end; ++item) {
        const long long klat = item / ngr;
        const long long p0 = (item - klat * ngr) * RB;
        const int nr = (int)min((long long)RB, planes - p0);
        T* xr = x + (p0 * nlat + klat) * (long long)N;
        if (!PF) prefetch(item);

#pragma unroll
        for (int q = 0; q < NSPEC; ++q) {
            const int idx = tid + q * NT;
            const int r = idx % RB, m = idx / RB;
            if (m <= N2) {
                float2 X = spec[q];
                if (m == 0)
                    X = make_float2(w_dc * X.x, 0.f);
                else if (m == N2)
                    X = make_float2(w_nyq * X.x, 0.f);
                else
                    X = make_float2(0.5f * w_pos * X.x, 0.5f * w_pos * X.y);
                buf[r * LS + m] = X;
            }
        }
        __syncthreads();
        if (PF && item + 1 < it.end) prefetch(item + 1);       // in flight during the pre-twiddle and the passes

        // in-place pre-twiddle on the pairs (j, N2-j): Zs[j] = (A + Bc) + i conj(W^j)(A - Bc); store conj(Zs)
        for (int idx = tid; idx < RB * (N2 / 2 + 1); idx += NT) {
            const int row = idx / (N2 / 2 + 1), j = idx % (N2 / 2 + 1);
            const int j2 = N2 - j;                         // partner (j = 0 pairs with N2, j = N2/2 with itself)
            const float2 Xa = buf[row * LS + j], Xb = buf[row * LS + j2];
            {
                const float2 u = cadd(Xa, cconj(Xb)), t = csub(Xa, cconj(Xb));
                const float2 wt = cmul(cconj(tw[j]), t);
                buf[row * LS + j] = make_float2(u.x - wt.y, -(u.y + wt.x));
            }
            if (j != 0 && j2 != j) {
                const float2 u = cadd(Xb, cconj(Xa)), t = csub(Xb, cconj(Xa));
                const float2 wt = cmul(cconj(tw[j2]), t);
                buf[row * LS + j2] = make_float2(u.x - wt.y, -(u.y + wt.x));
            }
        }
        __syncthreads();

        auto ld_lds = [&](int row, int pos) -> float2 { return buf[row * LS + pos]; };
        auto st_lds = [&](int row, int pos, float2 val) { buf[row * LS + pos] = val; };
        auto st_global = [&](int row, int pos, float2 val) {
            if (row < nr) store_pair<T>(xr + (long long)row * rstride + 2 * pos, val.x, -val.y);     // conj
        };

        fft_pass<N2, R1, 1, RB, true>(tw, ld_lds, st_lds, tid);
        __syncthreads();
        if constexpr (R3 > 1) {
            fft_pass<N2, R2, R1, RB, true>(tw, ld_lds, st_lds, tid);
            __syncthreads();
            fft_pass<N2, R3, R1 * R2, RB, false>(tw, ld_lds, st_global, tid);
        } else {
            fft_pass<N2, R2, R1, RB, false>(tw, ld_lds, st_global, tid);
        }
        __syncthreads();
    }
}

template <int N2, int R1, int R2, int R3, int RB>
int launch(bool inverse, const void* in, void* out, int dtype, const float2* tw, int B, int C, int Cp, int nlat, int mmax,
           float w_dc, float w_pos, float w_nyq, hipStream_t s) {
    const long long planes = (long long)B * C;
    const int ngr = (int)((planes + RB - 1) / RB);
    const long long nitems = (long long)nlat * ngr;
    const long long rows = (long long)B * Cp;
    // persistent grid: a few workgroups per CU, each owning a contiguous item range
    constexpr size_t lds = (size_t)(RB * (N2 + 1) + 2 * N2) * 8;
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 6) per_cu = 6;
    if (per_cu < 1) per_cu = 1;
    long long grid = 256ll * per_cu;
    if (grid > nitems) grid = nitems;
    dim3 g((unsigned)grid), b(NT);
    if (!inverse) {
        if (dtype == MK_F32)
            hipLaunchKernelGGL((rfft_fast_kernel<N2, R1, R2, R3, RB, float>), g, b, 0, s, (const float*)in, (float*)out, tw, C,
                               Cp, rows, planes, nlat, mmax, ngr, nitems, w_dc, w_pos, w_nyq);
        else
            hipLaunchKernelGGL((rfft_fast_kernel<N2, R1, R2, R3, RB, u16>), g, b, 0, s, (const u16*)in, (float*)out, tw, C,
                               Cp, rows, planes, nlat, mmax, ngr, nitems, w_dc, w_pos, w_nyq);
    } else {
        if (dtype == MK_F32)
            hipLaunchKernelGGL((irfft_fast_kernel<N2, R1, R2, R3, RB, float>), g, b, 0, s, (const float*)in, (float*)out, tw,
                               C, Cp, rows, planes, nlat, mmax, ngr, nitems, w_dc, w_pos, w_nyq);
        else
            hipLaunchKernelGGL((irfft_fast_kernel<N2, R1, R2, R3, RB, u16>), g, b, 0, s, (const float*)in, (u16*)out, tw, C,
                               Cp, rows, planes, nlat, mmax, ngr, nitems, w_dc, w_pos, w_nyq);
    }
    return mk_check_launch(inverse ? "mk_irfft_rows(fast)" : "mk_rfft_rows(fast)");
}

}  // namespace

// returns -1000 if nlon has no specialised kernel (caller falls through to the generic one)
int mk_fft_fast_dispatch(bool inverse, const void* in, void* out, int dtype, const float* twiddle, int B, int C, int Cp,
                         int nlat, int nlon, int mmax, float w_dc, float w_pos, float w_nyq, void* stream) {
    const float2* tw = reinterpret_cast<const float2*>(twiddle);
    hipStream_t s = (hipStream_t)stream;
    switch (nlon) {
        case 1440: return launch<720, 10, 9, 8, 8>(inverse, in, out, dtype, tw, B, C, Cp, nlat, mmax, w_dc, w_pos, w_nyq, s);
        case 480: return launch<240, 10, 6, 4, 16>(inverse, in, out, dtype, tw, B, C, Cp, nlat, mmax, w_dc, w_pos, w_nyq, s);
        case 360: return launch<180, 6, 6, 5, 16>(inverse, in, out, dtype, tw, B, C, Cp, nlat, mmax, w_dc, w_pos, w_nyq, s);
        case 128: return launch<64, 4, 4, 4, 16>(inverse, in, out, dtype, tw, B, C, Cp, nlat, mmax, w_dc, w_pos, w_nyq, s);
        case 72: return launch<36, 6, 6, 1, 16>(inverse, in, out, dtype, tw, B, C, Cp, nlat, mmax, w_dc, w_pos, w_nyq, s);
        default: return -1000;
    }
}
