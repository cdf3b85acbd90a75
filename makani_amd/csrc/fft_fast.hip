// Specialised longitude FFTs for the row lengths of the BASELINE grids (nlon = 1440, 480, ...).
//
// Same contract as the generic kernels in fft.hip, but everything that decides speed is a
// compile-time constant:
//   * the complex length N2 = nlon/2 is factored into 2-3 LARGE radices (720 = 30*24,
//     240 = 10*6*4) whose butterflies run entirely in registers (PDft<R>, fft_packed.h), so a row
//     crosses LDS only between passes (in place: read -> barrier -> write);
//   * the last pass of the inverse transform writes rows straight to global memory;
//   * index arithmetic is by constants, loops are fully unrolled;
//   * persistent workgroups own a CONTIGUOUS range of (plane, latitude-group) items, so the
//     RB-float runs they write to / read from the lat-major F-layout are adjacent in time and
//     address (they merge in the XCD's L2), and the twiddle table is staged into LDS once;
//   * complex values are register PAIRS worked on by the packed fp32 instructions (fft_packed.h): the kernels are bound
//     by VALU issue, and the packed butterflies need about half the instructions of the struct-of-two-floats form.
#include <stdlib.h>

#include "fft_packed.h"

// Build-time plan knobs (A/B builds: tools/ab.py; what each alternative measured is in docs/LAB_NOTEBOOK.md and
// docs/DESIGN_rounds1-4.md section 4):
//   MK_FFT_1440_2PASS  1440 points as two passes 30 x 24 (default) instead of three 10 x 9 x 8
//   MK_FFT_480_FWD_OCC the forward 480-point bf16 kernel limited to 128 registers = two workgroups per CU (the only
//                      instantiation that does not spill there)
//   MK_FFT_HV          two half-workgroups of 256 threads with their own LDS work buffers, half 1 running MK_FFT_SKEW barrier
//                      intervals behind half 0: 0 = never, 1 (default) = only the forward 480-point bf16 kernel (the one
//                      instantiation where it measured faster), 2 = every 512-thread instantiation.  Bit-identical results.
#ifndef MK_FFT_1440_2PASS
#define MK_FFT_1440_2PASS 1
#endif
#ifndef MK_FFT_480_FWD_OCC
#define MK_FFT_480_FWD_OCC 1
#endif
#ifndef MK_FFT_480_INV_OCC      // the inverse 480-point kernels held to 128 registers = two workgroups per CU (A/B: round 6)
#define MK_FFT_480_INV_OCC 0
#endif
#ifndef MK_FFT_HV
#define MK_FFT_HV 1
#endif
#ifndef MK_FFT_SKEW
#define MK_FFT_SKEW 3
#endif

// MK_FFT_DIAG=1 (tools/ab_fast.sh fft_fast diag:-DMK_FFT_DIAG=1, tools/fft_diag.py): s_memtime stamps at the phase boundaries of the
// forward kernel, summed over all waves of a launch: where a wave's cycles go (every stamp waits for the wave's own LDS traffic first)
#ifndef MK_FFT_DIAG
#define MK_FFT_DIAG 0
#endif

namespace {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

#if MK_FFT_DIAG
// 0 prefetch issue, 1 pass loads (LDS reads landed), 2 barriers, 3 twiddle + butterflies + LDS stores, 4 untangle (LDS reads, arithmetic,
// global store issue), 5 commit (wait for the prefetched row vectors, convert, LDS stores), 6 whole kernel, 7 waves
__device__ unsigned long long g_fft_diag[8];
#define MK_FFT_STAMP(k)                                             \
    do {                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          \
        const unsigned long long t_ = __builtin_readcyclecounter(); \
        dg[k] += t_ - tprev;                                        \
        tprev = t_;                                                 \
    } while (0)
#else
#define MK_FFT_STAMP(k)
#endif

template <int HV>
__device__ __forceinline__ void skew_barriers(bool mine) {
    if constexpr (HV > 1) {
        if (mine) {                     // (wave-uniform scalar condition: a real branch around the barriers)
#pragma unroll
            for (int q = 0; q < MK_FFT_SKEW; ++q) __syncthreads();
        }
    }
}

template <int ITEMS, int NT>
struct Rounds {
    static constexpr int value = (ITEMS + NT - 1) / NT;
};

// ---- segmented addressing (distributed transforms, makani_amd/distributed.py) -------------------------------------------
// SEG kernels address both sides of the transform through an MkFftSeg descriptor (include/makani_amd.h) so that the
// all-to-all exchanges around the transform need no pack / unpack pass:
//   * F side: per-peer slabs.  Slab (jw, ih) holds the orders m in [m_off[jw], m_off[jw+1]) and the rows (planes) in
//     [r_off[ih], r_off[ih+1]) of every latitude of this call, LATITUDE OUTERMOST: [lat][m][re/im][row] at F + base[jw][ih].
//     The forward transform writes the send buffers of the (h x w) exchange this way (one contiguous slab per destination
//     rank, and — latitude being outermost — the slabs of the ranks of one polar group concatenate into the Legendre
//     operand by landing next to each other); the inverse transform reads its receive buffers the same way.
//   * x side: a row (plane, latitude) of nlon points cut into `xseg` equal pieces, piece j of all rows at x + j * x_stride
//     (the pieces the planes <-> longitude exchange delivers / collects, one buffer per azimuth peer).
struct SegTab {                 // LDS copy of the F-side tables
    long long base[MK_FFT_SEG_MAX * MK_FFT_SEG_MAX];
    int m_off[MK_FFT_SEG_MAX + 1];
    int r_off[MK_FFT_SEG_MAX + 1];
};

template <int NT>
__device__ __forceinline__ void seg_fill(SegTab* t, const MkFftSeg& sg, int tid) {
    for (int q = tid; q < MK_FFT_SEG_MAX * MK_FFT_SEG_MAX; q += NT) t->base[q] = sg.base[q / MK_FFT_SEG_MAX][q % MK_FFT_SEG_MAX];
    for (int q = tid; q <= MK_FFT_SEG_MAX; q += NT) {
        t->m_off[q] = sg.m_off[q < sg.nw ? q : sg.nw];
        t->r_off[q] = sg.r_off[q < sg.nh ? q : sg.nh];
    }
}

__device__ __forceinline__ int seg_find(const int* off, int n, int v) {
    int j = 0;
#pragma unroll
    for (int k = 1; k < MK_FFT_SEG_MAX; ++k) j += (k < n && v >= off[k]) ? 1 : 0;
    return j;
}

// float offset of (m, latitude klat, re plane, row pr) from F, and the distance to the im plane
__device__ __forceinline__ long long seg_f_offset(const SegTab* t, int nw, int nh, int m, long long klat, long long pr, int* im_stride) {
    const int jw = seg_find(t->m_off, nw, m), ih = seg_find(t->r_off, nh, (int)pr);
    const int Mj = t->m_off[jw + 1] - t->m_off[jw], Pi = t->r_off[ih + 1] - t->r_off[ih];
    *im_stride = Pi;
    return t->base[jw * MK_FFT_SEG_MAX + ih] + ((klat * Mj + (m - t->m_off[jw])) * 2) * Pi + (pr - t->r_off[ih]);
}

// LDS address map of one "generation" of the work buffer (the rows as one pass leaves them and the next one reads them):
// position p of row r lives at r * LS + p + (p / BS) * D (complex values).  All loads of a pass happen before the barrier and
// all its stores after it, so every generation may have its own row stride and padding: what matters is that the 16 / 32 lanes
// the LDS serves in one cycle (MI355X_MICROARCH.md, LDS: ds_read_b64 = 2 x 32 lanes over 64 banks, ds_write_b64 = 4 x 16 lanes
// over 32 banks) touch different banks.  The first pass stores with a stride of R1 values between lanes — an even stride puts
// lanes l and l + 8 on one bank — so the generation it writes is padded to blocks of R1 + 1 (BS = R1, D = 1), which the next
// pass then reads as consecutive values; see LdsPlan below and tools/fft_lds_model.py (bank-conflict model of every phase).
template <int LS_, int BS_, int D_>
struct Gen {
    static constexpr int LS = LS_, BS = BS_, D = D_;
    __device__ static __forceinline__ int pad(int pos) {
        if constexpr (D_ == 0) return pos; else return pos + (pos / BS_) * D_;
    }
    template <int STEP>
    static constexpr int step() {                       // address step of a position step that is a multiple of the block
        if constexpr (D_ == 0) return STEP; else { static_assert(STEP % BS_ == 0, "step must cover whole blocks"); return STEP + (STEP / BS_) * D_; }
    }
};

// One in-place Stockham pass of radix R (Ns = product of earlier radices) over RB rows, split into
// its load half and its twiddle + butterfly + store half so that the loads of the NEXT work item can
// be issued early (software prefetch) and so that load and store may alias the same LDS buffer
// (all loads of a thread happen before the barrier, all stores after it).
// LPR: lanes per row.  0 = the N2 / R butterflies of the rows are dealt over consecutive threads; 32 / 64 = every row starts
// on a lane group of its own (lanes past N2 / R idle), so that no group of lanes the LDS serves together spans two rows.
template <int N2, int R, int RB, int NT, int LPR = 0>
struct PassShape {
    static constexpr int NB = N2 / R;
    static constexpr int LANES = LPR ? LPR : NB;
    static_assert(LANES >= NB, "lanes per row");
    static constexpr int ITEMS = RB * LANES;
    static constexpr int NR = Rounds<ITEMS, NT>::value;
};

template <int N2, int R, int RB, int NT, int LPR = 0>
using PassRegs = cf[PassShape<N2, R, RB, NT, LPR>::NR][R];

// load(row, position, LDS address)
template <int N2, int R, int RB, int NT, int LPR, typename GIN, typename LoadFn>
__device__ __forceinline__ void pass_load(PassRegs<N2, R, RB, NT, LPR>& v, LoadFn load, int tid) {
    using S = PassShape<N2, R, RB, NT, LPR>;
#pragma unroll
    for (int q = 0; q < S::NR; ++q) {
        const int idx = tid + q * NT;
        if (idx < S::ITEMS) {
            const int row = idx / S::LANES, j = idx % S::LANES;
            if (LPR == 0 || j < S::NB) {
                const int base = row * GIN::LS + GIN::pad(j);
#pragma unroll
                for (int r = 0; r < R; ++r) v[q][r] = load(row, j + r * S::NB, base + r * GIN::template step<S::NB>());
            }
        }
    }
}

// tp: this pass's twiddles, tp[(r-1)*NS + k] = exp(-2 pi i k r / (NS R)): consecutive lanes (k) read
// consecutive LDS words.  store(row, position, LDS address, value)
template <int N2, int R, int NS, int RB, int NT, int LPR, typename GOUT, typename StoreFn>
__device__ __forceinline__ void pass_compute_store(PassRegs<N2, R, RB, NT, LPR>& v, const cf* __restrict__ tp,
                                                   StoreFn store, int tid) {
    using S = PassShape<N2, R, RB, NT, LPR>;
    static_assert(GOUT::D == 0 || GOUT::BS == NS * R, "a padded generation is padded per output block of the pass that writes it");
#pragma unroll
    for (int q = 0; q < S::NR; ++q) {
        const int idx = tid + q * NT;
        if (idx < S::ITEMS) {
            const int row = idx / S::LANES, j = idx % S::LANES;
            if (LPR == 0 || j < S::NB) {
                const int k = j % NS;
                if (NS > 1) {
#pragma unroll
                    for (int r = 1; r < R; ++r) v[q][r] = cmul(v[q][r], tp[(r - 1) * NS + k]);
                }
                PDft<R>::run(v[q]);
                const int j0 = (j - k) * R + k;
                const int base = row * GOUT::LS + (j / NS) * (NS * R + GOUT::D) + k;      // = row * LS + pad(j0)
#pragma unroll
                for (int o = 0; o < R; ++o) store(row, j0 + o * NS, base + o * NS, v[q][PDft<R>::loc(o)]);
            }
        }
    }
}

template <int N2, int R, int NS, int RB, int NT, bool SYNC_BETWEEN, int LPR, typename GIN, typename GOUT, typename LoadFn,
          typename StoreFn>
__device__ __forceinline__ void fft_pass(const cf* __restrict__ tp, LoadFn load, StoreFn store, int tid) {
    cf v[PassShape<N2, R, RB, NT, LPR>::NR][R];
    pass_load<N2, R, RB, NT, LPR, GIN>(v, load, tid);
    if (SYNC_BETWEEN) __syncthreads();
    pass_compute_store<N2, R, NS, RB, NT, LPR, GOUT>(v, tp, store, tid);
}
#if MK_FFT_DIAG
// the same pass with the diagnostic stamps of the forward kernel
template <int N2, int R, int NS, int RB, int NT, int LPR, typename GIN, typename GOUT, typename LoadFn, typename StoreFn>
__device__ __forceinline__ void fft_pass_diag(const cf* __restrict__ tp, LoadFn load, StoreFn store, int tid, unsigned long long* dg,
                                              unsigned long long& tprev) {
    cf v[PassShape<N2, R, RB, NT, LPR>::NR][R];
    pass_load<N2, R, RB, NT, LPR, GIN>(v, load, tid);
    MK_FFT_STAMP(1);
    __syncthreads();
    MK_FFT_STAMP(2);
    pass_compute_store<N2, R, NS, RB, NT, LPR, GOUT>(v, tp, store, tid);
    MK_FFT_STAMP(3);
}
#endif

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

// LDS row stride (in complex values).  The step that moves between the [row][m] LDS image and the
// [m][row] global image touches LDS with the row index fastest across lanes, so the stride decides its
// bank pattern (MI355X_MICROARCH.md, LDS): ds_read_b64 works on 32-lane groups over 64 banks, ds_write_b64
// on 16-lane groups over 32 banks.
//   forward (reads):  RB = 8 -> stride = 4 mod 8;  RB = 16 -> stride = 2 mod 4
//   inverse (writes): RB = 8 -> stride = 2 mod 4;  RB = 16 -> stride odd
__host__ __device__ constexpr int row_stride(int n2, int rb, bool inverse) {
    int ls = n2 + 1;
    if (!inverse) {
        if (rb <= 8) { while (ls % 8 != 4) ++ls; } else { while (ls % 4 != 2) ++ls; }
    } else {
        if (rb <= 8) { while (ls % 4 != 2) ++ls; } else { while (ls % 2 != 1) ++ls; }
    }
    return ls;
}

// Row strides, padding and lanes per row of the generations of one kernel (g0 = what the first pass reads, g1 / g2 = what
// passes 1 / 2 leave, the last generation of the forward kernel = what the untangle step reads).  Default: one layout for all
// (row_stride above, chosen for the [row][m] <-> [m][row] steps).  The plans of the benchmark's kernels come out of the bank
// model (tools/fft_lds_model.py; tests/test_lds_layouts.py).  MK_FFT_LDSPLAN=0: the default layout everywhere (the A/B build).
#ifndef MK_FFT_LDSPLAN
#define MK_FFT_LDSPLAN 1
#endif
template <int N2, int R1, int R2, int R3, int RBH, int NTH, bool INV, bool ON = (MK_FFT_LDSPLAN != 0)>
struct LdsPlan {
    static constexpr int LS0 = row_stride(N2, RBH, INV), LS1 = LS0, LS2 = LS0, LS3 = LS0;
    static constexpr int D1 = 0, D2 = 0, LPR1 = 0, LPR2 = 0, LPR3 = 0;
    static constexpr bool SWAP = false;     // forward, bf16 rows: the two 16-byte halves of a vector stored in lane-dependent order
};
// SWAP: the two 16-byte halves of a bf16 vector stored in lane-dependent order by the commit (removes its 2-way bank conflict).
// OFF: its selects cost more vector instructions than the conflict costs cycles.  MK_FFT_SWAP=1 is the A/B build.
#ifndef MK_FFT_SWAP
#define MK_FFT_SWAP 0
#endif
template <>
struct LdsPlan<720, 30, 24, 1, 16, 512, false, true> {
    static constexpr int LS0 = 728, LS1 = 744, LS2 = 722, LS3 = 722, D1 = 1, D2 = 0, LPR1 = 0, LPR2 = 32, LPR3 = 0;
    static constexpr bool SWAP = MK_FFT_SWAP != 0;
};
template <>
struct LdsPlan<720, 30, 24, 1, 16, 512, true, true> {
    static constexpr int LS0 = 721, LS1 = 744, LS2 = 721, LS3 = 721, D1 = 1, D2 = 0, LPR1 = 0, LPR2 = 32, LPR3 = 0;
    static constexpr bool SWAP = false;
};
template <>
struct LdsPlan<240, 10, 6, 4, 16, 256, false, true> {      // (128 registers: 64 lanes per row in pass 3 would spill; 1 728 with it)
    static constexpr int LS0 = 248, LS1 = 264, LS2 = 296, LS3 = 242, D1 = 1, D2 = 14, LPR1 = 0, LPR2 = 0, LPR3 = 0;
    static constexpr bool SWAP = MK_FFT_SWAP != 0;
};
template <>
struct LdsPlan<240, 10, 6, 4, 32, 512, true, true> {
    static constexpr int LS0 = 249, LS1 = 264, LS2 = 296, LS3 = 249, D1 = 1, D2 = 14, LPR1 = 0, LPR2 = 0, LPR3 = 64;
    static constexpr bool SWAP = false;
};
template <typename PL>
constexpr int plan_max_stride() {
    int m = PL::LS0;
    if (PL::LS1 > m) m = PL::LS1;
    if (PL::LS2 > m) m = PL::LS2;
    if (PL::LS3 > m) m = PL::LS3;
    return m;
}

// twiddle tables in LDS: per-pass tables (see pass_compute_store) + U[m] = exp(-2 pi i m / N), m < UN
template <int N2, int R1, int R2, int R3, int UN>
struct Tables {
    static constexpr int T2 = (R2 - 1) * R1;
    static constexpr int T3 = (R3 > 1) ? (R3 - 1) * R1 * R2 : 0;
    static constexpr int U = UN;
    static constexpr int SIZE = T2 + T3 + U;
    template <int NT>
    __device__ static __forceinline__ void fill(cf* t, const cf* __restrict__ tw_g, int tid) {
        constexpr int N = 2 * N2;
        for (int q = tid; q < T2; q += NT) {
            const int r = q / R1 + 1, k = q % R1;
            t[q] = tw_g[k * r * (N / (R1 * R2))];
        }
        for (int q = tid; q < T3; q += NT) {
            const int r = q / (R1 * R2) + 1, k = q % (R1 * R2);
            t[T2 + q] = tw_g[k * r * (N / (R1 * R2 * (R3 > 1 ? R3 : 1)))];
        }
        for (int q = tid; q < U; q += NT) t[T2 + T3 + q] = tw_g[q];
    }
};

// ------------------------------------------------------------------------------------------
// Global <-> LDS traffic of both kernels moves 16 bytes per lane:
//   * x rows are copied whole (contiguous 16-byte vectors, every cache line requested once) into the LDS
//     work buffer, where the first pass then runs in place like the others; the copy of the NEXT item is
//     in flight (in registers) during the passes of the current one;
//   * the F side is touched as float4 = 4 consecutive rows of one (m, latitude, re/im).
// X[m] of one real row from the N2-point complex FFT Z of its (even, odd) pairs, truncated spectrum weights applied
// per order m: the positions of the pair (Z[m], Z[N2 - m]), the twiddle U[m] and the weights of the two output components
template <int N2>
struct Untangle {
    int ma, mb;
    cf tw, hw;         // hw = (w / 2, edge ? 0 : w / 2): X = (u + (-i) U t) / 2, scaled by the truncated-spectrum weight w
    __device__ __forceinline__ Untangle(const cf* __restrict__ twu, int m, float w_dc, float w_pos, float w_nyq) {
        ma = (m == N2) ? 0 : m;
        mb = (m == 0 || m == N2) ? 0 : N2 - m;
        tw = twu[m];
        const bool edge = (m == 0) || (m == N2);
        const float w = 0.5f * ((m == 0) ? w_dc : ((m == N2) ? w_nyq : w_pos));
        hw = cf_make(w, edge ? 0.f : w);
    }
    // unscaled X[m] of one row
    __device__ __forceinline__ cf raw(const cf* __restrict__ zrow) const {
        const cf A = zrow[ma], B = zrow[mb];
        const cf u = add_conj(A, B), t = sub_conj(A, B);
        return add_mi(u, cmul(t, tw));
    }
};

// weights of the two components of X[m] on the way into the inverse transform
template <int N2>
__device__ __forceinline__ cf inverse_weights(int m, float w_dc, float w_pos, float w_nyq) {
    const bool edge = (m == 0) || (m == N2);
    const float w = (m == 0) ? w_dc : ((m == N2) ? w_nyq : 0.5f * w_pos);
    return cf_make(w, edge ? 0.f : w);
}

template <typename T>
struct RowVec;                                  // 16-byte vector of a row
template <>
struct RowVec<float> {
    static constexpr int PAIRS = 2;             // (re, im)-style pairs = float2 slots per vector
};
template <>
struct RowVec<u16> {
    static constexpr int PAIRS = 4;
};

// WGS: the second __launch_bounds__ argument, which in HIP is the minimum number of WAVES per SIMD (not workgroups per CU).
template <int N2, int R1, int R2, int R3, int RB, int NT, int WGS, int MCAP, typename T, bool SEG, int HV>
__global__ __launch_bounds__(NT, (MK_FFT_480_FWD_OCC && N2 == 240 && sizeof(T) == 2 && !SEG) ? 4 : WGS) void rfft_fast_kernel(const T* __restrict__ x, float* __restrict__ F,
                                                            const cf* __restrict__ tw_g, int C, int Cp,
                                                            long long rows, long long planes, int nlat, int mmax,
                                                            int ngr, long long nitems, float w_dc, float w_pos,
                                                            float w_nyq, const MkFftSeg sg) {
    static_assert(R1 * R2 * R3 == N2, "radix product");
    // HV halves of NTH threads, RBH rows each (see MK_FFT_HV above); below, RBH / NTH / tid are the half's
    constexpr int NTH = NT / HV, RBH = RB / HV;
    using PL = LdsPlan<N2, R1, R2, R3, RBH, NTH, false>;
    using G0 = Gen<PL::LS0, 0, 0>;                                        // the committed rows
    using G1 = Gen<PL::LS1, R1, PL::D1>;                                  // after pass 1
    using G2 = Gen<PL::LS2, R1 * R2, (R3 > 1) ? PL::D2 : 0>;              // after pass 2 (two-pass plans: the last generation)
    using G3 = Gen<PL::LS3, 0, 0>;                                        // after pass 3
    constexpr int LS0 = PL::LS0, LS = (R3 > 1) ? PL::LS3 : PL::LS2;       // LS: the generation the untangle step reads
    constexpr int LSM = plan_max_stride<PL>();
    constexpr int N = 2 * N2;
    constexpr int VP = RowVec<T>::PAIRS, VROW = N2 / VP;        // vectors per row
    static_assert(N2 % VP == 0 && LS0 % 2 == 0 && RBH % 4 == 0 && NT % HV == 0 && RB % HV == 0 && NTH % 64 == 0, "vector layout");
    using Tb = Tables<N2, R1, R2, R3, MCAP>;        // mmax <= MCAP
    __shared__ __attribute__((aligned(16))) cf smem[RB * LSM + Tb::SIZE];
    __shared__ SegTab segtab_s;
    SegTab* segtab = &segtab_s;
    const int half = HV > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x / NTH) : 0;
    const int rofs = half * RBH;                    // first row of this half inside the item
    cf* buf = smem + half * (RBH * LSM);
    cf* tw2 = smem + RB * LSM;
    cf* tw3 = tw2 + Tb::T2;
    cf* twu = tw3 + Tb::T3;
    const int tid = (int)threadIdx.x % NTH;
    Tb::template fill<NT>(tw2, tw_g, (int)threadIdx.x);
    if constexpr (SEG) seg_fill<NT>(segtab, sg, (int)threadIdx.x);

    const ItemRange it = my_items(nitems);
    auto st_lds = [&](int, int, int addr, cf val) { buf[addr] = val; };
    auto ld_lds = [&](int, int, int addr) -> cf { return buf[addr]; };
    // work item = one latitude x RB consecutive (batch, channel) planes (k-major F layout, see fft.hip)
    // SEG: the row of a plane is cut into sg.xseg equal pieces in separate buffers (see SegTab): per (lane, q) the piece and
    // the offset inside it are fixed, only the (plane, latitude) part moves with the item
    const int xseg = SEG ? max(1, sg.xseg) : 1;
    const int vpp = VROW / xseg;                                 // vectors per piece
    const long long wl = (long long)N / xseg;                    // points per piece
    const long long xnl = (SEG && sg.x_nlat > 0) ? sg.x_nlat : nlat;      // latitudes per plane in the x buffers
    const long long rstride = xnl * wl;                          // distance between the rows of an item (inside a piece)
    constexpr int NV = (RBH * VROW + NTH - 1) / NTH;
    uint4 rawv[NV];
    // Everything below that depends on the LANE only — which vector of which row it fetches, where it lands in LDS, which order m
    // and which four rows it untangles, the twiddle and weights of that order, where its two stores go — is computed ONCE here and
    // kept in registers (the kernel runs one workgroup per CU: 256 registers per lane are there).  Round 4's stamps put 35 % of a
    // wave's cycles into vector instructions; the ISA showed that two thirds of the non-butterfly ones re-derived these constants
    // (divisions by constants, 64-bit address arithmetic, selects) for every item.  Per item only the uniform base moves.
    // (SEG: the F-side offsets depend on the item through the slab tables; that path keeps the per-item arithmetic.)
    // (HOIST off: instantiations held to 168 / 128 registers for three / four waves per SIMD — the 720- and 360-point kernels, the
    //  forward 480-point bf16 kernel — would spill the constants; they keep the per-item arithmetic)
    constexpr bool HOIST = !SEG && WGS <= 2 && !(MK_FFT_480_FWD_OCC && N2 == 240 && sizeof(T) == 2);      // (WGS: waves per SIMD asked of the compiler)
    unsigned pf_off[NV];                                        // element offset of this lane's vector q inside an item's rows
    int pf_row[NV], cm_off[NV];                                 // its row (rows >= nr hold zeros) and its LDS offset (complex values)
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const int idx = tid + q * NTH;
        const int row = idx / VROW, c = idx % VROW;
        pf_row[q] = row;
        pf_off[q] = HOIST ? (unsigned)((long long)row * rstride + (long long)c * (2 * VP)) : 0u;
        cm_off[q] = row * LS0 + c * VP;
    }
    auto prefetch = [&](long long itm) {
        const long long kl_ = (unsigned)itm / (unsigned)ngr;             // (nitems < 2^31: 32-bit division)
        const long long p0_ = (itm - kl_ * ngr) * RB + rofs;
        const int nr_ = (int)max(0ll, min((long long)RBH, planes - p0_));
        const T* xr_ = x + (p0_ * xnl + kl_) * wl;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            if constexpr (!HOIST) {
                const int idx = tid + q * NTH;
                const int row = idx / VROW, c = idx % VROW;
                const long long coff = SEG ? (long long)(c / vpp) * sg.x_stride + (long long)(c % vpp) * (2 * VP) : (long long)c * (2 * VP);
                rawv[q] = (row < nr_) ? *reinterpret_cast<const uint4*>(xr_ + (long long)row * rstride + coff) : make_uint4(0, 0, 0, 0);
            } else {
                rawv[q] = (pf_row[q] < nr_) ? *reinterpret_cast<const uint4*>(xr_ + pf_off[q]) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto commit = [&]() {                                       // registers -> work buffer as fp32 pairs
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int idx = tid + q * NTH;
            const int c = idx % VROW;
            float4* d = reinterpret_cast<float4*>(buf + (HOIST ? cm_off[q] : (idx / VROW) * LS0 + c * VP));
            const uint4 u = rawv[q];
            if (idx < RBH * VROW) {
                if constexpr (sizeof(T) == 4) {
                    d[0] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
                } else {
                    const float4 lo = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                                                  __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
                    const float4 hi = make_float4(__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u),
                                                  __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u));
                    if constexpr (PL::SWAP) {
                        // a lane's vector is 32 bytes of LDS: eight lanes storing their FIRST halves cover 256 bytes = every bank
                        // twice.  Every second group of four lanes stores its second half first: 128 bytes per instruction
                        const bool sw = (c >> 2) & 1;
                        d[sw ? 1 : 0] = make_float4(sw ? hi.x : lo.x, sw ? hi.y : lo.y, sw ? hi.z : lo.z, sw ? hi.w : lo.w);
                        d[sw ? 0 : 1] = make_float4(sw ? lo.x : hi.x, sw ? lo.y : hi.y, sw ? lo.z : hi.z, sw ? lo.w : hi.w);
                    } else {
                        d[0] = lo;
                        d[1] = hi;
                    }
                }
            }
        }
    };
    // 4 consecutive planes map to 4 consecutive F rows unless a quad straddles a batch boundary
    const bool vec = (C % 4 == 0) || (planes == C);
    if (it.begin < it.end) {
        prefetch(it.begin);
        commit();
    }
    __syncthreads();
    // untangle constants of this lane (the tables in LDS are complete after the barrier above): trip q handles order un_m, rows
    // r0 .. r0 + 3 — LDS offsets of (Z[m], Z[N2 - m]) in row r0, twiddle, output weights, float offset of its first store
    constexpr int NQU = (MCAP * (RBH / 4) + NTH - 1) / NTH;
    int un_za[NQU], un_zb[NQU], un_r0[NQU];
    cf un_tw[NQU], un_hw[NQU];
    unsigned un_fo[NQU];
    if constexpr (HOIST) {
#pragma unroll
        for (int q = 0; q < NQU; ++q) {
            const int idx = tid + q * NTH;
            const int r0 = (idx % (RBH / 4)) * 4, m = idx / (RBH / 4);
            const Untangle<N2> un(twu, min(m, mmax - 1), w_dc, w_pos, w_nyq);
            un_za[q] = r0 * LS + un.ma;
            un_zb[q] = r0 * LS + un.mb;
            un_tw[q] = un.tw;
            un_hw[q] = un.hw;
            un_r0[q] = (m < mmax) ? r0 : (1 << 20);                              // (orders past mmax: never "r0 < nr")
            un_fo[q] = (unsigned)(((long long)min(m, mmax - 1) * nlat) * 2 * rows + r0);
        }
    }
    skew_barriers<HV>(half == 1);                         // half 1 runs MK_FFT_SKEW barrier intervals behind half 0
#if MK_FFT_DIAG
    unsigned long long dg[7] = {0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_readcyclecounter();
    const unsigned long long tstart = tprev;
#endif
    for (long long item = it.begin; item < it.end; ++item) {
        const long long klat = (unsigned)item / (unsigned)ngr;
        const long long p0 = (item - klat * ngr) * RB + rofs;
        const int nr = (int)max(0ll, min((long long)RBH, planes - p0));

#if MK_FFT_DIAG
        if (item + 1 < it.end) prefetch(item + 1);
        MK_FFT_STAMP(0);
        fft_pass_diag<N2, R1, 1, RBH, NTH, PL::LPR1, G0, G1>(tw2, ld_lds, st_lds, tid, dg, tprev);
        __syncthreads();
        MK_FFT_STAMP(2);
        fft_pass_diag<N2, R2, R1, RBH, NTH, PL::LPR2, G1, G2>(tw2, ld_lds, st_lds, tid, dg, tprev);
        __syncthreads();
        MK_FFT_STAMP(2);
        if constexpr (R3 > 1) {
            fft_pass_diag<N2, R3, R1 * R2, RBH, NTH, PL::LPR3, G2, G3>(tw3, ld_lds, st_lds, tid, dg, tprev);
            __syncthreads();
            MK_FFT_STAMP(2);
        }
#else
        if (item + 1 < it.end) prefetch(item + 1);       // in flight during the passes and the untangle step
        fft_pass<N2, R1, 1, RBH, NTH, true, PL::LPR1, G0, G1>(tw2, ld_lds, st_lds, tid);
        __syncthreads();
        fft_pass<N2, R2, R1, RBH, NTH, true, PL::LPR2, G1, G2>(tw2, ld_lds, st_lds, tid);
        __syncthreads();
        if constexpr (R3 > 1) {
            fft_pass<N2, R3, R1 * R2, RBH, NTH, true, PL::LPR3, G2, G3>(tw3, ld_lds, st_lds, tid);
            __syncthreads();
        }

#endif

        // Hermitian untangle + truncation + weights
        if (vec && HOIST) {
            // (vec: C % 4 == 0 (then Cp == C) or one batch entry — either way plane pr IS row pr of the F layout)
            float* Fi = F + (klat * 2 * rows + p0);                                  // uniform: the item's part of the address
#pragma unroll
            for (int q = 0; q < NQU; ++q) {
                if (un_r0[q] < nr) {
                    const cf* za = buf + un_za[q];
                    const cf* zb = buf + un_zb[q];
                    cf R[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {                                    // (rows >= nr hold zeros)
                        const cf A = za[k * LS], B = zb[k * LS];
                        R[k] = add_mi(add_conj(A, B), cmul(sub_conj(A, B), un_tw[q]));
                    }
                    // the scaling is scalar on purpose: its results are the components of the two 16-byte stores
                    const float4 Xre = make_float4(R[0].x * un_hw[q].x, R[1].x * un_hw[q].x, R[2].x * un_hw[q].x, R[3].x * un_hw[q].x);
                    const float4 Xim = make_float4(R[0].y * un_hw[q].y, R[1].y * un_hw[q].y, R[2].y * un_hw[q].y, R[3].y * un_hw[q].y);
                    float* o = Fi + un_fo[q];
                    fft_st4(o, Xre);
                    fft_st4(o + rows, Xim);
                }
            }
        } else if (vec) {
            for (int idx = tid; idx < mmax * (RBH / 4); idx += NTH) {
                const int r0 = (idx % (RBH / 4)) * 4, m = idx / (RBH / 4);
                if (r0 >= nr) continue;
                const cf* z = buf + r0 * LS;                                        // rows >= nr hold zeros
                const Untangle<N2> un(twu, m, w_dc, w_pos, w_nyq);
                const cf R0 = un.raw(z), R1_ = un.raw(z + LS), R2_ = un.raw(z + 2 * LS), R3_ = un.raw(z + 3 * LS);
                // the scaling is scalar on purpose: its results are the components of the two 16-byte stores
                const float4 Xre = make_float4(R0.x * un.hw.x, R1_.x * un.hw.x, R2_.x * un.hw.x, R3_.x * un.hw.x);
                const float4 Xim = make_float4(R0.y * un.hw.y, R1_.y * un.hw.y, R2_.y * un.hw.y, R3_.y * un.hw.y);
                const long long pr = p0 + r0;
                if constexpr (SEG) {
                    int ims;
                    float* o = F + seg_f_offset(segtab, sg.nw, sg.nh, m, klat, pr, &ims);
                    fft_st4(o, Xre);
                    fft_st4(o + ims, Xim);
                } else {
                    // vec: C % 4 == 0 (then Cp == C) or one batch entry — either way plane pr IS row pr of the F layout
                    // (the general (pr / C) * Cp + pr % C costs two 64-bit divisions per lane and store)
                    float* o = F + ((long long)m * nlat + klat) * 2 * rows + pr;
                    fft_st4(o, Xre);
                    fft_st4(o + rows, Xim);
                }
            }
        } else {
            for (int idx = tid; idx < mmax * RBH; idx += NTH) {
                const int r = idx % RBH, m = idx / RBH;
                if (r >= nr) continue;
                const Untangle<N2> un(twu, m, w_dc, w_pos, w_nyq);
                const cf X = un.raw(buf + r * LS) * un.hw;
                const long long pr = p0 + r;
                float* o = F + ((long long)m * nlat + klat) * 2 * rows + (pr / C) * Cp + (pr % C);
                o[0] = X.x;
                o[rows] = X.y;
            }
        }
        MK_FFT_STAMP(4);
        __syncthreads();
        MK_FFT_STAMP(2);
        if (item + 1 < it.end) {
            commit();
            MK_FFT_STAMP(5);
            __syncthreads();
            MK_FFT_STAMP(2);
        }
    }
    skew_barriers<HV>(half == 0);
#if MK_FFT_DIAG
    dg[6] = tprev - tstart;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) atomicAdd(&g_fft_diag[k], dg[k]);
        atomicAdd(&g_fft_diag[7], 1ull);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------
// Forward 1440-point transform with WAVE-PRIVATE passes (round 6, MK_FFT_WAVE).  rfft_fast_kernel above runs its passes over the 16
// rows of an item with all eight waves in step: a barrier between the loads and the stores of every pass, one after it — five to
// six workgroup barriers per item, every wave in the same phase at the same time, LDS round trips that two waves per SIMD cannot
// cover (the kernel is bound by exactly that, not by its instruction count: docs/LAB_NOTEBOOK.md 6.4).  Here a wave owns TWO rows
// of the item and runs both passes on them alone: within one wave the LDS instructions of a pass execute in program order (all
// reads of a pass before its writes), so no barrier is needed and the eight waves drift apart — one waits for LDS while the other
// wave of its SIMD multiplies.  Each row starts on a 32-lane half of the wave (24 / 30 of the 32 lanes carry a butterfly).  What
// still needs the workgroup is the F side: 16 rows = 64-byte runs per (m, re / im), so the untangled spectrum goes through a
// 33 KB staging image and is stored by all 512 lanes — two barriers per item (image written / image free again).
// Same butterflies, twiddles and untangle arithmetic in the same order as rfft_fast_kernel: bit-identical outputs (tools/
// fft_plan_check.py).  RESULT: not faster (0.51 - 0.53 ms against 0.49), so not dispatched; kept for its diagnostic builds
// (MK_FFT_WAVE_DIAG): with neither loads nor stores the kernel still takes 0.344 of its 0.511 ms — the transform is bound by its
// ON-CHIP work (LDS instruction time of the CU + vector time of the SIMDs add up instead of overlapping), loads add 0.08 ms and
// stores 0.09 on top.
#ifndef MK_FFT_WAVE          // 0 (default): not dispatched — measured 0.51 - 0.53 ms against 0.49 for rfft_fast_kernel (docs/LAB_NOTEBOOK.md 6.4)
#define MK_FFT_WAVE 0
#endif
#ifndef MK_FFT_WAVE_TWR       // second-pass twiddles in registers (1: 38 spilled registers, 0.80 ms) or read from the LDS table per item (0)
#define MK_FFT_WAVE_TWR 0
#endif
#ifndef MK_FFT_WAVE_DIAG      // diagnostic builds: 1 = no global stores, 2 = no global loads, 4 = no untangle / staging at all
#define MK_FFT_WAVE_DIAG 0
#endif

template <typename T>
__global__ __launch_bounds__(512, 1) void rfft_wave_kernel(const T* __restrict__ x, float* __restrict__ F, const cf* __restrict__ tw_g,
                                                           long long rows, long long planes, int nlat, int mmax, int ngr,
                                                           long long nitems, float w_dc, float w_pos, float w_nyq) {
    constexpr int N2 = 720, N = 1440, R1 = 30, R2 = 24, RB = 16, NT = 512, NW = NT / 64, RW = RB / NW, MCAP = N2 / 3 + 1;
    static_assert(RW == 2, "two rows per wave: one per 32-lane half");
    // row strides (complex values) of the three generations of a wave's two-row region; each half-wave touches ONE row, so only the
    // pattern inside a row matters for the passes (see the bank notes at Gen): g1 is padded by one value per 30 (stride 31 between
    // the lanes' stores), g2's two rows are 128 bytes apart modulo 256 for the untangle reads, which alternate between the rows
    using G0 = Gen<728, 0, 0>;
    using G1 = Gen<744, R1, 1>;
    using G2 = Gen<736, 0, 0>;
    constexpr int LSM = 744, LS0 = G0::LS, LS2 = G2::LS;
    constexpr int VP = RowVec<T>::PAIRS, VROW = N2 / VP;
    using Tb = Tables<N2, R1, R2, 1, MCAP>;
    constexpr int SST = 17;                                     // floats per (m, re / im) line of the staging image (16 rows + 1: bank spread)
    __shared__ __attribute__((aligned(16))) cf smem[RB * LSM + Tb::SIZE];
    __shared__ float stage[MCAP * 2 * SST];
    const int lane = (int)threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    cf* buf = smem + wave * (RW * LSM);
    cf* tw2 = smem + RB * LSM;
    cf* twu = tw2 + Tb::T2;
    Tb::template fill<NT>(tw2, tw_g, (int)threadIdx.x);

    const ItemRange it = my_items(nitems);
    const long long rstride = (long long)nlat * N;              // distance between the rows (planes) of an item
    // ---- lane constants ------------------------------------------------------------------------------------------------
    constexpr int NV = (RW * VROW + 63) / 64;                   // 16-byte vectors of the wave's two rows per lane
    uint4 rawv[NV];
    // Global traffic goes through raw buffer instructions: a lane that has nothing to fetch / store gets an offset beyond the
    // buffer (reads return zeros — what rows past the last plane must hold —, writes are dropped), so every load and store of an
    // item is issued UNCONDITIONALLY.  With loads or stores under lane conditions the compiler cannot count what is in flight and
    // answers with s_waitcnt vmcnt(0) wherever a prefetched value is needed: the wave then also waits for the stores it issued a
    // moment ago (loads and stores retire through one in-order counter on gfx9) — a full memory round trip exposed per item.
    constexpr unsigned OOB = 0x80000000u;
    auto prefetch = [&](long long itm) {
        const long long kl_ = (unsigned)itm / (unsigned)ngr;
        const long long p0_ = (itm - kl_ * ngr) * RB;
        const int nr_ = (int)max(0ll, min((long long)RB, planes - p0_));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (p0_ * nlat + kl_) * N), 0, OOB, 0x00020000);
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int idx = lane + q * 64;
            const int row = idx / VROW, c = idx % VROW;         // (constants per lane, re-derived per item: registers are the scarce resource here)
            const bool ok = idx < RW * VROW && wave * RW + row < nr_ && !(MK_FFT_WAVE_DIAG & 2);
            const unsigned vo = ok ? (unsigned)(((long long)(wave * RW + row) * rstride + (long long)c * (2 * VP)) * (long long)sizeof(T)) : OOB;
            const u32x4_t r = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0);
            rawv[q] = make_uint4(r.x, r.y, r.z, r.w);
        }
    };
    auto commit = [&]() {                                       // registers -> this wave's rows as fp32 pairs
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int idx = lane + q * 64;
            if (idx < RW * VROW) {
                float4* d = reinterpret_cast<float4*>(buf + (idx / VROW) * LS0 + (idx % VROW) * VP);
                const uint4 u = rawv[q];
                if constexpr (sizeof(T) == 4) {
                    d[0] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
                } else {
                    d[0] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                                       __uint_as_float(u.y & 0xffff0000u));
                    d[1] = make_float4(__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16),
                                       __uint_as_float(u.w & 0xffff0000u));
                }
            }
        }
    };
    if (it.begin < it.end) {
        prefetch(it.begin);
        commit();
    }
    __syncthreads();                                            // the tables are complete
    const int lrow = lane >> 5, lj = lane & 31;                 // second pass: this lane's row and butterfly (k = lj < 30)
    cf twr[MK_FFT_WAVE_TWR ? R2 - 1 : 1];
    if constexpr (MK_FFT_WAVE_TWR) {
#pragma unroll
        for (int r = 1; r < R2; ++r) twr[r - 1] = tw2[(r - 1) * R1 + (lj < R1 ? lj : 0)];
    }
    // untangle: trip q of a lane handles order m = idx / 2 of row idx % 2 of its wave (offsets, twiddle and weights are re-derived per
    // item: eight trips' worth of constants would not fit the registers next to the 23 twiddles, and the vector unit has slack)
    constexpr int NQU = (MCAP * RW + 63) / 64;
    // store: trip q of a thread handles line mi = (m, re / im) and rows 4 quad .. 4 quad + 3 of the item
    constexpr int NST = (MCAP * 2 * (RB / 4) + NT - 1) / NT;
    auto ld_lds = [&](int, int, int addr) -> cf { return buf[addr]; };
    auto st_lds = [&](int, int, int addr, cf val) { buf[addr] = val; };
    for (long long item = it.begin; item < it.end; ++item) {
        const long long klat = (unsigned)item / (unsigned)ngr;
        const long long p0 = (item - klat * ngr) * RB;
        const int nr = (int)max(0ll, min((long long)RB, planes - p0));
        if (item + 1 < it.end) prefetch(item + 1);              // in flight during the passes
        // the two passes on this wave's rows: no workgroup barrier (one wave: LDS instructions in program order)
        fft_pass<N2, R1, 1, RW, 64, false, 32, G0, G1>(tw2, ld_lds, st_lds, lane);
        if constexpr (!MK_FFT_WAVE_TWR) {
            fft_pass<N2, R2, R1, RW, 64, false, 32, G1, G2>(tw2, ld_lds, st_lds, lane);
        } else {   // second pass: as fft_pass<N2, R2, R1, RW, 64, false, 32, G1, G2>, the lane's 23 twiddles from registers (twr) instead
            // of 12 dependent LDS round trips per item
            cf v[1][R2];
            pass_load<N2, R2, RW, 64, 32, G1>(v, ld_lds, lane);
            if (lj < N2 / R2) {
#pragma unroll
                for (int r = 1; r < R2; ++r) v[0][r] = cmul(v[0][r], twr[r - 1]);
                PDft<R2>::run(v[0]);
                cf* o = buf + lrow * G2::LS + lj;
#pragma unroll
                for (int q = 0; q < R2; ++q) o[q * R1] = v[0][PDft<R2>::loc(q)];
            }
        }
        __syncthreads();                                        // the staging image of the previous item has been stored
#pragma unroll
        for (int q = 0; q < NQU; ++q) {
            const int idx = lane + q * 64;
            const int m = idx >> 1, row = idx & 1;
            if (m < mmax) {
                const cf* z = buf + row * LS2;
                const cf A = z[m], B = z[m == 0 ? 0 : N2 - m], tw = twu[m];
                const float w = 0.5f * (m == 0 ? w_dc : w_pos);                 // (m < mmax <= N2 / 3 + 1: never the Nyquist order)
                const cf R = add_mi(add_conj(A, B), cmul(sub_conj(A, B), tw));
                float* sp = stage + (2 * m) * SST + wave * RW + row;
                sp[0] = R.x * w;
                sp[SST] = R.y * (m == 0 ? 0.f : w);
            }
        }
        if (item + 1 < it.end) commit();                        // this wave's rows of the next item (its own region: program order)
        __syncthreads();                                        // the staging image is complete
        const __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc((void*)(F + (klat * 2 * rows + p0)), 0, OOB, 0x00020000);
#pragma unroll
        for (int q = 0; q < NST; ++q) {
            const int idx = (int)threadIdx.x + q * NT;
            const int mi = idx / (RB / 4), quad = idx % (RB / 4);
            const int m = mi >> 1, ri = mi & 1;
            const float* sp = stage + min(mi, MCAP * 2 - 1) * SST + quad * 4;
            const u32x4_t v4 = {__float_as_uint(sp[0]), __float_as_uint(sp[1]), __float_as_uint(sp[2]), __float_as_uint(sp[3])};
            const bool ok = m < mmax && quad * 4 < nr && !(MK_FFT_WAVE_DIAG & 1);
            const unsigned vo = ok ? (unsigned)((((long long)m * nlat) * 2 * rows + (long long)ri * rows + quad * 4) * 4) : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(v4, rsF, vo, 0, 0);
        }
    }
}

// MCAP: compile-time bound on mmax (N2/3+1 for the 3x-truncated spectra of the scale-3 model, else N2+1);
// it sizes the registers that carry the next item's spectrum.
template <int N2, int R1, int R2, int R3, int RB, int NT, int WGS, int MCAP, typename T, bool SEG, int HV>
__global__ __launch_bounds__(NT, (MK_FFT_480_INV_OCC && N2 == 240 && !SEG) ? 4 : WGS) void irfft_fast_kernel(const float* __restrict__ F, T* __restrict__ x,
                                                             const cf* __restrict__ tw_g, int C, int Cp,
                                                             long long rows, long long planes, int nlat, int mmax,
                                                             int ngr, long long nitems, float w_dc, float w_pos,
                                                             float w_nyq, const MkFftSeg sg) {
    static_assert(R1 * R2 * R3 == N2, "radix product");
    static_assert(N2 <= 1024, "the piece index of the SEG stores is a multiply-shift valid for rows of at most 2048 points");
    constexpr int NTH = NT / HV, RBH = RB / HV;      // HV halves of NTH threads, RBH rows each (see MK_FFT_HV above)
    static_assert(NT % HV == 0 && RB % HV == 0 && RBH % 4 == 0 && NTH % 64 == 0, "halves");
    using PL = LdsPlan<N2, R1, R2, R3, RBH, NTH, true>;
    using G0 = Gen<PL::LS0, 0, 0>;                                        // the (pre-twiddled) spectrum rows
    using G1 = Gen<PL::LS1, R1, PL::D1>;                                  // after pass 1
    using G2 = Gen<PL::LS2, R1 * R2, (R3 > 1) ? PL::D2 : 0>;              // after pass 2 (three-pass plans)
    constexpr int N = 2 * N2, LS = PL::LS0, LSM = plan_max_stride<PL>();
    // PRUNED (mmax <= N2/2): the spectrum is zero for mmax <= m <= N2 - mmax ... N2, which the kernel never
    // touches: no zero fill, the pre-twiddle runs on the loaded registers, the first pass substitutes zeros
    constexpr bool PRUNED = MCAP <= N2 / 2;
    using Tb = Tables<N2, R1, R2, R3, PRUNED ? MCAP : N2 + 1>;
    __shared__ __attribute__((aligned(16))) cf smem[RB * LSM + Tb::SIZE];
    __shared__ SegTab segtab_s;
    SegTab* segtab = &segtab_s;
    const int half = HV > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x / NTH) : 0;
    const int rofs = half * RBH;
    cf* buf = smem + half * (RBH * LSM);
    cf* tw2 = smem + RB * LSM;
    cf* tw3 = tw2 + Tb::T2;
    cf* twu = tw3 + Tb::T3;
    const int tid = (int)threadIdx.x % NTH;
    Tb::template fill<NT>(tw2, tw_g, (int)threadIdx.x);
    if constexpr (SEG) seg_fill<NT>(segtab, sg, (int)threadIdx.x);
    __syncthreads();

    const ItemRange it = my_items(nitems);
    const int xseg = SEG ? max(1, sg.xseg) : 1;                  // x rows cut into xseg pieces in separate buffers (see SegTab)
    const int wl = N / xseg;                                     // points per piece
    const unsigned wl_magic = ((1u << 24) + (unsigned)wl - 1u) / (unsigned)wl;
    const long long xnl = (SEG && sg.x_nlat > 0) ? sg.x_nlat : nlat;      // latitudes per plane in the x buffers
    const long long rstride = xnl * wl;
    const bool vec = (C % 4 == 0) || (planes == C);
    // the next item's half spectrum X[m], m < mmax, rides in registers while this one is transformed:
    // vec: float4 = 4 rows per (m, re/im);  scalar fallback: one (row, m) pair per slot
    constexpr int NQ4 = (MCAP * (RBH / 4) + NTH - 1) / NTH;
    constexpr int NQ1 = (MCAP * RBH + NTH - 1) / NTH;
    float4 sre[NQ4], sim[NQ4];
    // lane constants hoisted out of the item loop (see rfft_fast_kernel): float offset of this lane's spectrum quad q inside an
    // item, its first row, and — for the pruned transform — where its pre-twiddled pair lands in LDS, the order's twiddle and weights
    constexpr bool HOIST = !SEG && WGS <= 2 && !(MK_FFT_480_INV_OCC && N2 == 240);
    unsigned pi_fo[NQ4];
    int pi_r0[NQ4], pt_off[NQ4], pt_m[NQ4];
    cf pt_tw[NQ4], pt_wv[NQ4];
    if constexpr (HOIST) {
#pragma unroll
        for (int q = 0; q < NQ4; ++q) {
            const int idx = tid + q * NTH;
            const int r0 = (idx % (RBH / 4)) * 4, m = idx / (RBH / 4);
            const int mc = min(m, mmax - 1);
            pi_r0[q] = r0;
            pi_fo[q] = (unsigned)(((long long)mc * nlat) * 2 * rows + r0);
            pt_m[q] = (m < mmax) ? m : -1;
            pt_off[q] = r0 * LS + mc;
            pt_tw[q] = twu[mc];                                   // (tables complete: the barrier above)
            pt_wv[q] = inverse_weights<N2>(mc, w_dc, w_pos, w_nyq);
        }
    }
    auto prefetch = [&](long long itm) {
        const long long kl_ = (unsigned)itm / (unsigned)ngr;             // (nitems < 2^31: 32-bit division)
        const long long p0_ = (itm - kl_ * ngr) * RB + rofs;
        const int nr_ = (int)max(0ll, min((long long)RBH, planes - p0_));
        if constexpr (HOIST) {
            // (same clamped, unconditional loads as below; rows >= nr_ read row 0 of the group, a half past the last plane reads plane 0)
            const float* Fi = F + (kl_ * 2 * rows + (nr_ > 0 ? p0_ : 0));
#pragma unroll
            for (int q = 0; q < NQ4; ++q) {
                const float* sp = Fi + ((pi_r0[q] < nr_) ? pi_fo[q] : pi_fo[q] - (unsigned)pi_r0[q]);
                sre[q] = *reinterpret_cast<const float4*>(sp);
                sim[q] = *reinterpret_cast<const float4*>(sp + rows);
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < NQ4; ++q) {
            const int idx = tid + q * NTH;
            const int r0 = (idx % (RBH / 4)) * 4, m = idx / (RBH / 4);
            // Unconditional loads from clamped (always valid) positions: a load under a lane condition, or a select on the
            // loaded value, makes hipcc wait for the memory round trip on the spot instead of leaving the loads in flight
            // during the passes of the current item.  Entries with m >= mmax or rows >= nr are zeroed where they are USED
            // (the same conditions are re-evaluated there for this item).
            const int mc = min(m, mmax - 1);
            const long long pr = (nr_ > 0 ? p0_ : 0) + ((r0 < nr_) ? r0 : 0);      // (a half past the last plane reads plane 0)
            float4 a, b;
            if constexpr (SEG) {
                int ims;
                const float* sp = F + seg_f_offset(segtab, sg.nw, sg.nh, mc, kl_, pr, &ims);
                a = *reinterpret_cast<const float4*>(sp);
                b = *reinterpret_cast<const float4*>(sp + ims);
            } else {
                const float* sp = F + ((long long)mc * nlat + kl_) * 2 * rows + pr;      // (float4 access: plane pr = row pr)
                a = *reinterpret_cast<const float4*>(sp);
                b = *reinterpret_cast<const float4*>(sp + rows);
            }
            sre[q] = a;
            sim[q] = b;
        }
    };
    if ((vec || PRUNED) && it.begin < it.end) prefetch(it.begin);
    skew_barriers<HV>(half == 1);                         // half 1 runs MK_FFT_SKEW barrier intervals behind half 0
    for (long long item = it.begin; item < it.end; ++item) {
        const long long klat = (unsigned)item / (unsigned)ngr;
        const long long p0 = (item - klat * ngr) * RB + rofs;
        const int nr = (int)max(0ll, min((long long)RBH, planes - p0));
        T* xr = x + (p0 * xnl + klat) * (long long)wl;

        if constexpr (PRUNED) {
            // registers -> pre-twiddled pairs (m, N2-m) with X[N2-m] = 0:  Zs[m] and Zs[N2-m] from X[m] alone
#pragma unroll
            for (int q = 0; q < NQ4; ++q) {
                const int idx = tid + q * NTH;
                const int r0 = HOIST ? pi_r0[q] : (idx % (RBH / 4)) * 4, m = HOIST ? pt_m[q] : idx / (RBH / 4);
                if (HOIST ? (m >= 0) : (m < mmax)) {
                    // rows >= nr were loaded from a clamped (valid) position: zero weights instead of a select per value
                    const cf wv = (r0 < nr) ? (HOIST ? pt_wv[q] : inverse_weights<N2>(m, w_dc, w_pos, w_nyq)) : cf_make(0.f, 0.f);
                    const cf tw = HOIST ? pt_tw[q] : twu[m];
                    cf* za = buf + (HOIST ? pt_off[q] : r0 * LS + m);              // Z[m] of row r0
                    cf* zb = za + (N2 - 2 * m);                                      // Z[N2 - m]
                    const float re[4] = {sre[q].x, sre[q].y, sre[q].z, sre[q].w};
                    const float im[4] = {sim[q].x, sim[q].y, sim[q].z, sim[q].w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const cf Xa = mul_parts(wv, re[i], im[i]);
                        const cf Xc = cf_make(Xa.x, -Xa.y);
                        const cf wt = cmulc(Xa, tw);                           // conj(U) X
                        za[i * LS] = sub_yx(Xc, wt);                               // conj(X + i conj(U) X)
                        if (m != 0) {
                            const cf w2 = cmul(Xc, tw);                        // U conj(X)
                            zb[i * LS] = sub_yx(Xa, w2);                           // conj(conj(X) + i U conj(X))
                        }
                    }
                }
            }
            __syncthreads();
            if (item + 1 < it.end) prefetch(item + 1);         // in flight during the passes
        } else {
            // weighted spectrum -> LDS rows, zero beyond mmax (and in rows >= nr)
            if (vec) {
    #pragma unroll
                for (int q = 0; q < NQ4; ++q) {
                    const int idx = tid + q * NTH;
                    const int r0 = HOIST ? pi_r0[q] : (idx % (RBH / 4)) * 4, m = HOIST ? pt_m[q] : idx / (RBH / 4);
                    if (HOIST ? (m >= 0) : (m < mmax)) {
                        const cf wv = (r0 < nr) ? (HOIST ? pt_wv[q] : inverse_weights<N2>(m, w_dc, w_pos, w_nyq)) : cf_make(0.f, 0.f);
                        cf* z = buf + (HOIST ? pt_off[q] : r0 * LS + m);
                        z[0 * LS] = mul_parts(wv, sre[q].x, sim[q].x);
                        z[1 * LS] = mul_parts(wv, sre[q].y, sim[q].y);
                        z[2 * LS] = mul_parts(wv, sre[q].z, sim[q].z);
                        z[3 * LS] = mul_parts(wv, sre[q].w, sim[q].w);
                    }
                }
            } else {
    #pragma unroll 4
                for (int q = 0; q < NQ1; ++q) {
                    const int idx = tid + q * NTH;
                    const int r = idx % RBH, m = idx / RBH;
                    if (m < mmax) {
                        cf X = cf_make(0.f, 0.f);
                        if (r < nr) {
                            const long long pr = p0 + r;
                            const float* sp = F + ((long long)m * nlat + klat) * 2 * rows + (pr / C) * Cp + (pr % C);
                            X = cf_make(sp[0], sp[rows]) * inverse_weights<N2>(m, w_dc, w_pos, w_nyq);
                        }
                        buf[r * LS + m] = X;
                    }
                }
            }
            for (int idx = tid + mmax * RBH; idx < (N2 + 1) * RBH; idx += NTH) buf[(idx % RBH) * LS + idx / RBH] = cf_make(0.f, 0.f);
            __syncthreads();
            if (vec && item + 1 < it.end) prefetch(item + 1);      // in flight during the pre-twiddle and the passes

            // in-place pre-twiddle on the pairs (j, N2-j): Zs[j] = (A + Bc) + i conj(W^j)(A - Bc); store conj(Zs)
            for (int idx = tid; idx < RBH * (N2 / 2 + 1); idx += NTH) {
                const int row = idx / (N2 / 2 + 1), j = idx % (N2 / 2 + 1);
                const int j2 = N2 - j;                         // partner (j = 0 pairs with N2, j = N2/2 with itself)
                const cf Xa = buf[row * LS + j], Xb = buf[row * LS + j2];
                const cf ua = add_conj(Xa, Xb), ub = add_conj(Xb, Xa);             // ub = conj(ua)
                {
                    const cf wt = cmulc(sub_conj(Xa, Xb), twu[j]);
                    buf[row * LS + j] = sub_yx(ub, wt);                                // conj(ua + i wt)
                }
                if (j != 0 && j2 != j) {
                    const cf wt = cmulc(sub_conj(Xb, Xa), twu[j2]);
                    buf[row * LS + j2] = sub_yx(ua, wt);
                }
            }
            __syncthreads();
        }

        auto ld_lds = [&](int, int, int addr) -> cf { return buf[addr]; };
        auto st_lds = [&](int, int, int addr, cf val) { buf[addr] = val; };
        // the last pass writes its rows straight to global memory (staging them through LDS for 16-byte
        // stores measured 10 % slower: the extra round trip costs more than the narrower stores)
        auto st_global = [&](int row, int pos, int, cf val) {
            long long e = 2 * pos;
            if constexpr (SEG) {
                // piece j = e / wl by a multiply-shift that is exact for e < 2 N2 <= 2^11 and wl >= 8 (e * wl < 2^24)
                const unsigned j = ((unsigned)(2 * pos) * wl_magic) >> 24;
                e = (long long)j * sg.x_stride + (2 * pos - (int)j * wl);
            }
            if (row < nr) store_pair<T>(xr + (long long)row * rstride + e, val.x, -val.y);     // conj
        };
        if constexpr (PRUNED) {
            const int z0 = mmax, z1 = N2 - mmax;               // never-written (zero) positions, inclusive
            fft_pass<N2, R1, 1, RBH, NTH, true, PL::LPR1, G0, G1>(tw2, [&](int, int pos, int addr) -> cf {
                return (pos >= z0 && pos <= z1) ? cf_make(0.f, 0.f) : buf[addr];
            }, st_lds, tid);
        } else {
            fft_pass<N2, R1, 1, RBH, NTH, true, PL::LPR1, G0, G1>(tw2, ld_lds, st_lds, tid);
        }
        __syncthreads();
        using GX = Gen<1, 0, 0>;                               // (the last pass stores to global memory: no LDS map)
        if constexpr (R3 > 1) {
            fft_pass<N2, R2, R1, RBH, NTH, true, PL::LPR2, G1, G2>(tw2, ld_lds, st_lds, tid);
            __syncthreads();
            fft_pass<N2, R3, R1 * R2, RBH, NTH, false, PL::LPR3, G2, GX>(tw3, ld_lds, st_global, tid);
        } else {
            fft_pass<N2, R2, R1, RBH, NTH, false, PL::LPR2, G1, GX>(tw2, ld_lds, st_global, tid);
        }
        __syncthreads();
    }
    skew_barriers<HV>(half == 0);
}

// halves: see MK_FFT_HV (two only when the workgroup has 512 threads and every half keeps whole groups of four rows)
template <int N2, int RB, int NT, typename T, bool SEG, bool INVERSE>
constexpr int fft_halves() {
    if (!(NT == 512 && RB % 8 == 0)) return 1;
    if (MK_FFT_HV == 2) return 2;
    return (MK_FFT_HV == 1 && N2 == 240 && !INVERSE && sizeof(T) == 2 && !SEG) ? 2 : 1;
}

template <int N2, int R1, int R2, int R3, int RB, int NT, int WGS, int MCAP, typename T, bool SEG>
int launch_inverse(const float* in, T* out, const float2* tw, int C, int Cp, long long rows, long long planes, int nlat,
                   int mmax, int ngr, long long nitems, float w_dc, float w_pos, float w_nyq, const MkFftSeg& sg, hipStream_t s) {
    auto kern = irfft_fast_kernel<N2, R1, R2, R3, RB, NT, WGS, MCAP, T, SEG, fft_halves<N2, RB, NT, T, SEG, true>()>;
    static int per_cu = 0;
    if (per_cu == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(kern), NT, 0) != hipSuccess || n < 1) n = WGS;
        per_cu = n > 8 ? 8 : n;
    }
    long long grid = 256ll * per_cu;            // persistent: every workgroup resident, contiguous item ranges
    if (grid > nitems) grid = nitems;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), 0, s, in, out, reinterpret_cast<const cf*>(tw), C, Cp, rows, planes, nlat, mmax, ngr, nitems,
                       w_dc, w_pos, w_nyq, sg);
    return mk_check_launch("mk_irfft_rows(fast)");
}

template <int N2, int R1, int R2, int R3, int RB, int NT, int WGS, int MCAP, typename T, bool SEG>
int launch_forward(const T* in, float* out, const float2* tw, int C, int Cp, long long rows, long long planes, int nlat,
                   int mmax, int ngr, long long nitems, float w_dc, float w_pos, float w_nyq, const MkFftSeg& sg, hipStream_t s) {
    if constexpr (MK_FFT_WAVE && N2 == 720 && R1 == 30 && R2 == 24 && RB == 16 && NT == 512 && MCAP == N2 / 3 + 1 && !SEG && sizeof(T) == 2) {      // (fp32 rows: 24 more registers in flight per lane: spills)
        // wave-private passes (rfft_wave_kernel): the float4 F access needs plane = row (C % 4 == 0 or one batch entry)
        static const bool off = [] { const char* e = getenv("MAKANI_AMD_FFT_WAVE"); return e && e[0] == '0'; }();
        if (!off && ((C % 4 == 0) || planes == C) && (long long)mmax * nlat * 2 * rows < (1ll << 32)) {
            long long grid = 256;
            if (grid > nitems) grid = nitems;
            hipLaunchKernelGGL(rfft_wave_kernel<T>, dim3((unsigned)grid), dim3(512), 0, s, in, out, reinterpret_cast<const cf*>(tw), rows, planes,
                               nlat, mmax, ngr, nitems, w_dc, w_pos, w_nyq);
            return mk_check_launch("mk_rfft_rows(wave)");
        }
    }
    auto kern = rfft_fast_kernel<N2, R1, R2, R3, RB, NT, WGS, MCAP, T, SEG, fft_halves<N2, RB, NT, T, SEG, false>()>;
    static int per_cu = 0;
    if (per_cu == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(kern), NT, 0) != hipSuccess || n < 1) n = WGS;
        per_cu = n > 8 ? 8 : n;
    }
    long long grid = 256ll * per_cu;
    if (grid > nitems) grid = nitems;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), 0, s, in, out, reinterpret_cast<const cf*>(tw), C, Cp, rows, planes, nlat, mmax, ngr, nitems,
                       w_dc, w_pos, w_nyq, sg);
    return mk_check_launch("mk_rfft_rows(fast)");
}

template <int N2, int R1, int R2, int R3, int RB, int NT, int WGS, bool SEG>
int launch_seg(bool inverse, const void* in, void* out, int dtype, const float2* tw, int B, int C, int Cp, int nlat, int mmax,
               float w_dc, float w_pos, float w_nyq, const MkFftSeg& sg, hipStream_t s) {
    const long long planes = (long long)B * C;
    const int ngr = (int)((planes + RB - 1) / RB);
    const long long nitems = (long long)nlat * ngr;
    const long long rows = (long long)B * Cp;
    MK_REQUIRE((((uintptr_t)in | (uintptr_t)out) & 15) == 0, "fft: x and F must be 16-byte aligned");
    MK_REQUIRE(nitems < (1ll << 31), "fft: too many (latitude, row group) items");
    constexpr int M3 = N2 / 3 + 1, MF = N2 + 1;
#define MK_FFT_TAIL tw, C, Cp, rows, planes, nlat, mmax, ngr, nitems, w_dc, w_pos, w_nyq, sg, s
    const bool vec = (C % 4 == 0) || (B == 1);
    if (SEG) {
        // per-peer slabs: rows in groups of 4 (float4 F access), row ranges on multiples of 4, x pieces of whole 16-byte vectors
        MK_REQUIRE(vec && B == 1, "fft(seg): one batch entry (the planes are the rows)");
        MK_REQUIRE(sg.nw >= 1 && sg.nw <= MK_FFT_SEG_MAX && sg.nh >= 1 && sg.nh <= MK_FFT_SEG_MAX, "fft(seg): 1..%d peers per direction", MK_FFT_SEG_MAX);
        MK_REQUIRE(sg.m_off[0] == 0 && sg.m_off[sg.nw] == mmax && sg.r_off[0] == 0 && sg.r_off[sg.nh] >= C, "fft(seg): ranges must cover m < mmax and every row");
        for (int i = 0; i <= sg.nh; ++i) MK_REQUIRE(sg.r_off[i] % 4 == 0, "fft(seg): row ranges must start on multiples of 4");
        for (int j = 0; j < sg.nw; ++j)
            for (int i = 0; i < sg.nh; ++i) MK_REQUIRE(sg.base[j][i] % 4 == 0, "fft(seg): slab offsets must be multiples of 4 floats");
        const int xs = sg.xseg < 1 ? 1 : sg.xseg;
        const int ev = dtype == MK_F32 ? 4 : 8;
        MK_REQUIRE(xs <= MK_FFT_SEG_MAX && (2 * N2) % xs == 0 && ((2 * N2) / xs) % ev == 0 && (xs == 1 || sg.x_stride % ev == 0),
                   "fft(seg): x pieces must be equal and hold whole 16-byte vectors");
    }
    if (!inverse) {
        if (mmax <= M3) {
            if (dtype == MK_F32) return launch_forward<N2, R1, R2, R3, RB, NT, WGS, M3, float, SEG>((const float*)in, (float*)out, MK_FFT_TAIL);
            return launch_forward<N2, R1, R2, R3, RB, NT, WGS, M3, u16, SEG>((const u16*)in, (float*)out, MK_FFT_TAIL);
        }
        if (dtype == MK_F32) return launch_forward<N2, R1, R2, R3, RB, NT, WGS, MF, float, SEG>((const float*)in, (float*)out, MK_FFT_TAIL);
        return launch_forward<N2, R1, R2, R3, RB, NT, WGS, MF, u16, SEG>((const u16*)in, (float*)out, MK_FFT_TAIL);
    }
    if (mmax <= M3 && vec) {      // truncated spectrum (pruned transform; it needs the float4 F access)
        if (dtype == MK_F32) return launch_inverse<N2, R1, R2, R3, RB, NT, WGS, M3, float, SEG>((const float*)in, (float*)out, MK_FFT_TAIL);
        return launch_inverse<N2, R1, R2, R3, RB, NT, WGS, M3, u16, SEG>((const float*)in, (u16*)out, MK_FFT_TAIL);
    }
    if (dtype == MK_F32) return launch_inverse<N2, R1, R2, R3, RB, NT, WGS, MF, float, SEG>((const float*)in, (float*)out, MK_FFT_TAIL);
    return launch_inverse<N2, R1, R2, R3, RB, NT, WGS, MF, u16, SEG>((const float*)in, (u16*)out, MK_FFT_TAIL);
#undef MK_FFT_TAIL
}

template <int N2, int R1, int R2, int R3, int RB, int NT, int WGS>
int launch(bool inverse, const void* in, void* out, int dtype, const float2* tw, int B, int C, int Cp, int nlat, int mmax,
           float w_dc, float w_pos, float w_nyq, const MkFftSeg* sg, hipStream_t s) {
    if (sg) return launch_seg<N2, R1, R2, R3, RB, NT, WGS, true>(inverse, in, out, dtype, tw, B, C, Cp, nlat, mmax, w_dc, w_pos, w_nyq, *sg, s);
    MkFftSeg none = {};
    return launch_seg<N2, R1, R2, R3, RB, NT, WGS, false>(inverse, in, out, dtype, tw, B, C, Cp, nlat, mmax, w_dc, w_pos, w_nyq, none, s);
}

}  // namespace

#define MK_FFT_ARGS inverse, in, out, dtype, tw, B, C, Cp, nlat, mmax, w_dc, w_pos, w_nyq, seg, s

// returns -1000 if nlon has no specialised kernel (caller falls through to the generic one)
// seg != nullptr: segmented addressing of both sides (distributed transforms), see SegTab
int mk_fft_fast_dispatch(bool inverse, const void* in, void* out, int dtype, const float* twiddle, int B, int C, int Cp,
                         int nlat, int nlon, int mmax, float w_dc, float w_pos, float w_nyq, const MkFftSeg* seg, void* stream) {
    const float2* tw = reinterpret_cast<const float2*>(twiddle);
    hipStream_t s = (hipStream_t)stream;
    //                         N2  radices   RB   NT  WG/CU
    switch (nlon) {
        // measured and rejected: 384-thread workgroups with 8 rows (3 waves/SIMD) 10-25 % slower than 256-thread ones;
        // forcing 3 workgroups/CU on the 8-row forward 1440 kernel spills (2.6x slower)
        // 16 rows / 512 threads, one workgroup per CU: the F side is touched in 64-byte runs (8 rows / 256 threads,
        // two workgroups per CU: 14-16 % slower)
        // 12 rows / 512 threads at 128 registers = TWO workgroups per CU (77 KB of LDS each; launch<720, 30, 24, 1, 12, 512, 4>):
        // 12-18 spilled registers and 48-byte runs on the F side, rfft bf16 0.49 -> 0.64 ms, irfft 0.41 -> 0.64 (round 4, rejected)
#if MK_FFT_1440_2PASS
        case 1440: return launch<720, 30, 24, 1, 16, 512, 1>(MK_FFT_ARGS);
#else
        case 1440: return launch<720, 10, 9, 8, 16, 512, 1>(MK_FFT_ARGS);
#endif
        case 480: return launch<240, 10, 6, 4, 32, 512, 2>(MK_FFT_ARGS);   // 32 rows: whole 128-byte lines on the F side (irfft +7 %)
        // FourCastNet3's internal grid (360 x 720, full spectrum mmax = 361: fourcastnet3.py:503-504); round 2 ran it on the
        // generic kernel at 0.11 of the HBM rate
        case 720: return launch<360, 10, 6, 6, 16, 256, 3>(MK_FFT_ARGS);
        case 360: return launch<180, 6, 6, 5, 16, 256, 3>(MK_FFT_ARGS);
        case 128: return launch<64, 4, 4, 4, 16, 256, 4>(MK_FFT_ARGS);
        case 72: return launch<36, 6, 6, 1, 16, 256, 4>(MK_FFT_ARGS);
        default: return -1000;
    }
}

#if MK_FFT_DIAG
extern "C" int mk_fft_diag_read(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fft_diag), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_fft_diag), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif
