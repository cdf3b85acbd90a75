// Second-generation split-bf16 GEMM kernels for the spectral hot shapes (same arithmetic as xgemm.hip: fp32 operands
// expanded into 3 (or 2) bf16 limbs, 6 (3) bf16 MFMAs per product, fp32 accumulation).
//
// What xgemm.hip's kernels lose (SQ counters, profiles/r02_pmc_sq_bench_raw.md): a workgroup alternates a "split the
// next fp32 tile into limbs" segment (VALU + LDS stores) with an MFMA segment, two barriers per 16-deep k-step, and the
// two workgroups that share a CU fall into lockstep, so that the matrix pipe and the VALU take turns instead of
// overlapping (MFMA busy + VALU busy = the whole SIMD time).  Here the overlap is built into one 512-thread workgroup:
//   * the LDS limb images are double buffered; tile kt+1 is split and stored while tile kt is multiplied;
//   * the 8 waves form two groups of 4 (one wave of each group per SIMD).  A k-step has two phases separated by
//     barriers: in phase g group g runs its MFMAs on the current LDS image while the other group splits and stores
//     ITS share of the next tile and issues the global loads of the tile after that ("ping-pong").  On every SIMD one
//     wave feeds the matrix pipe while its partner uses the VALU and the memory path;
//   * work is spread over the SIMDs by COLUMN slices (wave = 32 or 64 output columns x every second live 32-row
//     tile), so that the triangular structure of the spectral operators (rows l < m / m > l are skipped in 32-row
//     steps) shortens every SIMD's segment by the same amount instead of idling whole SIMDs;
//   * tiles are 128 x 128 (complex) and 256 x 256 (real) instead of 64 x 128 and 64 x 256: every fp32 element is
//     split by fewer workgroups;
//   * the real kernel takes its A operand (the constant Legendre matrices) ALREADY split into bf16 limb planes
//     (prepared once per matrix by the host side), copies them global -> LDS as 16-byte vectors and only splits the
//     data operand: 1.25 VALU instructions per MFMA instead of 5.
#include <type_traits>

#include "xsplit.h"

#ifndef MK_X2_ST_NT            // A/B knob: results of the Legendre / dhconv forward and data-gradient GEMMs with the streaming (nt) store policy
#define MK_X2_ST_NT 0
#endif
namespace {

using gemm::BlockCoord;
using gemm::decode_block;
using gemm::validate;
using namespace xsplit;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int NT2 = 512;
// timing diagnostics (tools/ab.py variants; results are WRONG with any of them), a bit mask: 1 = no limb split / LDS stores in
// the main loop, 2 = no MFMA segment, 4 = no global loads in the main loop, 8 = (complex kernel) operand fragments read
// from LDS in the first k-step only, 16 = (complex kernel) no result stores, 32 = (complex kernel) s_memtime stamps at the
// segment boundaries of every k-step, summed per wave group into g_x2_diag (read with mk_x2_diag_read; tools/x2_diag.py)
#ifndef MK_X2_DIAG
#define MK_X2_DIAG 0
#endif
// structure knobs (tools/ab.py variants)
#ifndef MK_X2_MIDBAR          // 1: barrier between the two phases of a k-step (strict ping-pong); 0: one barrier per k-step,
#define MK_X2_MIDBAR 0        //    group 0 = produce then compute, group 1 = compute then produce (soft ping-pong)
#endif
#ifndef MK_X2_PRIO            // s_setprio 1 around the MFMA segment: the matrix pipe's wave wins the issue arbitration
#define MK_X2_PRIO 1          // against its partner's split / store instructions
#endif
#ifndef MK_X2_REV             // heaviest batches first for the "<=" triangles (dhconv: l descending)
#define MK_X2_REV 1
#endif
#ifndef MK_X2_DEPTH           // k-steps a global load is issued ahead of its split (1 or 2 staging register sets);
#define MK_X2_DEPTH 2         // complex kernel.  Measured: 2 = 1 within noise (the kernels are bound by the bytes a CU
#endif                        // can ingest per clock, not by load latency); the real kernel stays at 1 (2 would spill)
#ifndef MK_X2_DEPTH_R
#define MK_X2_DEPTH_R 1
#endif
#ifndef MK_X2_WGRAD_NT         // interleaved-complex results (the dhconv weight gradient, read again only by the optimizer at the end
#define MK_X2_WGRAD_NT 1      // of the step) leave with non-temporal stores: 0.227 -> 0.223 ms, and 283 MB less cache turnover per launch
#endif
#ifndef MK_X2_ILV             // limb products issued round-robin over the independent accumulators
#define MK_X2_ILV 1
#endif

#if MK_X2_DIAG & 32
// [group][0 produce, 1 fragment reads, 2 MFMA segment, 3 barrier, 4 prologue, 5 epilogue, 6 whole kernel, 7 waves]
__device__ unsigned long long g_x2_diag[2][8];
#define MK_X2_STAMP(k)                                              \
    do {                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          \
        const unsigned long long t_ = __builtin_readcyclecounter(); \
        dg[k] += t_ - tprev;                                        \
        tprev = t_;                                                 \
    } while (0)
#else
#define MK_X2_STAMP(k)
#endif

__device__ __forceinline__ void prio_hi() {
#if MK_X2_PRIO
    __builtin_amdgcn_s_setprio(1);
#endif
}
__device__ __forceinline__ void prio_lo() {
#if MK_X2_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
}

// decode_block with the batch order reversed for the triangles that grow with the batch index: the long-running
// workgroups start first and the short ones fill the tail of the launch
template <int BM, int BN>
__device__ __forceinline__ BlockCoord decode_block2(const MkGemm& p, int tilesM, int tilesN) {
    BlockCoord c;
    const int xcd = blockIdx.x % MK_NUM_XCD, j = blockIdx.x / MK_NUM_XCD;
    const int tpb = tilesM * tilesN;
    c.b = (j / tpb) * MK_NUM_XCD + xcd;
    const int t = j % tpb;
    c.i0 = (t / tilesN) * BM;
    c.j0 = (t % tilesN) * BN;
    c.Meff = p.M;
    c.klo = 0;
    c.khi = p.K;
    c.active = c.b < p.batch;
    if (!c.active) return c;
    if (MK_X2_REV && (p.tri_mode == MK_TRI_ROW_LE || p.tri_mode == MK_TRI_K_LE)) c.b = p.batch - 1 - c.b;
    const int tt = c.b / p.inner + p.tri_off;
    switch (p.tri_mode) {
        case MK_TRI_ROW_GE:
            if (c.i0 + BM <= tt) c.active = false;
            break;
        case MK_TRI_K_GE:
            c.klo = max(0, min(tt, p.K));
            break;
        case MK_TRI_ROW_LE:
            c.Meff = max(0, min(p.M, tt + 1));
            if (c.i0 >= c.Meff) c.active = false;
            break;
        case MK_TRI_K_LE:
            c.khi = max(0, min(p.K, tt + 1));
            break;
        default:
            break;
    }
    return c;
}

// one 32 x 32 x 16 complex tile update, limb products round-robin over the three accumulators (a dependent MFMA on the
// same accumulator then finds its predecessor retired): re += ar br, ng += ai bi, im += ar bi + ai br
template <int NP>
__device__ __forceinline__ void cmma_split(const bf16x8* ar, const bf16x8* ai, const bf16x8* br, const bf16x8* bi,
                                           f32x16& cre, f32x16& cng, f32x16& cim) {
#if MK_X2_ILV
    constexpr int NPROD = NP == 3 ? 6 : 3;
    constexpr int IA[6] = {1, 0, 2, 0, 1, 0}, IB[6] = {1, 2, 0, 1, 0, 0};      // smallest terms first (as mma_split)
    constexpr int O = NP == 3 ? 0 : 3;
#pragma unroll
    for (int q = 0; q < NPROD; ++q) {
        const int a = IA[O + q], b = IB[O + q];
        cre = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[a], br[b], cre, 0, 0, 0);
        cim = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[a], bi[b], cim, 0, 0, 0);
        cng = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ai[a], bi[b], cng, 0, 0, 0);
        cim = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ai[a], br[b], cim, 0, 0, 0);
    }
#else
    cre = mma_split<NP>(ar, br, cre);
    cng = mma_split<NP>(ai, bi, cng);
    cim = mma_split<NP>(ar, bi, cim);
    cim = mma_split<NP>(ai, br, cim);
#endif
}

// two real 32 x 32 x 16 tile updates sharing the A fragment
template <int NP>
__device__ __forceinline__ void rmma_split2(const bf16x8* a, const bf16x8* b0, const bf16x8* b1, f32x16& c0, f32x16& c1) {
#if MK_X2_ILV
    constexpr int NPROD = NP == 3 ? 6 : 3;
    constexpr int IA[6] = {1, 0, 2, 0, 1, 0}, IB[6] = {1, 2, 0, 1, 0, 0};
    constexpr int O = NP == 3 ? 0 : 3;
#pragma unroll
    for (int q = 0; q < NPROD; ++q) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[IA[O + q]], b0[IB[O + q]], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[IA[O + q]], b1[IB[O + q]], c1, 0, 0, 0);
    }
#else
    c0 = mma_split<NP>(a, b0, c0);
    c1 = mma_split<NP>(a, b1, c1);
#endif
}
// explicit global address space on the staging loads: a pointer that may be either an operand address or the zero
// block is otherwise treated as generic and loaded with flat_load (which also ties up the LDS counter)
typedef __attribute__((address_space(1))) const f32x4 g_f32x4;
typedef __attribute__((address_space(1))) const u32x4 g_u32x4;

// 32 bytes of zeros: the address a lane loads from when its vector lies outside the operand.  Every global load of
// these kernels is UNCONDITIONAL (out-of-range lanes read zeros from here): a load under a lane condition, or one
// whose destination was zeroed first, makes hipcc wait for all loads in flight (vmcnt(0)) before it issues the next
// one, which serialises the memory round trips of a tile (seen in xgemm.hip's edge path and its interleaved-B path).
__device__ const f32x4 g_zero32[2] = {};

// fp32 tile staging: the tile of the NEXT k-step rides in registers while the current one is multiplied
// RQN: float4 vectors per k-row that are staged (default: the whole ROWS-wide tile).  A narrower operand (the real kernel's
// NARROW form: 160 of 256 columns) is dealt over the threads vector by vector, so that whole waves — not lanes — go without
// work for the columns that do not exist; the LDS image keeps the pitch of the ROWS-wide tile.
// SW (KC only): the [row][k] limb image has NO padding (pitch 16 elements = 32 B); the two 16-byte halves of a row are
// exchanged in rows whose bit 3 is set, which keeps the ds_read_b128 fragment reads (16 lanes = 16 rows per pass, 32 B apart)
// and the ds_write_b64 split stores conflict-free.  A third less LDS per [row][k] plane: what lets the 256-row complex tile
// keep two stages.
template <int ROWS, bool KC, int RQN = ROWS / 4, bool SW = false>
struct Stage2 {
    static constexpr int NVEC = KC ? ROWS * BK / 4 : RQN * BK;          // vectors of one k-step
    static constexpr int NV = (NVEC + NT2 - 1) / NT2;
    static constexpr bool RAGGED = NVEC % NT2 != 0;                       // the last round is not full
    static_assert(NV >= 1 && (KC ? RQN == ROWS / 4 : true), "tile too small");
    f32x4 v[NV];
    unsigned keep;              // KC: 4 bits per vector = elements inside [klo, khi), applied when the tile is stored
    bool interior;              // uniform: the tile lies inside [klo, khi) (then no element masks are applied)
    const float* p0[KC ? 1 : NV];   // address of every vector at k = 0 (KC: of vector 0; vector q lies q * qstep floats further)
    unsigned ok;                // bit q: the row(s) of vector q exist
    long long kstride;          // floats per unit of k
    long long qstep;            // KC: floats between the rows of consecutive vectors of a thread (uniform)

    // ilv: the operand is an interleaved complex tensor; this stage then loads 8 consecutive floats per vector
    // (v = first four, v2 = last four) and store_ilv() separates real and imaginary parts
    f32x4 v2[NV];

    __device__ __forceinline__ void init(const float* __restrict__ base, long long rs, long long ks, int r0, int rmax,
                                         int tid, bool ilv = false) {
        ok = 0u;
        const int u = ilv ? 2 : 1;
        kstride = KC ? u : ks;
        qstep = (long long)(NT2 / 4) * rs;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int f = tid + q * NT2;
            if constexpr (KC) {
                const int row = f >> 2, kq = f & 3;
                if (q == 0) p0[0] = base + (long long)(r0 + row) * rs + kq * 4 * u;
                ok |= (r0 + row < rmax ? 1u : 0u) << q;
            } else {
                constexpr int RQ = RQN;
                const int kk = (f / RQ) % BK, rq = f % RQ;            // (% BK: threads past the last vector of a ragged round)
                p0[q] = base + (long long)kk * ks + (long long)(r0 + rq * 4) * u;
                ok |= ((r0 + rq * 4 < rmax && f < NVEC) ? 1u : 0u) << q;
            }
        }
    }

    // load_at: the addresses of another stage (passed by value) plus a uniform offset: the imaginary part of a planar operand
    // lies `im` floats behind the real part, its stage then carries no pointers of its own
    template <bool ILV>
    __device__ __forceinline__ void load(int k0, int klo, int khi, int tid) {
        load_at<ILV>(k0, klo, khi, tid, p0[0], ok, kstride, qstep, 0);
    }
    template <bool ILV>
    __device__ __forceinline__ void load_from(const Stage2& ad, long long off, int k0, int klo, int khi, int tid) {
        static_assert(KC || NV == 1, "only vector 0 takes its address from the other stage");
        load_at<ILV>(k0, klo, khi, tid, ad.p0[0], ad.ok, ad.kstride, ad.qstep, off);
    }
    template <bool ILV>
    __device__ __forceinline__ void load_at(int k0, int klo, int khi, int tid, const float* a0, unsigned aok,
                                            long long akstride, long long aqstep, long long off) {
        const long long koff = (long long)k0 * akstride + off;
        interior = k0 >= klo && k0 + BK <= khi;                 // uniform
        keep = 0u;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int f = tid + q * NT2;
            bool valid = (aok >> q) & 1u;
            unsigned m = 0xfu;
            if (!interior) {
                if constexpr (KC) {
                    const int k = k0 + (f & 3) * 4;
                    valid = valid && k < khi && k + 3 >= klo;
                    m = 0u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) m |= in_range(k + e, klo, khi) ? (1u << e) : 0u;
                } else {
                    constexpr int RQ = RQN;
                    valid = valid && in_range(k0 + f / RQ, klo, khi);
                }
            }
            keep |= (valid ? m : 0u) << (4 * q);
            const float* pq = KC ? a0 + q * aqstep : (q == 0 ? a0 : p0[KC ? 0 : q]);
            const g_f32x4* src = valid ? (const g_f32x4*)(pq + koff) : (const g_f32x4*)g_zero32;
            v[q] = src[0];
            if constexpr (ILV) v2[q] = src[1];
        }
    }

    __device__ static __forceinline__ int lds_off(int f) {
        if constexpr (KC && SW) {
            const int row = f >> 2, kq = f & 3;
            return row * BK + ((((kq >> 1) ^ (row >> 3)) & 1) << 3) + (kq & 1) * 4;
        } else if constexpr (KC) {
            return (f >> 2) * PK + (f & 3) * 4;
        } else {
            constexpr int RQ = RQN;
            return (f / RQ) * (ROWS + 32) + (f % RQ) * 4;
        }
    }

    // split into NP bf16 limbs and store; limb plane p lives at lds + p * PLANE (elements)
    template <int NP, int PLANE>
    __device__ __forceinline__ void store(u16* lds, int tid, float sign) const {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            if (RAGGED && q == NV - 1 && tid + q * NT2 >= NVEC) continue;      // (whole waves: NVEC is a multiple of 64)
            f32x4 r = v[q] * sign;
            if constexpr (KC) {
                if (!interior) r = mask4(r, keep >> (4 * q));
            }
            split_store4<NP, PLANE>(lds, lds_off(tid + q * NT2), r.xy, r.zw);
        }
    }

    // interleaved operand: real parts -> lre, imaginary parts (times sign) -> lim
    template <int NP, int PLANE>
    __device__ __forceinline__ void store_ilv(u16* lre, u16* lim, int tid, float sign) const {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            f32x4 re = {v[q][0], v[q][2], v2[q][0], v2[q][2]};
            f32x4 im = f32x4{v[q][1], v[q][3], v2[q][1], v2[q][3]} * sign;
            if constexpr (KC) {
                if (!interior) {
                    re = mask4(re, keep >> (4 * q));
                    im = mask4(im, keep >> (4 * q));
                }
            }
            const int off = lds_off(tid + q * NT2);
            split_store4<NP, PLANE>(lre, off, re.xy, re.zw);
            split_store4<NP, PLANE>(lim, off, im.xy, im.zw);
        }
    }
};

template <int ROWS, bool KC, bool SW>
constexpr int plane_elems2() {
    return (KC && SW) ? ROWS * BK : plane_elems<ROWS, KC>();
}
template <int ROWS, bool KC, bool SW>
__device__ __forceinline__ bf16x8 frag2(const u16* plane, int r0, int lane) {
    if constexpr (KC && SW) {
        const int row = r0 + (lane & 31);
        return __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(plane + row * BK + ((((lane >> 5) ^ (row >> 3)) & 1) << 3)));
    } else {
        return frag<ROWS, KC>(plane, r0, lane);
    }
}
// the limbs of -x are the negated limbs of x (round to nearest even is symmetric): flip the sign bits of a fragment
__device__ __forceinline__ bf16x8 neg_frag(bf16x8 v) {
    u32x4 u = __builtin_bit_cast(u32x4, v);
    u ^= 0x80008000u;
    return __builtin_bit_cast(bf16x8, u);
}

// 32 x 32 x 16 complex tile update on TWO accumulators: im += ai br, re += ar br; then the sign bits of the ai fragments are
// flipped in place (no registers for a negated copy) and re += (-ai) bi, im += ar bi.  Both halves alternate between the two
// accumulators.
template <int NP>
__device__ __forceinline__ void cmma_split2acc(const bf16x8* ar, bf16x8* ai, const bf16x8* br, const bf16x8* bi, f32x16& cre,
                                               f32x16& cim) {
    constexpr int NPROD = NP == 3 ? 6 : 3;
    constexpr int IA[6] = {1, 0, 2, 0, 1, 0}, IB[6] = {1, 2, 0, 1, 0, 0};
    constexpr int O = NP == 3 ? 0 : 3;
#pragma unroll
    for (int q = 0; q < NPROD; ++q) {
        const int a = IA[O + q], b = IB[O + q];
        cim = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ai[a], br[b], cim, 0, 0, 0);
        cre = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[a], br[b], cre, 0, 0, 0);
    }
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) ai[pl] = neg_frag(ai[pl]);
#pragma unroll
    for (int q = 0; q < NPROD; ++q) {
        const int a = IA[O + q], b = IB[O + q];
        cre = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ai[a], bi[b], cre, 0, 0, 0);
        cim = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[a], bi[b], cim, 0, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// complex kernel: tile 128 x 128, wave (g = wave >> 2, cs = wave & 3) owns columns [32 cs, 32 cs + 32) of row tiles g, g + 2
// ---------------------------------------------------------------------------------------------------------------
// RT = 4 ("tall"): tile 256 x 128, wave (g, cs) owns columns [32 cs, 32 cs + 32) of row tiles g, g + 2, g + 4, g + 6.  One k-step
// then ingests 48 KB of fp32 for 6 144 matrix-pipe cycles per SIMD (7.8 B/clk) where the 128 x 128 tile ingests 32 KB for 3 072
// (10.7 B/clk, more than a CU's memory path delivers: DESIGN.md section 4), the weight tile is read by half as many workgroups
// and its fragments are read from LDS once per four row tiles.  What makes it fit: the [row][k] limb images are unpadded and
// swizzled (two stages of a 256-row A and a 128-column B = 144-156 KB), and the complex product runs on TWO accumulators per
// tile (re += ar br + ai (-bi) with the sign bits of the bi fragment flipped once per k-step: 128 accumulator registers
// instead of 192), with one staging register set.
template <bool A_KC, bool B_KC, int NP, bool B_ILV, int RT = 2>
__global__ __launch_bounds__(NT2) void xcgemm2_kernel(const MkGemm p, int tilesM, int tilesN, float* __restrict__ ssq) {
    constexpr int BM = 64 * RT, BN = 128;
    constexpr bool TALL = RT == 4;
    static_assert(RT == 2 || (RT == 4 && A_KC), "row tiles per wave: 2, or 4 with a k-contiguous A");
    constexpr int PLA = plane_elems2<BM, A_KC, TALL>(), PLB = plane_elems2<BN, B_KC, TALL>();
    constexpr int STG = 2 * NP * (PLA + PLB);             // elements per stage: (re, im) x limbs x (A, B)
    static_assert(2 * STG * 2 <= 160 * 1024, "two stages must fit the LDS of a CU");
    __shared__ __attribute__((aligned(16))) u16 smem[2 * STG];

    const BlockCoord c = decode_block2<BM, BN>(p, tilesM, tilesN);
    if (!c.active) {
        if (ssq && threadIdx.x == 0) ssq[blockIdx.x] = 0.f;
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, cs = wave & 3;
    const int l31 = lane & 31, lh = lane >> 5;
#if MK_X2_DIAG & 32
    unsigned long long dg[7] = {0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_readcyclecounter();
    const unsigned long long tstart = tprev;
#endif
    const long long bo = c.b / p.inner, bi = c.b % p.inner;
    const float* Ab = p.A + bo * p.a_batch + bi * p.a_inner;
    const float* Bb = p.B + bo * p.b_batch + bi * p.b_inner;
    const float sgn_a = p.conj_a ? -1.f : 1.f;
    const float sgn_b = p.conj_b ? -1.f : 1.f;
    bool live[RT];
#pragma unroll
    for (int j = 0; j < RT; ++j) live[j] = c.i0 + 32 * (grp + 2 * j) < c.Meff;

    constexpr int NNG = TALL ? 1 : RT;                      // (the tall form has no separate accumulator for - ai bi)
    f32x16 cre[RT], cim[RT], cng[NNG];
#pragma unroll
    for (int j = 0; j < RT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            cre[j][r] = 0.f;
            cim[j][r] = 0.f;
            if (j < NNG) cng[j][r] = 0.f;
        }

    const int kt0 = c.klo / BK, kt1 = (c.khi + BK - 1) / BK;
    const int nk = kt1 - kt0;
    const int a_rmax = A_KC ? c.Meff : p.M;
    constexpr int D = TALL ? 1 : MK_X2_DEPTH;               // staging register sets: tile t rides in set t % D
    Stage2<BM, A_KC, BM / 4, TALL> sar[D], sai[D];
    Stage2<BN, B_KC, BN / 4, TALL> sbr[D], sbi[D];          // B_ILV: sbr carries both parts, sbi is unused
#pragma unroll
    for (int d = 0; d < D; ++d) {
        sar[d].init(Ab, p.a_row, p.a_k, c.i0, a_rmax, tid);
        if constexpr (!TALL) sai[d].init(Ab + p.a_im, p.a_row, p.a_k, c.i0, a_rmax, tid);
        if constexpr (B_ILV) {
            sbr[d].init(Bb, p.b_col, p.b_k, c.j0, p.N, tid, true);
        } else {
            sbr[d].init(Bb, p.b_col, p.b_k, c.j0, p.N, tid);
            if constexpr (!TALL) sbi[d].init(Bb + p.b_im, p.b_col, p.b_k, c.j0, p.N, tid);
        }
    }
    auto ld = [&](auto set, int kt) __attribute__((always_inline)) {
        constexpr int d = decltype(set)::value;
        sar[d].template load<false>(kt * BK, c.klo, c.khi, tid);
        if constexpr (TALL)
            sai[d].template load_from<false>(sar[d], p.a_im, kt * BK, c.klo, c.khi, tid);
        else
            sai[d].template load<false>(kt * BK, c.klo, c.khi, tid);
        if constexpr (B_ILV) {
            sbr[d].template load<true>(kt * BK, c.klo, c.khi, tid);
        } else {
            sbr[d].template load<false>(kt * BK, c.klo, c.khi, tid);
            if constexpr (TALL)
                sbi[d].template load_from<false>(sbr[d], p.b_im, kt * BK, c.klo, c.khi, tid);
            else
                sbi[d].template load<false>(kt * BK, c.klo, c.khi, tid);
        }
    };
    auto st = [&](auto set, int buf) __attribute__((always_inline)) {
        constexpr int d = decltype(set)::value;
        u16* Are = smem + buf * STG;
        u16* Aim = Are + NP * PLA;
        u16* Bre = Aim + NP * PLA;
        u16* Bim = Bre + NP * PLB;
        sar[d].template store<NP, PLA>(Are, tid, 1.f);
        sai[d].template store<NP, PLA>(Aim, tid, sgn_a);
        if constexpr (B_ILV) {
            sbr[d].template store_ilv<NP, PLB>(Bre, Bim, tid, sgn_b);
        } else {
            sbr[d].template store<NP, PLB>(Bre, tid, 1.f);
            sbi[d].template store<NP, PLB>(Bim, tid, sgn_b);
        }
    };
#if MK_X2_DIAG & 8
    bf16x8 d_br[NP], d_bim[NP], d_ar[RT][NP], d_ai[RT][NP];
    bool d_have = false;
#endif
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const u16* Are = smem + buf * STG;
        const u16* Aim = Are + NP * PLA;
        const u16* Bre = Aim + NP * PLA;
        const u16* Bim = Bre + NP * PLB;
        if (!live[0]) return;                               // row tile g + 2 is dead whenever row tile g is
#if MK_X2_DIAG & 8
        if (!d_have) {
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                d_br[pl] = frag2<BN, B_KC, TALL>(Bre + pl * PLB, cs * 32, lane);
                d_bim[pl] = frag2<BN, B_KC, TALL>(Bim + pl * PLB, cs * 32, lane);
#pragma unroll
                for (int j = 0; j < RT; ++j) {
                    d_ar[j][pl] = frag2<BM, A_KC, TALL>(Are + pl * PLA, (grp + 2 * j) * 32, lane);
                    d_ai[j][pl] = frag2<BM, A_KC, TALL>(Aim + pl * PLA, (grp + 2 * j) * 32, lane);
                }
            }
            d_have = true;
        }
#pragma unroll
        for (int j = 0; j < RT; ++j) {
            if (!live[j]) continue;
            prio_hi();
            cmma_split<NP>(d_ar[j], d_ai[j], d_br, d_bim, cre[j], cng[TALL ? 0 : j], cim[j]);
            prio_lo();
        }
#else
        bf16x8 br[NP], bim[NP];
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            br[pl] = frag2<BN, B_KC, TALL>(Bre + pl * PLB, cs * 32, lane);
            bim[pl] = frag2<BN, B_KC, TALL>(Bim + pl * PLB, cs * 32, lane);
        }
#pragma unroll
        for (int j = 0; j < RT; ++j) {
            if (!live[j]) continue;
            bf16x8 ar[NP], ai[NP];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                ar[pl] = frag2<BM, A_KC, TALL>(Are + pl * PLA, (grp + 2 * j) * 32, lane);
                ai[pl] = frag2<BM, A_KC, TALL>(Aim + pl * PLA, (grp + 2 * j) * 32, lane);
            }
            // (ar + i ai)(br + i bi): re = ar br - ai bi (second part accumulated apart), im = ar bi + ai br
            MK_X2_STAMP(1);
            prio_hi();
            if constexpr (TALL)
                cmma_split2acc<NP>(ar, ai, br, bim, cre[j], cim[j]);
            else
                cmma_split<NP>(ar, ai, br, bim, cre[j], cng[j], cim[j]);
            prio_lo();
            MK_X2_STAMP(2);
        }
#endif
    };
    auto produce = [&](auto set, int i) __attribute__((always_inline)) {                  // tile kt0 + i + 1 sits in staging set `set` = (i + 1) % D
        if (!(MK_X2_DIAG & 1) && i + 1 < nk) st(set, (i + 1) & 1);
        if (!(MK_X2_DIAG & 4) && i + 1 + D < nk) ld(set, kt0 + i + 1 + D);
    };

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, D - 1>;          // set of the odd tiles (= S0 when D == 1)
    if (nk > 0) {
        ld(S0{}, kt0);
        st(S0{}, 0);
        if (nk > 1) ld(S1{}, kt0 + 1);
        if (D == 2 && nk > 2) ld(S0{}, kt0 + 2);
    }
    __syncthreads();
    auto step = [&](auto set, int i) __attribute__((always_inline)) {
#if MK_X2_MIDBAR
        if (grp == 0) { if (!(MK_X2_DIAG & 2)) compute(i & 1); } else produce(set, i);
        __syncthreads();
        if (grp == 1) { if (!(MK_X2_DIAG & 2)) compute(i & 1); } else produce(set, i);
#else
        // one barrier per k-step; the groups run the two segments in opposite order, so that on every SIMD one wave
        // starts on the matrix pipe while its partner starts on the VALU / memory path.
        // (`grp` is a per-lane value to the compiler, so the two orders are laid out one after the other under exec masks;
        // hipcc's wait-count pass then takes the loads of the masked-off path for the newest ones in flight on the same
        // staging registers and every split waits for ALL loads in flight (vmcnt(3..0)): the effective prefetch distance is
        // one k-step.  Branching on readfirstlane(grp) gives exact counts but the register allocator then spills the
        // accumulators (384-612 bytes of scratch in every instantiation).  ONE LOOP PER GROUP (if (grp == 0) { loop of
        // produce, compute, barrier } else { loop of compute, produce, barrier }, split and loads unconditional) gives exact
        // counts without spilling — vmcnt(7, 7, 6, 5, 4): only the set requested one step ago may still be in flight, a true
        // distance of two k-steps — and runs at the same speed (dhconv fwd 0.281 -> 0.275 ms, wgrad 0.237 -> 0.241,
        // profiles/r03_ab_twoloops.txt): the wait in front of the split is not latency, the kernel is bound by the bytes a
        // CU ingests per clock (32 KB per k-step against 3072 MFMA cycles per SIMD, tools/x2_diag.py))
        if (grp == 0) {
            produce(set, i);
            MK_X2_STAMP(0);
            if (!(MK_X2_DIAG & 2)) compute(i & 1);
        } else {
            if (!(MK_X2_DIAG & 2)) compute(i & 1);
            produce(set, i);
            MK_X2_STAMP(0);
        }
#endif
        __syncthreads();
        MK_X2_STAMP(3);
    };
    MK_X2_STAMP(4);
    for (int i = 0; i < nk; i += 2) {                       // step i splits tile i + 1, which rides in set (i + 1) % D
        step(S1{}, i);
        if (i + 1 < nk) step(S0{}, i + 1);
    }

    float* Cb = p.C + bo * p.c_batch + bi * p.c_inner;
    const int col = c.j0 + cs * 32 + l31;
    float sq = 0.f;                                         // sum of re^2 + im^2 over the entries this lane writes (ssq)
#pragma unroll
    for (int j = 0; j < RT; ++j) {
        if (!live[j]) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = c.i0 + (grp + 2 * j) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row < c.Meff && col < p.N && (!(MK_X2_DIAG & 16) || p.K == -12345)) {
                float* dr = Cb + (long long)row * p.c_row + (long long)col * p.c_col;
                float* di = dr + p.c_im;
                float vr = TALL ? cre[j][r] : cre[j][r] - cng[TALL ? 0 : j][r], vi = cim[j][r];
                if (p.beta) {
                    vr += *dr;
                    vi += *di;
                }
                sq = fmaf(vr, vr, fmaf(vi, vi, sq));
                if (p.c_col == 2) {          // interleaved complex C: one 8-byte store per entry
#if MK_X2_WGRAD_NT
                    typedef float f32x2_t __attribute__((ext_vector_type(2)));
                    __builtin_nontemporal_store(f32x2_t{vr, vi}, reinterpret_cast<f32x2_t*>(dr));
#else
                    *reinterpret_cast<float2*>(dr) = make_float2(vr, vi);
#endif
                } else if constexpr (MK_X2_ST_NT) {
                    __builtin_nontemporal_store(vr, dr);
                    __builtin_nontemporal_store(vi, di);
                } else {
                    *dr = vr;
                    *di = vi;
                }
            }
        }
    }
    // Sum of squares of everything this workgroup wrote, in a fixed order (lanes by shuffle tree, waves 0..7 in sequence): the
    // weight gradient's contribution to the global gradient norm, so that the clipping pass need not read 283 MB per layer again
    if (ssq) {                                              // (kernel argument: uniform)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_down(sq, o, 64);
        float* red = reinterpret_cast<float*>(smem);        // the last k-step ended with a barrier: the stages are dead
        if (lane == 0) red[wave] = sq;
        __syncthreads();
        if (tid == 0) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NT2 / 64; ++w) t += red[w];
            ssq[blockIdx.x] = t;
        }
    }
#if MK_X2_DIAG & 32
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MK_X2_STAMP(5);
    dg[6] = tprev - tstart;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) atomicAdd(&g_x2_diag[grp][k], dg[k]);
        atomicAdd(&g_x2_diag[grp][7], 1ull);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// real kernel, A pre-split: tile 256 x 256, wave (g, cs) owns columns [64 cs, 64 cs + 64) of the live row tiles
// t0 + g, t0 + g + 2, t0 + g + 4, t0 + g + 6 (t0 = first live 32-row tile of this block)
//   A: NP bf16 limb planes, [k][row] with the row index contiguous (pitch pl_k elements, a multiple of 8; rows past M
//      inside the pitch hold zeros), plane q at Apl + q * pl_stride, batch b at + b * pl_batch
//   B: fp32, [k][col] with the column index contiguous (the F / S layouts), split on the fly
// ---------------------------------------------------------------------------------------------------------------
struct PreA {
    const u16* planes;
    long long pl_stride, pl_batch, pl_k;
    int rows_valid;            // rows readable per k-row (multiple of 8)
    // optional numerical band of the constant matrix per batch (device arrays of `batch` ints, or null): outside
    // [lo[b], hi[b]) every entry of A[b] is below the caller's threshold (the Legendre functions of order m vanish
    // towards the poles like sin^m).  mode 1: the band is a range of k (analysis: contraction over latitude) ->
    // the k-loop is clipped; mode 2: a range of rows (synthesis: output latitudes) -> rows outside are written as
    // exact zeros without being computed.  Skips A AND B traffic of the dead part.
    const int* band_lo;
    const int* band_hi;
    int band_mode;
};

// NARROW (round 4; the ERA5-shaped transform of BASELINE's "fwd SHT GB/s" metric: 73 channels = 152 columns): at most 160
// columns exist.  In the column-slice mapping above the waves of column slices 2 and 3 would then multiply (mostly) padding —
// 146 TF dense-equivalent at C = 73 against 272 TF at C = 384.  Here wave w owns ROW tile t0 + w (all eight live row tiles of the
// block, counted from the first live one, so that the triangle still shortens the waves' work evenly) x the five 32-column
// tiles that exist: 30 MFMAs per wave and k-step instead of 48 on the busiest SIMD, and the data operand is split for 160
// columns instead of 256 (whole waves skip the columns that do not exist).  Same arithmetic and accumulation order per
// accumulator as the wide form.
template <int NP, bool NARROW>
__global__ __launch_bounds__(NT2) void xgemm2_kernel(const MkGemm p, const PreA a, int tilesM, int tilesN) {
    constexpr int BM = 256, BN = 256;
    constexpr int NCT = 5;                                 // NARROW: live 32-column tiles
    constexpr int PR = BM + 32;                            // pitch of the [k][row] limb tiles (both operands)
    constexpr int PL = BK * PR;                            // elements per limb plane
    constexpr int STG = 2 * NP * PL;
    __shared__ __attribute__((aligned(16))) u16 smem[2 * STG];

    BlockCoord c = decode_block2<BM, BN>(p, tilesM, tilesN);
    if (!c.active) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, cs = wave & 3;
    const int l31 = lane & 31, lh = lane >> 5;
    const long long bo = c.b / p.inner, bi = c.b % p.inner;
    const float* Bb = p.B + bo * p.b_batch + bi * p.b_inner;
    const u16* Ab = a.planes + bo * a.pl_batch;        // the constant matrix (and its band) belong to the OUTER batch index

    // rows of this block that are stored: 32-row tiles [s0, s1); rows that are computed: tiles [t0, t1) inside them
    const int rows_end = min(c.Meff, p.M) - c.i0;          // rows of this block that exist
    int s0 = 0;
    if (p.tri_mode == MK_TRI_ROW_GE) s0 = max(0, (c.b / p.inner + p.tri_off - c.i0) / 32);
    const int s1 = min(BM / 32, (rows_end + 31) / 32);
    int t0 = s0, t1 = s1;
    if (a.band_mode == 1) {
        c.klo = max(c.klo, a.band_lo[bo]);
        c.khi = min(c.khi, a.band_hi[bo]);
        if (c.khi < c.klo) c.khi = c.klo;
    } else if (a.band_mode == 2) {
        t0 = max(s0, (a.band_lo[bo] - c.i0) >> 5);                     // floor (arithmetic shift of a possibly negative value)
        t1 = min(s1, (a.band_hi[bo] - c.i0 + 31) >> 5);
        if (t1 < t0) t1 = t0;
    }
    int tile[4];
    bool live[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        tile[j] = NARROW ? t0 + wave : t0 + grp + 2 * j;   // NARROW: one row tile per wave (slot 0)
        live[j] = tile[j] < t1 && (!NARROW || j == 0);
    }

    // wide: acc[j][n] = row tile j x column tile n of the wave's slice; NARROW: acc[0 .. 4] (flattened) = the five column tiles
    f32x16 acc[NARROW ? 3 : 4][2];
#pragma unroll
    for (int j = 0; j < (NARROW ? 3 : 4); ++j)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][n][r] = 0.f;

    const int kt0 = c.klo / BK, kt1 = (c.khi + BK - 1) / BK;
    const int nk = kt1 - kt0;
    constexpr int D = MK_X2_DEPTH_R;
    Stage2<BN, false, NARROW ? NCT * 8 : BN / 4> sb[D];
#pragma unroll
    for (int d = 0; d < D; ++d) sb[d].init(Bb, p.b_col, p.b_k, c.j0, p.N, tid);
    // A limb vectors of this thread: k-row ak (0..15), rows [ar0, ar0 + 8) of the block
    const int ak = tid >> 5, ar0 = (tid & 31) * 8;
    const bool a_ok = c.i0 + ar0 < a.rows_valid && ar0 < 32 * t1 && ar0 + 8 > 32 * t0;
    const u16* ap = Ab + (long long)ak * a.pl_k + c.i0 + ar0;
    u32x4 av[D][NP];
    auto ld = [&](auto set, int kt) __attribute__((always_inline)) {
        constexpr int d = decltype(set)::value;
        sb[d].template load<false>(kt * BK, c.klo, c.khi, tid);
        const int k = kt * BK + ak;
        const bool ok = a_ok && k < p.K;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const g_u32x4* src = ok ? (const g_u32x4*)(ap + (long long)kt * BK * a.pl_k + q * a.pl_stride) : (const g_u32x4*)g_zero32;
            av[d][q] = src[0];
        }
    };
    auto st = [&](auto set, int buf) __attribute__((always_inline)) {
        constexpr int d = decltype(set)::value;
        u16* As = smem + buf * STG;
        u16* Bs = As + NP * PL;
        sb[d].template store<NP, PL>(Bs, tid, 1.f);
#pragma unroll
        for (int q = 0; q < NP; ++q) *reinterpret_cast<u32x4*>(As + q * PL + ak * PR + ar0) = av[d][q];
    };
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const u16* As = smem + buf * STG;
        const u16* Bs = As + NP * PL;
        if (!live[0]) return;
        if constexpr (NARROW) {
            bf16x8 bn[NCT][NP], an[NP];
#pragma unroll
            for (int n = 0; n < NCT; ++n)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) bn[n][pl] = frag<BN, false>(Bs + pl * PL, n * 32, lane);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) an[pl] = frag<BM, false>(As + pl * PL, tile[0] * 32, lane);
            constexpr int NPROD = NP == 3 ? 6 : 3;
            constexpr int IA[6] = {1, 0, 2, 0, 1, 0}, IB[6] = {1, 2, 0, 1, 0, 0};      // smallest terms first (as mma_split)
            constexpr int O = NP == 3 ? 0 : 3;
            prio_hi();
#pragma unroll
            for (int q = 0; q < NPROD; ++q)
#pragma unroll
                for (int n = 0; n < NCT; ++n)
                    acc[n >> 1][n & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(an[IA[O + q]], bn[n][IB[O + q]], acc[n >> 1][n & 1], 0, 0, 0);
            prio_lo();
            return;
        }
        bf16x8 bf[2][NP];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) bf[n][pl] = frag<BN, false>(Bs + pl * PL, cs * 64 + n * 32, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!live[j]) continue;
            bf16x8 af[NP];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) af[pl] = frag<BM, false>(As + pl * PL, tile[j] * 32, lane);
            prio_hi();
            rmma_split2<NP>(af, bf[0], bf[1], acc[j][0], acc[j][1]);
            prio_lo();
        }
    };
    auto produce = [&](auto set, int i) __attribute__((always_inline)) {                  // tile kt0 + i + 1 sits in staging set `set` = (i + 1) % D
        if (!(MK_X2_DIAG & 1) && i + 1 < nk) st(set, (i + 1) & 1);
        if (!(MK_X2_DIAG & 4) && i + 1 + D < nk) ld(set, kt0 + i + 1 + D);
    };

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, D - 1>;          // set of the odd tiles (= S0 when D == 1)
    if (nk > 0) {
        ld(S0{}, kt0);
        st(S0{}, 0);
        if (nk > 1) ld(S1{}, kt0 + 1);
        if (D == 2 && nk > 2) ld(S0{}, kt0 + 2);
    }
    __syncthreads();
    auto step = [&](auto set, int i) __attribute__((always_inline)) {
#if MK_X2_MIDBAR
        if (grp == 0) { if (!(MK_X2_DIAG & 2)) compute(i & 1); } else produce(set, i);
        __syncthreads();
        if (grp == 1) { if (!(MK_X2_DIAG & 2)) compute(i & 1); } else produce(set, i);
#else
        // one barrier per k-step; the groups run the two segments in opposite order, so that on every SIMD one wave
        // starts on the matrix pipe while its partner starts on the VALU / memory path
        if (grp == 0) {
            produce(set, i);
            if (!(MK_X2_DIAG & 2)) compute(i & 1);
        } else {
            if (!(MK_X2_DIAG & 2)) compute(i & 1);
            produce(set, i);
        }
#endif
        __syncthreads();
    };
    for (int i = 0; i < nk; i += 2) {                       // step i splits tile i + 1, which rides in set (i + 1) % D
        step(S1{}, i);
        if (i + 1 < nk) step(S0{}, i + 1);
    }

    float* Cb = p.C + bo * p.c_batch + bi * p.c_inner;
    if (a.band_mode == 2 && !p.beta) {                     // stored rows outside the band: exact zeros
        for (int t = s0 + (NARROW ? wave : grp); t < s1; t += (NARROW ? 8 : 2)) {
            if (t >= t0 && t < t1) continue;
#pragma unroll
            for (int n = 0; n < (NARROW ? NCT : 2); ++n) {
                const int col = NARROW ? c.j0 + n * 32 + l31 : c.j0 + cs * 64 + n * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = c.i0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < c.Meff && col < p.N) Cb[(long long)row * p.c_row + col] = 0.f;
                }
            }
        }
    }
    if constexpr (NARROW) {
        if (live[0]) {
#pragma unroll
            for (int n = 0; n < NCT; ++n) {
                const int col = c.j0 + n * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = c.i0 + tile[0] * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < c.Meff && col < p.N) {
                        float* dst = Cb + (long long)row * p.c_row + col;
                        float val = acc[n >> 1][n & 1][r];
                        if (p.beta) val += *dst;
                        if constexpr (MK_X2_ST_NT) __builtin_nontemporal_store(val, dst);
                        else *dst = val;
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (!live[j]) continue;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int col = c.j0 + cs * 64 + n * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = c.i0 + tile[j] * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < c.Meff && col < p.N) {
                    float* dst = Cb + (long long)row * p.c_row + col;
                    float val = acc[j][n][r];
                    if (p.beta) val += *dst;
                    if constexpr (MK_X2_ST_NT) __builtin_nontemporal_store(val, dst);
                    else *dst = val;
                }
            }
        }
    }
}

template <int NP>
int launch_cplx2(const MkGemm* g, bool a_kc, bool b_kc, bool b_ilv, hipStream_t s, float* ssq = nullptr) {
    // more than 128 rows and a k-contiguous A (the dhconv forward / data gradient: A = the coefficients of one degree, rows = orders):
    // the 256 x 128 form.  MAKANI_AMD_X2_TALL=0 keeps the 128 x 128 tile everywhere
    static const bool tall_ok = [] { const char* e = getenv("MAKANI_AMD_X2_TALL"); return !(e && e[0] == '0'); }();
    const bool tall = tall_ok && a_kc && g->M > 128;
    const int BM = tall ? 256 : 128, BN = 128;
    const int tm = (g->M + BM - 1) / BM, tn = (g->N + BN - 1) / BN;
    const long long nb = (long long)((g->batch + MK_NUM_XCD - 1) / MK_NUM_XCD) * MK_NUM_XCD * tm * tn;
    MK_REQUIRE(nb < (1ll << 31), "xcgemm2: grid too large");
    dim3 grid((unsigned)nb), block(NT2);
#define MK_XC2(AK, BK_, IL) hipLaunchKernelGGL((xcgemm2_kernel<AK, BK_, NP, IL>), grid, block, 0, s, *g, tm, tn, ssq)
#define MK_XC2T(BK_, IL) hipLaunchKernelGGL((xcgemm2_kernel<true, BK_, NP, IL, 4>), grid, block, 0, s, *g, tm, tn, ssq)
    if (b_ilv) {
        MK_REQUIRE(a_kc, "xcgemm2: an interleaved B operand needs a k-contiguous A");
        if (tall) {
            if (b_kc)
                MK_XC2T(true, true);
            else
                MK_XC2T(false, true);
        } else if (b_kc)
            MK_XC2(true, true, true);
        else
            MK_XC2(true, false, true);
    } else if (tall) {
        if (b_kc)
            MK_XC2T(true, false);
        else
            MK_XC2T(false, false);
    } else if (a_kc && b_kc)
        MK_XC2(true, true, false);
    else if (a_kc && !b_kc)
        MK_XC2(true, false, false);
    else if (!a_kc && b_kc)
        MK_XC2(false, true, false);
    else
        MK_XC2(false, false, false);
#undef MK_XC2
#undef MK_XC2T
    return mk_check_launch("mk_cgemm_split2_batched");
}

}  // namespace

#if MK_X2_DIAG & 32
extern "C" int mk_x2_diag_read(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_x2_diag), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_x2_diag), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

extern "C" int mk_cgemm_split2_batched(const MkGemm* g, int limbs, void* stream) {
    bool a_kc, b_kc, b_ilv;
    int rc = validate(g, true, &a_kc, &b_kc, true, &b_ilv);
    if (rc) return rc;
    MK_REQUIRE(limbs == 2 || limbs == 3, "split gemm: limbs must be 2 or 3");
    hipStream_t s = (hipStream_t)stream;
    return limbs == 3 ? launch_cplx2<3>(g, a_kc, b_kc, b_ilv, s) : launch_cplx2<2>(g, a_kc, b_kc, b_ilv, s);
}

// grid of launch_cplx2 (one partial per workgroup)
static long long cplx2_blocks(const MkGemm* g, bool a_kc) {
    static const bool tall_ok = [] { const char* e = getenv("MAKANI_AMD_X2_TALL"); return !(e && e[0] == '0'); }();
    const bool tall = tall_ok && a_kc && g->M > 128;
    const int BM = tall ? 256 : 128, BN = 128;
    const int tm = (g->M + BM - 1) / BM, tn = (g->N + BN - 1) / BN;
    return (long long)((g->batch + MK_NUM_XCD - 1) / MK_NUM_XCD) * MK_NUM_XCD * tm * tn;
}

extern "C" long long mk_cgemm_split2_ssq_count(const MkGemm* g) {
    bool a_kc, b_kc, b_ilv;
    if (validate(g, true, &a_kc, &b_kc, true, &b_ilv)) return -1;
    return cplx2_blocks(g, a_kc);
}

extern "C" int mk_cgemm_split2_batched_ssq(const MkGemm* g, int limbs, float* ssq_part, void* stream) {
    bool a_kc, b_kc, b_ilv;
    int rc = validate(g, true, &a_kc, &b_kc, true, &b_ilv);
    if (rc) return rc;
    MK_REQUIRE(limbs == 2 || limbs == 3, "split gemm: limbs must be 2 or 3");
    MK_REQUIRE(ssq_part, "split gemm with sums of squares: null partial buffer");
    hipStream_t s = (hipStream_t)stream;
    return limbs == 3 ? launch_cplx2<3>(g, a_kc, b_kc, b_ilv, s, ssq_part) : launch_cplx2<2>(g, a_kc, b_kc, b_ilv, s, ssq_part);
}

extern "C" int mk_sgemm_presplit_batched(const MkGemm* g, const void* a_planes, long long pl_stride, long long pl_batch,
                                         long long pl_k, int limbs, const int* band_lo, const int* band_hi, int band_mode,
                                         void* stream) {
    MK_REQUIRE(g && a_planes && g->B && g->C, "presplit gemm: null pointer");
    MK_REQUIRE(g->M > 0 && g->N > 0 && g->K >= 0 && g->batch > 0, "presplit gemm: bad extents");
    MK_REQUIRE(limbs == 2 || limbs == 3, "presplit gemm: limbs must be 2 or 3");
    MK_REQUIRE(g->inner >= 1 && g->batch % g->inner == 0 && (g->b_inner & 3) == 0, "presplit gemm: inner must divide batch, b_inner % 4 == 0");
    MK_REQUIRE(g->b_col == 1 && (g->b_k & 3) == 0 && g->b_k >= ((g->N + 3) & ~3), "presplit gemm: B must be [k][col], col contiguous");
    MK_REQUIRE((g->b_batch & 3) == 0 && ((uintptr_t)g->B & 15) == 0, "presplit gemm: B alignment");
    MK_REQUIRE(g->c_col == 1, "presplit gemm: c_col must be 1");
    MK_REQUIRE((pl_k & 7) == 0 && (pl_batch & 7) == 0 && (pl_stride & 7) == 0 && ((uintptr_t)a_planes & 15) == 0 && pl_k >= g->M,
               "presplit gemm: A limb planes need 16-byte aligned k-rows of at least M elements");
    constexpr int BM = 256, BN = 256;
    const int tm = (g->M + BM - 1) / BM, tn = (g->N + BN - 1) / BN;
    const long long nb = (long long)((g->batch + MK_NUM_XCD - 1) / MK_NUM_XCD) * MK_NUM_XCD * tm * tn;
    MK_REQUIRE(nb < (1ll << 31), "presplit gemm: grid too large");
    MK_REQUIRE(band_mode == 0 || ((band_mode == 1 || band_mode == 2) && band_lo && band_hi),
               "presplit gemm: band_mode must be 0, or 1 / 2 with both band arrays");
    PreA a{(const u16*)a_planes, pl_stride, pl_batch, pl_k, (int)(pl_k & ~7ll), band_lo, band_hi, band_mode};
    dim3 grid((unsigned)nb), block(NT2);
    hipStream_t s = (hipStream_t)stream;
    // at most 160 columns (one column tile, five 32-column MFMA tiles): the row-tile-per-wave form
    if (g->N <= 160) {
        if (limbs == 3)
            hipLaunchKernelGGL((xgemm2_kernel<3, true>), grid, block, 0, s, *g, a, tm, tn);
        else
            hipLaunchKernelGGL((xgemm2_kernel<2, true>), grid, block, 0, s, *g, a, tm, tn);
    } else if (limbs == 3)
        hipLaunchKernelGGL((xgemm2_kernel<3, false>), grid, block, 0, s, *g, a, tm, tn);
    else
        hipLaunchKernelGGL((xgemm2_kernel<2, false>), grid, block, 0, s, *g, a, tm, tn);
    return mk_check_launch("mk_sgemm_presplit_batched");
}
