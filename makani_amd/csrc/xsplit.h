// Tile staging, limb splitting and MFMA fragments shared by the split-bf16 GEMM engines (xgemm.hip, xgemm2.hip).
// See xgemm.hip for the numerics of the bf16 limb expansion.
#pragma once
#include "gemm_common.h"

namespace xsplit {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int BK = 16;
constexpr int PK = 24;   // pitch (elements) of [row][k] limb tiles: 48 B -> conflict-free ds_read_b128

__device__ __forceinline__ bool in_range(int k, int lo, int hi) { return k >= lo && k < hi; }

// ---- the limb split on packed instructions ---------------------------------------------------------------------------
// Four consecutive fp32 values (two register pairs) -> NP bf16 limbs each, limb p of the four as one 8-byte LDS store in
// plane p.  Per pair and limb: one v_cvt_pk_bf16_f32 (round to nearest even, result already packed for the store), two
// integer instructions that widen the two limbs back to fp32, one v_pk_add_f32 for the residual: 5 VALU instructions per
// value for three limbs.  The element-at-a-time form of the same arithmetic compiled to 9 (one conversion per element
// with the second source unused, scalar subtractions, a re-pack) and dominated the engines, which are bound by VALU issue.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t limb_pair(f32x2 a) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf16x2)); }
__device__ __forceinline__ f32x2 widen_pair(uint32_t p) {
    const f32x2 r = {__uint_as_float(p << 16), __uint_as_float(p & 0xffff0000u)};
    return r;
}

template <int NP, int PLANE>
__device__ __forceinline__ void split_store4(u16* lds, int off, f32x2 a, f32x2 b) {
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
        const uint32_t pa = limb_pair(a), pb = limb_pair(b);
        *reinterpret_cast<uint2*>(lds + pl * PLANE + off) = make_uint2(pa, pb);
        if (pl + 1 < NP) {
            a -= widen_pair(pa);
            b -= widen_pair(pb);
        }
    }
}

// element masks of a vector (4 bits of `keep`): only tiles cut by the contraction range carry them
__device__ __forceinline__ f32x4 mask4(f32x4 r, unsigned bits) {
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = ((bits >> e) & 1u) ? r[e] : 0.f;
    return r;
}

// fp32 tile staging registers (same addressing as sgemm.hip's TileStage)
template <int ROWS, bool KC, int NT = 256>
struct Stage {
    static constexpr int NV = (ROWS * BK / 4) / NT;
    static_assert(NV >= 1, "tile too small");
    f32x4 v[NV];
    unsigned keep;      // KC: 4 bits per vector = elements inside [klo, khi); applied when the tile is stored, NOT
                        // on the freshly loaded registers (that would put an s_waitcnt right behind every load)
    bool interior;      // uniform: the tile lies inside [klo, khi), no element masks to apply
    // loop-invariant addressing, set up once per tile by init(): pointer of every vector at k = 0 and whether its
    // row exists.  Inside the k-loop a load is then "pointer + uniform offset" (the 64-bit index arithmetic and the
    // row checks done per load used to cost as much as a third of the kernel)
    const float* p0[NV];
    unsigned ok;
    long long kstride;  // floats per unit of k

    __device__ __forceinline__ void init(const float* __restrict__ base, long long rs, long long ks, int r0, int rmax,
                                         int tid) {
        ok = 0u;
        kstride = KC ? 1 : ks;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int f = tid + q * NT;
            if constexpr (KC) {
                const int row = f >> 2, kq = f & 3;
                p0[q] = base + (long long)(r0 + row) * rs + kq * 4;
                ok |= (r0 + row < rmax ? 1u : 0u) << q;
            } else {
                constexpr int RQ = ROWS / 4;
                const int kk = f / RQ, rq = f % RQ;
                p0[q] = base + (long long)kk * ks + (r0 + rq * 4);
                ok |= (r0 + rq * 4 < rmax ? 1u : 0u) << q;
            }
        }
    }

    // interleaved complex operand ((re, im) pairs, strides rs / ks in floats): this stage holds the addresses, its
    // partner `im` receives the odd floats (load_ilv)
    __device__ __forceinline__ void init_ilv(const float* __restrict__ base, long long rs, long long ks, int r0, int rmax,
                                             int tid) {
        ok = 0u;
        kstride = KC ? 2 : ks;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int f = tid + q * NT;
            if constexpr (KC) {
                const int row = f >> 2, kq = f & 3;
                p0[q] = base + (long long)(r0 + row) * rs + kq * 8;
                ok |= (r0 + row < rmax ? 1u : 0u) << q;
            } else {
                constexpr int RQ = ROWS / 4;
                const int kk = f / RQ, rq = f % RQ;
                p0[q] = base + (long long)kk * ks + (long long)(r0 + rq * 4) * 2;
                ok |= (r0 + rq * 4 < rmax ? 1u : 0u) << q;
            }
        }
    }

    __device__ __forceinline__ void load_pair(int q, long long koff, Stage& im) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p0[q] + koff);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p0[q] + koff + 4);
        v[q] = f32x4{a[0], a[2], b[0], b[2]};
        im.v[q] = f32x4{a[1], a[3], b[1], b[3]};
    }

    // load() of an interleaved operand into the (re = *this, im) pair of stages
    __device__ __forceinline__ void load_ilv(Stage& im, int k0, int klo, int khi, int tid) {
        const long long koff = (long long)k0 * kstride;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        interior = im.interior = k0 >= klo && k0 + BK <= khi;
        if (interior) {
            keep = im.keep = 0xffffffffu;
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                v[q] = zero;
                im.v[q] = zero;
                if ((ok >> q) & 1u) load_pair(q, koff, im);
            }
            return;
        }
        keep = 0u;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int f = tid + q * NT;
            v[q] = zero;
            im.v[q] = zero;
            if constexpr (KC) {
                const int k = k0 + (f & 3) * 4;
                if (((ok >> q) & 1u) && k < khi && k + 3 >= klo) {
                    load_pair(q, koff, im);
                    unsigned m = 0u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) m |= in_range(k + e, klo, khi) ? (1u << e) : 0u;
                    keep |= m << (4 * q);
                }
            } else {
                constexpr int RQ = ROWS / 4;
                const int k = k0 + f / RQ;
                if (((ok >> q) & 1u) && in_range(k, klo, khi)) load_pair(q, koff, im);
            }
        }
        im.keep = keep;
    }

    // tile [k0, k0 + BK) of the operand, zero outside [klo, khi)
    __device__ __forceinline__ void load(int k0, int klo, int khi, int tid) {
        const long long koff = (long long)k0 * kstride;                 // uniform
        interior = k0 >= klo && k0 + BK <= khi;
        if (interior) {                                                 // interior tile (uniform branch)
            keep = 0xffffffffu;
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                f32x4 val = {0.f, 0.f, 0.f, 0.f};
                if ((ok >> q) & 1u) val = *reinterpret_cast<const f32x4*>(p0[q] + koff);
                v[q] = val;
            }
            return;
        }
        keep = 0u;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int f = tid + q * NT;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if constexpr (KC) {
                const int k = k0 + (f & 3) * 4;
                if (((ok >> q) & 1u) && k < khi && k + 3 >= klo) {
                    val = *reinterpret_cast<const f32x4*>(p0[q] + koff);
                    unsigned m = 0u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) m |= in_range(k + e, klo, khi) ? (1u << e) : 0u;
                    keep |= m << (4 * q);
                }
            } else {
                constexpr int RQ = ROWS / 4;
                const int k = k0 + f / RQ;
                if (((ok >> q) & 1u) && in_range(k, klo, khi)) val = *reinterpret_cast<const f32x4*>(p0[q] + koff);
            }
            v[q] = val;
        }
    }

    // split into NP bf16 limbs and store; limb plane p lives at lds + p * PLANE (elements)
    template <int NP, int PLANE>
    __device__ __forceinline__ void store(u16* lds, int tid, float sign) const {
        constexpr int PR = ROWS + 32;   // pitch of [k][row] tiles
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int f = tid + q * NT;
            int off;
            if constexpr (KC) {
                const int row = f >> 2, kq = f & 3;
                off = row * PK + kq * 4;
            } else {
                constexpr int RQ = ROWS / 4;
                const int kk = f / RQ, rq = f % RQ;
                off = kk * PR + rq * 4;
            }
            f32x4 r = v[q] * sign;
            if constexpr (KC) {
                if (!interior) r = mask4(r, keep >> (4 * q));
            }
            split_store4<NP, PLANE>(lds, off, r.xy, r.zw);
        }
    }
};

template <int ROWS, bool KC>
constexpr int plane_elems() {
    return KC ? ROWS * PK : BK * (ROWS + 32);
}

// MFMA operand fragment (8 consecutive k for row/col `r0 + (lane & 31)`, k-block lane>>5) of limb plane
template <int ROWS, bool KC>
__device__ __forceinline__ bf16x8 frag(const u16* plane, int r0, int lane) {
    if constexpr (KC) {
        return __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(plane + (r0 + (lane & 31)) * PK + (lane >> 5) * 8));
    } else {
        constexpr int PR = ROWS + 32;
        const int s = lane & 15, g1 = (lane >> 4) & 1, lh = lane >> 5;
        const u16* q0 = plane + (lh * 8 + (s >> 2)) * PR + r0 + g1 * 16 + (s & 3) * 4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q0 + 4 * PR));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
}

// acc += sum over the limb products kept for NP limbs
template <int NP>
__device__ __forceinline__ f32x16 mma_split(const bf16x8* a, const bf16x8* b, f32x16 acc) {
    if constexpr (NP == 3) {   // smallest terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    return acc;
}

}  // namespace xsplit
