// fp32-MFMA strided batched GEMM engine (real + planar complex) for gfx950.
//
//   C[b][i][j] (+)= sum_k A[b][i][k] * B[b][j][k]
//
// This one engine carries every fp32-matrix op of the SFNO spectral path:
//   Legendre analysis / synthesis and their adjoints  (real, batched over m)
//   dhconv channel contraction fwd / dgrad / wgrad    (complex, batched over l)
// exploiting the triangular structure P_l^m = 0 (l < m) through tri_mode.
//
// Design (CDNA4):
//   * v_mfma_f32_32x32x2_f32 — exact fp32 (== fmaf chain), 64 cycles / SIMD.  A/B operands are
//     ONE VGPR per lane with no k-contiguity requirement, so both operand layouts
//     (k-contiguous "KC" and row-contiguous) are served by the same k-major LDS image
//     As[k][row]; half-waves read 32 consecutive floats -> conflict-free ds_read_b32.
//   * 256 threads = 4 waves, each wave owns a 64x64 (real) / 32x64 (complex) output tile =
//     2x2 / 1x2 MFMA tiles; BK = 16, double-buffered LDS, register-staged global prefetch,
//     one __syncthreads per k-tile.  The kernel is MFMA-bound by construction: per k-step a
//     wave issues 4 ds_read_b32 for 4 (real) / 6 for 8 (complex) MFMAs of 64 cycles each.
//   * XCD-aware grid: batch b runs on XCD b % 8 (all its tiles share A[b]/B[b] in that XCD's L2),
//     and every XCD gets every 8th batch so the triangular work is balanced.
#include "gemm_common.h"

namespace {

using gemm::BlockCoord;
using gemm::decode_block;
using gemm::validate;

constexpr int BK = 16;
constexpr int NT = 256;

__device__ __forceinline__ bool in_range(int k, int lo, int hi) { return k >= lo && k < hi; }

// ---- global -> register -> LDS tile staging ------------------------------------
template <int ROWS, bool KC>
struct TileStage {
    static constexpr int NV = (ROWS * BK / 4) / NT;  // float4 per thread
    static_assert(NV >= 1, "tile too small");
    f32x4 v[NV];
    unsigned keep;      // KC: 4 bits per vector = elements inside [klo, khi), applied in store() so that no
                        // s_waitcnt lands right behind the loads

    // base already includes the batch offset.  rs = row stride, ks = k stride (the other one is 1).
    __device__ __forceinline__ void load(const float* __restrict__ base, long long rs, long long ks, int r0, int rmax,
                                         int k0, int klo, int khi, int tid) {
        keep = 0u;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int f = tid + q * NT;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if constexpr (KC) {
                const int row = f >> 2, kq = f & 3;
                const int k = k0 + kq * 4;
                if (r0 + row < rmax && k < khi && k + 3 >= klo) {
                    val = *reinterpret_cast<const f32x4*>(base + (long long)(r0 + row) * rs + k);
                    unsigned m = 0u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) m |= in_range(k + e, klo, khi) ? (1u << e) : 0u;
                    keep |= m << (4 * q);
                }
            } else {
                constexpr int RQ = ROWS / 4;
                const int kk = f / RQ, rq = f % RQ;
                const int k = k0 + kk;
                const int r = r0 + rq * 4;
                if (in_range(k, klo, khi) && r < rmax) {
                    val = *reinterpret_cast<const f32x4*>(base + (long long)k * ks + r);
                }
            }
            v[q] = val;
        }
    }

    // lds: [BK][S] floats
    template <int S>
    __device__ __forceinline__ void store(float* lds, int tid, float sign) const {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int f = tid + q * NT;
            if constexpr (KC) {
                const int row = f >> 2, kq = f & 3;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    lds[(kq * 4 + e) * S + row] = ((keep >> (4 * q + e)) & 1u) ? v[q][e] * sign : 0.f;
            } else {
                constexpr int RQ = ROWS / 4;
                const int kk = f / RQ, rq = f % RQ;
                *reinterpret_cast<f32x4*>(lds + kk * S + rq * 4) = v[q] * sign;
            }
        }
    }
};

// ---- real kernel -----------------------------------------------------------------
template <int BM, int BN, bool A_KC, bool B_KC>
__global__ __launch_bounds__(NT, 2) void sgemm_kernel(const MkGemm p, int tilesM, int tilesN) {
    constexpr int SA = BM + (A_KC ? 2 : 4);
    constexpr int SB = BN + (B_KC ? 2 : 4);
    constexpr int WAVES_N = BN / 64;
    static_assert((BM / 64) * (BN / 64) == 4, "4 waves of 64x64");
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (SA + SB)];
    float* As = smem;                 // [2][BK][SA]
    float* Bs = smem + 2 * BK * SA;   // [2][BK][SB]

    const BlockCoord c = decode_block<BM, BN>(p, tilesM, tilesN);
    if (!c.active) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, lh = lane >> 5;

    const long long bo = c.b / p.inner, bi = c.b % p.inner;
    const float* Ab = p.A + bo * p.a_batch + bi * p.a_inner;
    const float* Bb = p.B + bo * p.b_batch + bi * p.b_inner;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int kt0 = c.klo / BK, kt1 = (c.khi + BK - 1) / BK;
    TileStage<BM, A_KC> sa;
    TileStage<BN, B_KC> sb;

    if (kt0 < kt1) {
        sa.load(Ab, p.a_row, p.a_k, c.i0, A_KC ? c.Meff : p.M, kt0 * BK, c.klo, c.khi, tid);
        sb.load(Bb, p.b_col, p.b_k, c.j0, p.N, kt0 * BK, c.klo, c.khi, tid);
        sa.template store<SA>(As, tid, 1.f);
        sb.template store<SB>(Bs, tid, 1.f);
    }
    __syncthreads();

    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        const bool more = (kt + 1) < kt1;
        if (more) {
            sa.load(Ab, p.a_row, p.a_k, c.i0, A_KC ? c.Meff : p.M, (kt + 1) * BK, c.klo, c.khi, tid);
            sb.load(Bb, p.b_col, p.b_k, c.j0, p.N, (kt + 1) * BK, c.klo, c.khi, tid);
        }
        const float* Ac = As + buf * BK * SA + wm * 64 + l31;
        const float* Bc = Bs + buf * BK * SB + wn * 64 + l31;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int krow = 2 * kk + lh;
            const float a0 = Ac[krow * SA], a1 = Ac[krow * SA + 32];
            const float b0 = Bc[krow * SB], b1 = Bc[krow * SB + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) {
            sa.template store<SA>(As + (buf ^ 1) * BK * SA, tid, 1.f);
            sb.template store<SB>(Bs + (buf ^ 1) * BK * SB, tid, 1.f);
        }
        __syncthreads();
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* Cb = p.C + bo * p.c_batch + bi * p.c_inner;
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const int col = c.j0 + wn * 64 + tb * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = c.i0 + wm * 64 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < c.Meff && col < p.N) {
                    float* dst = Cb + (long long)row * p.c_row + col;
                    float val = acc[ta][tb][r];
                    if (p.beta) val += *dst;
                    *dst = val;
                }
            }
        }
}

// ---- complex kernel (planar re/im) -------------------------------------------------
// block tile 64 (rows) x 128 (cols), waves 2x2, wave tile 32 x 64.
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(NT, 2) void cgemm_kernel(const MkGemm p, int tilesM, int tilesN) {
    constexpr int BM = 64, BN = 128;
    constexpr int SA = BM + (A_KC ? 2 : 4);
    constexpr int SB = BN + (B_KC ? 2 : 4);
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (2 * SA + 2 * SB)];
    float* Are = smem;                    // [2][BK][SA]
    float* Aim = Are + 2 * BK * SA;
    float* Bre = Aim + 2 * BK * SA;       // [2][BK][SB]
    float* Bim = Bre + 2 * BK * SB;

    const BlockCoord c = decode_block<BM, BN>(p, tilesM, tilesN);
    if (!c.active) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    const long long bo = c.b / p.inner, bi = c.b % p.inner;
    const float* Ab = p.A + bo * p.a_batch + bi * p.a_inner;
    const float* Bb = p.B + bo * p.b_batch + bi * p.b_inner;
    const float sgn_a = p.conj_a ? -1.f : 1.f;
    const float sgn_b = p.conj_b ? -1.f : 1.f;

    f32x16 cre[2], cim[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            cre[b][r] = 0.f;
            cim[b][r] = 0.f;
        }

    const int kt0 = c.klo / BK, kt1 = (c.khi + BK - 1) / BK;
    TileStage<BM, A_KC> sar, sai;
    TileStage<BN, B_KC> sbr, sbi;
    const int a_rmax = A_KC ? c.Meff : p.M;

    if (kt0 < kt1) {
        sar.load(Ab, p.a_row, p.a_k, c.i0, a_rmax, kt0 * BK, c.klo, c.khi, tid);
        sai.load(Ab + p.a_im, p.a_row, p.a_k, c.i0, a_rmax, kt0 * BK, c.klo, c.khi, tid);
        sbr.load(Bb, p.b_col, p.b_k, c.j0, p.N, kt0 * BK, c.klo, c.khi, tid);
        sbi.load(Bb + p.b_im, p.b_col, p.b_k, c.j0, p.N, kt0 * BK, c.klo, c.khi, tid);
        sar.template store<SA>(Are, tid, 1.f);
        sai.template store<SA>(Aim, tid, sgn_a);
        sbr.template store<SB>(Bre, tid, 1.f);
        sbi.template store<SB>(Bim, tid, sgn_b);
    }
    __syncthreads();

    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        const bool more = (kt + 1) < kt1;
        if (more) {
            sar.load(Ab, p.a_row, p.a_k, c.i0, a_rmax, (kt + 1) * BK, c.klo, c.khi, tid);
            sai.load(Ab + p.a_im, p.a_row, p.a_k, c.i0, a_rmax, (kt + 1) * BK, c.klo, c.khi, tid);
            sbr.load(Bb, p.b_col, p.b_k, c.j0, p.N, (kt + 1) * BK, c.klo, c.khi, tid);
            sbi.load(Bb + p.b_im, p.b_col, p.b_k, c.j0, p.N, (kt + 1) * BK, c.klo, c.khi, tid);
        }
        const int aoff = buf * BK * SA + wm * 32 + l31;
        const int boff = buf * BK * SB + wn * 64 + l31;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int krow = 2 * kk + lh;
            const float ar = Are[aoff + krow * SA], ai = Aim[aoff + krow * SA];
            const float nai = -ai;
            const float br0 = Bre[boff + krow * SB], br1 = Bre[boff + krow * SB + 32];
            const float bi0 = Bim[boff + krow * SB], bi1 = Bim[boff + krow * SB + 32];
            // (ar + i ai)(br + i bi) = (ar br - ai bi) + i (ar bi + ai br)
            cre[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, br0, cre[0], 0, 0, 0);
            cim[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, bi0, cim[0], 0, 0, 0);
            cre[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, br1, cre[1], 0, 0, 0);
            cim[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, bi1, cim[1], 0, 0, 0);
            cre[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(nai, bi0, cre[0], 0, 0, 0);
            cim[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, br0, cim[0], 0, 0, 0);
            cre[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(nai, bi1, cre[1], 0, 0, 0);
            cim[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, br1, cim[1], 0, 0, 0);
        }
        if (more) {
            const int nb = buf ^ 1;
            sar.template store<SA>(Are + nb * BK * SA, tid, 1.f);
            sai.template store<SA>(Aim + nb * BK * SA, tid, sgn_a);
            sbr.template store<SB>(Bre + nb * BK * SB, tid, 1.f);
            sbi.template store<SB>(Bim + nb * BK * SB, tid, sgn_b);
        }
        __syncthreads();
    }

    float* Cb = p.C + bo * p.c_batch + bi * p.c_inner;
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
        const int col = c.j0 + wn * 64 + tb * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = c.i0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row < c.Meff && col < p.N) {
                float* dr = Cb + (long long)row * p.c_row + col;
                float* di = dr + p.c_im;
                float vr = cre[tb][r], vi = cim[tb][r];
                if (p.beta) {
                    vr += *dr;
                    vi += *di;
                }
                *dr = vr;
                *di = vi;
            }
        }
    }
}

template <int BM, int BN>
int launch_real(const MkGemm* g, bool a_kc, bool b_kc, hipStream_t s) {
    const int tm = (g->M + BM - 1) / BM, tn = (g->N + BN - 1) / BN;
    const long long nb = (long long)((g->batch + MK_NUM_XCD - 1) / MK_NUM_XCD) * MK_NUM_XCD * tm * tn;
    MK_REQUIRE(nb < (1ll << 31), "gemm: grid too large");
    dim3 grid((unsigned)nb), block(NT);
    if (a_kc && b_kc)
        hipLaunchKernelGGL((sgemm_kernel<BM, BN, true, true>), grid, block, 0, s, *g, tm, tn);
    else if (a_kc && !b_kc)
        hipLaunchKernelGGL((sgemm_kernel<BM, BN, true, false>), grid, block, 0, s, *g, tm, tn);
    else if (!a_kc && b_kc)
        hipLaunchKernelGGL((sgemm_kernel<BM, BN, false, true>), grid, block, 0, s, *g, tm, tn);
    else
        hipLaunchKernelGGL((sgemm_kernel<BM, BN, false, false>), grid, block, 0, s, *g, tm, tn);
    return mk_check_launch("mk_sgemm_batched");
}

}  // namespace

extern "C" int mk_sgemm_batched(const MkGemm* g, void* stream) {
    bool a_kc, b_kc;
    int rc = validate(g, false, &a_kc, &b_kc);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    // rows are the triangular index for analysis-shaped calls: a finer row tile skips more
    if (g->tri_mode == MK_TRI_ROW_GE || g->tri_mode == MK_TRI_ROW_LE || g->M <= 64)
        return launch_real<64, 256>(g, a_kc, b_kc, s);
    if (g->N <= 64) return launch_real<256, 64>(g, a_kc, b_kc, s);
    return launch_real<128, 128>(g, a_kc, b_kc, s);
}

extern "C" int mk_cgemm_batched(const MkGemm* g, void* stream) {
    bool a_kc, b_kc;
    int rc = validate(g, true, &a_kc, &b_kc);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    constexpr int BM = 64, BN = 128;
    const int tm = (g->M + BM - 1) / BM, tn = (g->N + BN - 1) / BN;
    const long long nb = (long long)((g->batch + MK_NUM_XCD - 1) / MK_NUM_XCD) * MK_NUM_XCD * tm * tn;
    MK_REQUIRE(nb < (1ll << 31), "cgemm: grid too large");
    dim3 grid((unsigned)nb), block(NT);
    if (a_kc && b_kc)
        hipLaunchKernelGGL((cgemm_kernel<true, true>), grid, block, 0, s, *g, tm, tn);
    else if (a_kc && !b_kc)
        hipLaunchKernelGGL((cgemm_kernel<true, false>), grid, block, 0, s, *g, tm, tn);
    else if (!a_kc && b_kc)
        hipLaunchKernelGGL((cgemm_kernel<false, true>), grid, block, 0, s, *g, tm, tn);
    else
        hipLaunchKernelGGL((cgemm_kernel<false, false>), grid, block, 0, s, *g, tm, tn);
    return mk_check_launch("mk_cgemm_batched");
}
