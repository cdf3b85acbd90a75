// Split-bf16 ("bf16x6" / "bf16x3") batched GEMM engine for fp32 operands on the bf16 matrix cores (gfx950).
//
//   C[b][i][j] (+)= sum_k A[b][i][k] * B[b][j][k]          same MkGemm descriptor as sgemm.hip
//
// gfx950 has no TF32/xf32 path and its exact-fp32 MFMA runs at 1/16 of the bf16 rate.  Here every
// fp32 operand element is split on the fly (while it is staged into LDS) into NP bf16 limbs
//      x = hi + mid (+ lo),   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)
// and the product is expanded into bf16 MFMAs with fp32 accumulation:
//      NP = 3:  hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid    (6 MFMAs, dropped terms <= 2^-24 relative)
//      NP = 2:  hi*hi + hi*mid + mid*hi                                (3 MFMAs, dropped terms ~ 2^-16)
// bf16 x bf16 products are exact in fp32, so NP = 3 reproduces an fp32 GEMM to fp32 round-off
// (measured rel-L2 1.8e-7 vs fp64 on the Legendre matrices, the exact-fp32 MFMA kernel gives 2.8e-7) at
// 6/16 of its MFMA time; NP = 2 gives ~4e-6 at 3/16.
//
// Structure: 256 threads = 4 waves, v_mfma_f32_32x32x16_bf16, BK = 16 (one MFMA k-step per tile),
// single LDS stage + register prefetch of the next fp32 tile (two barriers per k-tile of >= 768 MFMA
// cycles per wave; >= 2 workgroups per CU overlap them).  Operand limbs live in LDS either as
// [row][k] (k-contiguous global operand; fragments = one ds_read_b128) or as [k][row] (row-contiguous
// global operand; fragments = two ds_read_b64_tr_b16 transpose reads), so no layout shuffling is
// ever done in registers.  Triangular skipping and the XCD-aware grid are those of sgemm.hip.
#include "gemm_common.h"

#include "xsplit.h"

namespace {

using gemm::BlockCoord;
using gemm::decode_block;
using gemm::validate;
using namespace xsplit;

constexpr int NT = 256;

// ---- real kernel: block tile BM x BN, 4 waves of 64 x 64 ---------------------------------
template <int BM, int BN, bool A_KC, bool B_KC, int NP>
__global__ __launch_bounds__(NT, 2) void xgemm_kernel(const MkGemm p, int tilesM, int tilesN) {
    constexpr int PLA = plane_elems<BM, A_KC>(), PLB = plane_elems<BN, B_KC>();
    constexpr int WAVES_N = BN / 64;
    static_assert((BM / 64) * (BN / 64) == 4, "4 waves of 64x64");
    __shared__ __attribute__((aligned(16))) u16 smem[NP * (PLA + PLB)];
    u16* As = smem;
    u16* Bs = smem + NP * PLA;

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
    const int a_rmax = A_KC ? c.Meff : p.M;
    Stage<BM, A_KC> sa;
    Stage<BN, B_KC> sb;
    sa.init(Ab, p.a_row, p.a_k, c.i0, a_rmax, tid);
    sb.init(Bb, p.b_col, p.b_k, c.j0, p.N, tid);
    if (kt0 < kt1) {
        sa.load(kt0 * BK, c.klo, c.khi, tid);
        sb.load(kt0 * BK, c.klo, c.khi, tid);
    }
    for (int kt = kt0; kt < kt1; ++kt) {
        sa.template store<NP, PLA>(As, tid, 1.f);
        sb.template store<NP, PLB>(Bs, tid, 1.f);
        __syncthreads();
        if (kt + 1 < kt1) {   // next fp32 tile in flight while this one is multiplied
            sa.load((kt + 1) * BK, c.klo, c.khi, tid);
            sb.load((kt + 1) * BK, c.klo, c.khi, tid);
        }
        bf16x8 af[2][NP], bfr[2][NP];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                af[t][pl] = frag<BM, A_KC>(As + pl * PLA, wm * 64 + t * 32, lane);
                bfr[t][pl] = frag<BN, B_KC>(Bs + pl * PLB, wn * 64 + t * 32, lane);
            }
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) acc[ta][tb] = mma_split<NP>(af[ta], bfr[tb], acc[ta][tb]);
        __syncthreads();
    }

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

template <int BM, int BN, int NP>
int launch_real(const MkGemm* g, bool a_kc, bool b_kc, hipStream_t s) {
    const int tm = (g->M + BM - 1) / BM, tn = (g->N + BN - 1) / BN;
    const long long nb = (long long)((g->batch + MK_NUM_XCD - 1) / MK_NUM_XCD) * MK_NUM_XCD * tm * tn;
    MK_REQUIRE(nb < (1ll << 31), "xgemm: grid too large");
    dim3 grid((unsigned)nb), block(NT);
    if (a_kc && b_kc)
        hipLaunchKernelGGL((xgemm_kernel<BM, BN, true, true, NP>), grid, block, 0, s, *g, tm, tn);
    else if (a_kc && !b_kc)
        hipLaunchKernelGGL((xgemm_kernel<BM, BN, true, false, NP>), grid, block, 0, s, *g, tm, tn);
    else if (!a_kc && b_kc)
        hipLaunchKernelGGL((xgemm_kernel<BM, BN, false, true, NP>), grid, block, 0, s, *g, tm, tn);
    else
        hipLaunchKernelGGL((xgemm_kernel<BM, BN, false, false, NP>), grid, block, 0, s, *g, tm, tn);
    return mk_check_launch("mk_sgemm_split_batched");
}

}  // namespace

extern "C" int mk_sgemm_split_batched(const MkGemm* g, int limbs, void* stream) {
    bool a_kc, b_kc;
    int rc = validate(g, false, &a_kc, &b_kc);
    if (rc) return rc;
    MK_REQUIRE(limbs == 2 || limbs == 3, "split gemm: limbs must be 2 or 3");
    hipStream_t s = (hipStream_t)stream;
    const bool rows_tri = g->tri_mode == MK_TRI_ROW_GE || g->tri_mode == MK_TRI_ROW_LE || g->M <= 64;
    if (limbs == 3) return rows_tri ? launch_real<64, 256, 3>(g, a_kc, b_kc, s) : launch_real<128, 128, 3>(g, a_kc, b_kc, s);
    return rows_tri ? launch_real<64, 256, 2>(g, a_kc, b_kc, s) : launch_real<128, 128, 2>(g, a_kc, b_kc, s);
}

// The first-generation complex kernel (64 x 128 tile, one LDS stage) was retired in round 5: every descriptor it took is served
// by the second-generation kernel (csrc/xgemm2.hip); the entry point stays in the ABI as an alias.
extern "C" int mk_cgemm_split2_batched(const MkGemm* g, int limbs, void* stream);
extern "C" int mk_cgemm_split_batched(const MkGemm* g, int limbs, void* stream) { return mk_cgemm_split2_batched(g, limbs, stream); }
