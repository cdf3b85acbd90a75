// Shared pieces of the batched GEMM engines (fp32-MFMA sgemm.hip, split-bf16 xgemm.hip).
#pragma once
#include "common.h"

namespace gemm {

struct BlockCoord {
    int b, i0, j0, Meff, klo, khi;
    bool active;
};

template <int BM, int BN>
__device__ __forceinline__ BlockCoord decode_block(const MkGemm& p, int tilesM, int tilesN) {
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

// Complex operands are planar (re / im planes `*_im` floats apart) except, where the engine says it can (ilv_ok):
//   B interleaved  (b_im == 1): b_col / b_k count floats, the contiguous direction has stride 2 (re, im pairs);
//   C interleaved  (c_im == 1, c_col == 2): (re, im) pairs are stored with one 8-byte write.
// This is the memory order of a complex64 torch tensor: the dhconv weight and its gradient are used in place.
static inline int validate(const MkGemm* g, bool cplx, bool* a_kc, bool* b_kc, bool ilv_ok = false, bool* b_ilv = nullptr) {
    MK_REQUIRE(g && g->A && g->B && g->C, "gemm: null pointer");
    MK_REQUIRE(g->M > 0 && g->N > 0 && g->K >= 0 && g->batch > 0, "gemm: bad extents M=%d N=%d K=%d batch=%d", g->M,
               g->N, g->K, g->batch);
    const bool bi = cplx && ilv_ok && g->b_im == 1;
    const bool ci = cplx && ilv_ok && g->c_im == 1;
    if (b_ilv) *b_ilv = bi;
    const long long bu = bi ? 2 : 1;
    MK_REQUIRE(ci ? (g->c_col == 2 && (g->c_row & 1) == 0 && (g->c_batch & 1) == 0 && (g->c_inner & 1) == 0 &&
                     ((uintptr_t)g->C & 7) == 0)
                  : g->c_col == 1,
               "gemm: c_col must be 1 (planar C) or 2 with c_im == 1 (interleaved complex C, split engine only)");
    MK_REQUIRE(g->a_k == 1 || g->a_row == 1, "gemm: A needs a unit stride");
    MK_REQUIRE(g->b_k == bu || g->b_col == bu, "gemm: B needs a unit stride (2 floats when interleaved)");
    MK_REQUIRE(g->inner >= 1 && g->batch % g->inner == 0, "gemm: inner must be >= 1 and divide batch");
    *a_kc = (g->a_k == 1);
    *b_kc = (g->b_k == bu);
    // vector loads along the unit-stride dim: 16-byte alignment of everything else
    auto al = [](long long s) { return (s & 3) == 0; };
    MK_REQUIRE(((uintptr_t)g->A & 15) == 0 && ((uintptr_t)g->B & 15) == 0, "gemm: A/B must be 16-byte aligned");
    MK_REQUIRE(al(g->a_batch) && al(g->b_batch) && al(g->a_inner) && al(g->b_inner),
               "gemm: batch strides must be multiples of 4");
    if (*a_kc) {
        MK_REQUIRE(al(g->a_row), "gemm: a_row must be a multiple of 4");
    } else {
        MK_REQUIRE(al(g->a_k) && g->a_k >= ((g->M + 3) & ~3),
                   "gemm: row-contiguous A needs a_k %% 4 == 0 and a_k >= M rounded up to 4 (16-byte row vectors)");
    }
    if (*b_kc) {
        MK_REQUIRE(al(g->b_col), "gemm: b_col must be a multiple of 4");
    } else {
        MK_REQUIRE(al(g->b_k) && g->b_k >= bu * ((g->N + 3) & ~3),
                   "gemm: col-contiguous B needs b_k %% 4 == 0 and b_k >= N rounded up to 4 (16-byte row vectors)");
    }
    if (cplx) MK_REQUIRE(al(g->a_im) && (bi || al(g->b_im)), "gemm: plane offsets must be multiples of 4");
    return 0;
}

}  // namespace gemm
