// bf16-MFMA channel GEMMs for the pointwise (1x1 convolution) blocks on NCHW planes (gfx950).
//
//   mk_conv1x1_nn :  Y[b][m][n] = epi( sum_k A[m][k] * X[b][k][n] )      (forward and data-gradient)
//        A = weights (M x Kp, k contiguous, zero padded), X = activations with the PIXEL index contiguous.
//        The activation operand is k-strided for an MFMA fragment (a lane needs 8 consecutive k at one
//        pixel); it is staged row-major [k][n] into LDS with plain 16-byte writes and fetched with the
//        CDNA4 transpose read ds_read_b64_tr_b16 (two per fragment), so no shuffling instructions are spent.
//        epilogue: + bias[m], exact GELU (optionally also storing the pre-activation for backward),
//        + residual R[b][m][n] (the skip connection), * gelu'(G[b][m][n]) (activation backward).
//   mk_conv1x1_wgrad : dW[m][k] = sum_{b,n} G[b][m][n] * X[b][k][n]       (weight gradient)
//        both operands pixel-contiguous = k-contiguous for the contraction: plain ds_read_b128 fragments;
//        the huge contraction (pixels) is split over workgroups, fp32 partial tiles are reduced by a
//        second tiny kernel (deterministic, no atomics).
//
// Tiles: 256 threads = 4 waves (2 x 2); v_mfma_f32_32x32x16_bf16, fp32 accumulate; BK = 32;
// double-buffered LDS with register-staged prefetch, one barrier per k-tile.
#include <stdlib.h>

#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int NT = 256;

struct ConvNN {
    const u16* A;        // (M, lda) weights, k contiguous, lda % 8 == 0, zero beyond K
    const u16* X;        // (B, K, N)
    u16* Y;              // (B, M, N)
    u16* Ypre;           // optional: pre-activation (B, M, N)
    const float* bias;   // optional (M)
    const u16* R;        // optional residual (B, M, N), added after activation
    const u16* G;        // optional: multiply result by gelu'(G)  (B, M, N)
    int M, K, lda, B;
    long long N;
    int act;
};

__device__ __forceinline__ uint4 ld16(const u16* p) { return *reinterpret_cast<const uint4*>(p); }

// ------------------------------------------------------------------------------------------
// Main loop: LDS double buffer + TWO register sets (loads run two k-tiles ahead: one k-tile is only 16 MFMAs =
// 0.2 us per wave), interior tiles load without per-vector branches.  Epilogue: the fp32 accumulators cross LDS
// (64 rows at a time) so that Y, Ypre, G and R are all touched as 16-byte vectors along the pixel axis; direct
// stores from the MFMA layout would be 2-byte scattered writes (a lane owns one pixel column).
template <int BM, int BN, int DEPTH, int BKT, int WGS>
__global__ __launch_bounds__(NT, WGS) void conv_nn_kernel(const ConvNN p, int tilesM, long long tilesN) {
    constexpr int WM = BM / 2, WN = BN / 2;          // wave tile
    constexpr int TM = WM / 32, TN = WN / 32;        // MFMA tiles per wave
    constexpr int BK = BKT;                          // shadows the file-level default
    constexpr int LDSB = (BKT == 32) ? 2 : 1;        // LDS stages: BK = 64 uses ONE stage (+ register prefetch)
    constexpr int PA = BK + 8;                       // A pitch (u16): 80 / 144 B -> conflict-free b128 reads
    constexpr int PB = BN + 32;                      // X pitch (u16): rows 16 dwords apart mod 64 -> conflict-free tr reads
    constexpr int NA = (BM * BK / 8) / NT;           // 16-byte chunks per thread
    constexpr int NB_ = (BK * BN / 8) / NT;
    constexpr int PE = BN + 4;                       // epilogue pitch (floats)
    static_assert(NA >= 1 && NB_ >= 1, "tile too small");
    static_assert(WM == 64, "the epilogue stages one wave row (64 output channels) at a time");
    constexpr int OPER = LDSB * (BM * PA + BK * PB) * 2;   // bytes
    constexpr int EPI = WM * PE * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[OPER > EPI ? OPER : EPI];
    u16* As = reinterpret_cast<u16*>(smem_raw);      // [2][BM][PA]
    u16* Bs = As + LDSB * BM * PA;                   // [LDSB][BK][PB]
    float* Es = reinterpret_cast<float*>(smem_raw);  // [WM][PE]  (after the k-loop)

    // tile decode: all M-tiles of one pixel range are adjacent block ids; xcd_remap keeps consecutive virtual ids
    // on one XCD, so the tilesM tiles that re-read the same X columns hit that XCD's L2 instead of HBM
    const long long bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = (int)(bid % tilesM);
    const long long tnb = bid / tilesM;
    const int b = (int)(tnb / tilesN);
    const long long tn = tnb % tilesN;
    const int m0 = tm * BM;
    const long long n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    const u16* Xb = p.X + (long long)b * p.K * p.N;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    uint4 ra[DEPTH][NA], rb[DEPTH][NB_];
    const bool tile_full = (m0 + BM <= p.M) && (n0 + BN <= p.N);
    auto load_tiles = [&](uint4* qa, uint4* qb, int k0) {
        if (tile_full && k0 + BK <= p.K) {
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                const int f = tid + q * NT;
                qa[q] = ld16(p.A + (long long)(m0 + f / (BK / 8)) * p.lda + k0 + (f % (BK / 8)) * 8);
            }
#pragma unroll
            for (int q = 0; q < NB_; ++q) {
                const int f = tid + q * NT;
                qb[q] = ld16(Xb + (long long)(k0 + f / (BN / 8)) * p.N + n0 + (f % (BN / 8)) * 8);
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int f = tid + q * NT;
            const int row = f / (BK / 8), c = f % (BK / 8);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (m0 + row < p.M && k0 + c * 8 < p.lda) v = ld16(p.A + (long long)(m0 + row) * p.lda + k0 + c * 8);
            qa[q] = v;
        }
#pragma unroll
        for (int q = 0; q < NB_; ++q) {
            const int f = tid + q * NT;
            const int kk = f / (BN / 8), c = f % (BN / 8);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (k0 + kk < p.K && n0 + c * 8 < p.N) v = ld16(Xb + (long long)(k0 + kk) * p.N + n0 + c * 8);
            qb[q] = v;
        }
    };
    auto store_tiles = [&](const uint4* qa, const uint4* qb, int buf) {
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int f = tid + q * NT;
            *reinterpret_cast<uint4*>(As + buf * BM * PA + (f / (BK / 8)) * PA + (f % (BK / 8)) * 8) = qa[q];
        }
#pragma unroll
        for (int q = 0; q < NB_; ++q) {
            const int f = tid + q * NT;
            *reinterpret_cast<uint4*>(Bs + buf * BK * PB + (f / (BN / 8)) * PB + (f % (BN / 8)) * 8) = qb[q];
        }
    };

    // per-lane LDS offsets of the fragments
    const int s = lane & 15, g1 = (lane >> 4) & 1;
    const int a_off = (wm * WM + l31) * PA + lh * 8;                                   // + i*32*PA + ks*16
    const int b_off = (lh * 8 + (s >> 2)) * PB + wn * WN + g1 * 16 + (s & 3) * 4;      // + (ks*16 [+4])*PB + j*32
    auto compute = [&](int buf) {
        const u16* Ab = As + buf * BM * PA;
        const u16* Bb = Bs + buf * BK * PB;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(Ab + a_off + i * 32 * PA + ks * 16));
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const u16* q0 = Bb + b_off + (ks * 16) * PB + j * 32;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q0));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q0 + 4 * PB));
                const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                bfr[j] = __builtin_bit_cast(bf16x8, v);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    };

    const int nk = (p.K + BK - 1) / BK;
    load_tiles(ra[0], rb[0], 0);
    if constexpr (DEPTH == 2) {
        if (nk > 1) load_tiles(ra[1], rb[1], BK);
        store_tiles(ra[0], rb[0], 0);
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {
            if (kt + 2 < nk) load_tiles(ra[0], rb[0], (kt + 2) * BK);
            compute(0);
            if (kt + 1 < nk) store_tiles(ra[1], rb[1], 1);
            __syncthreads();
            if (kt + 1 >= nk) break;
            if (kt + 3 < nk) load_tiles(ra[1], rb[1], (kt + 3) * BK);
            compute(1);
            if (kt + 2 < nk) store_tiles(ra[0], rb[0], 0);
            __syncthreads();
        }
    } else if constexpr (LDSB == 2) {
        store_tiles(ra[0], rb[0], 0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) load_tiles(ra[0], rb[0], (kt + 1) * BK);
            compute(kt & 1);
            if (kt + 1 < nk) store_tiles(ra[0], rb[0], (kt & 1) ^ 1);
            __syncthreads();
        }
    } else {       // one LDS stage: twice the bytes in flight per k-tile, half as many exposed round trips
        store_tiles(ra[0], rb[0], 0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) load_tiles(ra[0], rb[0], (kt + 1) * BK);
            compute(0);
            __syncthreads();
            if (kt + 1 < nk) {
                store_tiles(ra[0], rb[0], 0);
                __syncthreads();
            }
        }
    }

    // ---- epilogue: 64 output channels at a time through LDS, then 8-pixel vectors per thread ----
    const long long plane = (long long)b * p.M * p.N;
    const bool vec_ok = (p.N % 8) == 0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (wm == half) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        Es[row * PE + wn * WN + j * 32 + l31] = acc[i][j][r];
                    }
        }
        __syncthreads();
        for (int f = tid; f < WM * (BN / 8); f += NT) {
            const int row = f / (BN / 8), c = f % (BN / 8);
            const int m = m0 + half * WM + row;
            const long long n = n0 + c * 8;
            if (m >= p.M || n >= p.N) continue;
            const float bv = p.bias ? p.bias[m] : 0.f;
            const long long o = plane + (long long)m * p.N + n;
            float v[8];
            const float4 e0 = *reinterpret_cast<const float4*>(Es + row * PE + c * 8);
            const float4 e1 = *reinterpret_cast<const float4*>(Es + row * PE + c * 8 + 4);
            v[0] = e0.x + bv, v[1] = e0.y + bv, v[2] = e0.z + bv, v[3] = e0.w + bv;
            v[4] = e1.x + bv, v[5] = e1.y + bv, v[6] = e1.z + bv, v[7] = e1.w + bv;
            const bool full = vec_ok && n + 8 <= p.N;
            auto put = [&](u16* dst, const float* x) {
                if (full) {
                    uint4 u;
                    u.x = (uint32_t)f32_to_bf16(x[0]) | ((uint32_t)f32_to_bf16(x[1]) << 16);
                    u.y = (uint32_t)f32_to_bf16(x[2]) | ((uint32_t)f32_to_bf16(x[3]) << 16);
                    u.z = (uint32_t)f32_to_bf16(x[4]) | ((uint32_t)f32_to_bf16(x[5]) << 16);
                    u.w = (uint32_t)f32_to_bf16(x[6]) | ((uint32_t)f32_to_bf16(x[7]) << 16);
                    *reinterpret_cast<uint4*>(dst + o) = u;
                } else {
                    for (int e = 0; e < 8 && n + e < p.N; ++e) dst[o + e] = f32_to_bf16(x[e]);
                }
            };
            auto get = [&](const u16* src, float* x) {
                if (full) {
                    const uint4 u = ld16(src + o);
                    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        x[2 * e] = __uint_as_float(w[e] << 16);
                        x[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
                    }
                } else {
                    for (int e = 0; e < 8; ++e) x[e] = (n + e < p.N) ? bf16_to_f32(src[o + e]) : 0.f;
                }
            };
            if (p.act) {
                if (p.Ypre) put(p.Ypre, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
            }
            if (p.G) {
                float g[8];
                get(p.G, g);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= gelu_grad_f(g[e]);
            }
            if (p.R) {
                float rr[8];
                get(p.R, rr);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rr[e];
            }
            put(p.Y, v);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// wgrad: part[s][m][k] = sum_{n in split s} G[b][m][n] X[b][k][n];   tile 128 (m) x 128 (k-channel)
struct ConvWg {
    const u16* G;   // (B, M, N)
    const u16* X;   // (B, K, N)
    float* part;    // (S, M, K) fp32 partials
    int M, K, B, S;
    long long N;
    long long chunk;   // pixels per split (multiple of BK)
};

// BKT = 64: two LDS stages, two register sets (loads two pixel tiles ahead);
// BKT = 128: one LDS stage, one register set of twice the size (64 MFMAs per barrier pair)
template <int BKT>
__global__ __launch_bounds__(NT, 2) void conv_wgrad_kernel(const ConvWg p, int tilesM, int tilesK) {
    constexpr int BM = 128, BN = 128;
    constexpr int BK = BKT;                  // >= 64: whole 128-byte cache lines per row
    constexpr int LDSB = BKT == 64 ? 2 : 1, SETS = BKT == 64 ? 2 : 1;
    constexpr int PA = BK + 8;
    constexpr int NA = (BM * BK / 8) / NT;   // 16-byte vectors per thread and operand
    __shared__ __attribute__((aligned(16))) u16 smem[LDSB * (BM + BN) * PA];
    u16* As = smem;
    u16* Bs = smem + LDSB * BM * PA;

    // the tilesM*tilesK tiles of one pixel split re-read the same G / X columns: keep them on one XCD (L2)
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid % tilesM;
    const int tk = (bid / tilesM) % tilesK;
    const int sp = bid / (tilesM * tilesK);          // split index over (b, pixel chunk)
    const int m0 = tm * BM, c0 = tk * BN;
    const int splits_per_b = p.S / p.B;
    const int b = sp / splits_per_b;
    const long long nbeg = (long long)(sp % splits_per_b) * p.chunk;
    const long long nend = min(p.N, nbeg + p.chunk);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const u16* Gb = p.G + (long long)b * p.M * p.N;
    const u16* Xb = p.X + (long long)b * p.K * p.N;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // two register sets: the loads of pixel tile kt+2 are issued while tile kt is multiplied (one k-step of 32 MFMAs
    // = 0.43 us does not cover an HBM round trip under load), tile kt+1 is copied to the other LDS stage after the
    // MFMAs.  Interior tiles load unconditionally (no per-vector branches).
    uint4 ra[SETS][NA], rb[SETS][NA];
    const bool rows_full = (m0 + BM <= p.M) && (c0 + BN <= p.K);
    auto load_tiles = [&](uint4* qa, uint4* qb, long long n) {
        if (rows_full && n + BK <= nend) {
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                const int f = tid + q * NT;
                const int row = f / (BK / 8), c = f % (BK / 8);
                qa[q] = ld16(Gb + (long long)(m0 + row) * p.N + n + c * 8);
                qb[q] = ld16(Xb + (long long)(c0 + row) * p.N + n + c * 8);
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int f = tid + q * NT;
            const int row = f / (BK / 8), c = f % (BK / 8);
            uint4 va = make_uint4(0, 0, 0, 0), vb = make_uint4(0, 0, 0, 0);
            if (n + c * 8 < nend) {
                if (m0 + row < p.M) va = ld16(Gb + (long long)(m0 + row) * p.N + n + c * 8);
                if (c0 + row < p.K) vb = ld16(Xb + (long long)(c0 + row) * p.N + n + c * 8);
            }
            qa[q] = va;
            qb[q] = vb;
        }
    };
    auto store_tiles = [&](const uint4* qa, const uint4* qb, int buf) {
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int f = tid + q * NT;
            const int row = f / (BK / 8), c = f % (BK / 8);
            *reinterpret_cast<uint4*>(As + buf * BM * PA + row * PA + c * 8) = qa[q];
            *reinterpret_cast<uint4*>(Bs + buf * BN * PA + row * PA + c * 8) = qb[q];
        }
    };
    const int a_off = (wm * 64 + l31) * PA + lh * 8;
    const int b_off = (wn * 64 + l31) * PA + lh * 8;
    auto compute = [&](int buf) {
        const u16* Ab = As + buf * BM * PA;
        const u16* Bb = Bs + buf * BN * PA;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(Ab + a_off + i * 32 * PA + ks * 16));
                bfr[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(Bb + b_off + i * 32 * PA + ks * 16));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    };

    const int nk = (int)((nend - nbeg + BK - 1) / BK);
    if constexpr (SETS == 2) {
        if (nk > 0) load_tiles(ra[0], rb[0], nbeg);
        if (nk > 1) load_tiles(ra[1], rb[1], nbeg + BK);
        if (nk > 0) store_tiles(ra[0], rb[0], 0);
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {
            // even tile: LDS stage 0, its registers (set 0) are free again
            if (kt + 2 < nk) load_tiles(ra[0], rb[0], nbeg + (long long)(kt + 2) * BK);
            compute(0);
            if (kt + 1 < nk) store_tiles(ra[1], rb[1], 1);
            __syncthreads();
            if (kt + 1 >= nk) break;
            // odd tile: LDS stage 1
            if (kt + 3 < nk) load_tiles(ra[1], rb[1], nbeg + (long long)(kt + 3) * BK);
            compute(1);
            if (kt + 2 < nk) store_tiles(ra[0], rb[0], 0);
            __syncthreads();
        }
    } else {
        if (nk > 0) {
            load_tiles(ra[0], rb[0], nbeg);
            store_tiles(ra[0], rb[0], 0);
        }
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) load_tiles(ra[0], rb[0], nbeg + (long long)(kt + 1) * BK);
            compute(0);
            __syncthreads();
            if (kt + 1 < nk) {
                store_tiles(ra[0], rb[0], 0);
                __syncthreads();
            }
        }
    }

    float* out = p.part + (long long)sp * p.M * p.K;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = c0 + wn * 64 + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < p.M && c < p.K) out[(long long)m * p.K + c] = acc[i][j][r];
            }
        }
}

__global__ void reduce_splits(const float* __restrict__ part, float* __restrict__ out, long long n, int S, int accumulate) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = accumulate ? out[i] : 0.f;
    for (int k = 0; k < S; ++k) s += part[(long long)k * n + i];
    out[i] = s;
}

}  // namespace

extern "C" int mk_conv1x1_nn(const void* A, const void* X, void* Y, void* Ypre, const float* bias, const void* R,
                             const void* G, int M, int K, int lda, int B, long long N, int act, void* stream) {
    MK_REQUIRE(A && X && Y, "conv1x1_nn: null pointer");
    MK_REQUIRE(M > 0 && K > 0 && B > 0 && N > 0, "conv1x1_nn: bad shape M=%d K=%d B=%d N=%lld", M, K, B, N);
    MK_REQUIRE((lda % 8) == 0 && lda >= K, "conv1x1_nn: lda=%d must be a multiple of 8 and >= K=%d", lda, K);
    MK_REQUIRE((N % 8) == 0, "conv1x1_nn: pixel count %lld must be a multiple of 8", N);
    MK_REQUIRE((((uintptr_t)A | (uintptr_t)X) & 15) == 0, "conv1x1_nn: operands must be 16-byte aligned");
    ConvNN p{(const u16*)A, (const u16*)X, (u16*)Y, (u16*)Ypre, bias, (const u16*)R, (const u16*)G, M, K, lda, B, N, act};
    // 128 x 256 tile, BK = 64 with one LDS stage (measured 10-15 % faster at 721x1440 than BK = 32 double-buffered,
    // equal at 240x480; a 128 x 128 tile with two register sets was 10-20 % slower)
    // BK = 64 with one LDS stage (10-15 % faster at 721x1440 than BK = 32 double-buffered).  Tile 128 x 128 with
    // 3 workgroups / CU on the internal grid (0.118 ms vs 0.128 for 128 x 256 at 240x480, 768 <- 384; the library:
    // 0.104), 128 x 256 (longer contiguous row segments) on the full-resolution planes; a 4-workgroup build spills.
    constexpr int BM = 128;
    const int tm = (M + BM - 1) / BM;
    if (N >= (1ll << 19)) {
        constexpr int BN = 256;
        const long long tn = (N + BN - 1) / BN;
        const long long nb = (long long)tm * tn * B;
        MK_REQUIRE(nb < (1ll << 31), "conv1x1_nn: grid too large");
        hipLaunchKernelGGL((conv_nn_kernel<BM, BN, 1, 64, 2>), dim3((unsigned)nb), dim3(NT), 0, (hipStream_t)stream, p, tm, tn);
    } else {
        constexpr int BN = 128;
        const long long tn = (N + BN - 1) / BN;
        const long long nb = (long long)tm * tn * B;
        MK_REQUIRE(nb < (1ll << 31), "conv1x1_nn: grid too large");
        hipLaunchKernelGGL((conv_nn_kernel<BM, BN, 1, 64, 3>), dim3((unsigned)nb), dim3(NT), 0, (hipStream_t)stream, p, tm, tn);
    }
    return mk_check_launch("mk_conv1x1_nn");
}

extern "C" long long mk_conv1x1_wgrad_workspace(int M, int K, int B, long long N) {
    // number of fp32 elements the caller must provide as `part`
    const int tiles = ((M + 127) / 128) * ((K + 127) / 128);
    long long per_b = 1024 / ((long long)tiles * B);
    if (per_b < 1) per_b = 1;
    const long long maxs = (N + 2047) / 2048;      // at least 2048 pixels per split
    if (per_b > maxs) per_b = maxs;
    return per_b * B * (long long)M * K;
}

extern "C" int mk_conv1x1_wgrad(const void* G, const void* X, float* dW, float* part, int M, int K, int B, long long N,
                                int accumulate, void* stream) {
    MK_REQUIRE(G && X && dW && part, "conv1x1_wgrad: null pointer");
    MK_REQUIRE(M > 0 && K > 0 && B > 0 && N > 0 && (N % 8) == 0, "conv1x1_wgrad: bad shape");
    const int tm = (M + 127) / 128, tk = (K + 127) / 128;
    const long long S = mk_conv1x1_wgrad_workspace(M, K, B, N) / ((long long)M * K);
    const long long per_b = S / B;
    long long chunk = (N + per_b - 1) / per_b;
    chunk = (chunk + 127) / 128 * 128;
    ConvWg p{(const u16*)G, (const u16*)X, part, M, K, B, (int)S, N, chunk};
    hipStream_t s = (hipStream_t)stream;
    // BK = 128 / one LDS stage and BK = 64 / two stages + two register sets measure the same (+-2 %) on the 384/768
    // channel shapes; the former is 8 % faster on the 73-channel ones and needs 36 fewer VGPRs
    hipLaunchKernelGGL(conv_wgrad_kernel<128>, dim3((unsigned)(tm * tk * S)), dim3(NT), 0, s, p, tm, tk);
    const long long n = (long long)M * K;
    hipLaunchKernelGGL(reduce_splits, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, dW, n, (int)S, accumulate);
    return mk_check_launch("mk_conv1x1_wgrad");
}
