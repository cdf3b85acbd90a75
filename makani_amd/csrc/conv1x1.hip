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
#include <type_traits>

#include "common.h"

#ifndef MK_ASTAT_EPI_DRAIN
#define MK_ASTAT_EPI_DRAIN 0
#endif
#ifndef MK_A2_PRIO              // conv_nn_astat2_kernel: s_setprio 1 around the multiplication phases (A/B knob)
#define MK_A2_PRIO 1
#endif
// timing diagnostic of the weight-stationary kernel (tools/astat_diag.py; build with -DMK_ASTAT_DIAG=1): s_memtime stamps at the
// segment boundaries of every pixel tile, summed per wave into g_astat_diag:
// [0 wait for the chunk + barrier, 1 fragment reads + MFMAs, 2 next chunk's DMA issue, 3 epilogue: convert + stage, 4 barrier,
//  5 epilogue: read back + epilogue math + stores, 6 barrier, 7 prologue (weights), 8 whole kernel, 9 waves, 10 tiles]
#ifndef MK_ASTAT_DIAG
#define MK_ASTAT_DIAG 0
#endif
#if MK_ASTAT_DIAG
__device__ unsigned long long g_astat_diag[16];
#define MK_AS_STAMP(k)                                              \
    do {                                                            \
        const unsigned long long t_ = __builtin_readcyclecounter(); \
        dg[k] += t_ - tprev;                                        \
        tprev = t_;                                                 \
    } while (0)
#else
#define MK_AS_STAMP(k)
#endif

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
    int nt;              // output stores with the streaming (nt) cache policy: set by the launcher for outputs that fit the memory-side cache
};

__device__ __forceinline__ uint4 ld16(const u16* p) { return *reinterpret_cast<const uint4*>(p); }
// Output stores.  Measured (gpurun_out/r07l, same box): with the streaming (nt) policy the K = 384 / 768 kernels are 7-15 % faster at
// 115 200 pixels — the 88-177 MB outputs stop evicting the activations the previous kernel left in the 256 MB memory-side cache — and
// 0-3 % slower at 1 038 240 pixels (0.8-1.6 GB outputs); the launcher sets p.nt by the output size.  One store is issued either way
// (uniform branch): the counted waits of the ring kernels do not change.
__device__ __forceinline__ void mk_st16(u16* p, uint4 v, int nt) {
    typedef unsigned int u32x4_nt __attribute__((ext_vector_type(4)));
    if (nt) __builtin_nontemporal_store(u32x4_nt{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4_nt*>(p));
    else *reinterpret_cast<uint4*>(p) = v;
}
#define MK_BUF_ST16(VAL, RS, VOFF, NT)                                            \
    do {                                                                          \
        if (NT) __builtin_amdgcn_raw_buffer_store_b128(VAL, RS, VOFF, 0, 2);      \
        else __builtin_amdgcn_raw_buffer_store_b128(VAL, RS, VOFF, 0, 0);         \
    } while (0)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef int v4i_t __attribute__((ext_vector_type(4)));

template <int N_> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

// ---- LDS-DMA (buffer_load ... lds) as inline assembly -------------------------------------------------------------
// The ring kernels below keep one or two stages of LDS-DMA in flight across barriers and wait for them with counted
// s_waitcnt vmcnt(N).  hipcc's own wait-count pass knows the builtin form of these loads and puts a full
// s_waitcnt vmcnt(0) in front of the next LDS read (it cannot tell which stage a ds_read touches), which drains the
// ring at every step; as inline assembly the loads are outside its bookkeeping and the kernel counts them itself.
// One statement = 3 or 4 pieces of 1 KB (64 lanes x 16 B) to LDS addresses lds0 + i * STEP; M0 (the DMA destination
// base) is written and restored inside the statement; s_nop 4 covers an SGPR operand written by the instruction just
// before, s_nop 0 the M0 write -> DMA read hazard.
__device__ __forceinline__ v4i_t make_rsrc(const void* ptr) {      // raw buffer, stride 0: voffset >= 2^31 reads zeros
    const unsigned long long a = (unsigned long long)ptr;
    v4i_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    r[2] = (int)0x80000000u;
    r[3] = 0x00020000;
    return r;
}
__device__ __forceinline__ v4i_t make_rsrc_n(const void* ptr, unsigned bytes) {      // raw buffer of `bytes`: offsets beyond it read zeros
    v4i_t r = make_rsrc(ptr);
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    return r;
}
__device__ __forceinline__ unsigned lds_addr(const void* ptr) { return (unsigned)(unsigned long long)(lds_void_t*)ptr; }

template <int STEP>
__device__ __forceinline__ void dma4(unsigned lds0, v4i_t rs, unsigned soff, unsigned v0, unsigned v1, unsigned v2, unsigned v3) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_nop 4\n\t"
        "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %2, %3 offen lds\n\t"
        "s_add_u32 m0, m0, %8\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %2, %3 offen lds\n\t"
        "s_add_u32 m0, m0, %8\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %2, %3 offen lds\n\t"
        "s_add_u32 m0, m0, %8\n\ts_nop 0\n\tbuffer_load_dwordx4 %7, %2, %3 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds0), "s"(rs), "s"(soff), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "i"(STEP)
        : "memory", "scc");
}
__device__ __forceinline__ void dma1(unsigned lds0, v4i_t rs, unsigned soff, unsigned v0) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_nop 4\n\t"
        "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %2, %3 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds0), "s"(rs), "s"(soff), "v"(v0)
        : "memory");
}
template <int STEP>
__device__ __forceinline__ void dma2(unsigned lds0, v4i_t rs, unsigned soff, unsigned v0, unsigned v1) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_nop 4\n\t"
        "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %2, %3 offen lds\n\t"
        "s_add_u32 m0, m0, %6\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %2, %3 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds0), "s"(rs), "s"(soff), "v"(v0), "v"(v1), "i"(STEP)
        : "memory", "scc");
}
template <int STEP>
__device__ __forceinline__ void dma3(unsigned lds0, v4i_t rs, unsigned soff, unsigned v0, unsigned v1, unsigned v2) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_nop 4\n\t"
        "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %2, %3 offen lds\n\t"
        "s_add_u32 m0, m0, %7\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %2, %3 offen lds\n\t"
        "s_add_u32 m0, m0, %7\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %2, %3 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds0), "s"(rs), "s"(soff), "v"(v0), "v"(v1), "v"(v2), "i"(STEP)
        : "memory", "scc");
}

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
                    u.x = pack_bf16x2(x[0], x[1]);
                    u.y = pack_bf16x2(x[2], x[3]);
                    u.z = pack_bf16x2(x[4], x[5]);
                    u.w = pack_bf16x2(x[6], x[7]);
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
// Forward / data-gradient channel GEMM, ring version (K a multiple of 64, M >= 192): a persistent grid of 512-thread
// workgroups, one per CU, walks over (pixel tile, channel slab) pairs.  Tile = BM (256 or 192) output channels x 256
// pixels; both operands stream through a two-stage LDS ring filled by LDS-DMA (a stage = BM x 64 weights + 64 x 256
// activations = 56-64 KB, so that much is in flight per CU while the other stage is multiplied):
//   weights     rows of 128 B (64 k), chunk index XOR (row >> 1) & 7      -> conflict-free ds_read_b128 fragments
//   activations rows of 512 B (256 pixels of one input channel), chunk index XOR (k & 3) << 2
//                                                                          -> conflict-free ds_read_b64_tr_b16 fragments
// (the XOR is applied to the per-lane source address; the DMA destination is lane-linear).
// The product is formed transposed, D[pixel][channel] (activation fragment as the MFMA's first operand), so that a lane
// owns ONE output channel and runs of 4 consecutive pixels: bias is a per-lane scalar and the accumulators go to a
// 32 KB bf16 staging image in LDS with 8-byte stores; from there every thread handles whole 16-byte pixel vectors of
// one channel row (epilogue math, coalesced 512-byte row segments to global memory).
// The first two stages of the next tile are requested before the epilogue of the current one starts.
template <int TM>                // TM = 32-row channel tiles per wave (BM = 2 * TM * 32)
__global__ __launch_bounds__(512, 2) void conv_nn_ring_kernel(const ConvNN p, int tilesM, long long tilesN, long long ntiles) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TN = 2;                               // 8 waves = 2 (channels) x 4 (pixels); wave tile (TM*32) x 64
    constexpr int BM = 2 * TM * 32, BN = 256, BK = 64;
    constexpr int ASZ = BM * 128, XSZ = BK * BN * 2, STAGE = ASZ + XSZ;
    constexpr int NIA = BM / 64, NIX = 4, NI = NIA + NIX;     // LDS-DMA instructions per wave and stage
    constexpr int EROWS = 64;                           // staging image: 64 channel rows x 256 pixels bf16 = 32 KB
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE + EROWS * 512];
    unsigned char* const stg = smem + 2 * STAGE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, lh = lane >> 5;
    const int nk = (p.K + BK - 1) / BK;                 // the last k-tile may be ragged: its missing activation rows read zeros
    const unsigned rowbytes = (unsigned)(p.N * 2);

    // ---- DMA addressing ----
    // weights: instruction i fills row group rg = wave + 8 i (8 rows x 128 B); lane -> (row = lane >> 3, physical chunk = lane & 7)
    const int ca_log = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);
    // activations: instruction j fills rows kk = 2 (wave + 8 j) + (lane >> 5); lane -> physical chunk lane & 31 of a 512-byte row
    const int kx3 = (2 * (wave & 1) + (lane >> 5)) & 3;                 // kk & 3, the same for every j
    const int cx_log = (lane & 31) ^ (kx3 << 2);
    unsigned voffx[NIX];
#pragma unroll
    for (int j = 0; j < NIX; ++j) voffx[j] = (unsigned)(2 * (wave + 8 * j) + (lane >> 5)) * rowbytes + (unsigned)cx_log * 16u;
    // weights: the buffer ends with the matrix (a ragged last k-tile reads columns k >= lda: the next row's first values,
    // finite, against activation rows that read zero; behind the last row: zeros)
    const v4i_t rsA = make_rsrc_n(p.A, (unsigned)((long long)p.M * p.lda * 2));
    const unsigned lds_w = lds_addr(smem) + __builtin_amdgcn_readfirstlane(wave) * 1024;    // this wave's 1 KB slot of a row-group stripe

    // ---- fragment addressing ----
    const int swa = (l31 >> 1) & 7;
    int aoff[4];                                        // weight fragment (MFMA second operand): row l31 of a 32-row tile, chunk ks*2 + lh
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) aoff[ks] = (wm * TM * 32 + l31) * 128 + (((ks * 2 + lh) ^ swa) * 16);
    const int s15 = lane & 15, g1 = (lane >> 4) & 1;
    int xoff[TN];                                       // activation fragment (first operand): transpose read, rows ks*16 + lh*8 + (s15>>2) [+4]
#pragma unroll
    for (int j = 0; j < TN; ++j)
        xoff[j] = ASZ + (lh * 8 + (s15 >> 2)) * 512 + ((((wn * TN + j) ^ ((s15 >> 2) & 3)) * 4 + g1 * 2 + ((s15 & 3) >> 1)) * 16) + (s15 & 1) * 8;

    const long long first = xcd_remap(blockIdx.x, gridDim.x);
    unsigned voffa[NIA];
    unsigned voffx_t[NIX];
    const u16* xb = p.X;                                // activations of the batch entry being multiplied
    int m0 = 0, bcur = -1;
    long long n0 = 0;

    auto setup_tile = [&](long long t) {                // per-tile DMA addresses (tiles of one pixel range are consecutive ids)
        const int tm = (int)(t % tilesM);
        const long long tnb = t / tilesM;
        const int b = (int)(tnb / tilesN);
        m0 = tm * BM;
        n0 = (tnb % tilesN) * BN;
        if (b != bcur) {
            bcur = b;
            xb = p.X + (long long)b * p.K * p.N;
        }
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
            const int row = min(m0 + (wave + 8 * i) * 8 + (lane >> 3), p.M - 1);     // rows past M: any valid row (never stored)
            voffa[i] = (unsigned)row * (unsigned)(p.lda * 2) + (unsigned)ca_log * 16u;
        }
        // pixels past N (last pixel tile): re-read the last valid chunk of the row (those columns are never stored)
        const int cmax = (int)((min(p.N, n0 + BN) - n0) / 8) - 1;
#pragma unroll
        for (int j = 0; j < NIX; ++j) voffx_t[j] = voffx[j] - (unsigned)max(0, cx_log - cmax) * 16u;
    };
    auto issue = [&](int kt, int stage) {
        const unsigned dst = lds_w + stage * STAGE;
        const unsigned soffa = (unsigned)(kt * BK * 2);
        // activations: a buffer per k-tile that ends with the tile's last existing input channel (rows k >= K read zeros);
        // offsets inside it stay below 64 rows (any K x N fits)
        const v4i_t rsX = make_rsrc_n(xb + (long long)kt * BK * p.N, (unsigned)min(BK, p.K - kt * BK) * rowbytes);
        const unsigned soffx = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(n0 * 2));
        if constexpr (NIA == 4) dma4<8192>(dst, rsA, soffa, voffa[0], voffa[1], voffa[2], voffa[3]);
        else dma3<8192>(dst, rsA, soffa, voffa[0], voffa[1], voffa[2]);
        dma4<8192>(dst + ASZ, rsX, soffx, voffx_t[0], voffx_t[1], voffx_t[2], voffx_t[3]);
    };

    f32x16 acc[TM][TN];
    auto compute = [&](int stage) {
        const unsigned char* sb = smem + stage * STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[TM], xf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(sb + aoff[ks] + i * 4096));
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const unsigned char* q0 = sb + xoff[j] + ks * 16 * 512;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q0));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q0 + 4 * 512));
                const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                xf[j] = __builtin_bit_cast(bf16x8, v);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[j], af[i], acc[i][j], 0, 0, 0);     // D[pixel][channel]
        }
    };

    long long t = first;
    if (t < ntiles) {
        setup_tile(t);
        issue(0, 0);
        if (nk > 1) issue(1, 1);
    }
    for (; t < ntiles; t += gridDim.x) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int cm0 = m0;                               // coordinates of the tile being multiplied (setup_tile moves on below)
        const long long cn0 = n0;
        const int cb = bcur;
        // this lane's bias values, fetched and WAITED FOR here: the wait hipcc emits in front of the first use of an
        // ordinary load is a full vmcnt drain as far as the hardware counter is concerned (the DMA pieces are not in its
        // books); at this point that drain coincides with the wait for the tile's first stage, in the epilogue it would
        // stall on the next tile's prefetch
        float bvr[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mrow = cm0 + (wm * TM + i) * 32 + l31;
            bvr[i] = (p.bias && mrow < p.M) ? p.bias[mrow] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(bvr[i]));
        for (int kt = 0; kt < nk; ++kt) {
            if (kt == 0 || kt + 1 >= nk) wait_vmcnt<0>(); else wait_vmcnt<NI>();     // kt == 0 also drains the previous tile's stores
            __builtin_amdgcn_s_barrier();
            compute(kt & 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < nk) issue(kt + 2, kt & 1);
        }
        // next tile: its first two stages fly during this tile's epilogue.  With a fused multiplicand / residual the
        // epilogue has loads of its own, and the wait in front of their first use drains the whole counter (see above):
        // there the prefetch is issued in the last round, after those loads have been consumed
        const bool have_next = t + gridDim.x < ntiles;
        const bool epi_loads = p.G || p.R;
        if (have_next) setup_tile(t + gridDim.x);
        if (have_next && !epi_loads) {
            issue(0, 0);                                  // (both stages are free: the k-loop ended with a barrier)
            if (nk > 1) issue(1, 1);
        }

        // ---- epilogue: TM rounds of 64 channel rows (32 per wave row) through the staging image ----
        const long long plane = (long long)cb * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            // (a) this thread's four 8-pixel vectors of the round and the operands it has to fetch for them
            long long off[4];
            bool live[4];
            uint4 gq[4], rq[4];
#pragma unroll
            for (int u4 = 0; u4 < 4; ++u4) {
                const int idx = tid + 512 * u4;
                const int row = idx >> 5, ch = idx & 31;
                const int m = cm0 + ((row >> 5) * TM + i) * 32 + (row & 31);
                const long long n = cn0 + ch * 8;
                live[u4] = m < p.M && n < p.N;
                off[u4] = plane + (long long)m * p.N + n;
                gq[u4] = rq[u4] = make_uint4(0, 0, 0, 0);
                if (live[u4] && p.G) gq[u4] = ld16(p.G + off[u4]);
                if (live[u4] && p.R) rq[u4] = ld16(p.R + off[u4]);
            }
            // (b) accumulators (+ bias) -> bf16 -> staging image
            const float bv = bvr[i];
            const int lrow = wm * 32 + l31;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int px = (wn * TN + j) * 32 + 8 * q + 4 * lh;           // 4 consecutive pixels of this lane's channel
                    uint2 u;
                    u.x = pack_bf16x2(acc[i][j][4 * q] + bv, acc[i][j][4 * q + 1] + bv);
                    u.y = pack_bf16x2(acc[i][j][4 * q + 2] + bv, acc[i][j][4 * q + 3] + bv);
                    *reinterpret_cast<uint2*>(stg + lrow * 512 + (((px >> 3) ^ (lrow & 15)) * 16) + ((px >> 2) & 1) * 8) = u;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (epi_loads) {
#pragma unroll
                for (int u4 = 0; u4 < 4; ++u4) {          // consume the fetched operands here (one wait), then the prefetch may go
                    asm volatile("" : "+v"(gq[u4].x), "+v"(gq[u4].y), "+v"(gq[u4].z), "+v"(gq[u4].w));
                    asm volatile("" : "+v"(rq[u4].x), "+v"(rq[u4].y), "+v"(rq[u4].z), "+v"(rq[u4].w));
                }
                if (i == TM - 1 && have_next) {
                    issue(0, 0);
                    if (nk > 1) issue(1, 1);
                }
            }
            // (c) staging image -> epilogue math -> 16-byte stores
#pragma unroll
            for (int u4 = 0; u4 < 4; ++u4) {
                const int idx = tid + 512 * u4;
                const int row = idx >> 5, ch = idx & 31;
                const uint4 raw = *reinterpret_cast<const uint4*>(stg + row * 512 + ((ch ^ (row & 15)) * 16));
                if (live[u4]) {
                    const long long o = off[u4];
                    if (!p.act && !p.G && !p.R) {
                        mk_st16(p.Y + o, raw, p.nt);
                    } else {
                        if (p.act && p.Ypre) mk_st16(p.Ypre + o, raw, p.nt);
                        const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[2 * e] = __uint_as_float(w[e] << 16);
                            v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
                        }
                        if (p.act) {
                            gelu_fast_n<8>(v);
                        }
                        if (p.G) {
                            const uint32_t gw[4] = {gq[u4].x, gq[u4].y, gq[u4].z, gq[u4].w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float ga2[2] = {__uint_as_float(gw[e] << 16), __uint_as_float(gw[e] & 0xffff0000u)};
                                float gd2[2];
                                gelu_grad_fast_n<2>(ga2, gd2);
                                v[2 * e] *= gd2[0];
                                v[2 * e + 1] *= gd2[1];
                            }
                        }
                        if (p.R) {
                            const uint32_t rw[4] = {rq[u4].x, rq[u4].y, rq[u4].z, rq[u4].w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                v[2 * e] += __uint_as_float(rw[e] << 16);
                                v[2 * e + 1] += __uint_as_float(rw[e] & 0xffff0000u);
                            }
                        }
                        uint4 out;
                        out.x = pack_bf16x2(v[0], v[1]);
                        out.y = pack_bf16x2(v[2], v[3]);
                        out.z = pack_bf16x2(v[4], v[5]);
                        out.w = pack_bf16x2(v[6], v[7]);
                        mk_st16(p.Y + o, out, p.nt);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                   // the staging image is free for the next round
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------
// Forward / data-gradient channel GEMM with the WEIGHTS STATIONARY IN REGISTERS (K = 384, the embedding width).
// The ring kernel above re-ingests its weight tile for every pixel tile: 196 KB of weights + 196 KB of activations
// per 128 KB of output, and the memory path of a CU (about 27 GB/s, loads + stores, L2 hits included) is what bounds
// it.  Here a 256-thread workgroup (one wave per SIMD, the full 512-register file per lane) owns a slab of 384 output
// channels for its whole life: wave w keeps W[96 w .. 96 w + 95][0 .. 383] as 72 MFMA operand fragments (288 VGPRs),
// loaded once.  Only activations stream: pixel tiles of 64 pixels, cut into chunks of 64 input channels (8 KB) that
// travel through a ring of 16 LDS slots by LDS-DMA, up to LOOK chunks (a whole pixel tile and more) ahead of the
// multiplication and across tile boundaries.  Ingest per output element: 2 bytes (M = 384: every activation is read
// once) instead of 6.  Output: D[pixel][channel] accumulators -> bf16 -> 16 KB staging image -> whole 128-byte rows.
//   activations in LDS: rows of 128 B (64 pixels of one input channel), chunk index XOR ((k >> 1) & 1) << 2
//   staging image:      rows of 128 B (64 pixels of one output channel), chunk index XOR (row >> 1) & 7
// The stores of the epilogue are buffer stores that are always issued (lanes without a valid target carry an
// out-of-range offset), so the number of memory instructions between a DMA piece and the wait for it is a constant
// and the counted s_waitcnt vmcnt(N) stays exact across tile boundaries.
template <int N_> __device__ __forceinline__ void wait_vmcnt_le() {      // N_ may exceed the 6-bit field only in dead branches
    if constexpr (N_ <= 63) wait_vmcnt<N_>(); else wait_vmcnt<63>();
}

// TM: 32-row channel tiles per wave (slab = 4 waves x TM x 32 = 256 or 384 channels); KCH: input channels per ring
// chunk and barrier (64 or 128)
// KS: k16-steps of the contraction.  24 = the embedding width K = 384.  5 (round 4) = the 73-channel edges of the network
// (encoder 384 <- 73 with bias + GELU + pre-activation, data gradient of the decoder's last layer 384 <- 73 with gelu'): K = 73 is
// padded to lda = 80 by the weight image; the ONE chunk of a pixel tile is 96 input-channel rows (whole DMA pieces of 8 rows per
// wave), of which rows >= K arrive as zeros (the activation descriptor ends with row K - 1) and rows >= 80 are never multiplied.
// Those launches are pure output streams (9 KB in, 48 KB out per tile, 30 MFMAs per wave): with 60 weight registers instead of
// 288 two workgroups share a CU (the variant without epilogue operand), so that one's epilogue overlaps the other's loads.
template <int TM, int KCH, bool PRE, bool EPI_LOADS, int KS = 24>
__global__ __launch_bounds__(256, (KS < 24 && !EPI_LOADS) ? 2 : 1) void conv_nn_astat_kernel(const ConvNN p, int slabs, long long tilesN) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr bool SMALLK = KS < 24;
    static_assert(KS == 24 || (KS * 16 <= KCH && KCH % 32 == 0), "small K: one chunk per pixel tile");
    constexpr int TN = 2;                               // wave tile: TM * 32 channels x 64 pixels
    constexpr int BM = 4 * TM * 32, BN = 64, NT_ = 256;
    constexpr int CH = KCH * 128;                       // one chunk: KCH input channels x 64 pixels, bf16
    constexpr int NCH = SMALLK ? 1 : 384 / KCH;         // chunks per pixel tile
    constexpr int K4 = KS / NCH;                        // k16-steps per chunk
    constexpr int ROUNDS_ = TM;
    // EPI_LOADS: the epilogue operand (gelu'(G) factor or skip tensor R) of every staging round arrives by LDS-DMA in its
    // own 16 KB image, requested at the START of the pixel tile, so that it lands behind the tile's multiplications;
    // the ring gives up one slot's worth of space per round for that (128 -> 96 KB with three rounds)
    constexpr int EBYTES = EPI_LOADS ? ROUNDS_ * 128 * 128 : 0;
    // ring slots (small K without epilogue operand: 5 slots = 76 KB with the staging image, two workgroups per CU)
    constexpr int NSLOT = SMALLK ? (EPI_LOADS ? 8 : 5) : (EPI_LOADS ? (144 * 1024 - EBYTES) : 128 * 1024) / CH;
    constexpr int NP = KCH / 32;                        // DMA pieces (8 rows x 128 B) per wave and chunk
    constexpr int ROUNDS = TM;                          // staging rounds of 128 channel rows (32 per wave)
    constexpr int NSTORE = ROUNDS * 4 * (PRE ? 2 : 1);  // epilogue memory instructions per thread and tile (always issued)
    constexpr int EPIECES = EPI_LOADS ? ROUNDS * 4 : 0; // DMA pieces of the epilogue operand per wave and tile
    // chunks in flight ahead of the one being multiplied: as many as the ring and the 6-bit vmcnt field allow
    constexpr int LOOK = SMALLK ? ((PRE || EPI_LOADS) ? 2 : 3) : (KCH == 64 ? 8 : ((NP * 4 + 2 * NSTORE <= 63) ? 5 : 4));
    // small K with an epilogue operand: memory instructions a wave issues between the pieces of chunk c (requested in tile
    // c - LOOK, after that tile's multiplications) and the wait for them at the start of tile c — the stores of tile c - LOOK,
    // then operand pieces + chunk pieces + stores of each of the LOOK - 1 tiles in between
    constexpr int SMALLK_EPI_WAIT = NSTORE + (LOOK - 1) * (EPIECES + NP + NSTORE);
    static_assert(!(SMALLK && EPI_LOADS) || SMALLK_EPI_WAIT <= 63, "vmcnt is a 6-bit counter");
    constexpr int NEPI_MAX = (LOOK + NCH - 1) / NCH;    // tile ends the look-ahead window can span
    static_assert(LOOK <= NSLOT - 1, "a slot is refilled only after every wave has left it");
    static_assert(EPI_LOADS || NP * (LOOK - 1) + NEPI_MAX * NSTORE <= 63, "vmcnt is a 6-bit counter");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NSLOT * CH + EBYTES + 128 * 128];
    unsigned char* const ebuf = smem + NSLOT * CH;      // [ROUNDS][128 rows][128 B], linear (as the DMA writes it)
    unsigned char* const stg = ebuf + EBYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool nt_st = p.nt != 0;                       // (kernel argument: scalar, the branch around each store is uniform)
    const int l31 = lane & 31, lh = lane >> 5;
    const int s15 = lane & 15, g1 = (lane >> 4) & 1;
#if MK_ASTAT_DIAG
    unsigned long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_readcyclecounter();
    const unsigned long long tstart = tprev;
#endif

    // ---- work: blockIdx.y = batch entry; slab = id % slabs keeps the weights; pixel tiles id / slabs, + gridDim.x / slabs, ... ----
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int slab = vid % slabs;
    const int pstride = gridDim.x / slabs;              // the launcher makes gridDim.x a multiple of slabs
    const int pfirst = vid / slabs;
    const int tilesN32 = (int)tilesN;
    const int my_tiles = pfirst < tilesN32 ? (tilesN32 - pfirst + pstride - 1) / pstride : 0;
    const int cb = blockIdx.y;
    const int m_base = slab * BM + wave * (TM * 32);
    const unsigned nbytes = (unsigned)(p.N * 2);        // one channel row; all in-plane byte offsets fit 32 bits (checked by the launcher)

    // ---- the stationary operand: W[m_base + i*32 + l31][ks*16 + lh*8 .. +7] ----
    bf16x8 af[TM][KS];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = min(m_base + i * 32 + l31, p.M - 1);
        const u16* src = p.A + (long long)row * p.lda + lh * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) af[i][ks] = __builtin_bit_cast(bf16x8, ld16(src + ks * 16));
    }
    float bvr[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mrow = m_base + i * 32 + l31;
        bvr[i] = (p.bias && mrow < p.M) ? p.bias[mrow] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {                      // wait for the weights here, before any DMA piece is in flight
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(af[i][ks]));
        asm volatile("" : "+v"(bvr[i]));
    }

    // ---- DMA addressing: a chunk = KCH / 8 pieces of 1 KB (8 rows x 128 B); wave w issues pieces NP w .. NP w + NP - 1 ----
    const unsigned lds0 = lds_addr(smem) + __builtin_amdgcn_readfirstlane(wave) * (NP * 1024);
    unsigned vx[NP];
    // logical 16-byte chunk this lane fetches: (lane & 7) ^ swizzle(row), swizzle = ((row >> 1) & 1) << 2 with
    // row = 8 * piece + (lane >> 3): the same for every piece
    const int cxl = (lane & 7) ^ ((((lane >> 3) >> 1) & 1) << 2);
#pragma unroll
    for (int q = 0; q < NP; ++q) vx[q] = (unsigned)((wave * NP + q) * 8 + (lane >> 3)) * nbytes + (unsigned)cxl * 16u;
    // the DMA stream: chunks in the order they are multiplied (tile after tile, NCH chunks each), kept as running scalar
    // state so that issuing one costs a handful of scalar instructions
    // small K: the descriptor ends with input channel K - 1, so that the chunk's rows K .. KCH - 1 arrive as zeros
    const v4i_t rsX = SMALLK ? make_rsrc_n(p.X + (long long)cb * p.K * p.N, (unsigned)p.K * nbytes) : make_rsrc(p.X + (long long)cb * p.K * p.N);
    const unsigned kstride = (unsigned)KCH * nbytes;    // KCH input channels further
    int i_pt = pfirst, i_kc = 0, i_slot = 0;            // next chunk to issue: pixel tile, chunk within it, ring slot
    auto issue_next = [&]() {
        const unsigned n0b = (unsigned)i_pt * (BN * 2);                              // first pixel of the tile, in bytes
        const int cmax = min(7, (int)((nbytes - n0b) / 16) - 1);                     // pixels past N: re-read the last valid chunk
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)i_kc * kstride + n0b));
        const unsigned back = (unsigned)max(0, cxl - cmax) * 16u;
        if constexpr (NP == 2) dma2<1024>(lds0 + (unsigned)i_slot * CH, rsX, soff, vx[0] - back, vx[1] - back);
        else if constexpr (NP == 3) dma3<1024>(lds0 + (unsigned)i_slot * CH, rsX, soff, vx[0] - back, vx[1] - back, vx[2] - back);
        else dma4<1024>(lds0 + (unsigned)i_slot * CH, rsX, soff, vx[0] - back, vx[1] - back, vx[2] - back, vx[3] - back);
        i_slot = i_slot + 1 == NSLOT ? 0 : i_slot + 1;
        if (++i_kc == NCH) {
            i_kc = 0;
            i_pt += pstride;
        }
    };

    // ---- epilogue operand by DMA (EPI_LOADS): G if present, else R; round i, wave w: channel rows m_base + 32 i .. + 31 as
    // four pieces of 8 rows x 128 B into ebuf + i * 16 KB + w * 4 KB.  Rows past M re-read row M - 1 (their results are
    // never stored), pixel chunks past N the last valid chunk (as the activation stream does)
    const u16* const eop = p.G ? p.G : p.R;
    const v4i_t rsE = make_rsrc(EPI_LOADS ? (const void*)(eop + (long long)cb * p.M * p.N) : (const void*)p.X);
    const unsigned ldsE = lds_addr(ebuf) + __builtin_amdgcn_readfirstlane(wave) * 4096;
    auto issue_epi = [&](long long cn0) {               // (offsets recomputed per tile: the kernel has no registers to spare)
        const unsigned n0b = (unsigned)(cn0 * 2);
        const int cmax = min(7, (int)((nbytes - n0b) / 16) - 1);
        const unsigned col = (unsigned)min(lane & 7, cmax) * 16u;
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)n0b);
#pragma unroll
        for (int i = 0; i < (EPI_LOADS ? ROUNDS_ : 0); ++i) {
            unsigned v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = (unsigned)min(m_base + i * 32 + q * 8 + (lane >> 3), p.M - 1) * nbytes + col;
            dma4<1024>(ldsE + (unsigned)i * 16384u, rsE, soff, v[0], v[1], v[2], v[3]);
        }
    };

    // ---- fragment addressing inside a chunk: transpose read, rows k4*16 + lh*8 + (s15 >> 2) [+4] ----
    int xoff[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int kk = lh * 8 + (s15 >> 2);
        const int chunk = (j * 4 + g1 * 2 + ((s15 & 3) >> 1)) ^ (((kk >> 1) & 1) << 2);
        xoff[j] = kk * 128 + chunk * 16 + (s15 & 1) * 8;
    }

    const int nchunks = my_tiles * NCH;
    MK_AS_STAMP(7);
    for (int c = 0; c < LOOK && c < nchunks; ++c) issue_next();
    int c_slot = 0;                                     // ring slot of the chunk being multiplied

    f32x16 acc[TM][TN];
    const long long plane0 = (long long)cb * p.M * p.N;
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Y + plane0), 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)((PRE ? p.Ypre : p.Y) + plane0), 0, 0x80000000u, 0x00020000);

    for (int ts = 0; ts < my_tiles; ++ts) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const long long cn0 = (long long)(pfirst + ts * pstride) * BN;

#pragma unroll
        for (int kc = 0; kc < NCH; ++kc) {
            const int c = ts * NCH + kc;
            // Memory instructions this wave issued after the NP pieces of chunk c (at step c - LOOK, or in the prologue):
            // NP pieces for each of the LOOK - 1 younger chunks, plus NSTORE for every epilogue that ran in between: the
            // tile ends among the steps c - LOOK .. c - 1, i.e. ceil((LOOK - kc) / NCH) of them once the stream is that old,
            // fewer within the first tiles (never more than ts).  vmcnt retires in order (loads and stores alike), so "at
            // most that many outstanding" means chunk c has landed.  Near the end of the stream fewer chunks are in flight:
            // wait for everything.
            const int nfull = (LOOK - kc + NCH - 1) / NCH;      // a constant once the kc loop is unrolled
            if (c + LOOK > nchunks) {
                wait_vmcnt<0>();
            } else if (SMALLK && EPI_LOADS) {
                // (the K = 384 form below drains the counter at every tile start, which is harmless with six or three chunks per
                // tile requested a whole tile ahead; with ONE chunk per tile it would wait for the chunk requested a moment ago)
                if (ts < LOOK) wait_vmcnt<0>(); else wait_vmcnt_le<SMALLK_EPI_WAIT>();
            } else if (EPI_LOADS) {
                // Round 4: counted like the other variants (the drain that stood here — s_waitcnt vmcnt(0) at the start of every
                // tile — waited for the chunk requested ONE step earlier, i.e. exposed a full memory round trip per pixel tile;
                // tools/vmcnt_check.py reports the slack of every wait and checks this count against a model of the stream).
                // Behind the pieces of chunk c (requested in step c - LOOK, after that step's operand images) a wave issued:
                // the LOOK - 1 younger chunks, the stores of the nfull tile ends in between and the operand images of the tile
                // starts strictly in between (nfull of them, one fewer when this step is itself a tile start: its images are
                // requested behind this wait).  The first tiles have fewer epilogues behind them: drained.
#if MK_ASTAT_EPI_DRAIN          // the round-3 behaviour, kept for the same-box A/B (tools/ab_fast.sh conv1x1 drain:-DMK_ASTAT_EPI_DRAIN=1)
                if (kc == 0) wait_vmcnt<0>();
                if (true) {
                } else
#endif
                if (ts < NEPI_MAX) {
                    wait_vmcnt<0>();
                } else if (kc == 0) {
                    if (nfull == 1) wait_vmcnt_le<NP * (LOOK - 1) + NSTORE>();
                    else if (nfull == 2) wait_vmcnt_le<NP * (LOOK - 1) + 2 * NSTORE + EPIECES>();
                    else wait_vmcnt_le<NP * (LOOK - 1) + 3 * NSTORE + 2 * EPIECES>();
                } else {
                    if (nfull == 1) wait_vmcnt_le<NP * (LOOK - 1) + NSTORE + EPIECES>();
                    else if (nfull == 2) wait_vmcnt_le<NP * (LOOK - 1) + 2 * NSTORE + 2 * EPIECES>();
                    else wait_vmcnt_le<NP * (LOOK - 1) + 3 * NSTORE + 3 * EPIECES>();
                }
            } else {
                const int nepi = min(nfull, ts);
                if (nepi == 0) wait_vmcnt_le<NP * (LOOK - 1)>();
                else if (nepi == 1) wait_vmcnt_le<NP * (LOOK - 1) + NSTORE>();
                else if (nepi == 2) wait_vmcnt_le<NP * (LOOK - 1) + 2 * NSTORE>();
                else wait_vmcnt_le<NP * (LOOK - 1) + 3 * NSTORE>();
            }
            __builtin_amdgcn_s_barrier();
            MK_AS_STAMP(0);
            if (EPI_LOADS && kc == 0) issue_epi(cn0);      // the previous tile's last round left the operand images behind this barrier
            const unsigned char* sb = smem + c_slot * CH;
            c_slot = c_slot + 1 == NSLOT ? 0 : c_slot + 1;
#pragma unroll
            for (int k4 = 0; k4 < K4; ++k4) {
                bf16x8 xf[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const unsigned char* q0 = sb + xoff[j] + k4 * 16 * 128;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q0));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q0 + 4 * 128));
                    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    xf[j] = __builtin_bit_cast(bf16x8, v);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[j], af[i][kc * K4 + k4], acc[i][j], 0, 0, 0);
            }
            // chunk c + LOOK goes to slot (c + LOOK) % NSLOT, last multiplied at step c + LOOK - NSLOT <= c - 1: every wave
            // finished that step before it reached the barrier of the step after it, which lies behind us
#if MK_ASTAT_DIAG
            asm volatile("s_nop 0" : "+v"(acc[TM - 1][TN - 1]));      // (the stamp must not move in front of the last MFMA)
#endif
            MK_AS_STAMP(1);
            if (c + LOOK < nchunks) issue_next();
            MK_AS_STAMP(2);
        }

        // ---- epilogue: TM rounds of 128 channel rows (32 per wave) x 64 pixels through the staging image ----
        if constexpr (EPI_LOADS) {
            // the operand images were requested before this tile's NCH chunk requests (NP pieces each): in-order retirement
            // makes "at most NP * NCH outstanding" mean they have landed; at the end of the stream fewer chunks follow them
            if ((ts + 1) * NCH - 1 + LOOK < nchunks) wait_vmcnt_le<NP * NCH>(); else wait_vmcnt<0>();
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            unsigned voff[4];
#pragma unroll
            for (int u4 = 0; u4 < 4; ++u4) {
                const int idx = tid + NT_ * u4;
                const int row = idx >> 3, ch = idx & 7;
                const int m = slab * BM + (row >> 5) * (TM * 32) + i * 32 + (row & 31);
                const long long n = cn0 + ch * 8;
                const bool live = m < p.M && n < p.N;
                const long long o = (long long)m * p.N + n;
                voff[u4] = live ? (unsigned)(o * 2) : 0xC0000000u;          // out of range: the store is dropped
            }
            const float bv = bvr[i];
            const int lrow = wave * 32 + l31;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int px = j * 32 + 8 * q + 4 * lh;
                    uint2 u;
                    u.x = pack_bf16x2(acc[i][j][4 * q] + bv, acc[i][j][4 * q + 1] + bv);
                    u.y = pack_bf16x2(acc[i][j][4 * q + 2] + bv, acc[i][j][4 * q + 3] + bv);
                    *reinterpret_cast<uint2*>(stg + lrow * 128 + (((px >> 3) ^ ((lrow >> 1) & 7)) * 16) + ((px >> 2) & 1) * 8) = u;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            MK_AS_STAMP(3);
            __builtin_amdgcn_s_barrier();
            MK_AS_STAMP(4);
#pragma unroll
            for (int u4 = 0; u4 < 4; ++u4) {
                const int idx = tid + NT_ * u4;
                const int row = idx >> 3, ch = idx & 7;
                const uint4 raw = *reinterpret_cast<const uint4*>(stg + row * 128 + ((ch ^ ((row >> 1) & 7)) * 16));
                typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
                if (PRE) MK_BUF_ST16((u32x4_t{raw.x, raw.y, raw.z, raw.w}), rsP, voff[u4], nt_st);
                uint4 ev = make_uint4(0, 0, 0, 0);          // EPI_LOADS: the DMA image of this round (rows of 128 B, linear)
                if constexpr (EPI_LOADS) ev = *reinterpret_cast<const uint4*>(ebuf + i * 16384 + row * 128 + ch * 16);
                uint4 out = raw;
                if (p.act || EPI_LOADS) {
                    const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e] = __uint_as_float(w[e] << 16);
                        v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
                    }
                    if (p.act) {
                        gelu_fast_n<8>(v);
                    }
                    if (EPI_LOADS && p.G) {
                        const uint32_t gw[4] = {ev.x, ev.y, ev.z, ev.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float ga2[2] = {__uint_as_float(gw[e] << 16), __uint_as_float(gw[e] & 0xffff0000u)};
                            float gd2[2];
                            gelu_grad_fast_n<2>(ga2, gd2);
                            v[2 * e] *= gd2[0];
                            v[2 * e + 1] *= gd2[1];
                        }
                    }
                    if (EPI_LOADS && !p.G) {
                        const uint32_t rw[4] = {ev.x, ev.y, ev.z, ev.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[2 * e] += __uint_as_float(rw[e] << 16);
                            v[2 * e + 1] += __uint_as_float(rw[e] & 0xffff0000u);
                        }
                    }
                    out.x = pack_bf16x2(v[0], v[1]);
                    out.y = pack_bf16x2(v[2], v[3]);
                    out.z = pack_bf16x2(v[4], v[5]);
                    out.w = pack_bf16x2(v[6], v[7]);
                }
                MK_BUF_ST16((u32x4_t{out.x, out.y, out.z, out.w}), rsY, voff[u4], nt_st);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            MK_AS_STAMP(5);
            __builtin_amdgcn_s_barrier();
            MK_AS_STAMP(6);
        }
    }
#if MK_ASTAT_DIAG
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&g_astat_diag[k], dg[k]);
        atomicAdd(&g_astat_diag[8], tprev - tstart);
        atomicAdd(&g_astat_diag[9], 1ull);
        atomicAdd(&g_astat_diag[10], (unsigned long long)my_tiles);
    }
#endif
#endif
}

// ------------------------------------------------------------------------------------------
// Weight-stationary channel GEMM, second form (round 4): TWO wave groups, one multiplying while the other runs its epilogue.
//
// s_memtime stamps of the kernel above (profiles/r04_astat_diag.txt): a wave spends 46 % of its cycles in the MFMA segment and
// 35 % (plain) to 55 % (bias + GELU + pre-activation, gelu') in the epilogue — convert, stage, read back, epilogue math, stores —
// and only 7 % waiting for memory: with ONE wave per SIMD the matrix pipe idles through every epilogue and the vector ALU through
// every multiplication.  Here the 384-row slab is held by EIGHT waves (48 rows each = three 16-row tiles of
// v_mfma_f32_16x16x32_bf16, 36 weight fragments = 144 registers, so that two waves fit a SIMD's register file); waves 0-3
// (group 0) and 4-7 (group 1) run the SAME seven phases per 64-pixel tile — three chunk multiplications, one staging phase
// (all 48 rows of a wave; the accumulators are dead from then on, which is what lets the epilogue math fit beside 144 weight
// registers), three rounds of {read back 16 rows per wave, epilogue math, stores} — with group 1 THREE phases behind group 0.
// Every phase boundary is a workgroup barrier that all eight waves execute ("tick"), but on every SIMD one wave is in the matrix
// pipe while its partner is in the vector ALU / LDS / store path:
//
//   tick mod 7      0    1    2    3    4    5    6
//   group 0         M0   M1   M2   S    R0   R1   R2         M = multiply chunk kc, S = stage, R = read back + math + store
//   group 1         R0'  R1'  R2'  M0   M1   M2   S          (' = previous tile)
//
// Activations: chunks of 128 input channels x 64 pixels (16 KB) in a ring of NSLOT slots (6; 4 beside the epilogue operand
// images); chunk c + NSLOT is requested by all eight waves (two 1 KB pieces each) one tick after group 1 multiplied chunk c,
// i.e. at ticks 4, 5, 6 of the period, at least four (eight with 6 slots) ticks before group 0 needs it.  The instructions a wave issues between a
// chunk's request and the wait for it are a fixed multiset per (group, chunk position) once the stream is two tiles old —
// a1 NP + b1 NS3 + e1 EPIECES with the small tables below, derived from the schedule and checked by tools/vmcnt_check.py against
// an in-order retirement model of both groups (first tiles and end of stream: drained).
// LDS: ring 96 / 64 KB + 24 KB staging image per group + (epilogue operand) 24 KB per group = 144 / 160 KB.
// Fragment reads: ds_read_b64_tr_b16 on [k][pixel] rows of 128 B; the 16-byte chunk index of a row is XOR-swizzled with
// ((k >> 1) & 1) << 2 (as above) and (k >> 3 & 1) << 1 (the two 16-lane groups of a half-wave read k-blocks 8 rows apart).
template <bool PRE, bool EPI_LOADS>
__global__ __launch_bounds__(512) void conv_nn_astat2_kernel(const ConvNN p, int slabs, long long tilesN) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    constexpr int KS = 12;                              // k32-steps: K = 384
    constexpr int KCH = 128, NCH = 3, K4 = 4;           // chunk: 128 input channels = 4 k32-steps; 3 chunks per pixel tile
    constexpr int CH = KCH * 128;                       // 16 KB
    constexpr int NSLOT = 4, NP = 2;                    // ring slots (six measured slower); DMA pieces per wave and chunk
    constexpr int PT = 4, CT = 3;                       // 16-pixel tiles per pixel tile; 16-row channel tiles per wave
    constexpr int PER = 7, OFF = 3;                     // ticks per tile; ticks group 1 runs behind group 0
    constexpr int NS3 = 2 * (PRE ? 2 : 1);              // stores per thread and read-back round (always issued)
    constexpr int EPIECES = EPI_LOADS ? 6 : 0;          // DMA pieces of the epilogue operand per wave and tile (192 rows / 4 waves / 8)
    // steady-state count of memory instructions between a chunk's request and the wait for it, per (group, chunk position)
    constexpr int WA[2][3] = {{3, 2, 1}, {3, 2, 1}};    // [group][kc]: chunk requests
    constexpr int WB[2][3] = {{3, 2, 1}, {3, 1, 2}};    //              read-back rounds (stores)
    constexpr int WE[2][3] = {{1, 1, 1}, {1, 0, 0}};    //              operand requests
#define MK_A2_WAIT(G, K) (WA[G][K] * NP + WB[G][K] * NS3 + WE[G][K] * EPIECES)
    static_assert(MK_A2_WAIT(0, 0) <= 63 && MK_A2_WAIT(1, 2) <= 63, "vmcnt is a 6-bit counter");
    constexpr int EBYTES = EPI_LOADS ? 2 * 192 * 128 : 0;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NSLOT * CH + 2 * 192 * 128 + EBYTES];
    unsigned char* const stg = smem + NSLOT * CH;       // [group][192 rows][128 B]: row = 64 ct + 16 (wave in group) + channel in tile
    unsigned char* const ebuf = stg + 2 * 192 * 128;    // [group][192 rows][128 B], linear (as the DMA writes it)

    const int tid = threadIdx.x, lane = tid & 63;
    const bool nt_st = p.nt != 0;                       // (kernel argument: scalar, the branch around each store is uniform)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wg = wave & 3;
    const int c16 = lane & 15, q4 = lane >> 4;
#if MK_ASTAT_DIAG          // [0 chunk wait + barrier, 1 multiply, 2 chunk / operand requests, 3 stage, 4 (operand wait), 5 read back + math + stores, 7 prologue]
    unsigned long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_readcyclecounter();
    const unsigned long long tstart = tprev;
#endif
    // The per-lane addressing of every phase is a handful of integer instructions on the lane id.  hipcc hoists all of it out of
    // the tile loop, and the kernel — 144 weight + 48 accumulator registers per lane of 256 — then spills those values; an opaque
    // copy of the lane id per phase keeps the arithmetic inside the phase (where the vector ALU has slots to spare).
    auto lane_here = [&]() __attribute__((always_inline)) {
        int l = lane;
        asm volatile("" : "+v"(l));
        return l;
    };

    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int slab = vid % slabs;
    const int pstride = gridDim.x / slabs;
    const int pfirst = vid / slabs;
    const int tilesN32 = (int)tilesN;
    const int T = pfirst < tilesN32 ? (tilesN32 - pfirst + pstride - 1) / pstride : 0;
    const int nchunks = NCH * T;
    const int cb = blockIdx.y;
    const int m_base = slab * 384 + wave * 48;          // first output channel of this wave
    const unsigned nbytes = (unsigned)(p.N * 2);

    // ---- the stationary operand: W[m_base + 16 ct + c16][32 ks + 8 q4 .. + 7] ----
    bf16x8 wf[CT][KS];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int row = min(m_base + ct * 16 + c16, p.M - 1);
        const u16* src = p.A + (long long)row * p.lda + q4 * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[ct][ks] = __builtin_bit_cast(bf16x8, ld16(src + ks * 32));
    }
    float bvr[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int mrow = m_base + ct * 16 + c16;
        bvr[ct] = (p.bias && mrow < p.M) ? p.bias[mrow] : 0.f;
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {                   // wait for the weights here, before any DMA piece is in flight
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(wf[ct][ks]));
        asm volatile("" : "+v"(bvr[ct]));
    }

    // ---- DMA of the activation chunks: wave w issues pieces 2 w, 2 w + 1 (rows 16 w .. 16 w + 15 of the chunk) ----
    const unsigned lds0 = lds_addr(smem) + (unsigned)wave * (NP * 1024);
    const v4i_t rsX = make_rsrc(p.X + (long long)cb * p.K * p.N);
    const unsigned kstride = (unsigned)KCH * nbytes;
    auto issue_chunk = [&](int ts, int kc, int slot) __attribute__((always_inline)) {  // chunk 3 ts + kc of this workgroup's stream -> slot (3 ts + kc) % 4
        const unsigned n0b = (unsigned)(pfirst + ts * pstride) * 128u;              // first pixel of the tile, in bytes
        const int cmax = min(7, (int)((nbytes - n0b) / 16) - 1);                    // pixels past N: re-read the last valid chunk
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)kc * kstride + n0b));
        // (per-lane addressing recomputed per request: the kernel has no registers to spare)
        // row = 8 (2 w + q) + (lane >> 3): (row >> 1) & 1 = (lane >> 4) & 1, (row >> 3) & 1 = q
        const int ln = lane_here();
        const int c0 = (ln & 7) ^ ((((ln >> 3) >> 1) & 1) << 2), c1 = c0 ^ 2;
        const unsigned r0 = (unsigned)(wave * 16 + (ln >> 3)) * nbytes;
        dma2<1024>((unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)slot * CH)), rsX, soff,
                   r0 + (unsigned)min(c0, cmax) * 16u, r0 + 8u * nbytes + (unsigned)min(c1, cmax) * 16u);
    };

    // ---- epilogue operand by DMA (G if present, else R): the 192 channel rows of this GROUP, 6 pieces of 8 rows per wave ----
    const u16* const eop = p.G ? p.G : p.R;
    const v4i_t rsE = make_rsrc(EPI_LOADS ? (const void*)(eop + (long long)cb * p.M * p.N) : (const void*)p.X);
    const unsigned ldsE = lds_addr(ebuf) + (unsigned)grp * (192 * 128) + (unsigned)wg * (6 * 1024);
    auto issue_epi = [&](int ts) __attribute__((always_inline)) {
        const unsigned n0b = (unsigned)(pfirst + ts * pstride) * 128u;
        const int cmax = min(7, (int)((nbytes - n0b) / 16) - 1);
        const int ln = lane_here();
        const unsigned col = (unsigned)min(ln & 7, cmax) * 16u;
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)n0b);
        const unsigned le = (unsigned)__builtin_amdgcn_readfirstlane((int)ldsE);      // (wave-uniform: the DMA base goes through M0)
#pragma unroll
        for (int h = 0; h < 2; ++h) {                   // image row r = 48 wg + 8 q + (lane >> 3) <-> channel slab * 384 + 192 grp + r
            unsigned v[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int r = 48 * wg + 8 * (3 * h + q) + (ln >> 3);
                v[q] = (unsigned)min(slab * 384 + 192 * grp + r, p.M - 1) * nbytes + col;
            }
            dma3<1024>(le + (unsigned)h * 3072u, rsE, soff, v[0], v[1], v[2]);
            __builtin_amdgcn_sched_barrier(0);          // (three offsets at a time: registers)
        }
    };

    // ---- fragment addressing inside a chunk: transpose read of rows 32 k4 + 8 q4 + (c16 >> 2) [+ 4], pixels 16 pt + 4 (c16 & 3) .. ----
    int xoff[PT];
    {
        const int kk = q4 * 8 + (c16 >> 2);
        const int swz = ((((kk >> 1) & 1) << 2) ^ ((q4 & 1) << 1));
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) xoff[pt] = kk * 128 + (((2 * pt + ((c16 & 3) >> 1)) ^ swz) * 16) + (c16 & 1) * 8;
    }

    f32x4_t acc[PT][CT];
    const long long plane0 = (long long)cb * p.M * p.N;
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Y + plane0), 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)((PRE ? p.Ypre : p.Y) + plane0), 0, 0x80000000u, 0x00020000);
    unsigned char* const stgG = stg + grp * (192 * 128);
    const unsigned char* const ebufG = ebuf + grp * (192 * 128);

    auto mult = [&](int ts, auto kc_) __attribute__((always_inline)) {                 // multiply chunk kc of tile ts
        constexpr int kc = decltype(kc_)::value;
        if constexpr (kc == 0) {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[pt][ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        const unsigned char* sb = smem + ((NCH * ts + kc) & (NSLOT - 1)) * CH;
        // eight sub-steps (k32-step x pixel half): the fragments of sub-step s + 1 are requested before the six MFMAs of sub-step
        // s are issued (two fragment sets = 16 registers; no deeper: the kernel sits at 256 registers)
        auto frags = [&](bf16x8* xf, int sub) __attribute__((always_inline)) {
            const int k4 = sub >> 1, hp = sub & 1;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned char* q0 = sb + xoff[2 * hp + j] + k4 * 32 * 128;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q0));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q0 + 4 * 128));
                const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                xf[j] = __builtin_bit_cast(bf16x8, v);
            }
        };
        bf16x8 xf[2][2];
        frags(xf[0], 0);
#if MK_A2_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int sub = 0; sub < 2 * K4; ++sub) {
            if (sub + 1 < 2 * K4) frags(xf[(sub + 1) & 1], sub + 1);
            const int k4 = sub >> 1, hp = sub & 1;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[2 * hp + j][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[sub & 1][j], wf[ct][kc * K4 + k4], acc[2 * hp + j][ct], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#if MK_A2_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };
    auto stage = [&]() __attribute__((always_inline)) {                                // 48 rows of this wave x 64 pixels -> the group's staging image (bf16)
        const int ln = lane_here();
        const int c16 = ln & 15, q4 = ln >> 4;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const float bv = bvr[ct];
            const int lrow = ct * 64 + wg * 16 + c16;
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                uint2 u;
                u.x = pack_bf16x2(acc[pt][ct][0] + bv, acc[pt][ct][1] + bv);
                u.y = pack_bf16x2(acc[pt][ct][2] + bv, acc[pt][ct][3] + bv);
                const int chunk = 2 * pt + (q4 >> 1);   // pixels 16 pt + 4 q4 .. + 3
                *reinterpret_cast<uint2*>(stgG + lrow * 128 + ((chunk ^ ((lrow >> 1) & 7)) * 16) + (q4 & 1) * 8) = u;
            }
        }
    };
    auto readback_store = [&](int ts, int ct) __attribute__((always_inline)) {         // 64 staged rows of round ct as whole 128-byte rows: epilogue math + stores
        const unsigned cn0b = (unsigned)(pfirst + ts * pstride) * 128u;           // first pixel of the tile, in bytes (32-bit: M N 2 < 2^31)
        typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
        const int tg = (wg << 6) | lane_here();         // thread inside the group
#pragma unroll
        for (int u2 = 0; u2 < 2; ++u2) {
            const int idx = tg + 256 * u2;
            const int row = idx >> 3, ch = idx & 7;     // row of the round = 16 (wave in group) + channel in tile
            const int srow = ct * 64 + row;             // row of the staging image
            const int grow = (row >> 4) * 48 + ct * 16 + (row & 15);                 // row inside the group's 192 channels
            const int m = slab * 384 + grp * 192 + grow;
            const unsigned nb = cn0b + (unsigned)ch * 16u;
            const bool live = m < p.M && nb < nbytes;
            const unsigned voff = live ? (unsigned)m * nbytes + nb : 0xC0000000u;   // out of range: the store is dropped
            const uint4 raw = *reinterpret_cast<const uint4*>(stgG + srow * 128 + ((ch ^ ((srow >> 1) & 7)) * 16));
            if (PRE) MK_BUF_ST16((u32x4_t{raw.x, raw.y, raw.z, raw.w}), rsP, voff, nt_st);
            uint4 ev = make_uint4(0, 0, 0, 0);
            if constexpr (EPI_LOADS) ev = *reinterpret_cast<const uint4*>(ebufG + grow * 128 + ch * 16);
            uint4 out = raw;
            if (p.act || EPI_LOADS) {
                const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] = __uint_as_float(w[e] << 16);
                    v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
                }
                if (p.act) {
                    gelu_fast_n<8>(v);
                }
                if (EPI_LOADS && p.G) {
                    const uint32_t gw[4] = {ev.x, ev.y, ev.z, ev.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float ga2[2] = {__uint_as_float(gw[e] << 16), __uint_as_float(gw[e] & 0xffff0000u)};
                        float gd2[2];
                        gelu_grad_fast_n<2>(ga2, gd2);
                        v[2 * e] *= gd2[0];
                        v[2 * e + 1] *= gd2[1];
                    }
                }
                if (EPI_LOADS && !p.G) {
                    const uint32_t rw[4] = {ev.x, ev.y, ev.z, ev.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e] += __uint_as_float(rw[e] << 16);
                        v[2 * e + 1] += __uint_as_float(rw[e] & 0xffff0000u);
                    }
                }
                out.x = pack_bf16x2(v[0], v[1]);
                out.y = pack_bf16x2(v[2], v[3]);
                out.z = pack_bf16x2(v[4], v[5]);
                out.w = pack_bf16x2(v[6], v[7]);
            }
            MK_BUF_ST16((u32x4_t{out.x, out.y, out.z, out.w}), rsY, voff, nt_st);
        }
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;

    MK_AS_STAMP(7);
    for (int c = 0; c < NSLOT && c < nchunks; ++c) issue_chunk(c / NCH, c % NCH, c);

    // One tick: (1) wait for this wave's pieces of the chunk that group 0 multiplies first in this tick, (2) barrier, (3) request
    // the chunk whose slot group 1 freed in the previous tick.  gph / gts: position of the tick in the workgroup's stream
    // (tick = 7 gts + gph).  The first two tiles and the last one have fewer instructions behind the chunk (group 1 idles through
    // its first OFF ticks, no request follows the last chunks): drained.
    auto tick = [&](int gts, auto gph_, auto grp_) __attribute__((always_inline)) {
        constexpr int gph = decltype(gph_)::value, g = decltype(grp_)::value;
        if constexpr (gph < NCH) {
            if (gts < T) {
                if (gts <= 1 || gts >= T - 1) wait_vmcnt<0>(); else wait_vmcnt<MK_A2_WAIT(g, gph)>();
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // my LDS reads of the last phase have returned (slots / images are re-used)
        __builtin_amdgcn_s_barrier();
        MK_AS_STAMP(0);
    };
    // ... and at the END of the phase work of ticks 4, 5, 6: group 1 multiplied chunk 3 gts + (gph - 4) in the previous tick, every
    // wave has passed this tick's barrier: the slot is free.  Requested behind the phase work, so that the group that finishes its
    // phase first issues while the other one is still busy (all eight waves issuing right behind the barrier cost 540 cycles per
    // chunk, 13 % of a wave's time: profiles/r04_astat2_diag.txt)
    auto tick_end = [&](int gts, auto gph_) __attribute__((always_inline)) {
        constexpr int gph = decltype(gph_)::value;
        if constexpr (gph > OFF) {
            // chunk 3 gts + gph = tile gts + 1, chunks 1, 2 (ticks 4, 5) and tile gts + 2, chunk 0 (tick 6); slot = chunk & 3
            const int ts = gts + (gph == 6 ? 2 : 1);
            if (ts < T) issue_chunk(ts, gph == 6 ? 0 : gph - 3, (NCH * gts + gph) & (NSLOT - 1));
        }
        MK_AS_STAMP(2);
    };
    // the seven phases of one tile of this group (lts), each behind its tick; grp_: the group as a compile-time constant
    auto tile = [&](int lts, auto grp_) __attribute__((always_inline)) {
        constexpr int g = decltype(grp_)::value;
        // phase p of group g is tick (p + 3 g) mod 7 of global tile lts + (p + 3 g) / 7
#define MK_A2_TICK(P) tick(lts + ((P) + OFF * g) / PER, std::integral_constant<int, ((P) + OFF * g) % PER>{}, grp_)
#define MK_A2_END(P) tick_end(lts + ((P) + OFF * g) / PER, std::integral_constant<int, ((P) + OFF * g) % PER>{})
#if MK_ASTAT_DIAG
#define MK_A2_SEG(K) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); asm volatile("s_nop 0" : "+v"(acc[PT - 1][CT - 1])); MK_AS_STAMP(K); } while (0)
#else
#define MK_A2_SEG(K)
#endif
        MK_A2_TICK(0);
        if constexpr (EPI_LOADS) issue_epi(lts);                // (the previous tile's last round left the images a barrier ago)
        MK_AS_STAMP(2);
        mult(lts, I0{});
        MK_A2_SEG(1);
        MK_A2_END(0);
        MK_A2_TICK(1);
        mult(lts, I1{});
        MK_A2_SEG(1);
        MK_A2_END(1);
        MK_A2_TICK(2);
        mult(lts, I2{});
        MK_A2_SEG(1);
        MK_A2_END(2);
        MK_A2_TICK(3);
        stage();
        MK_A2_SEG(3);
        MK_A2_END(3);
        if constexpr (EPI_LOADS) {
            // behind the operand pieces (requested in this group's phase 0): group 0 nothing (the chunk requests of ticks 4 - 6 sit
            // behind its read-back phases), group 1 those three chunk requests; near the end of the stream fewer: drained.
            // IN FRONT of the phase-4 barrier: vmcnt covers this wave's own LDS-DMA pieces only, and read-back round 0 reads
            // image rows that sibling waves of the group requested (wave wg reads rows 8 wg.. and 32 + 8 wg..) — every wave's
            // pieces must have landed before any wave passes the barrier
            if (g == 0 || lts >= T - 2) wait_vmcnt<0>(); else wait_vmcnt<3 * NP>();
        }
        MK_A2_TICK(4);
        MK_AS_STAMP(4);
        readback_store(lts, 0);
        MK_A2_SEG(5);
        MK_A2_END(4);
        MK_A2_TICK(5);
        readback_store(lts, 1);
        MK_A2_SEG(5);
        MK_A2_END(5);
        MK_A2_TICK(6);
        readback_store(lts, 2);
        MK_A2_SEG(5);
        MK_A2_END(6);
#undef MK_A2_SEG
#undef MK_A2_TICK
#undef MK_A2_END
    };
    if (T > 0) {
        if (grp == 0) {
            for (int lts = 0; lts < T; ++lts) tile(lts, I0{});
            tick(T, I0{}, I0{});                                // group 1 is three ticks behind: its last phases' barriers
            tick(T, I1{}, I0{});
            tick(T, I2{}, I0{});
        } else {
            tick(0, I0{}, I1{});                                // group 0's first three ticks (tile 0's chunks: drained waits)
            tick(0, I1{}, I1{});
            tick(0, I2{}, I1{});
            for (int lts = 0; lts < T; ++lts) tile(lts, I1{});
        }
    }
#if MK_ASTAT_DIAG
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) atomicAdd(&g_astat_diag[q], dg[q]);
        atomicAdd(&g_astat_diag[8], tprev - tstart);
        atomicAdd(&g_astat_diag[9], 1ull);
        atomicAdd(&g_astat_diag[10], (unsigned long long)T);
    }
#endif
#undef MK_A2_WAIT
#endif
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

// ------------------------------------------------------------------------------------------
// wgrad, ring version: one 512-thread workgroup per CU owns an output tile that spans the WHOLE smaller channel
// dimension (TQ = 384 rows of operand Q) and a 256- or 192-row slab of the other operand (P), so every pixel of P is
// read from memory exactly once and Q's pixels are shared by only 2-3 co-scheduled workgroups (the 128 x 128 tiles above
// re-read each operand 3-6 times; measured 1.67x the algorithmic HBM bytes).  Both operands stream through a two-stage
// LDS ring filled by LDS-DMA (buffer_load ... lds: no staging registers, no ds_write pass); a stage holds
// (TP + TQ) rows x 64 pixels = 72-80 KB, i.e. 72-80 KB per CU are in flight at any time while the other stage is
// multiplied.  LDS rows are 128 bytes (one cache line of one channel); the 16-byte chunk index of a row is XOR-swizzled
// with (row >> 1) & 7 — applied to the per-lane SOURCE address, the DMA destination is lane-linear — which makes the
// ds_read_b128 fragment reads conflict-free.
struct ConvWgR {
    const u16* P;    // (B, RP, N)   slab operand (slabs of TP rows)
    const u16* Q;    // (B, RQ, N)   the other operand: slabs of TQ = 384 rows (one slab when RQ <= 384: held in full)
    float* part;     // (S, M, K) fp32 partials in OUTPUT orientation
    int RP, RQ, B, S, slabs, qslabs;
    int ldo;         // row length of the output (= K)
    long long N;
    long long chunk;   // pixels per split (multiple of 64)
    float* bias_part;  // optional (S, M): per-split row sums of G (the bias gradient), see below
};


// SWAP = false: P = G (rows m), Q = X (rows k): tile D[p][q] = out[m][k]
// SWAP = true : P = X (rows k), Q = G (rows m): MFMA operands exchanged so that D[q][p] = out[m][k]
// (either way the lanes of an accumulator row run along k, the contiguous output index)
template <int WP, int WQ, int WTP, int WTQ, bool SWAP>
__global__ __launch_bounds__(512, 2) void conv_wgrad_ring_kernel(const ConvWgR p) {
#if defined(__HIP_DEVICE_COMPILE__)      // buffer resources / LDS-DMA builtins exist in the device pass only (the host pass just needs the stub)
    static_assert(WP * WQ == 8, "8 waves");
    constexpr int TP = WP * WTP * 32, TQ = WQ * WTQ * 32;
    constexpr int BK = 64;                              // pixels per stage: one 128-byte line per row
    constexpr int STAGE = (TP + TQ) * 128;              // bytes
    constexpr int NI = (TP + TQ) / 64;                  // LDS-DMA instructions per wave and stage (8 rows each)
    constexpr int NIP = TP / 64;                        // the first NIP of them fetch P rows
    static_assert(TP % 64 == 0 && TQ == 384, "row groups; Q is fetched by 2 x 3 DMA pieces per wave");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];

    const int bid = xcd_remap(blockIdx.x, gridDim.x);   // the slabs of one pixel split are neighbours on one XCD (share Q in L2)
    const int slab = bid % p.slabs;
    const int qs = (bid / p.slabs) % p.qslabs;
    const int q0 = qs * TQ;                             // first Q row of this tile
    const int sp = bid / (p.slabs * p.qslabs);
    const int splits_per_b = p.S / p.B;
    const int b = sp / splits_per_b;
    const long long nbeg = (long long)(sp % splits_per_b) * p.chunk;
    const long long nend = min(p.N, nbeg + p.chunk);
    const int nk = nbeg < nend ? (int)((nend - nbeg + BK - 1) / BK) : 0;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave / WQ, wq = wave % WQ;
    const int l31 = lane & 31, lh = lane >> 5;

    // ---- LDS-DMA addressing: instruction i of this wave fills row group rg = wave + 8 i (8 rows x 128 B) ----
    // lane -> (row in group = lane >> 3, physical chunk = lane & 7); it fetches the LOGICAL chunk c = phys ^ swizzle(row)
    const int c_log = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);
    const unsigned rowbytes = (unsigned)(p.N * 2);
    unsigned voff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int rg = wave + 8 * i;
        int row;                                                                      // relative to the tile's first row
        if (i < NIP) row = min(rg * 8 + (lane >> 3), p.RP - 1 - slab * TP);           // rows past the operand: any valid row
        else row = min((rg - TP / 8) * 8 + (lane >> 3), p.RQ - 1 - q0);               // (their products are never stored)
        voff[i] = (unsigned)row * rowbytes + (unsigned)c_log * 16u;
    }
    // raw buffers (stride 0, 2^31 records) that start at the tile's first row: offsets stay below 384 rows whatever the
    // channel count; a lane whose voffset is >= 2^31 reads zeros — used for the ragged last pixel tile
    const v4i_t rsP = make_rsrc(p.P + ((long long)b * p.RP + slab * TP) * p.N);
    const v4i_t rsQ = make_rsrc(p.Q + ((long long)b * p.RQ + q0) * p.N);
    const unsigned lds_w = lds_addr(smem) + __builtin_amdgcn_readfirstlane(wave) * 1024;

    auto issue = [&](int kt, int stage) {
        const long long n = nbeg + (long long)kt * BK;
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(n * 2));
        const unsigned dst = lds_w + stage * STAGE;
        unsigned v[NI];
        // ragged last pixel tile of the plane (at most once per workgroup): chunks past nend get a voffset beyond the
        // buffer's 2^31 records and come back as zeros
        const bool ok = n + BK <= nend || n + c_log * 8 < nend;
#pragma unroll
        for (int i = 0; i < NI; ++i) v[i] = ok ? voff[i] : 0xC0000000u;
        if constexpr (NIP == 4) dma4<8192>(dst, rsP, soff, v[0], v[1], v[2], v[3]);
        else dma3<8192>(dst, rsP, soff, v[0], v[1], v[2]);
        dma3<8192>(dst + NIP * 8192, rsQ, soff, v[NIP], v[NIP + 1], v[NIP + 2]);
        dma3<8192>(dst + (NIP + 3) * 8192, rsQ, soff, v[NIP + 3], v[NIP + 4], v[NIP + 5]);
    };

    // ---- fragment addressing: lane (l31, lh) reads row l31 of a 32-row tile, logical chunk ks*2 + lh ----
    const int swz = (l31 >> 1) & 7;
    int pa[4], qa[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int ch = ((ks * 2 + lh) ^ swz) * 16;
        pa[ks] = (wp * WTP * 32 + l31) * 128 + ch;
        qa[ks] = (TP + wq * WTQ * 32 + l31) * 128 + ch;
    }

    f32x16 acc[WTP][WTQ];
#pragma unroll
    for (int i = 0; i < WTP; ++i)
#pragma unroll
        for (int j = 0; j < WTQ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Bias gradient (round 4): db[m] = sum over pixels of G[m] — the same rows this kernel streams anyway; the separate
    // plane-sum pass re-reads every gradient tensor (0.87 ms per step, at 5 TB/s).  G is operand P (rows m) when not swapped, Q
    // when swapped.  The waves of ONE column (row) of the wave grid add up the fragments they read for the MFMAs with
    // v_dot2c_f32_bf16 against (1, 1); only the workgroups of Q slab 0 (P slab 0 when swapped) write their split's sums.
    // Measured, same box (profiles/r04_step_ab_waits_bias_v1.txt, r04_step_ab_bias_v2.txt): the dot products are NOT free beside
    // the MFMAs (about ten cycles each, dependent chains of four): weight-gradient time per step 5.04 -> 5.44 ms with one wave per
    // row doing them (this form: net -0.45 ms of kernel time, -0.15 ms of step time; FourCastNet3 539.6 -> 534.7 ms), 5.05 -> 6.0 ms
    // when the four waves of a row share them by k16-step (every wave then carries a dependent VALU chain: net zero) — so the
    // work stays on one wave per row.
    constexpr int WTG = SWAP ? WTQ : WTP;
    constexpr int NSH = 1;                              // waves sharing the row sums of one set of G rows (see above)
    const int bshare = 0;
    const bool bias_wave = p.bias_part != nullptr && (SWAP ? (wp == 0 && slab == 0) : (wq == 0 && qs == 0));
    float rsum[WTG];
#pragma unroll
    for (int i = 0; i < WTG; ++i) rsum[i] = 0.f;
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    auto add8 = [](float s, bf16x8 v) {
        const bf16x2_t one = {(__bf16)1.0f, (__bf16)1.0f};
        s = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{v[0], v[1]}, one, s, false);
        s = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{v[2], v[3]}, one, s, false);
        s = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{v[4], v[5]}, one, s, false);
        return __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{v[6], v[7]}, one, s, false);
    };

    auto compute = [&](int stage) {
        const unsigned char* sb = smem + stage * STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 pf[WTP], qf[WTQ];
#pragma unroll
            for (int i = 0; i < WTP; ++i) pf[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(sb + pa[ks] + i * 4096));
#pragma unroll
            for (int j = 0; j < WTQ; ++j) qf[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(sb + qa[ks] + j * 4096));
            if (bias_wave && (ks & (NSH - 1)) == bshare) {      // (wave-uniform)
#pragma unroll
                for (int i = 0; i < WTG; ++i) rsum[i] = add8(rsum[i], SWAP ? qf[i] : pf[i]);
            }
#pragma unroll
            for (int i = 0; i < WTP; ++i)
#pragma unroll
                for (int j = 0; j < WTQ; ++j)
                    acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[j], pf[i], acc[i][j], 0, 0, 0)
                                     : __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[i], qf[j], acc[i][j], 0, 0, 0);
        }
    };

    if (nk > 0) issue(0, 0);
    if (nk > 1) issue(1, 1);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) wait_vmcnt<NI>(); else wait_vmcnt<0>();      // this wave's share of stage kt has landed ...
        __builtin_amdgcn_s_barrier();                                   // ... and so has everybody else's
        compute(kt & 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // my fragment reads have RETURNED (a raw s_barrier does not wait) ...
        __builtin_amdgcn_s_barrier();                                   // ... and everybody's: the stage may be overwritten
        if (kt + 2 < nk) issue(kt + 2, kt & 1);
    }

    if (bias_wave) {                                    // lane (l31, lh) holds the pixels of k-block lh of row l31: add the halves
        const int RG = SWAP ? p.RQ : p.RP;              // rows of G = M
#pragma unroll
        for (int i = 0; i < WTG; ++i) {
            const float tot = rsum[i] + __shfl_xor(rsum[i], 32);
            const int m = SWAP ? q0 + (wq * WTQ + i) * 32 + l31 : slab * TP + (wp * WTP + i) * 32 + l31;
            if (lh == 0 && m < RG) p.bias_part[((long long)sp * NSH + bshare) * RG + m] = tot;
        }
    }

    // ---- partial tile -> part[sp] in output orientation (lanes along k) ----
    float* out = p.part + (long long)sp * ((long long)(SWAP ? p.RQ : p.RP) * p.ldo);
#pragma unroll
    for (int i = 0; i < WTP; ++i)
#pragma unroll
        for (int j = 0; j < WTQ; ++j) {
            const int prow0 = slab * TP + (wp * WTP + i) * 32, qrow0 = q0 + (wq * WTQ + j) * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
                // not swapped: D rows = p (m), lanes = q (k);  swapped: D rows = q (m), lanes = p (k)
                const int m = SWAP ? qrow0 + rr : prow0 + rr;
                const int k = SWAP ? prow0 + l31 : qrow0 + l31;
                const bool ok = SWAP ? (m < p.RQ && k < p.RP) : (m < p.RP && k < p.RQ);
                if (ok) out[(long long)m * p.ldo + k] = acc[i][j][r];
            }
        }
#endif
}

// one output per thread (counts that are not multiples of 4, and the M bias sums of a weight gradient: three blocks whose
// threads each walk S partials — with one load in flight that was S dependent memory latencies, 34 us per launch; eight
// splits in flight as below, same summation order)
__global__ void reduce_splits(const float* __restrict__ part, float* __restrict__ out, long long n, int S, int accumulate) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = accumulate ? out[i] : 0.f;
    int k = 0;
    for (; k + 8 <= S; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(long long)(k + u) * n + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < S; ++k) s += part[(long long)k * n + i];
    out[i] = s;
}

// n % 4 == 0: four consecutive outputs per thread, eight splits in flight
__global__ void reduce_splits4(const float4* __restrict__ part, float4* __restrict__ out, long long n4, int S, int accumulate) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 s = accumulate ? out[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    for (; k + 8 <= S; k += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(long long)(k + u) * n4 + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s.x += v[u].x, s.y += v[u].y, s.z += v[u].z, s.w += v[u].w;
    }
    for (; k < S; ++k) {
        const float4 v = part[(long long)k * n4 + i];
        s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
    out[i] = s;
}

}  // namespace

#if MK_ASTAT_DIAG
extern "C" int mk_astat_diag_read(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_astat_diag), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_astat_diag), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

extern "C" int mk_conv1x1_nn(const void* A, const void* X, void* Y, void* Ypre, const float* bias, const void* R,
                             const void* G, int M, int K, int lda, int B, long long N, int act, void* stream) {
    MK_REQUIRE(A && X && Y, "conv1x1_nn: null pointer");
    MK_REQUIRE(M > 0 && K > 0 && B > 0 && N > 0, "conv1x1_nn: bad shape M=%d K=%d B=%d N=%lld", M, K, B, N);
    MK_REQUIRE((lda % 8) == 0 && lda >= K, "conv1x1_nn: lda=%d must be a multiple of 8 and >= K=%d", lda, K);
    MK_REQUIRE((N % 8) == 0, "conv1x1_nn: pixel count %lld must be a multiple of 8", N);
    MK_REQUIRE((((uintptr_t)A | (uintptr_t)X) & 15) == 0, "conv1x1_nn: operands must be 16-byte aligned");
    // streaming stores for outputs that fit the memory-side cache (256 MB) — see mk_st16.  MAKANI_AMD_CONV_NT=0 / 1: never / always (2: by the sum of both outputs)
    static const int nt_env = [] { const char* e = getenv("MAKANI_AMD_CONV_NT"); return e ? atoi(e) : -1; }();
    const long long out_bytes = (long long)B * M * N * 2;          // per output tensor (with the pre-activation there are two)
    const int nt = nt_env >= 0 ? (nt_env == 1 || (nt_env == 2 && out_bytes * ((act && Ypre) ? 2 : 1) <= (256ll << 20))) : (out_bytes <= (256ll << 20));
    ConvNN p{(const u16*)A, (const u16*)X, (u16*)Y, (u16*)Ypre, bias, (const u16*)R, (const u16*)G, M, K, lda, B, N, act, nt};
    static const bool force_tile = [] { const char* e = getenv("MAKANI_AMD_CONV_NN"); return e && e[0] == 't'; }();
    static const bool no_astat = [] { const char* e = getenv("MAKANI_AMD_CONV_NN"); return e && e[0] == 'r'; }();   // "ring": no weight-stationary kernel
    if (!force_tile && !no_astat && lda == 80 && K > 64 && K <= 80 && M >= 256 && (long long)M * N * 2 < (1ll << 31) &&
        (long long)K * N * 2 < (1ll << 32) && N >= 64 && !(R && G) && !((R || G) && act && Ypre)) {
        // the 73-channel edges (K padded to lda = 80): the weight-stationary kernel with 5 k16-steps and one 96-row chunk per tile
        const bool epi_loads = R || G;
        const int slabs = (M + 383) / 384;
        const long long tn = (N + 63) / 64;
        const int wgs = epi_loads ? 256 : 512;                     // two workgroups per CU without the epilogue operand images
        const long long streams = tn < wgs / slabs ? tn : wgs / slabs;
        const dim3 grid((unsigned)(streams * slabs), (unsigned)B), blk(256);
        const bool pre = act && Ypre;
        hipStream_t s = (hipStream_t)stream;
        if (epi_loads) hipLaunchKernelGGL((conv_nn_astat_kernel<3, 96, false, true, 5>), grid, blk, 0, s, p, slabs, tn);
        else if (pre) hipLaunchKernelGGL((conv_nn_astat_kernel<3, 96, true, false, 5>), grid, blk, 0, s, p, slabs, tn);
        else hipLaunchKernelGGL((conv_nn_astat_kernel<3, 96, false, false, 5>), grid, blk, 0, s, p, slabs, tn);
        return mk_check_launch("mk_conv1x1_nn");
    }
    // MAKANI_AMD_ASTAT2: 0 = never, unset / 1 = every K = 384 launch, 3 = round 4's rule (profiles/r04_ab_astat2.txt: plain -7 %, gelu'
    // -14 ... -22 %, skip operand -10 ... -21 %; bias + GELU + pre-activation — two output streams, then bound by its stores —
    // +0 ... 2 %, so that variant stayed on the one-group kernel).  Round 6, with the streaming stores and the one-exponential GELU
    // (gpurun_out/r07y, same box): that variant now runs 10 - 11 % faster on the two-group kernel at 115 200 pixels (0.134 -> 0.119,
    // 0.072 -> 0.065 ms), 4 % at 384 <- 384 / 1 038 240 pixels, equal at 768 <- 384 / 1 038 240: every K = 384 launch takes it
    static const int astat2 = [] { const char* e = getenv("MAKANI_AMD_ASTAT2"); return e ? atoi(e) : 2; }();
    // shard-sized grids (one rank of h4 w2 holds 14 400 ... 32 400 pixels of the internal grid): for 384 <- 384 the ring kernel beats the
    // weight-stationary ones by 10 - 25 % there (plain 14.5 / 16.3 us against 18.8 / 21.8, + skip operand 16.3 / 18.4 against 18.2 / 24.3:
    // profiles/r05_ab_conv_shard_kernel_choice.txt); from 115 200 pixels on the stationary kernels win everywhere
    const bool small_ring = K == 384 && M == 384 && (long long)B * N <= 32768 && !(act && Ypre) && astat2 != 1;
    if (astat2 && (astat2 != 3 || !(act && Ypre && !(R || G))) && !small_ring && !force_tile && !no_astat && K == 384 && M >= 256 &&
        (long long)M * N * 2 < (1ll << 31) && N >= 64 && !(R && G)) {
        // two wave groups, one multiplying while the other runs its epilogue (conv_nn_astat2_kernel): 384-channel slabs
        const bool epi_loads = R || G;
        const int slabs = (M + 383) / 384;
        const long long tn = (N + 63) / 64;
        const long long streams = tn < 256 / slabs ? tn : 256 / slabs;
        const dim3 grid((unsigned)(streams * slabs), (unsigned)B), blk(512);
        const bool pre = act && Ypre;
        hipStream_t s = (hipStream_t)stream;
        if (epi_loads && pre) hipLaunchKernelGGL((conv_nn_astat2_kernel<true, true>), grid, blk, 0, s, p, slabs, tn);
        else if (epi_loads) hipLaunchKernelGGL((conv_nn_astat2_kernel<false, true>), grid, blk, 0, s, p, slabs, tn);
        else if (pre) hipLaunchKernelGGL((conv_nn_astat2_kernel<true, false>), grid, blk, 0, s, p, slabs, tn);
        else hipLaunchKernelGGL((conv_nn_astat2_kernel<false, false>), grid, blk, 0, s, p, slabs, tn);
        return mk_check_launch("mk_conv1x1_nn");
    }
    if (!force_tile && !no_astat && !small_ring && K == 384 && M >= 256 && (long long)M * N * 2 < (1ll << 31) && N >= 64 && !(R && G)) {
        // weights stationary in registers: 384-channel slabs (4 waves x 3 row tiles; 256-row slabs measured slower), 64-pixel tiles,
        // persistent grid of 256-thread workgroups
        const bool epi_loads = R || G;
        constexpr int tmv = 3;
        const int kch = epi_loads ? 64 : 128;          // (the 128-channel chunk form of the epilogue-operand variant runs out of registers)
        const int bm = 128 * tmv;
        const int slabs = (M + bm - 1) / bm;
        const long long tn = (N + 63) / 64;
        const long long streams = tn < 256 / slabs ? tn : 256 / slabs;
        const dim3 grid((unsigned)(streams * slabs), (unsigned)B), blk(256);
        const bool pre = act && Ypre;
        hipStream_t s = (hipStream_t)stream;
#define MK_ASTAT(TM_, KCH_)                                                                                              \
    do {                                                                                                                  \
        if (epi_loads && pre) hipLaunchKernelGGL((conv_nn_astat_kernel<TM_, KCH_, true, true>), grid, blk, 0, s, p, slabs, tn);   \
        else if (epi_loads) hipLaunchKernelGGL((conv_nn_astat_kernel<TM_, KCH_, false, true>), grid, blk, 0, s, p, slabs, tn);    \
        else if (pre) hipLaunchKernelGGL((conv_nn_astat_kernel<TM_, KCH_, true, false>), grid, blk, 0, s, p, slabs, tn);          \
        else hipLaunchKernelGGL((conv_nn_astat_kernel<TM_, KCH_, false, false>), grid, blk, 0, s, p, slabs, tn);                  \
    } while (0)
        if (kch == 128) MK_ASTAT(3, 128);
        else MK_ASTAT(3, 64);
#undef MK_ASTAT
        return mk_check_launch("mk_conv1x1_nn");
    }
    static const bool ring_any_k = [] { const char* e = getenv("MAKANI_AMD_CONV_RINGK"); return !(e && e[0] == '0'); }();
    if (!force_tile && ((K % 64) == 0 || (ring_any_k && K >= 64)) && M >= 192 && N * 2 * 64 < (1ll << 31) &&
        (long long)M * lda * 2 < (1ll << 31) && N >= 256) {
        // ring kernel: persistent grid, one 512-thread workgroup per CU
        const bool big = (M % 256 == 0) || M > 576;
        const int bm = big ? 256 : 192;
        const int tm = (M + bm - 1) / bm;
        const long long tn = (N + 255) / 256;
        const long long nt = (long long)tm * tn * B;
        const unsigned grid = (unsigned)(nt < 256 ? nt : 256);
        if (big) hipLaunchKernelGGL((conv_nn_ring_kernel<4>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p, tm, tn, nt);
        else hipLaunchKernelGGL((conv_nn_ring_kernel<3>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p, tm, tn, nt);
        return mk_check_launch("mk_conv1x1_nn");
    }
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

// ---- host-side plan of the weight gradient -------------------------------------------------
namespace {
struct WgPlan {
    bool ring;          // ring kernel (big tiles) or the 128 x 128 tile kernel
    bool swap;          // ring: P = X, Q = G
    int tp;             // ring: slab height (256 / 192)
    int slabs;
    int qslabs;         // ring: 384-row slabs of Q
    long long S;        // pixel splits (all batch entries)
    long long chunk;    // pixels per split
};

int wgrad_kernel_choice() {      // MAKANI_AMD_WGRAD=tile forces the 128 x 128 tile kernel (A/B measurements)
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("MAKANI_AMD_WGRAD");
        v = (e && e[0] == 't') ? 1 : 0;
    }
    return v;
}

WgPlan wgrad_plan(int M, int K, int B, long long N) {
    WgPlan pl{};
    const int big = M > K ? M : K, small = M > K ? K : M;
    // ring kernel: the smaller channel count fits one 384-row tile, the other one is cut into slabs; 32-bit byte
    // offsets inside one batch entry; at least a few pixel tiles per split
    static const bool ring2d = [] { const char* e = getenv("MAKANI_AMD_WGRAD_2D"); return !(e && e[0] == '0'); }();
    const bool ok = wgrad_kernel_choice() == 0 && big >= 96 && N * 2 * 384 < (1ll << 31) && N >= 2048;
    if (ok && (small <= 384 || ring2d)) {
        pl.ring = true;
        int q, pr;                                   // rows of Q (384-row slabs; one slab = held in full) and of P (slabs of tp)
        if (big <= 384) { q = big; pr = small; }
        else if (small <= 384) { q = small; pr = big; }
        else { q = big; pr = small; }                // both large (FourCastNet3): the longer operand in 384-row slabs
        // Q = X (k) unless that puts the larger operand in P's place the wrong way round
        const bool q_is_x = (K == q) && !(M == q && M > K);
        pl.swap = !q_is_x;
        pl.tp = (pr > 192 && pr % 256 != 192 && (pr % 192 != 0 || pr % 256 == 0)) ? 256 : 192;
        if (pr <= 192) pl.tp = 192;
        pl.slabs = (pr + pl.tp - 1) / pl.tp;
        pl.qslabs = (q + 383) / 384;
        long long per_b = 256 / ((long long)pl.slabs * pl.qslabs * B);
        if (per_b < 1) per_b = 1;
        // at least 512 pixels (8 tiles of 64) per split.  1024 left most of the chip idle on shard-sized grids (14 400 pixels per rank
        // at h4 w2: 45 workgroups); 512: -20 ... -25 % per launch there, +-1 % at 115 200 pixels and above, 256 / 128 no better
        // (profiles/r05_ab_wgrad_shard_minpx.txt)
        const long long maxs = (N + 511) / 512;
        if (per_b > maxs) per_b = maxs;
        long long chunk = ((N + per_b - 1) / per_b + 63) / 64 * 64;
        per_b = (N + chunk - 1) / chunk;             // no empty splits
        pl.chunk = chunk;
        pl.S = per_b * B;
        return pl;
    }
    const int tiles = ((M + 127) / 128) * ((K + 127) / 128);
    long long per_b = 1024 / ((long long)tiles * B);
    if (per_b < 1) per_b = 1;
    const long long maxs = (N + 2047) / 2048;        // at least 2048 pixels per split
    if (per_b > maxs) per_b = maxs;
    long long chunk = (N + per_b - 1) / per_b;
    chunk = (chunk + 127) / 128 * 128;
    pl.chunk = chunk;
    pl.S = per_b * B;
    return pl;
}
}  // namespace

extern "C" long long mk_conv1x1_wgrad_workspace(int M, int K, int B, long long N) {
    // number of fp32 elements the caller must provide as `part` (the (S, M, K) partial products, then (S, M) partial row sums
    // of G for mk_conv1x1_wgrad_bias)
    const WgPlan pl = wgrad_plan(M, K, B, N);
    return pl.S * (long long)M * K + pl.S * (long long)M;
}

// 1 if mk_conv1x1_wgrad_bias computes the bias gradient inside the weight-gradient kernel for this shape (the ring kernel),
// 0 if the caller has to take the plane sums itself (mk_plane_sums)
extern "C" int mk_conv1x1_wgrad_fuses_bias(int M, int K, int B, long long N) {
    return wgrad_plan(M, K, B, N).ring ? 1 : 0;
}

static int conv1x1_wgrad_impl(const void* G, const void* X, float* dW, float* dbias, float* part, int M, int K, int B, long long N,
                              int accumulate, void* stream);

extern "C" int mk_conv1x1_wgrad(const void* G, const void* X, float* dW, float* part, int M, int K, int B, long long N,
                                int accumulate, void* stream) {
    return conv1x1_wgrad_impl(G, X, dW, nullptr, part, M, K, B, N, accumulate, stream);
}

// the weight gradient AND the bias gradient db[m] = sum_{b,n} G[b][m][n] (fp32, M entries, overwritten) from one pass over G;
// only where mk_conv1x1_wgrad_fuses_bias says so
extern "C" int mk_conv1x1_wgrad_bias(const void* G, const void* X, float* dW, float* dbias, float* part, int M, int K, int B,
                                     long long N, int accumulate, void* stream) {
    MK_REQUIRE(dbias, "conv1x1_wgrad_bias: null pointer");
    MK_REQUIRE(wgrad_plan(M, K, B, N).ring, "conv1x1_wgrad_bias: this shape runs the tile kernel, which does not form the bias gradient");
    return conv1x1_wgrad_impl(G, X, dW, dbias, part, M, K, B, N, accumulate, stream);
}

static int conv1x1_wgrad_impl(const void* G, const void* X, float* dW, float* dbias, float* part, int M, int K, int B, long long N,
                              int accumulate, void* stream) {
    MK_REQUIRE(G && X && dW && part, "conv1x1_wgrad: null pointer");
    MK_REQUIRE(M > 0 && K > 0 && B > 0 && N > 0 && (N % 8) == 0, "conv1x1_wgrad: bad shape");
    const WgPlan pl = wgrad_plan(M, K, B, N);
    hipStream_t s = (hipStream_t)stream;
    if (pl.ring) {
        ConvWgR p{};
        p.P = (const u16*)(pl.swap ? X : G);
        p.Q = (const u16*)(pl.swap ? G : X);
        p.RP = pl.swap ? K : M;
        p.RQ = pl.swap ? M : K;
        p.part = part, p.B = B, p.S = (int)pl.S, p.slabs = pl.slabs, p.qslabs = pl.qslabs, p.ldo = K, p.N = N, p.chunk = pl.chunk;
        p.bias_part = dbias ? part + pl.S * (long long)M * K : nullptr;
        const dim3 grid((unsigned)(pl.slabs * pl.qslabs * pl.S)), blk(512);
        if (pl.tp == 256) {
            if (pl.swap) hipLaunchKernelGGL((conv_wgrad_ring_kernel<2, 4, 4, 3, true>), grid, blk, 0, s, p);
            else hipLaunchKernelGGL((conv_wgrad_ring_kernel<2, 4, 4, 3, false>), grid, blk, 0, s, p);
        } else {
            if (pl.swap) hipLaunchKernelGGL((conv_wgrad_ring_kernel<2, 4, 3, 3, true>), grid, blk, 0, s, p);
            else hipLaunchKernelGGL((conv_wgrad_ring_kernel<2, 4, 3, 3, false>), grid, blk, 0, s, p);
        }
    } else {
        const int tm = (M + 127) / 128, tk = (K + 127) / 128;
        ConvWg p{(const u16*)G, (const u16*)X, part, M, K, B, (int)pl.S, N, pl.chunk};
        // BK = 128 / one LDS stage and BK = 64 / two stages + two register sets measure the same (+-2 %) on the 384/768
        // channel shapes; the former is 8 % faster on the 73-channel ones and needs 36 fewer VGPRs
        hipLaunchKernelGGL(conv_wgrad_kernel<128>, dim3((unsigned)(tm * tk * pl.S)), dim3(NT), 0, s, p, tm, tk);
    }
    const long long n = (long long)M * K;
    if (n % 4 == 0 && (((uintptr_t)part | (uintptr_t)dW) & 15) == 0)
        hipLaunchKernelGGL(reduce_splits4, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, (const float4*)part, (float4*)dW,
                           n / 4, (int)pl.S, accumulate);
    else
        hipLaunchKernelGGL(reduce_splits, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, dW, n, (int)pl.S, accumulate);
    if (dbias)      // (S, M) partial row sums
        hipLaunchKernelGGL(reduce_splits, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, part + pl.S * n, dbias, (long long)M,
                           (int)pl.S, 0);
    return mk_check_launch("mk_conv1x1_wgrad");
}
