// MK_HIPCC_FLAGS: -fno-slp-vectorize
// (gfx950: packed-fp32 instructions whose VGPR src1 is read through op_sel return wrong results while certain matrix-core
//  kernels run on the same compute unit — tools/pk_hazard_probe.py, docs/LAB_NOTEBOOK.md round 6.  The SLP vectoriser emits
//  those forms from plain scalar code, so this file is compiled without it; tools/pk_opsel_scan.py checks the ISA.)
// Grouped channel mix with a handful of channels per group (gfx950): z[b][g][r][n] = sum_c W[g][r][c] * x[b][g][c][n].
//
// The channel mixes of FourCastNet3's encoders / decoders [th.DiscreteContinuousConvS2 with groups > 1, torch-harmonics
// un-vendored; constructed at makani/models/networks/fourcastnet3.py:189-205,356-381] are per-group products with 8-9 input and
// output planes (one variable of one pressure level x the nine basis functions): as batched GEMMs they are M = 9 problems that
// the library runs at a quarter of the memory rate.  They are streaming operations — every activation is read once, every
// output written once, 2 * RG * CG flops per pixel — so here a thread owns VEC consecutive pixels of one (batch, group), issues
// the CG vector loads together, forms the RG outputs in registers (the RG x CG weights are wave-uniform: scalar loads) and
// stores RG vectors.  The data gradient is the same kernel with the transposed weights; the weight gradient
// dW[g][r][c] = sum_{b, n} dz[b][g][r][n] x[b][g][c][n] keeps RG * CG partial sums per thread and reduces them per block
// (deterministic two-stage sum, the second stage on the host side of the C ABI's caller).
#include "common.h"

namespace {

constexpr int GNT = 256;
typedef unsigned int gu32x4 __attribute__((ext_vector_type(4)));

template <typename T>
struct GVec;
template <>
struct GVec<float> {
    static constexpr int N = 4;
    typedef f32x4 Raw;
    __device__ static __forceinline__ void unpack(const Raw& r, float* v) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = r[i];
    }
    __device__ static __forceinline__ Raw pack(const float* v) { return Raw{v[0], v[1], v[2], v[3]}; }
};
template <>
struct GVec<u16> {
    static constexpr int N = 8;
    typedef gu32x4 Raw;
    __device__ static __forceinline__ void unpack(const Raw& r, float* v) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(r[i] << 16);
            v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u);
        }
    }
    __device__ static __forceinline__ Raw pack(const float* v) {
        Raw r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
        return r;
    }
};

// z = W x.  grid (pixel chunks, B * G); x (B*G, CG, N), z (B*G, RG, N), W (G, RG, CG) fp32.  N % VEC == 0.
template <typename T, int CG, int RG>
__global__ __launch_bounds__(GNT) void group_mix_kernel(const T* __restrict__ x, const float* __restrict__ W, T* __restrict__ z,
                                                        int G, long long N) {
    constexpr int VEC = GVec<T>::N;
    typedef typename GVec<T>::Raw Raw;
    const int bg = blockIdx.y, g = bg % G;
    const float* w = W + (long long)g * RG * CG;
    const T* xp = x + (long long)bg * CG * N;
    T* zp = z + (long long)bg * RG * N;
    const long long nvec = N / VEC;
    for (long long e = (long long)blockIdx.x * GNT + threadIdx.x; e < nvec; e += (long long)gridDim.x * GNT) {
        Raw raw[CG];
#pragma unroll
        for (int c = 0; c < CG; ++c) raw[c] = *reinterpret_cast<const Raw*>(xp + (long long)c * N + e * VEC);
        float acc[RG][VEC];
#pragma unroll
        for (int r = 0; r < RG; ++r)
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[r][i] = 0.f;
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            float v[VEC];
            GVec<T>::unpack(raw[c], v);
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                const float wv = w[r * CG + c];
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[r][i] = fmaf(wv, v[i], acc[r][i]);
            }
        }
#pragma unroll
        for (int r = 0; r < RG; ++r) *reinterpret_cast<Raw*>(zp + (long long)r * N + e * VEC) = GVec<T>::pack(acc[r]);
    }
}

// partial[blockIdx.y * gridDim.x + blockIdx.x][r][c] = sum over this block's pixels of dz[r][n] * x[c][n]
template <typename T, int CG, int RG>
__global__ __launch_bounds__(GNT) void group_mix_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dz, float* __restrict__ partial,
                                                              long long N) {
    constexpr int VEC = GVec<T>::N;
    typedef typename GVec<T>::Raw Raw;
    __shared__ float red[GNT / 64][RG * CG];
    const int bg = blockIdx.y;
    const T* xp = x + (long long)bg * CG * N;
    const T* dp = dz + (long long)bg * RG * N;
    float acc[RG][CG];
#pragma unroll
    for (int r = 0; r < RG; ++r)
#pragma unroll
        for (int c = 0; c < CG; ++c) acc[r][c] = 0.f;
    const long long nvec = N / VEC;
    for (long long e = (long long)blockIdx.x * GNT + threadIdx.x; e < nvec; e += (long long)gridDim.x * GNT) {
        Raw rx[CG], rd[RG];
#pragma unroll
        for (int c = 0; c < CG; ++c) rx[c] = *reinterpret_cast<const Raw*>(xp + (long long)c * N + e * VEC);
#pragma unroll
        for (int r = 0; r < RG; ++r) rd[r] = *reinterpret_cast<const Raw*>(dp + (long long)r * N + e * VEC);
        float xv[CG][VEC];
#pragma unroll
        for (int c = 0; c < CG; ++c) GVec<T>::unpack(rx[c], xv[c]);
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            float dv[VEC];
            GVec<T>::unpack(rd[r], dv);
#pragma unroll
            for (int c = 0; c < CG; ++c)
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[r][c] = fmaf(dv[i], xv[c][i], acc[r][c]);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < RG; ++r)
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            float v = acc[r][c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            if (lane == 0) red[wave][r * CG + c] = v;
        }
    __syncthreads();
    if (threadIdx.x < RG * CG) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < GNT / 64; ++q) s += red[q][threadIdx.x];
        partial[((long long)blockIdx.y * gridDim.x + blockIdx.x) * (RG * CG) + threadIdx.x] = s;
    }
}

inline int mix_blocks(long long nvec, int bg) {          // pixel chunks per (batch, group): ~8 resident blocks per CU over the launch
    long long want = (2048 + bg - 1) / bg;
    const long long maxb = (nvec + GNT - 1) / GNT;
    if (want > maxb) want = maxb;
    return (int)(want < 1 ? 1 : want);
}

}  // namespace

// 1 when the (CG, RG) pair is instantiated
extern "C" int mk_group_mix_supported(int CG, int RG) {
    return (CG == 9 && RG == 9) || (CG == 9 && RG == 8) || (CG == 8 && RG == 9) || (CG == 8 && RG == 8);
}

extern "C" int mk_group_mix_blocks(long long N, int dtype, int BG) { return mix_blocks(N / (dtype == MK_BF16 ? 8 : 4), BG); }

#define MK_MIX_DISPATCH(KERNEL, T, ...)                                                                \
    do {                                                                                                \
        if (CG == 9 && RG == 9) hipLaunchKernelGGL((KERNEL<T, 9, 9>), grid, dim3(GNT), 0, s, __VA_ARGS__);      \
        else if (CG == 9 && RG == 8) hipLaunchKernelGGL((KERNEL<T, 9, 8>), grid, dim3(GNT), 0, s, __VA_ARGS__); \
        else if (CG == 8 && RG == 9) hipLaunchKernelGGL((KERNEL<T, 8, 9>), grid, dim3(GNT), 0, s, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<T, 8, 8>), grid, dim3(GNT), 0, s, __VA_ARGS__);                         \
    } while (0)

// z (B*G, RG, N) = W (G, RG, CG) x (B*G, CG, N); dtype f32 | bf16 (fp32 accumulation), N a multiple of 4 (f32) / 8 (bf16)
extern "C" int mk_group_mix(const void* x, const float* W, void* z, int dtype, int B, int G, int CG, int RG, long long N, void* stream) {
    MK_REQUIRE(x && W && z, "group_mix: null pointer");
    MK_REQUIRE(mk_group_mix_supported(CG, RG), "group_mix: (%d -> %d) channels per group is not instantiated", CG, RG);
    const int vec = dtype == MK_BF16 ? 8 : 4;
    MK_REQUIRE(B > 0 && G > 0 && N > 0 && N % vec == 0 && (long long)B * G <= 65535, "group_mix: bad extents");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(mix_blocks(N / vec, B * G), B * G);
    if (dtype == MK_F32) {
        const float* xx = (const float*)x;
        float* zz = (float*)z;
        MK_MIX_DISPATCH(group_mix_kernel, float, xx, W, zz, G, N);
    } else {
        const u16* xx = (const u16*)x;
        u16* zz = (u16*)z;
        MK_MIX_DISPATCH(group_mix_kernel, u16, xx, W, zz, G, N);
    }
    return mk_check_launch("mk_group_mix");
}

// partial (B*G * mk_group_mix_blocks, RG * CG) fp32: the caller sums over the blocks of a group and over the batch
extern "C" int mk_group_mix_wgrad(const void* x, const void* dz, float* partial, int dtype, int B, int G, int CG, int RG, long long N,
                                  void* stream) {
    MK_REQUIRE(x && dz && partial, "group_mix_wgrad: null pointer");
    MK_REQUIRE(mk_group_mix_supported(CG, RG), "group_mix_wgrad: (%d -> %d) channels per group is not instantiated", CG, RG);
    const int vec = dtype == MK_BF16 ? 8 : 4;
    MK_REQUIRE(B > 0 && G > 0 && N > 0 && N % vec == 0 && (long long)B * G <= 65535, "group_mix_wgrad: bad extents");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(mix_blocks(N / vec, B * G), B * G);
    if (dtype == MK_F32) {
        const float *xx = (const float*)x, *dd = (const float*)dz;
        MK_MIX_DISPATCH(group_mix_wgrad_kernel, float, xx, dd, partial, N);
    } else {
        const u16 *xx = (const u16*)x, *dd = (const u16*)dz;
        MK_MIX_DISPATCH(group_mix_wgrad_kernel, u16, xx, dd, partial, N);
    }
    return mk_check_launch("mk_group_mix_wgrad");
}
