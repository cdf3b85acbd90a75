// Layout changes between torch-facing tensors and the internal spectral layouts (gfx950).
//   complex64 dhconv parameter (Cin, Cout, L)  <->  W-layout W[l][ri][i][o]   (o padded to Cop)
//   complex64 coefficients (B, C, L, M)        <->  S-layout S[l][m][ri][b][c] (c padded to Cp)
// LDS-tiled 32x32 transposes: both the read and the write side are coalesced.
#include "common.h"

namespace {

constexpr int TS = 32;

// grid: (ceil(L/32), ceil(Cout/32), Cin), block (32, 8)
__global__ void weight_to_w(const float2* __restrict__ w, float* __restrict__ W, int cin, int cout, int cip, int cop, int L) {
    __shared__ float tre[TS][TS + 1], tim[TS][TS + 1];
    const int i = blockIdx.z, o0 = blockIdx.y * TS, l0 = blockIdx.x * TS;
    for (int oo = threadIdx.y; oo < TS; oo += 8) {
        const int o = o0 + oo, l = l0 + threadIdx.x;
        float2 v = make_float2(0.f, 0.f);
        if (o < cout && l < L) v = w[((long long)i * cout + o) * L + l];
        tre[oo][threadIdx.x] = v.x;
        tim[oo][threadIdx.x] = v.y;
    }
    __syncthreads();
    for (int ll = threadIdx.y; ll < TS; ll += 8) {
        const int l = l0 + ll, o = o0 + threadIdx.x;
        if (l < L && o < cout) {
            W[(((long long)l * 2 + 0) * cip + i) * cop + o] = tre[threadIdx.x][ll];
            W[(((long long)l * 2 + 1) * cip + i) * cop + o] = tim[threadIdx.x][ll];
        }
    }
}

__global__ void w_to_weight_grad(const float* __restrict__ gW, float2* __restrict__ gw, int cin, int cout, int cip, int cop, int L) {
    __shared__ float tre[TS][TS + 1], tim[TS][TS + 1];
    const int i = blockIdx.z, o0 = blockIdx.y * TS, l0 = blockIdx.x * TS;
    for (int ll = threadIdx.y; ll < TS; ll += 8) {
        const int l = l0 + ll, o = o0 + threadIdx.x;
        float re = 0.f, im = 0.f;
        if (l < L && o < cout) {
            re = gW[(((long long)l * 2 + 0) * cip + i) * cop + o];
            im = gW[(((long long)l * 2 + 1) * cip + i) * cop + o];
        }
        tre[ll][threadIdx.x] = re;
        tim[ll][threadIdx.x] = im;
    }
    __syncthreads();
    for (int oo = threadIdx.y; oo < TS; oo += 8) {
        const int o = o0 + oo, l = l0 + threadIdx.x;
        if (o < cout && l < L) gw[((long long)i * cout + o) * L + l] = make_float2(tre[threadIdx.x][oo], tim[threadIdx.x][oo]);
    }
}

// S[l][m][ri][b][cp] -> out[b][c][l][m] (complex64); entries with l + l_off < m + m_off are exact zeros
// (l_off / m_off = first degree / order of this shard; 0 when not sharded)
// grid: (ceil(M/32), ceil(C/32), B*L), block (32, 8)
__global__ void s_to_complex(const float* __restrict__ S, float2* __restrict__ out, int B, int C, int Cp, int L, int M,
                             int l_off, int m_off) {
    __shared__ float tre[TS][TS + 1], tim[TS][TS + 1];
    const int b = blockIdx.z / L, l = blockIdx.z % L;
    const int c0 = blockIdx.y * TS, m0 = blockIdx.x * TS;
    const long long R = (long long)B * Cp;
    for (int mm = threadIdx.y; mm < TS; mm += 8) {
        const int m = m0 + mm, c = c0 + threadIdx.x;
        float re = 0.f, im = 0.f;
        if (m < M && c < C && m + m_off <= l + l_off) {
            const long long base = (((long long)l * M + m) * 2) * R + (long long)b * Cp + c;
            re = S[base];
            im = S[base + R];
        }
        tre[mm][threadIdx.x] = re;
        tim[mm][threadIdx.x] = im;
    }
    __syncthreads();
    for (int cc = threadIdx.y; cc < TS; cc += 8) {
        const int c = c0 + cc, m = m0 + threadIdx.x;
        if (c < C && m < M) out[(((long long)b * C + c) * L + l) * M + m] = make_float2(tre[threadIdx.x][cc], tim[threadIdx.x][cc]);
    }
}

// in[b][c][l][m] (complex64) -> S[l][m][ri][b][cp]; pad channels c in [C, Cp) are written as zeros
__global__ void complex_to_s(const float2* __restrict__ in, float* __restrict__ S, int B, int C, int Cp, int L, int M) {
    __shared__ float tre[TS][TS + 1], tim[TS][TS + 1];
    const int b = blockIdx.z / L, l = blockIdx.z % L;
    const int c0 = blockIdx.y * TS, m0 = blockIdx.x * TS;
    const long long R = (long long)B * Cp;
    for (int cc = threadIdx.y; cc < TS; cc += 8) {
        const int c = c0 + cc, m = m0 + threadIdx.x;
        float2 v = make_float2(0.f, 0.f);
        if (c < C && m < M) v = in[(((long long)b * C + c) * L + l) * M + m];
        tre[cc][threadIdx.x] = v.x;
        tim[cc][threadIdx.x] = v.y;
    }
    __syncthreads();
    for (int mm = threadIdx.y; mm < TS; mm += 8) {
        const int m = m0 + mm, c = c0 + threadIdx.x;
        if (m < M && c < Cp) {
            const long long base = (((long long)l * M + m) * 2) * R + (long long)b * Cp + c;
            S[base] = tre[threadIdx.x][mm];
            S[base + R] = tim[threadIdx.x][mm];
        }
    }
}

}  // namespace

extern "C" int mk_weight_to_wlayout(const float* w_c64, float* W, int cin, int cout, int cip, int cop, int L, void* stream) {
    MK_REQUIRE(w_c64 && W && cin > 0 && cout > 0 && cip >= cin && cop >= cout && L > 0, "weight_to_wlayout: bad args");
    dim3 grid((L + TS - 1) / TS, (cout + TS - 1) / TS, cin), block(TS, 8);
    hipLaunchKernelGGL(weight_to_w, grid, block, 0, (hipStream_t)stream, (const float2*)w_c64, W, cin, cout, cip, cop, L);
    return mk_check_launch("mk_weight_to_wlayout");
}

extern "C" int mk_wlayout_to_weight_grad(const float* gW, float* gw_c64, int cin, int cout, int cip, int cop, int L, void* stream) {
    MK_REQUIRE(gW && gw_c64 && cin > 0 && cout > 0 && cip >= cin && cop >= cout && L > 0, "wlayout_to_weight_grad: bad args");
    dim3 grid((L + TS - 1) / TS, (cout + TS - 1) / TS, cin), block(TS, 8);
    hipLaunchKernelGGL(w_to_weight_grad, grid, block, 0, (hipStream_t)stream, gW, (float2*)gw_c64, cin, cout, cip, cop, L);
    return mk_check_launch("mk_wlayout_to_weight_grad");
}

extern "C" int mk_slayout_to_complex(const float* S, float* out_c64, int B, int C, int Cp, int L, int M, int l_off,
                                     int m_off, void* stream) {
    MK_REQUIRE(S && out_c64 && B > 0 && C > 0 && Cp >= C && L > 0 && M > 0, "slayout_to_complex: bad args");
    MK_REQUIRE((long long)B * L < 65536, "slayout_to_complex: B*L too large for grid.z");
    dim3 grid((M + TS - 1) / TS, (C + TS - 1) / TS, B * L), block(TS, 8);
    hipLaunchKernelGGL(s_to_complex, grid, block, 0, (hipStream_t)stream, S, (float2*)out_c64, B, C, Cp, L, M, l_off, m_off);
    return mk_check_launch("mk_slayout_to_complex");
}

extern "C" int mk_complex_to_slayout(const float* in_c64, float* S, int B, int C, int Cp, int L, int M, void* stream) {
    MK_REQUIRE(S && in_c64 && B > 0 && C > 0 && Cp >= C && L > 0 && M > 0, "complex_to_slayout: bad args");
    MK_REQUIRE((long long)B * L < 65536, "complex_to_slayout: B*L too large for grid.z");
    dim3 grid((M + TS - 1) / TS, (Cp + TS - 1) / TS, B * L), block(TS, 8);
    hipLaunchKernelGGL(complex_to_s, grid, block, 0, (hipStream_t)stream, (const float2*)in_c64, S, B, C, Cp, L, M);
    return mk_check_launch("mk_complex_to_slayout");
}
