// MK_HIPCC_FLAGS: -fno-slp-vectorize
// (gfx950: v_pk_mul_f32 / v_pk_add_f32 whose src1 is a VGPR pair read through op_sel return wrong results while certain
//  matrix-core kernels run on the same compute unit — tools/pk_hazard_probe.py, docs/LAB_NOTEBOOK.md round 6.  The SLP vectoriser
//  emits exactly those forms from plain scalar code, so this file is compiled without it; tools/pk_opsel_scan.py checks the ISA.)
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


// ---- spectral L^p sums on the S-layout ------------------------------------------------------------
// partial[b][row] = sum over the (l, m) pairs of block b with l >= m of  w(m) * |c|^p (* wgt[l][m][row]),
// c = S[l][m][0][row] + i S[l][m][1][row];  w(m) = w0 for the global order 0, w1 otherwise.
constexpr int SP_PAIRS = 32;
constexpr int SP_NT = 128;

__device__ __forceinline__ float sp_pow(float a2, float p) {      // |c|^p from |c|^2
    return p == 2.f ? a2 : (a2 > 0.f ? powf(a2, 0.5f * p) : 0.f);
}

__global__ __launch_bounds__(SP_NT) void spec_lp_partial(const float* __restrict__ S, const float* __restrict__ wgt,
                                                         float* __restrict__ partial, int L, int M, long long R,
                                                         int tri_off, int m_off, float p, float w0, float w1) {
    const long long r = (long long)blockIdx.y * SP_NT + threadIdx.x;
    if (r >= R) return;
    const long long pair0 = (long long)blockIdx.x * SP_PAIRS;
    float acc = 0.f;
    for (int i = 0; i < SP_PAIRS; ++i) {
        const long long pr = pair0 + i;
        if (pr >= (long long)L * M) break;
        const int l = (int)(pr / M), m = (int)(pr % M);
        if (l + tri_off < m) continue;                         // structurally zero (never written by the SHT)
        const float re = S[(pr * 2) * R + r], im = S[(pr * 2 + 1) * R + r];
        float t = sp_pow(re * re + im * im, p) * ((m + m_off) == 0 ? w0 : w1);
        if (wgt) t *= wgt[pr * R + r];
        acc += t;
    }
    partial[(long long)blockIdx.x * R + r] = acc;
}

// dS = g[row] * w(m) * p * |c|^(p-2) * c (* wgt); zeros where l < m
__global__ __launch_bounds__(SP_NT) void spec_lp_bwd(const float* __restrict__ S, const float* __restrict__ wgt,
                                                     const float* __restrict__ g, float* __restrict__ dS, int L, int M,
                                                     long long R, int tri_off, int m_off, float p, float w0, float w1) {
    const long long r = (long long)blockIdx.y * SP_NT + threadIdx.x;
    if (r >= R) return;
    const long long pair0 = (long long)blockIdx.x * SP_PAIRS;
    const float gr = g[r];
    for (int i = 0; i < SP_PAIRS; ++i) {
        const long long pr = pair0 + i;
        if (pr >= (long long)L * M) break;
        const int l = (int)(pr / M), m = (int)(pr % M);
        float dre = 0.f, dim = 0.f;
        if (l + tri_off >= m) {
            const float re = S[(pr * 2) * R + r], im = S[(pr * 2 + 1) * R + r];
            const float a2 = re * re + im * im;
            float f = p == 2.f ? 2.f : (a2 > 0.f ? p * powf(a2, 0.5f * p - 1.f) : 0.f);
            f *= gr * ((m + m_off) == 0 ? w0 : w1);
            if (wgt) f *= wgt[pr * R + r];
            dre = f * re, dim = f * im;
        }
        dS[(pr * 2) * R + r] = dre;
        dS[(pr * 2 + 1) * R + r] = dim;
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

extern "C" long long mk_spec_lp_blocks(int L, int M) { return ((long long)L * M + SP_PAIRS - 1) / SP_PAIRS; }

extern "C" int mk_spec_lp_fwd(const float* S, const float* wgt, float* partial, int L, int M, long long R, int tri_off,
                              int m_off, float p, float w0, float w1, void* stream) {
    MK_REQUIRE(S && partial && L > 0 && M > 0 && R > 0 && p > 0.f, "spec_lp_fwd: bad args");
    const long long nb = mk_spec_lp_blocks(L, M);
    MK_REQUIRE(nb < (1ll << 31), "spec_lp_fwd: grid too large");
    dim3 grid((unsigned)nb, (unsigned)((R + SP_NT - 1) / SP_NT));
    hipLaunchKernelGGL(spec_lp_partial, grid, dim3(SP_NT), 0, (hipStream_t)stream, S, wgt, partial, L, M, R, tri_off, m_off, p, w0, w1);
    return mk_check_launch("mk_spec_lp_fwd");
}

extern "C" int mk_spec_lp_bwd(const float* S, const float* wgt, const float* g, float* dS, int L, int M, long long R,
                              int tri_off, int m_off, float p, float w0, float w1, void* stream) {
    MK_REQUIRE(S && g && dS && L > 0 && M > 0 && R > 0 && p > 0.f, "spec_lp_bwd: bad args");
    const long long nb = mk_spec_lp_blocks(L, M);
    MK_REQUIRE(nb < (1ll << 31), "spec_lp_bwd: grid too large");
    dim3 grid((unsigned)nb, (unsigned)((R + SP_NT - 1) / SP_NT));
    hipLaunchKernelGGL(spec_lp_bwd, grid, dim3(SP_NT), 0, (hipStream_t)stream, S, wgt, g, dS, L, M, R, tri_off, m_off, p, w0, w1);
    return mk_check_launch("mk_spec_lp_bwd");
}
