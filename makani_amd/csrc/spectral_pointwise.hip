// The spectral contractions of SpectralConv that are NOT batched matrix products over l:
//   separable   _contract_sep_lmwise / _contract_sep_lwise   (makani/models/common/contractions.py:26-31)
//   diagonal    _contract_lmwise                             (contractions.py:17-18)
// and their autograd.  All of them touch every weight exactly once per sample and do O(1) .. O(B) flops per weight
// byte: they are HBM streams, not matrix-core work, and are written as such (coalesced position-major reads of the
// weights in their PARAMETER layout, activations staged through LDS, structural zeros m > l skipped).
//
// Activations are in the S-layout of the rest of the library: x[((l*M + m)*2 + ri)*R + b*Cp + c], R = B*Cp.
// A position (l, m) is live when m <= l + tri_off (tri_off = first l of this shard - first m of this shard).
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------
// separable: y[l][m][b][c] = x[l][m][b][c] * w[l][mw][c]   (w in S-layout too: (L, Mw, 2, Cp), Mw = M or 1)
// one thread = 4 consecutive channels of one (position, b)
template <bool CONJ>
__global__ void __launch_bounds__(256) sep_mul_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      float* __restrict__ y, int L, int M, int Mw, int B, int Cp, int tri_off) {
    const int c4n = Cp >> 2;
    const long long total = (long long)L * M * B * c4n;
    const long long R = (long long)B * Cp;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4n) * 4;
        const long long t = idx / c4n;
        const int b = (int)(t % B);
        const long long p = t / B;
        const int l = (int)(p / M), m = (int)(p % M);
        const long long xo = (p * 2) * R + (long long)b * Cp + c;
        f32x4 yr = {0.f, 0.f, 0.f, 0.f}, yi = {0.f, 0.f, 0.f, 0.f};
        if (m <= l + tri_off) {
            const f32x4 xr = *(const f32x4*)(x + xo), xi = *(const f32x4*)(x + xo + R);
            const long long wo = (((long long)l * Mw + (Mw == 1 ? 0 : m)) * 2) * Cp + c;
            const f32x4 wr = *(const f32x4*)(w + wo), wi = *(const f32x4*)(w + wo + Cp);
            if (CONJ) {
                yr = xr * wr + xi * wi;
                yi = xi * wr - xr * wi;
            } else {
                yr = xr * wr - xi * wi;
                yi = xr * wi + xi * wr;
            }
        }
        *(f32x4*)(y + xo) = yr;
        *(f32x4*)(y + xo + R) = yi;
    }
}

// gw[l][mw][c] = sum_b (sum_m when Mw == 1) conj(x) * gy ; one thread = 4 channels of one weight position
__global__ void __launch_bounds__(256) sep_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                        float* __restrict__ gw, int L, int M, int Mw, int B, int Cp, int tri_off) {
    const int c4n = Cp >> 2;
    const long long total = (long long)L * Mw * c4n;
    const long long R = (long long)B * Cp;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4n) * 4;
        const long long pw = idx / c4n;
        const int l = (int)(pw / Mw);
        int m0 = (int)(pw % Mw), m1 = m0 + 1;
        if (Mw == 1) m0 = 0, m1 = M;
        const int mlim = l + tri_off + 1;          // live orders: m < mlim
        if (m1 > mlim) m1 = mlim;
        f32x4 ar = {0.f, 0.f, 0.f, 0.f}, ai = {0.f, 0.f, 0.f, 0.f};
        for (int m = m0; m < m1; ++m)
            for (int b = 0; b < B; ++b) {
                const long long xo = (((long long)l * M + m) * 2) * R + (long long)b * Cp + c;
                const f32x4 xr = *(const f32x4*)(x + xo), xi = *(const f32x4*)(x + xo + R);
                const f32x4 gr = *(const f32x4*)(gy + xo), gi = *(const f32x4*)(gy + xo + R);
                ar += xr * gr + xi * gi;
                ai += xr * gi - xi * gr;
            }
        const long long wo = (pw * 2) * Cp + c;
        *(f32x4*)(gw + wo) = ar;
        *(f32x4*)(gw + wo + Cp) = ai;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// diagonal (dense in channels, one matrix per (l, m)):
//   y[p][b][n] = sum_k x[p][b][k] * w[k][n][p]          (forward: k = i, n = o;  wk = Cout*LM, wn = LM)
//   gx[p][b][n] = sum_k gy[p][b][k] * conj(w[n][k][p])  (dgrad:   k = o, n = i;  wk = LM, wn = Cout*LM, CONJ)
// w is the PARAMETER (Cin, Cout, L, M) complex64, read in place: for a fixed (k, n) consecutive positions are
// consecutive float2, so a wavefront that owns 64 consecutive positions reads 512 contiguous bytes per (k, n).
// Workgroup: 256 threads = 64 positions x 4 waves; wave q owns 4 of the workgroup's 16 output channels.
// grid: (ceil(LM/64), ceil(N/16), B)
constexpr int DP = 64, DN = 16, DK = 32;

template <bool CONJ>
__global__ void __launch_bounds__(256) diag_kernel(const float* __restrict__ x, const float2* __restrict__ w, float* __restrict__ y,
                                                   int L, int M, int B, int K, int x_ld, int N, int Np, int y_ld, long long wk,
                                                   long long wn, int tri_off) {
    __shared__ float xs[2][DK][DP + 1];
    const int LM = L * M;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int p0 = blockIdx.x * DP, n0 = blockIdx.y * DN + q * 4, b = blockIdx.z;
    const long long Rx = (long long)B * x_ld, Ry = (long long)B * y_ld;
    const int p = p0 + lane;
    const bool live = p < LM && (p % M) <= (p / M) + tri_off;
    float ar[4] = {0.f, 0.f, 0.f, 0.f}, ai[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += DK) {
        __syncthreads();
        // stage x[p0 .. p0+63][b][k0 .. k0+31]: lanes run along k (contiguous in the S-layout)
        for (int e = threadIdx.x; e < DP * DK; e += 256) {
            const int pp = e / DK, kk = e % DK;
            const int gp = p0 + pp, gk = k0 + kk;
            float re = 0.f, im = 0.f;
            if (gp < LM && gk < K && (gp % M) <= (gp / M) + tri_off) {
                const long long o = ((long long)gp * 2) * Rx + (long long)b * x_ld + gk;
                re = x[o];
                im = x[o + Rx];
            }
            xs[0][kk][pp] = re;
            xs[1][kk][pp] = im;
        }
        __syncthreads();
        if (live) {
            const int kn = min(DK, K - k0);
#pragma unroll 4                                   // 16 independent 8-byte weight loads in flight per lane
            for (int kk = 0; kk < kn; ++kk) {
                const float xr = xs[0][kk][lane], xi = xs[1][kk][lane];
                const float2* wp = w + (long long)(k0 + kk) * wk + p;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (n0 + j < N) {
                        const float2 wv = wp[(long long)(n0 + j) * wn];
                        if (CONJ) {
                            ar[j] += xr * wv.x + xi * wv.y;
                            ai[j] += xi * wv.x - xr * wv.y;
                        } else {
                            ar[j] += xr * wv.x - xi * wv.y;
                            ai[j] += xr * wv.y + xi * wv.x;
                        }
                    }
                }
            }
        }
    }
    if (p < LM) {
        const long long o = ((long long)p * 2) * Ry + (long long)b * y_ld + n0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (n0 + j < Np) {                       // channels [N, Np) (layout padding) are written as zeros
                y[o + j] = ar[j];
                y[o + Ry + j] = ai[j];
            }
    }
}

// gw[i][o][p] = sum_b conj(x[p][b][i]) * gy[p][b][o], written in the parameter layout (coalesced along p).
// Workgroup: 64 positions x 4 waves; tile 8 input x 16 output channels; wave q owns inputs 2q, 2q+1.
// grid: (ceil(LM/64), ceil(Cin/8), ceil(Cout/16))
constexpr int WI = 8, WO = 16;

__global__ void __launch_bounds__(256) diag_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy, float2* __restrict__ gw,
                                                         int L, int M, int B, int Cin, int x_ld, int Cout, int y_ld, int tri_off) {
    __shared__ float xs[2][WI][DP + 1], gs[2][WO][DP + 1];
    const int LM = L * M;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int p0 = blockIdx.x * DP, i0 = blockIdx.y * WI, o0 = blockIdx.z * WO;
    const long long Rx = (long long)B * x_ld, Ry = (long long)B * y_ld;
    float ar[2][WO], ai[2][WO];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int o = 0; o < WO; ++o) ar[a][o] = ai[a][o] = 0.f;
    for (int b = 0; b < B; ++b) {
        __syncthreads();
        for (int e = threadIdx.x; e < DP * WI; e += 256) {
            const int pp = e / WI, ii = e % WI;
            const int gp = p0 + pp, gi = i0 + ii;
            float re = 0.f, im = 0.f;
            if (gp < LM && gi < Cin && (gp % M) <= (gp / M) + tri_off) {
                const long long o = ((long long)gp * 2) * Rx + (long long)b * x_ld + gi;
                re = x[o];
                im = x[o + Rx];
            }
            xs[0][ii][pp] = re;
            xs[1][ii][pp] = im;
        }
        for (int e = threadIdx.x; e < DP * WO; e += 256) {
            const int pp = e / WO, oo = e % WO;
            const int gp = p0 + pp, go = o0 + oo;
            float re = 0.f, im = 0.f;
            if (gp < LM && go < Cout && (gp % M) <= (gp / M) + tri_off) {
                const long long o = ((long long)gp * 2) * Ry + (long long)b * y_ld + go;
                re = gy[o];
                im = gy[o + Ry];
            }
            gs[0][oo][pp] = re;
            gs[1][oo][pp] = im;
        }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const float xr = xs[0][q * 2 + a][lane], xi = xs[1][q * 2 + a][lane];
#pragma unroll
            for (int o = 0; o < WO; ++o) {
                const float gr = gs[0][o][lane], gi = gs[1][o][lane];
                ar[a][o] += xr * gr + xi * gi;
                ai[a][o] += xr * gi - xi * gr;
            }
        }
    }
    const int p = p0 + lane;
    if (p < LM) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int i = i0 + q * 2 + a;
            if (i >= Cin) continue;
#pragma unroll
            for (int o = 0; o < WO; ++o)
                if (o0 + o < Cout) gw[((long long)i * Cout + (o0 + o)) * LM + p] = make_float2(ar[a][o], ai[a][o]);
        }
    }
}

inline unsigned stream_grid(long long total) {
    long long nb = (total + 255) / 256;
    if (nb > 256 * 16) nb = 256 * 16;          // 16 workgroups of 4 waves per CU, grid-stride beyond that
    return (unsigned)(nb < 1 ? 1 : nb);
}

}  // namespace

extern "C" int mk_spec_sep_mul(const float* x, const float* w, float* y, int L, int M, int Mw, int B, int Cp, int tri_off,
                               int conj_w, void* stream) {
    MK_REQUIRE(x && w && y && L > 0 && M > 0 && B > 0 && Cp > 0 && Cp % 4 == 0 && (Mw == M || Mw == 1), "spec_sep_mul: bad args");
    const long long total = (long long)L * M * B * (Cp / 4);
    if (conj_w)
        hipLaunchKernelGGL(sep_mul_kernel<true>, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, x, w, y, L, M, Mw, B, Cp, tri_off);
    else
        hipLaunchKernelGGL(sep_mul_kernel<false>, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, x, w, y, L, M, Mw, B, Cp, tri_off);
    return mk_check_launch("mk_spec_sep_mul");
}

extern "C" int mk_spec_sep_wgrad(const float* x, const float* gy, float* gw, int L, int M, int Mw, int B, int Cp, int tri_off,
                                 void* stream) {
    MK_REQUIRE(x && gy && gw && L > 0 && M > 0 && B > 0 && Cp > 0 && Cp % 4 == 0 && (Mw == M || Mw == 1), "spec_sep_wgrad: bad args");
    const long long total = (long long)L * Mw * (Cp / 4);
    hipLaunchKernelGGL(sep_wgrad_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, x, gy, gw, L, M, Mw, B, Cp, tri_off);
    return mk_check_launch("mk_spec_sep_wgrad");
}

extern "C" int mk_spec_diag_apply(const float* x, const float* w_c64, float* y, int L, int M, int B, int Cin, int Cout, int x_ld,
                                  int y_ld, int y_pad, int tri_off, int dgrad, void* stream) {
    MK_REQUIRE(x && w_c64 && y && L > 0 && M > 0 && B > 0 && Cin > 0 && Cout > 0 && x_ld > 0 && y_ld > 0 && y_pad >= 0,
               "spec_diag_apply: bad args");
    MK_REQUIRE(B < 65536, "spec_diag_apply: B too large for grid.z");
    const long long LM = (long long)L * M;
    MK_REQUIRE(LM < (1ll << 31), "spec_diag_apply: L*M too large");
    const unsigned gx = (unsigned)((LM + DP - 1) / DP);
    if (!dgrad) {       // x: Cin channels (row stride x_ld) -> y: Cout channels (+ y_pad zeroed), row stride y_ld
        MK_REQUIRE(Cin <= x_ld && Cout + y_pad <= y_ld, "spec_diag_apply: channel slices exceed the row strides");
        dim3 grid(gx, (unsigned)((Cout + y_pad + DN - 1) / DN), (unsigned)B);
        hipLaunchKernelGGL(diag_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, (const float2*)w_c64, y, L, M, B, Cin, x_ld,
                           Cout, Cout + y_pad, y_ld, (long long)Cout * LM, LM, tri_off);
    } else {            // x = gy: Cout channels -> y = gx: Cin channels (+ y_pad zeroed)
        MK_REQUIRE(Cout <= x_ld && Cin + y_pad <= y_ld, "spec_diag_apply: channel slices exceed the row strides");
        dim3 grid(gx, (unsigned)((Cin + y_pad + DN - 1) / DN), (unsigned)B);
        hipLaunchKernelGGL(diag_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, (const float2*)w_c64, y, L, M, B, Cout, x_ld,
                           Cin, Cin + y_pad, y_ld, LM, (long long)Cout * LM, tri_off);
    }
    return mk_check_launch("mk_spec_diag_apply");
}

extern "C" int mk_spec_diag_wgrad(const float* x, const float* gy, float* gw_c64, int L, int M, int B, int Cin, int Cout, int x_ld,
                                  int y_ld, int tri_off, void* stream) {
    MK_REQUIRE(x && gy && gw_c64 && L > 0 && M > 0 && B > 0 && Cin > 0 && Cout > 0 && x_ld >= Cin && y_ld >= Cout, "spec_diag_wgrad: bad args");
    const long long LM = (long long)L * M;
    MK_REQUIRE(LM < (1ll << 31), "spec_diag_wgrad: L*M too large");
    const unsigned gy_ = (unsigned)((Cin + WI - 1) / WI), gz = (unsigned)((Cout + WO - 1) / WO);
    MK_REQUIRE(gy_ < 65536 && gz < 65536, "spec_diag_wgrad: too many channels for the grid");
    dim3 grid((unsigned)((LM + DP - 1) / DP), gy_, gz);
    hipLaunchKernelGGL(diag_wgrad_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, gy, (float2*)gw_c64, L, M, B, Cin, x_ld, Cout, y_ld, tri_off);
    return mk_check_launch("mk_spec_diag_wgrad");
}
