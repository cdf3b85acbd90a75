/*
 * makani_amd.h — C ABI of libmakani_amd.so (gfx950 / MI355X).
 *
 * The reference (NVIDIA/makani) has no FFI of its own: its hot path is Python
 * calling torch ops.  This header is therefore the boundary a
 * maintainer would bind *behind* the reference's nn.Module interface; every
 * entry point names the reference call it replaces (paths relative to
 * /root/reference).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *  - all pointers are DEVICE pointers (hipMalloc'ed / torch-allocated), 16-byte aligned;
 *  - `stream` is a hipStream_t passed as void* (0 = default stream); every call
 *    only enqueues work, nothing synchronises;
 *  - return value: 0 on success, negative MK_E* code on invalid arguments,
 *    positive hipError_t if the launch failed. mk_last_error() gives a message;
 *  - dtype codes: MK_F32 = 0, MK_BF16 = 1.
 *
 * Internal spectral layouts (all fp32, planar re/im, padded to multiples of 4):
 *  F-layout  F[m][k][ri][row]   m < M, k < nlat, ri in {re,im}, row < R (R = B*Cp)  (k-major: rows contiguous)
 *  S-layout  S[l][m][ri][row]   l < L, m < M, row < R                                (coefficients)
 *  W-layout  W[l][ri][i][o]     dhconv weights, i < Cip, o < Cop (zero padded)
 */
#ifndef MAKANI_AMD_H
#define MAKANI_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define MK_F32 0
#define MK_BF16 1

#define MK_EINVAL -1   /* bad argument (shape / alignment / dtype)            */
#define MK_EUNSUP -2   /* unsupported size (e.g. odd nlon, prime factor > 31)  */

/* triangular structure of the spherical-harmonic index space (P_l^m = 0 for l < m) */
#define MK_TRI_NONE 0
#define MK_TRI_ROW_GE 1 /* only output rows i >= t are computed                (Legendre analysis, t = m)   */
#define MK_TRI_K_GE 2   /* contraction runs over k >= t only                   (Legendre synthesis, t = m)  */
#define MK_TRI_ROW_LE 3 /* only output rows i <= t are computed                (dhconv fwd/dgrad, t = l)    */
#define MK_TRI_K_LE 4   /* contraction runs over k <= t only                   (dhconv wgrad, t = l)        */
/* with t = batch_index / inner + tri_off  (the OUTER batch index, shifted by the shard offset when the
 * triangular index space is sharded across ranks; t may be negative or exceed the extent) */

/* Strided batched GEMM descriptor:  C[b][i][j] (+)= sum_k A[b][i][k] * B[b][j][k]
 * Element strides.  One of {a_row, a_k} must be 1 (same for B); c_col must be 1.
 * Two-level batch: b = outer * inner + in;  offset = outer * x_batch + in * x_inner.
 * Complex variant: *_im = element offset from the real plane to the imaginary plane. */
typedef struct MkGemm {
    const float* A;
    const float* B;
    float* C;
    long long a_batch, a_row, a_k;
    long long b_batch, b_col, b_k;
    long long c_batch, c_row, c_col;
    long long a_inner, b_inner, c_inner;
    long long a_im, b_im, c_im;
    int M, N, K, batch; /* batch = outer count * inner */
    int inner;          /* >= 1 */
    int tri_mode;
    int tri_off;
    int conj_a, conj_b;
    int beta; /* 0: C = A*B^T ; 1: C += A*B^T */
} MkGemm;

const char* mk_last_error(void);
int mk_version(void);

/* ---- fp32 MFMA batched GEMMs ------------------------------------------------
 * mk_sgemm_batched : real.  Replaces the two einsums of th.RealSHT.forward /
 *   th.InverseRealSHT.forward [torch-harmonics, un-vendored; call sites
 *   makani/models/common/spectral_convolution.py:239,241,253] and their autograd.
 * mk_cgemm_batched : complex (planar).  Replaces _contract_lwise
 *   (makani/models/common/contractions.py:23-24) and its autograd. */
int mk_sgemm_batched(const MkGemm* g, void* stream);
int mk_cgemm_batched(const MkGemm* g, void* stream);
/* Same contracts on the bf16 matrix cores: every fp32 operand is split on the fly into `limbs` bf16 limbs
 * (3: x = hi+mid+lo, 6 MFMAs per product, fp32 round-off class, measured rel-L2 1.8e-7 vs fp64;
 *  2: x = hi+mid, 3 MFMAs, ~4e-6), fp32 accumulation.  6/16 resp. 3/16 of the exact-fp32 MFMA time. */
int mk_sgemm_split_batched(const MkGemm* g, int limbs, void* stream);
int mk_cgemm_split_batched(const MkGemm* g, int limbs, void* stream);
/* Second-generation kernels of the same arithmetic for the hot shapes (csrc/xgemm2.hip: 512-thread workgroups whose two
 * wave groups alternate between the matrix pipe and the limb split, double-buffered LDS images).
 * mk_cgemm_split2_batched: the complex split engine (mk_cgemm_split_batched is an alias of it since round 5, when the
 *   first-generation complex kernel was retired): any complex descriptor (the dhconv forward / data-gradient /
 *   weight-gradient GEMMs of _contract_lwise, makani/models/common/contractions.py:23-24).
 * mk_sgemm_presplit_batched: real GEMM whose A operand is a CONSTANT matrix handed over already split into `limbs`
 *   bf16 limb planes — the Legendre matrices of th.RealSHT / th.InverseRealSHT [un-vendored; precomputed in
 *   makani_amd/legendre.py].  g->A is ignored; plane q of batch b holds A[b][k][row] at
 *   a_planes + q*pl_stride + b*pl_batch + k*pl_k + row (bf16 elements; row contiguous, pl_k % 8 == 0, rows beyond M
 *   inside pl_k are zero).  B must be [k][col] with the column index contiguous (b_col == 1).  With inner > 1 the constant
 *   matrix, its band and the triangle belong to the OUTER index bo = b / inner (planes at + bo*pl_batch, band_lo[bo]); the
 *   inner index only moves B and C (b_inner / c_inner): several column blocks that live in separate buffers (the plane blocks
 *   of the h x w distributed transform, makani_amd/dist_pipeline.py) then share ONE launch and one pass over the matrix.
 *   band_lo / band_hi (device int[batch / inner], optional): outside [lo[b], hi[b]) every entry of A[b] is numerically zero by the
 *   caller's threshold (the Legendre functions of order m vanish towards the poles like sin^m theta).  band_mode 1: a range
 *   of k (analysis: the latitude sum is clipped); 2: a range of rows (synthesis: output latitudes outside the band are
 *   written as exact zeros); 0: no band.  Neither A nor B is read outside the band. */
int mk_cgemm_split2_batched(const MkGemm* g, int limbs, void* stream);
/* mk_cgemm_split2_batched that also hands back the sum of re^2 + im^2 over everything it writes, as one fp32 partial per workgroup
 * (ssq_part[0 .. mk_cgemm_split2_ssq_count(g)), fixed summation order; workgroups without work write 0): the weight gradient of
 * _contract_lwise (makani/models/common/contractions.py:23-24) is 283 MB per layer, and the global-norm clipping of
 * makani/utils/training/training_helpers.py:123-165 would otherwise read it once more just to square it (mk_grad_clip_coef_pre
 * takes the partials instead).  With beta = 1 the sums are those of the accumulated values. */
long long mk_cgemm_split2_ssq_count(const MkGemm* g);
int mk_cgemm_split2_batched_ssq(const MkGemm* g, int limbs, float* ssq_part, void* stream);
int mk_sgemm_presplit_batched(const MkGemm* g, const void* a_planes, long long pl_stride, long long pl_batch,
                              long long pl_k, int limbs, const int* band_lo, const int* band_hi, int band_mode,
                              void* stream);

/* ---- longitude FFTs ----------------------------------------------------------
 * mk_rfft_rows: x[row][lat][lon] (f32|bf16)  ->  F-layout, modes m < mmax:
 *     X_m = w_m * sum_n x_n exp(-2 pi i m n / nlon),  w = (w_dc, w_pos, w_nyq)
 *   Replaces `2*pi*torch.fft.rfft(x, norm="forward")[..., :mmax]` (th.RealSHT.forward;
 *   in-tree twin makani/mpu/fft.py:157-160) and, with other weights, the adjoint of irfft.
 * mk_irfft_rows: F-layout -> x[row][lat][lon] (f32|bf16):
 *     x_n = sum_m w_m * ( Re X_m cos(2 pi m n/nlon) - Im X_m sin(2 pi m n/nlon) )
 *   Replaces `torch.fft.irfft(X, n=nlon, norm="forward")` incl. the Im(m=0)/Im(Nyquist)
 *   drop (th.InverseRealSHT.forward; twin makani/mpu/fft.py:242) and the adjoint of rfft.
 * x holds B*C planes; plane (b, c) maps to F row b*Cp + c (R = B*Cp rows, pad rows untouched).
 * A workgroup transforms one latitude of 8-16 consecutive planes, so the F side is touched in runs of
 * 8-16 consecutive rows and the Legendre GEMMs see both operands row-contiguous.
 * `twiddle` = device table of nlon float2: exp(-2 pi i q / nlon) (host fp64 -> fp32).
 * `radix`   = host array of `nradix` radices whose product is nlon/2.               */
int mk_rfft_rows(const void* x, int x_dtype, float* F, const float* twiddle, const int* radix, int nradix,
                 int B, int C, int Cp, int nlat, int nlon, int mmax, float w_dc, float w_pos, float w_nyq,
                 void* stream);
int mk_irfft_rows(const float* F, void* x, int x_dtype, const float* twiddle, const int* radix, int nradix,
                  int B, int C, int Cp, int nlat, int nlon, int mmax, float w_dc, float w_pos, float w_nyq,
                  void* stream);

/* Segmented addressing for the distributed transforms (makani_amd/distributed.py; the reference schedule: torch-harmonics'
 * DistributedRealSHT / DistributedInverseRealSHT [un-vendored], in-tree twin makani/mpu/fft.py:148-182,214-249 with the
 * split / all_to_all / cat of makani/mpu/mappings.py:38-67).  Instead of packing send chunks and concatenating received ones
 * in separate passes, the FFT kernels address both of their sides through this descriptor:
 *   F side = per-peer slabs: slab (jw, ih) = orders m in [m_off[jw], m_off[jw+1]) x rows in [r_off[ih], r_off[ih+1]) of every
 *     latitude of the call, latitude outermost: [lat][m][re/im][row] at F + base[jw][ih] (floats).  nw / nh = number of
 *     m ranges / row ranges (<= MK_FFT_SEG_MAX); r_off and base entries are multiples of 4.
 *   x side = every row (plane, latitude) of nlon points cut into `xseg` equal pieces; piece j of all rows at
 *     x + j * x_stride elements, rows of a piece [plane][x_nlat][nlon / xseg] (xseg <= 1: whole rows, as mk_rfft_rows).
 *   A call may transform a latitude sub-range (chunked overlap of transform and exchange): nlat = its length, x and the slab
 *   offsets point at its first latitude, x_nlat = latitudes per plane of the x buffers.
 * One batch entry; C planes; implemented by the specialised row lengths only (mk_fft_seg_supported). */
#define MK_FFT_SEG_MAX 8
typedef struct MkFftSeg {
    int nw, nh;
    int m_off[MK_FFT_SEG_MAX + 1];
    int r_off[MK_FFT_SEG_MAX + 1];
    long long base[MK_FFT_SEG_MAX][MK_FFT_SEG_MAX];
    int xseg;
    long long x_stride;
    int x_nlat; /* latitudes per plane in the x buffers (the call may cover a sub-range: x then points at its first latitude);
                   0 = nlat of the call */
} MkFftSeg;
int mk_fft_seg_supported(int nlon);
int mk_rfft_rows_seg(const void* x, int x_dtype, float* F, const float* twiddle, int C, int nlat, int nlon, int mmax,
                     float w_dc, float w_pos, float w_nyq, const MkFftSeg* seg, void* stream);
int mk_irfft_rows_seg(const float* F, void* x, int x_dtype, const float* twiddle, int C, int nlat, int nlon, int mmax,
                      float w_dc, float w_pos, float w_nyq, const MkFftSeg* seg, void* stream);

/* ---- layout changes ------------------------------------------------------------
 * complex64 dhconv parameter (Cin, Cout, L) [makani/models/common/spectral_convolution.py:164-193]
 * <-> W-layout, and S-layout <-> complex64 (rows, L, M) tensors at the RealSHT API boundary. */
int mk_weight_to_wlayout(const float* w_c64, float* W, int cin, int cout, int cip, int cop, int L, void* stream);
int mk_wlayout_to_weight_grad(const float* gW, float* gw_c64, int cin, int cout, int cip, int cop, int L, void* stream);
int mk_slayout_to_complex(const float* S, float* out_c64, int B, int C, int Cp, int L, int M, int l_off, int m_off,
                          void* stream);
int mk_complex_to_slayout(const float* in_c64, float* S, int B, int C, int Cp, int L, int M, void* stream);

/* ---- pointwise blocks -----------------------------------------------------------
 * All operate on NCHW planes: x[(b*channels + c)][hw], dtype f32 | bf16, fp32 arithmetic.
 * `ws` is a caller-provided scratch of >= planes * mk_pointwise_chunks(hw, dtype, planes) * 2 floats (a plane is cut
 * into as many chunks as keep every workgroup of the launch resident at once).
 *
 * Instance norm = nn.InstanceNorm2d(C, eps, affine=True) at makani/models/networks/sfnonet.py:618-620
 * (statistics in fp32 as in makani/mpu/layer_norm.py:147-168); optional fused GELU
 * (nn.GELU, sfnonet.py:392-393; exact erff on fp32 tensors, the A&S 7.1.26 erf — |err| < 1.5e-7 — on bf16 tensors).  stats: (planes, 2) f32 = {mean, rstd}.
 * Backward: sums (2, planes) f32, planar: row 0 = sum ga (dbeta per plane), row 1 = sum ga * xhat (dgamma per plane),
 * gx = rstd*gamma*(ga - mean(ga) - xhat*mean(ga*xhat)), ga = gy * (gelu'(a) if fused).
 * phase 0 = reduce + apply (serial); 1 = reduce only (writes the local `sums`); 2 = apply only with the
 * caller-provided (all-reduced) `sums` and `hw_total` = pixels of the whole plane over all spatial ranks —
 * the split DistributedInstanceNorm2d (makani/mpu/layer_norm.py:108-170) needs.                      */
int mk_pointwise_chunks(long long hw, int dtype, long long planes);
/* sums[p] = sum of plane p (fp32; `sums` holds 2 * planes floats, the second row is zero): the bias gradient of a 1x1
 * convolution, `grad_output.sum((0, 2, 3))` in the autograd of nn.Conv2d (makani/models/common/layers.py:603-643). */
int mk_plane_sums(const void* x, int dtype, float* sums, float* ws, long long planes, long long hw, void* stream);
/* `quad` (hw fp32 weights of this shard, may be NULL) with `quad_sum` = their sum: quadrature-weighted local moments
 * with count = quad_sum — what DistributedGeometricInstanceNormS2 merges (makani/mpu/layer_norm.py:207-222). */
int mk_instnorm_stats(const void* x, int dtype, float* stats, float* ws, long long planes, long long hw, float eps,
                      const float* quad, float quad_sum, void* stream);
/* merged statistics of planes sharded over `nranks` ranks: all_stats (nranks, planes, 2) = every rank's local {mean, rstd}
 * (the all-gathered outputs of mk_instnorm_stats), counts (nranks) = every rank's pixel count (or sum of quadrature weights);
 * stats (planes, 2) = {mean, rstd} of the whole plane — the pairwise moment merge of DistributedInstanceNorm2d
 * (makani/mpu/layer_norm.py:31-81,140-152), fp64 inside, no host round trip. */
int mk_instnorm_merge(const float* all_stats, const float* counts, float* stats, long long planes, int nranks, float eps,
                      void* stream);
int mk_instnorm_apply(const void* x, void* y, int dtype, const float* stats, const float* gamma, const float* beta,
                      long long planes, int channels, long long hw, int fuse_gelu, void* stream);
/* stats + apply in two launches (the apply kernel finishes the statistics reduction itself and writes `stats` for the
 * backward): what the serial InstanceNorm2d forward uses.  `pre_bias` (channels, may be NULL) folds the bias of the
 * convolution in front of the norm (the MLP's fc2, makani/models/common/layers.py:768-823) into both passes: the norm
 * sees (x + pre_bias[c]) rounded to the tensor dtype, exactly the tensor the reference materialises; same in backward.
 * `quad` (hw floats, may be NULL; `quad_sum` = their sum Q, 1 on a full grid, < 1 on a crop) switches to
 * quadrature-weighted statistics mean = sum q x, var = sum q (x - mean)^2: GeometricInstanceNormS2
 * (makani/models/common/layer_norm.py:30-160); the backward then is
 * gx_i = rstd gamma (ga_i - q_i (S1 + (n_i - mean rstd (1 - Q)) S2)) with the unweighted sums S1 = sum ga, S2 = sum ga n. */
int mk_instnorm_fwd(const void* x, void* y, int dtype, float* stats, float* ws, const float* gamma, const float* beta,
                    const float* pre_bias, const float* quad, float quad_sum, long long planes, int channels, long long hw,
                    float eps, int fuse_gelu, void* stream);
int mk_instnorm_bwd(const void* x, const void* gy, void* gx, int dtype, const float* stats, const float* gamma,
                    const float* beta, const float* pre_bias, const float* quad, float quad_sum, float* sums, float* ws, long long planes, int channels, long long hw,
                    long long hw_total, int phase, int fuse_gelu, void* stream);
/* The same norm in ONE pass over the plane (round 6): a block keeps its chunk of the plane in registers between the
 * statistics and the apply phase — 2 plane-sized transfers forward instead of 3, 3 backward instead of 5 — and the blocks of a
 * plane hand their partial sums over through 64-bit agent-scope atomics (csrc/pointwise.hip: in_fwd_fused).  Serves unweighted
 * statistics on planes whose size is a multiple of the 16-byte vector; mk_instnorm_fused_chunks returns 0 otherwise (use the
 * two-kernel entry points above).  `slots`: planes * mk_instnorm_fused_chunks(..., kind) 64-bit words, ALL ONES before the first
 * launch; `depart`: planes 32-bit words, ZERO before the first launch; the kernels leave both as they found them, so one pair
 * serves every later launch on the same stream (never two launches that may run concurrently).  Same arithmetic and outputs
 * as mk_instnorm_fwd / mk_instnorm_bwd (phase 0) — nn.InstanceNorm2d, makani/models/networks/sfnonet.py:618-620. */
int mk_instnorm_fused_chunks(long long hw, int dtype, long long planes, int kind);   /* kind: 0 forward, 1 backward, 2 backward + GELU */
int mk_instnorm_fwd_fused(const void* x, void* y, int dtype, float* stats, const float* gamma, const float* beta,
                          const float* pre_bias, void* slots, void* depart, long long planes, int channels, long long hw,
                          float eps, int fuse_gelu, void* stream);
int mk_instnorm_bwd_fused(const void* x, const void* gy, void* gx, int dtype, const float* stats, const float* gamma,
                          const float* beta, const float* pre_bias, float* sums, void* slots, void* depart, long long planes,
                          int channels, long long hw, int fuse_gelu, void* stream);
/* y = gelu(x + bias[c])  — the bias+activation of the 1x1 convolutions in MLP / EncoderDecoder
 * (makani/models/common/layers.py:603-643,768-823).  bias may be NULL (plain GELU).
 * Backward: gx = gy * gelu'(x + bias[c]); optional sums (2, planes): sums[0][p] = sum gx (bias grad). */
int mk_bias_gelu_fwd(const void* x, const float* bias, void* y, int dtype, long long planes, int channels,
                     long long hw, void* stream);
int mk_bias_gelu_bwd(const void* x, const float* bias, const void* gy, void* gx, float* sums, float* ws, int dtype,
                     long long planes, int channels, long long hw, void* stream);

/* ---- quadrature-weighted L^p plane sums (geometric losses) ---------------------------------
 * Replace GridQuadrature.forward (makani/utils/grids.py:185-191) and the elementwise chain of
 * GeometricLpLoss.abs/rel (makani/utils/losses/lp_loss.py:61-107) and their autograd:
 *   mode 0:  sums[0][plane] = sum_i q[i] * a[plane][i] (* wgt[plane][i])
 *   mode 1:  sums[0][plane] = sum_i q[i] * |a[plane][i] - b[plane][i]|^p (* wgt[plane][i])     (b NULL = 0)
 * a, b: (planes, hw) f32 | bf16 independently (prediction bf16, target f32); q: (hw) f32 quadrature weights;
 * wgt: optional (planes, hw) f32; sums: (2, planes) f32 (second row 0); ws: planes * mk_quad_lp_chunks(hw) * 2
 * floats of scratch.  Backward: da = g[plane] * q * wgt * d|d|^p/dd (mode 0: g * q * wgt), db = -da; either may be
 * NULL; gradients are written in the dtype of the tensor they belong to. */
int mk_quad_lp_chunks(long long hw);
int mk_quad_lp_fwd(const void* a, int a_dtype, const void* b, int b_dtype, const float* wgt, const float* q,
                   float* sums, float* ws, long long planes, long long hw, int mode, float p, void* stream);
int mk_quad_lp_bwd(const void* a, int a_dtype, const void* b, int b_dtype, const float* wgt, const float* q,
                   const float* g, void* da, void* db, long long planes, long long hw, int mode, float p, void* stream);

/* ---- spectral L^p sums on the S-layout (spectral losses) ------------------------------------
 * Replace the chain |coeffs|^p * wgt, the m = 0 once / m > 0 twice Parseval sum over m and the sum over l of
 * SpectralLpLoss.abs/rel (makani/utils/losses/lp_loss.py:141-247) and their autograd, directly on the SHT output:
 *   partial[b][row] = sum over the 32 (l, m) pairs of block b with l + tri_off >= m of w(m) |c_lm|^p (* wgt[l][m][row])
 * w(m) = w0 when m + m_off == 0 (global order 0), else w1.  S: (L, M, 2, R) f32; wgt: optional (L, M, R) f32;
 * partial: (mk_spec_lp_blocks(L, M), R) f32, summed over its first axis by the caller.
 * Backward: dS = g[row] * w(m) * p * |c|^(p-2) * c (* wgt), exact zeros where l + tri_off < m. */
long long mk_spec_lp_blocks(int L, int M);
int mk_spec_lp_fwd(const float* S, const float* wgt, float* partial, int L, int M, long long R, int tri_off, int m_off,
                   float p, float w0, float w1, void* stream);
int mk_spec_lp_bwd(const float* S, const float* wgt, const float* g, float* dS, int L, int M, long long R, int tri_off,
                   int m_off, float p, float w0, float w1, void* stream);

/* ---- spectral contractions that are not matrix products over l --------------------------------
 * Activations x, y, gy: S-layout (L, M, 2, R) f32; position (l, m) is live when m <= l + tri_off, dead positions
 * are written as exact zeros and never read.
 *
 * mk_spec_sep_mul / mk_spec_sep_wgrad: the separable operators _contract_sep_lmwise ("bgixy,gixy->bgixy") and
 *   _contract_sep_lwise ("bgixy,gix->bgixy") (makani/models/common/contractions.py:26-31) and their autograd.
 *   R = B*Cp, Cp % 4 == 0.  w, gw: the weight in S-layout (L, Mw, 2, Cp), Mw = M (lm-wise) or 1 (l-wise).
 *     sep_mul:   y = x * w            (conj_w = 0: forward)      y = x * conj(w)   (conj_w = 1: input gradient of gy)
 *     sep_wgrad: gw[l][mw][c] = sum_b (and sum over live m when Mw == 1) conj(x) * gy
 *
 * mk_spec_diag_apply / mk_spec_diag_wgrad: the dense diagonal operator _contract_lmwise ("bgixy,gioxy->bgoxy",
 *   contractions.py:17-18) for ONE group.  w_c64 / gw_c64: the group's slice of the PARAMETER, (Cin, Cout, L, M)
 *   complex64, read and written in place (every weight byte crosses HBM exactly once, coalesced along (l, m)).
 *   x: Cin channels inside rows of stride x_ld per batch entry (R_x = B*x_ld), y likewise with y_ld; pointers may be
 *   offset to a group's first channel.
 *     diag_apply dgrad = 0: y[p][b][o] = sum_i x[p][b][i] * w[i][o][p]; channels [Cout, Cout + y_pad) are zeroed
 *                dgrad = 1: x is gy (Cout channels), y is gx[p][b][i] = sum_o gy[p][b][o] * conj(w[i][o][p])
 *     diag_wgrad: gw[i][o][p] = sum_b conj(x[p][b][i]) * gy[p][b][o] */
int mk_spec_sep_mul(const float* x, const float* w, float* y, int L, int M, int Mw, int B, int Cp, int tri_off, int conj_w,
                    void* stream);
int mk_spec_sep_wgrad(const float* x, const float* gy, float* gw, int L, int M, int Mw, int B, int Cp, int tri_off, void* stream);
int mk_spec_diag_apply(const float* x, const float* w_c64, float* y, int L, int M, int B, int Cin, int Cout, int x_ld, int y_ld,
                       int y_pad, int tri_off, int dgrad, void* stream);
int mk_spec_diag_wgrad(const float* x, const float* gy, float* gw_c64, int L, int M, int B, int Cin, int Cout, int x_ld, int y_ld,
                       int tri_off, void* stream);

/* ---- bf16 channel GEMMs (1x1 convolutions on NCHW planes) ---------------------------------
 * Replace nn.Conv2d(kernel_size=1) of MLP / EncoderDecoder / outer_skip / residual_transform
 * (makani/models/common/layers.py:603-643,768-823; makani/models/networks/sfnonet.py:335-338,726-730)
 * and its autograd under bf16 autocast.  All tensors bf16 except bias (f32) and dW/part (f32).
 *   mk_conv1x1_nn:   Y[b][m][n] = epi( sum_k A[m][k] X[b][k][n] ),  n = pixel (contiguous), n % 8 == 0
 *       A: (M, lda) k-contiguous, zero padded to lda (multiple of 8).  Forward: A = W; dgrad: A = W^T.
 *       epi(v) = v + bias[m]; if act: (Ypre = v), v = gelu(v); if G: v *= gelu'(G[b][m][n]); if R: v += R[b][m][n].
 *   mk_conv1x1_wgrad: dW[m][k] (+)= sum_{b,n} G[b][m][n] X[b][k][n]; `part` = scratch of
 *       mk_conv1x1_wgrad_workspace(M, K, B, N) floats, laid out as (S, M, K) split-pixel partial tiles FOLLOWED BY (S, M)
 *       partial row sums of G (S = the kernel's pixel splits; both reduced deterministically).  The query returns
 *       S*M*K + S*M; mk_conv1x1_wgrad_bias REQUIRES a scratch of the size this version of the query returns (a caller that
 *       sized it as S*M*K, the layout before the bias sums existed, would be overrun by S*M floats).
 *   mk_conv1x1_wgrad_bias: the same and, from the same pass over G, the bias gradient of the convolution
 *       db[m] = sum_{b,n} G[b][m][n] (what torch's conv backward returns as grad_bias; fp32, M entries, overwritten) —
 *       available where mk_conv1x1_wgrad_fuses_bias(M, K, B, N) returns 1 (the streaming "ring" kernel), else the caller
 *       takes the plane sums with mk_plane_sums. */
int mk_conv1x1_nn(const void* A, const void* X, void* Y, void* Ypre, const float* bias, const void* R, const void* G,
                  int M, int K, int lda, int B, long long N, int act, void* stream);
long long mk_conv1x1_wgrad_workspace(int M, int K, int B, long long N);
int mk_conv1x1_wgrad(const void* G, const void* X, float* dW, float* part, int M, int K, int B, long long N,
                     int accumulate, void* stream);
int mk_conv1x1_wgrad_fuses_bias(int M, int K, int B, long long N);
int mk_conv1x1_wgrad_bias(const void* G, const void* X, float* dW, float* dbias, float* part, int M, int K, int B, long long N,
                          int accumulate, void* stream);

/* ---- optimizer ------------------------------------------------------------------------
 * One fused AdamW update (torch.optim.AdamW semantics: decoupled weight decay, bias correction with
 * `step` >= 1) over a flat fp32 tensor; complex64 parameters are passed as their real view
 * (as makani views them, makani/mpu/mappings.py:467-489).  The reference's train step uses AdamW
 * (config/sfnonet.yaml:50-54) after global-norm clipping (utils/training/training_helpers.py:123-165):
 * `grad_scale` (device pointer, may be NULL) is that clipping coefficient, applied to g on the fly. */
int mk_adamw_step(float* p, const float* g, float* m, float* v, long long n, const float* grad_scale, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int step, const float* step_state, void* stream);
/* Device-side step counter: `state` = 3 floats {step, 1 / (1 - beta1^step), 1 / sqrt(1 - beta2^step)} (zero-initialised by
 * the caller); one launch increments the step and refreshes the two bias corrections.  The update kernels read them when
 * `step_state` is non-NULL (and ignore `step`): no launch argument depends on the step number, which is what makes a
 * captured hipGraph of the whole train step replayable. */
int mk_adamw_advance(float* state, float beta1, float beta2, void* stream);
/* The same update over `count` tensors (host array of descriptors) in ceil(count / 48) launches: for the ~80 small
 * tensors of the model, where one launch each costs more in dispatch gaps than in HBM time. */
typedef struct MkAdamTensor {
    float* p;
    const float* g;
    float* m;
    float* v;
    long long n;
    void* p_bf16;   /* optional (mk_adamw_multi only): receives bf16(updated p), the autocast operand of the next step */
    void* p_bf16_t; /* optional: receives bf16(updated p) transposed (needs cols > 0): the data-gradient GEMM's operand */
    int cols;       /* > 0: p is an (n / cols, cols) matrix; then p_bf16 has row pitch ld, p_bf16_t row pitch ld_t */
    int ld, ld_t;
} MkAdamTensor;
int mk_adamw_multi(const MkAdamTensor* tensors, int count, const float* grad_scale, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int step, const float* step_state, void* stream);
/* Global gradient norm and the clipping coefficient of makani/utils/training/training_helpers.py:123-165 over all
 * `tensors[i].g` (only g and n are read): out[0] = min(1, max_norm / (norm + 1e-6)) (1 if max_norm <= 0), out[1] = norm.
 * `partial` needs mk_grad_norm_workspace() floats.  ceil(count / 48) + 1 launches, fixed summation order. */
long long mk_grad_norm_workspace(const MkAdamTensor* tensors, int count);
int mk_grad_clip_coef(const MkAdamTensor* tensors, int count, float max_norm, float* partial, float* out, void* stream);
/* The same with `npre` buffers of ALREADY SQUARED partial sums (pre[i].g = the buffer, pre[i].n = its length; e.g. the ssq_part of
 * mk_cgemm_split2_batched_ssq) standing in for the tensors they were formed from: norm^2 = sum of squares of tensors[] + sum of pre[].
 * `partial` needs mk_grad_norm_workspace_pre() floats; count or npre may be 0. */
long long mk_grad_norm_workspace_pre(const MkAdamTensor* tensors, int count, const MkAdamTensor* pre, int npre);
int mk_grad_clip_coef_pre(const MkAdamTensor* tensors, int count, const MkAdamTensor* pre, int npre, float max_norm, float* partial,
                          float* out, void* stream);

/* ---- layer norm over the channels of an NCHW tensor ------------------------------------------------------------------
 * Replaces DistributedLayerNorm (makani/mpu/layer_norm.py:256-290: nn.LayerNorm(C) between two transposes of the NCHW
 * activation; `normalization_layer="layer_norm"`, makani/models/networks/sfnonet.py:609-613, fourcastnet3.py:95-96) and its
 * autograd without the transposes.  x, y, gy, gx: (B, C, P) planes (P = H * W grid points); stats: (B, 2, P) f32 = [mean,
 * rstd] per grid point; gamma / beta: (C) f32 or NULL.  dtypes: x f32 -> y f32; x bf16 -> y f32 (nn.LayerNorm under
 * autocast) or bf16; gy has y's dtype, gx has x's.
 * wgrad: partial[(2, C, mk_chan_layernorm_chunks(C, P))] = per-chunk sums of gy * xhat (dgamma) and gy (dbeta). */
int mk_chan_layernorm_chunks(int C, long long P);
int mk_chan_layernorm_fwd(const void* x, int x_dtype, void* y, int y_dtype, float* stats, const float* gamma, const float* beta,
                          int B, int C, long long P, float eps, void* stream);
int mk_chan_layernorm_bwd(const void* x, int x_dtype, const void* gy, int g_dtype, void* gx, const float* stats,
                          const float* gamma, int B, int C, long long P, void* stream);
int mk_chan_layernorm_wgrad(const void* x, int x_dtype, const void* gy, int g_dtype, const float* stats, float* partial, int B,
                            int C, long long P, void* stream);

/* ---- ensemble CRPS on the sphere -----------------------------------------------------------------------------------
 * Pointwise score over the ensemble dimension fused with the quadrature over the plane: replaces the kernels of
 * makani/utils/losses/crps_loss.py:124-275 ("skillspread" = type 0, the default of CRPSLoss :277-452; "probability weighted
 * moment" = 1; "naive skillspread" = 2; "gauss" = 3) and the piecewise-integrated "cdf" form of :55-122 (type 4: members in
 * rank order, optional per-member ensemble weights `ens_w` (E) — `self.ensemble_weights[idx]` of :392-396), the weighted sum
 * of :435-438 and their autograd.
 * f: (B, E, C, hw) forecasts, obs: (B, C, hw), q: (hw) quadrature weights, w: optional (B, C, hw) spatial weights.
 * grad == 0: partial[(B * C) * mk_crps_chunks(hw)] chunk sums of q * w * crps (the caller adds the chunks of a plane);
 * grad == 1: gf (shape / dtype of f) = gout[b * C + c] * q * w * d crps / d f_e.  Any 2 <= E <= 32 (members live in registers). */
int mk_crps_chunks(long long hw);
int mk_crps(const void* f, int f_dtype, const void* obs, int o_dtype, const float* q, const float* w, const float* gout,
            float* partial, void* gf, int B, int E, int C, long long hw, int type, float alpha, float eps, int grad,
            const float* ens_w, void* stream);
/* The "naive skillspread" score on COMPLEX members — SpectralCRPSLoss(absolute=False), makani/utils/losses/crps_loss.py:205-243,
 * 536-545: |.| is the complex modulus.  f (B, E, C, hw) and obs (B, C, hw) complex64 (re, im interleaved); gf like f (torch's
 * complex-gradient convention); q / w / gout / partial as mk_crps. */
int mk_crps_complex(const void* f, const void* obs, const float* q, const float* w, const float* gout, float* partial, void* gf,
                    int B, int E, int C, long long hw, float alpha, int grad, void* stream);

/* ---- DISCO convolution and S2 resampling (FourCastNet3's local operators) ----------------------------------------
 * Replace th.DiscreteContinuousConvS2's sparse contraction and th.ResampleS2 [torch-harmonics, un-vendored; call sites
 * makani/models/networks/fourcastnet3.py:189-205 (encoder), :356-381 (decoder), :518-534 (local blocks)].
 * The convolution tensor psi[k][t][(i, j)] is handed over as lists (built by makani_amd/disco.py):
 *   forward lists, sorted by (t, k):  off[t * K + k] .. off[t * K + k + 1] index (nrow, nlon, nval) = (input latitude
 *     relative to lat_lo[t], input longitude, value); lat_n[t] rows are touched, max_rows = max_t lat_n[t];
 *   transposed lists for the adjoint: per input latitude (mk_disco_bwd: entries (k, t, lon, val), any nlon_in =
 *     s * nlon_out) or per (input latitude, k) with rows relative to t_lo[i * K + k] and negated longitudes
 *     (mk_disco_bwd_same: nlon_in == nlon_out).
 * mk_disco_fwd:  y[pl * K + k][t][p] = sum_n nval[n] * x[pl][lat_lo[t] + nrow[n]][(nlon[n] + p * s) mod nlon_in]
 *   x: (planes, nlat_in, nlon_in), y: (planes * K, nlat_out, nlon_out) — the NCHW input of the channel GEMM
 *   (mk_conv1x1_nn) that applies the (out, in * K) weight; dtype f32 | bf16, fp32 accumulation.
 * mk_disco_bwd / mk_disco_bwd_same: the adjoint (gradient with respect to x), deterministic gathers.
 * mk_resample_fwd: bilinear interpolation, latitude first (rows lat_a / lat_b with weight lat_w; a row index -1 / -2 is
 *   the longitude mean of the first / last input row: the pole extension), then periodic longitude (lon_l, lon_r, lon_w).
 * mk_resample_bwd: its adjoint from the inverse stencils (CSR per input row / input column / pole). */
int mk_disco_fwd(const void* x, void* y, int dtype, const int* off, const int* nrow, const int* nlon, const float* nval,
                 const int* lat_lo, const int* lat_n, int max_rows, int planes, int K, int nlat_in, int nlon_in,
                 int nlat_out, int nlon_out, void* stream);
int mk_disco_bwd(const void* gy, void* gx, int dtype, const int* off, const int* nk, const int* nt, const int* nlon,
                 const float* nval, int planes, int K, int nlat_in, int nlon_in, int nlat_out, int nlon_out, void* stream);
int mk_disco_bwd_same(const void* gy, void* gx, int dtype, const int* off, const int* nrow, const int* nlon_l,
                      const float* nval, const int* t_lo, const int* t_n, int max_rows, int planes, int K, int nlat_in,
                      int nlon, int nlat_out, void* stream);
/* Run form of the same contraction for nlon_in == nlon_out (FourCastNet3's local blocks and decoder convolutions,
 * fourcastnet3.py:356-381,518-534; forward and adjoint): csrc/disco_runs.hip, sliding-window correlations in registers.
 *   mk_disco_runs_shape: 1 when the run-form kernels take (nlon, max_rows image rows, planes, dtype, img_bf16), with the
 *     longitudes per lane R (4 | 8) and the planes per workgroup PB (4 | 2; *PB_out preset to 2 or 4 asks for exactly
 *     that value); 0 -> use the list kernels above.
 *   lists (makani_amd/disco.py: _build_runs): seg_off[s] .. seg_off[s + 1] index runs (n, 4) = {image row, first longitude,
 *     offset into vals, groups of R values}; vals zero-padded to whole groups plus R trailing zeros.
 *   mk_disco_fwd_runs: segments s = t * K + k, image rows relative to lat_lo[t] (lat_n[t] rows, max_rows their maximum).
 *   mk_disco_bwd_runs: segments s = i * K + k with negated longitudes, image rows = output latitudes relative to
 *     t_lo[(i / lat_group) * K + k] (t_n rows: lat_group = 2 | 4 consecutive latitudes share one staged image).
 *   img_bf16: keep bf16 tensors as bf16 in the LDS row image (half the LDS per workgroup, one conversion per read).
 *   mk_disco_fwd_fused (K = 9): one stream per (output latitude, image row) shared by all basis functions (their filters
 *     live on the same longitude interval): seg_off (nlat_out + 1), runs (n, 4) = {image row relative to lat_lo[t / LG], first
 *     slot = first longitude / 4 (runs aligned to 4 longitudes), value offset, groups}, vals per group K x 4 ([k][tau]);
 *     lat_lo / lat_n per group of LG output latitudes, LG as named by mk_disco_fused_shape for the longitude count. */
int mk_disco_runs_shape(int nlon, int max_rows, int planes, int dtype, int img_bf16, int* R_out, int* PB_out);
int mk_disco_fused_shape(int nlon, int K, int max_rows, int planes, int* LG_out, int* PB_out);
int mk_disco_fwd_fused(const void* x, void* y, int dtype, const int* seg_off, const int* runs, const float* vals,
                       const int* lat_lo, const int* lat_n, int max_rows, int planes, int K, int nlat_in, int nlon, int nlat_out,
                       void* stream);
int mk_disco_fwd_runs(const void* x, void* y, int dtype, const int* seg_off, const int* runs, const float* vals,
                      const int* lat_lo, const int* lat_n, int max_rows, int planes, int K, int nlat_in, int nlon, int nlat_out,
                      int R, int PB, int img_bf16, void* stream);
int mk_disco_bwd_runs(const void* gy, void* gx, int dtype, const int* seg_off, const int* runs, const float* vals,
                      const int* t_lo, const int* t_n, int max_rows, int planes, int K, int nlat_in, int nlon, int nlat_out,
                      int R, int PB, int img_bf16, int lat_group, void* stream);
/* Grouped channel mix with a handful of channels per group (csrc/groupmix.hip): the channel mixes of FourCastNet3's encoders /
 * decoders (th.DiscreteContinuousConvS2 with groups > 1, fourcastnet3.py:189-205,356-381: 8-9 planes in and out per group).
 *   mk_group_mix:        z (B*G, RG, N) = W (G, RG, CG) x (B*G, CG, N); dtype f32 | bf16, fp32 accumulation; the data gradient is
 *                        the same call with the transposed weights.  N a multiple of 4 (f32) / 8 (bf16).
 *   mk_group_mix_wgrad:  partial (B*G * mk_group_mix_blocks(N, dtype, B*G), RG * CG) = per-block sums of dz[r][n] x[c][n]; the caller
 *                        adds the blocks of a group and the batch entries (deterministic).
 *   mk_group_mix_supported: 1 for the instantiated (CG, RG) pairs (8 | 9 each). */
int mk_group_mix_supported(int CG, int RG);
int mk_group_mix_blocks(long long N, int dtype, int BG);
int mk_group_mix(const void* x, const float* W, void* z, int dtype, int B, int G, int CG, int RG, long long N, void* stream);
int mk_group_mix_wgrad(const void* x, const void* dz, float* partial, int dtype, int B, int G, int CG, int RG, long long N,
                       void* stream);
int mk_resample_fwd(const void* x, void* y, int dtype, const int* lat_a, const int* lat_b, const float* lat_w,
                    const int* lon_l, const int* lon_r, const float* lon_w, int planes, int nlat_in, int nlon_in,
                    int nlat_out, int nlon_out, void* stream);
int mk_resample_bwd(const void* gy, void* gx, int dtype, const int* lat_off, const int* lat_t, const float* lat_wt,
                    const int* lon_off, const int* lon_p, const float* lon_wt, const int* pole_off, const int* pole_t,
                    const float* pole_wt, int planes, int nlat_in, int nlon_in, int nlat_out, int nlon_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MAKANI_AMD_H */
