// DISCO contraction, run form (gfx950): the equal-longitude-count case (nlon_in == nlon_out: FourCastNet3's "local" blocks and
// its decoder convolutions, forward and adjoint) as sliding-window correlations in registers.
//
// Replaces th.DiscreteContinuousConvS2's sparse contraction for that case [torch-harmonics, un-vendored; call sites
// makani/models/networks/fourcastnet3.py:356-381,518-534]; csrc/disco.hip keeps the general (strided) kernels.
//
//   y[pl][k][t][p] = sum_{row, j} psi[k][t][row][j] * x[pl][lo(t) + row][(j + p) mod N]
//
// For a fixed (k, t, row) the non-zeros of psi in j form ONE circular run (the filter support is a disc), so the sum over j is a
// 1-D circular correlation of a short filter with a latitude row.  The list kernels of disco.hip read one LDS operand per
// multiply-add (x PB planes) and have a lane own output longitudes 256 apart; here a lane owns R CONSECUTIVE output
// longitudes, keeps a window of 2 R - 1 row elements in registers and slides it: R new LDS reads per R taps feed R * R * PB
// multiply-adds (R = 4: 16 per read instead of 4).  The filter values are wave-uniform and come through the scalar cache
// (s_load), not through LDS.
//
// Data layout.  The host turns every (segment, row) into runs {row, first longitude js, value offset, groups}: values padded
// with zeros to a multiple of R (a "group" = R taps).  The row image in LDS is de-interleaved by longitude class so that the
// 64 lanes of a wave, which read longitudes R apart, touch consecutive slots (no bank conflicts):
//     element (row, lon, plane)  ->  row * ROWB + (lon % R) * SEGB + (lon / R) * SLOT + plane * sizeof(IMG)
// with one duplicate slot behind every class segment (slot N/R = slot 0), so that "next slot" never needs a second wrap.
// A run whose first longitude is js = R bq + bm reads, for the element e of its stream, class (bm + e) % R and slot
// (bq + lane + (bm + e) / R) mod N/R: with bm a template parameter of the run body (a uniform switch per run) every class and
// carry is a compile-time constant, i.e. an immediate offset of the LDS read; per group a lane spends 3 integer instructions
// (advance + wrap of its slot address) beside R reads and R * R * PB multiply-adds.
//
//   forward, all K basis functions per stream (disco_fused_fwd_kernel, the default for K = 9): workgroup = (LG consecutive
//             output latitudes, 2 planes); the union of the x rows they touch is staged once; wave group lg of NW waves walks
//             the (row) streams of latitude t0 + lg, every stream feeding the K accumulator sets of its lanes.
//   forward, one stream per basis function (disco_runs_fwd_kernel): workgroup = (output latitude t, PB planes), WV waves take
//             the (k, 64-lane longitude segment) items round-robin.
//   adjoint / K planes in, one out (disco_runs_bwd_kernel): workgroup = (LG consecutive input latitudes, PB planes); for every k
//             the gradient rows of gy[k] that those latitudes touch are staged (two barriers per k), wave group lg accumulates
//             gx of latitude i0 + lg over all k.  The same kernel evaluates sum_k psi_k (*) z_k on the transposed tensor's lists.
// Measured on FourCastNet3's local block (360 x 720, 677 planes, 570 GFLOP): fused forward 7.0 ms = 81 TFLOP/s fp32 (0.51 of
// the packed-FMA vector peak; VALU issue 0.85 busy), per-k forward 10.4 ms, adjoint 13.1 ms; the list kernels of disco.hip 33 /
// 59 ms.
#include "common.h"

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const unsigned char lds_cb;

template <int PB>
struct FV;
template <>
struct FV<2> { typedef f32x2 type; };
template <>
struct FV<4> { typedef f32x4 type; };

// one image slot (PB planes of one longitude) -> PB floats
template <typename IMG, int PB>
__device__ __forceinline__ typename FV<PB>::type load_slot(lds_cb* p) {
    typedef typename FV<PB>::type fv;
    if constexpr (sizeof(IMG) == 4) {
        return *(__attribute__((address_space(3))) const fv*)p;
    } else if constexpr (PB == 4) {
        const u32x2 r = *(__attribute__((address_space(3))) const u32x2*)p;
        fv v;
        v[0] = __uint_as_float(r[0] << 16), v[1] = __uint_as_float(r[0] & 0xffff0000u);
        v[2] = __uint_as_float(r[1] << 16), v[3] = __uint_as_float(r[1] & 0xffff0000u);
        return v;
    } else {
        const unsigned r = *(__attribute__((address_space(3))) const unsigned*)p;
        fv v;
        v[0] = __uint_as_float(r << 16), v[1] = __uint_as_float(r & 0xffff0000u);
        return v;
    }
}

template <typename IMG, typename T>
__device__ __forceinline__ IMG img_cvt(T v) {
    if constexpr (sizeof(IMG) == sizeof(T)) return v;
    else return bf16_to_f32(v);                                 // bf16 tensor, fp32 image
}

// rows [0, nr) x all longitudes x PB planes of `src` (plane stride plane_step, row stride nlon) -> the de-interleaved image.
// Work item = (row, slot q): PB vector loads of the R consecutive longitudes q R .. q R + R - 1 (one per plane), transposed
// in registers into R image slots (one per longitude class, PB planes each) and written as whole slots: lanes with
// consecutive q read consecutive global memory and write consecutive LDS slots of every class segment.  Two items per
// thread are in flight (the loads of both before the first LDS write).
template <typename T, typename IMG, int PB, int R, int SEGB, int ROWB>
__device__ __forceinline__ void stage_image(unsigned char* img, const T* __restrict__ src, long long plane_step, int first,
                                            int planes, int nr, int nlon, int n4, int tid, int nthreads) {
    typedef T tvec __attribute__((ext_vector_type(R)));
    typedef IMG ivec __attribute__((ext_vector_type(PB)));
    constexpr int U = 2;
    const int total = nr * n4;
    const T* pl[PB];
#pragma unroll
    for (int b = 0; b < PB; ++b) pl[b] = src + (long long)min(first + b, planes - 1) * plane_step;
    for (int e0 = tid; e0 < total; e0 += U * nthreads) {
        tvec raw[U][PB];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = min(e0 + u * nthreads, total - 1);
            const int row = e / n4, q = e - row * n4;
#pragma unroll
            for (int b = 0; b < PB; ++b) raw[u][b] = *reinterpret_cast<const tvec*>(pl[b] + (long long)row * nlon + q * R);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * nthreads;
            if (e < total) {
                const int row = e / n4, q = e - row * n4;
                unsigned char* base = img + row * ROWB + q * (PB * (int)sizeof(IMG));
#pragma unroll
                for (int m = 0; m < R; ++m) {
                    ivec sv;
#pragma unroll
                    for (int b = 0; b < PB; ++b) sv[b] = img_cvt<IMG, T>(raw[u][b][m]);
                    *reinterpret_cast<ivec*>(base + m * SEGB) = sv;
                    if (q == 0) *reinterpret_cast<ivec*>(base + m * SEGB + n4 * (PB * (int)sizeof(IMG))) = sv;   // duplicate slot
                }
            }
        }
    }
}

// the taps of one run: groups of R filter values against the lane's sliding window
template <typename IMG, int PB, int R, int SEGB, int BM>
__device__ __forceinline__ void run_body(typename FV<PB>::type (&acc)[R], const float* __restrict__ vp, int ng, lds_cb* rowp,
                                         unsigned laneB, unsigned n4B, unsigned bqB) {
    typedef typename FV<PB>::type fv;
    constexpr int SLOT = PB * (int)sizeof(IMG);
    unsigned s = laneB + bqB;
    unsigned tB = min(s, s - n4B);
    auto advance = [&]() {
        s = tB + SLOT;
        tB = min(s, s - n4B);
    };
    auto fetch = [&](fv (&blk)[R]) {
#pragma unroll
        for (int j = 0; j < R; ++j) blk[j] = load_slot<IMG, PB>(rowp + tB + ((BM + j) % R) * SEGB + ((BM + j) / R) * SLOT);
    };
    // Three register blocks of R row elements rotate through the roles (current, next, the one after): group g multiplies
    // with blocks g and g + 1, which were requested during group g - 1 at the latest, and requests block g + 2 and the
    // filter values of group g + 1 before its first multiply-add, so that one wait per group finds everything a whole
    // group old.  Reads behind the end of the run hit valid image slots (the longitude wraps) and the value array ends with
    // R zeros.
    fv a[R], b[R], c[R];
    float va[R], vb[R];
    fetch(a);
    advance();
    fetch(b);
#pragma unroll
    for (int tau = 0; tau < R; ++tau) va[tau] = vp[tau];
    auto group = [&](const fv (&cur)[R], const fv (&nxt)[R], fv (&fut)[R], const float (&v)[R], float (&vn)[R],
                     const float* __restrict__ vnext) {
        advance();
        fetch(fut);
#pragma unroll
        for (int tau = 0; tau < R; ++tau) vn[tau] = vnext[tau];
#pragma unroll
        for (int tau = 0; tau < R; ++tau)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const fv xe = (tau + r < R) ? cur[tau + r] : nxt[tau + r - R];
                acc[r] = __builtin_elementwise_fma(fv(v[tau]), xe, acc[r]);
            }
    };
    int g = 0;
    for (; g + 6 <= ng; g += 6) {
        group(a, b, c, va, vb, vp + (g + 1) * R);
        group(b, c, a, vb, va, vp + (g + 2) * R);
        group(c, a, b, va, vb, vp + (g + 3) * R);
        group(a, b, c, vb, va, vp + (g + 4) * R);
        group(b, c, a, va, vb, vp + (g + 5) * R);
        group(c, a, b, vb, va, vp + (g + 6) * R);
    }
    // tail: up to five groups (the value roles alternate with the parity of the group, the blocks with g mod 3)
    if (g < ng) group(a, b, c, va, vb, vp + (g + 1) * R);
    if (g + 1 < ng) group(b, c, a, vb, va, vp + (g + 2) * R);
    if (g + 2 < ng) group(c, a, b, va, vb, vp + (g + 3) * R);
    if (g + 3 < ng) group(a, b, c, vb, va, vp + (g + 4) * R);
    if (g + 4 < ng) group(b, c, a, va, vb, vp + (g + 5) * R);
}

template <typename IMG, int PB, int R, int SEGB, int ROWB>
__device__ __forceinline__ void walk_runs(typename FV<PB>::type (&acc)[R], const i32x4* __restrict__ runs, int r0, int r1,
                                          const float* __restrict__ vals, lds_cb* img, unsigned laneB, unsigned n4B) {
    constexpr int SLOT = PB * (int)sizeof(IMG);
    for (int r = r0; r < r1; ++r) {
        const i32x4 h = runs[r];                                   // wave-uniform: scalar loads
        const int row = __builtin_amdgcn_readfirstlane(h[0]), js = __builtin_amdgcn_readfirstlane(h[1]);
        const int voff = __builtin_amdgcn_readfirstlane(h[2]), ng = __builtin_amdgcn_readfirstlane(h[3]);
        lds_cb* rowp = img + row * ROWB;
        const float* vp = vals + voff;
        const unsigned bqB = (unsigned)(js / R) * SLOT;
#define MK_RUN(BM_) run_body<IMG, PB, R, SEGB, BM_>(acc, vp, ng, rowp, laneB, n4B, bqB)
        if constexpr (R == 4) {
            switch (js & 3) {
                case 0: MK_RUN(0); break;
                case 1: MK_RUN(1); break;
                case 2: MK_RUN(2); break;
                default: MK_RUN(3); break;
            }
        } else {
            switch (js & 7) {
                case 0: MK_RUN(0); break;
                case 1: MK_RUN(1); break;
                case 2: MK_RUN(2); break;
                case 3: MK_RUN(3); break;
                case 4: MK_RUN(4); break;
                case 5: MK_RUN(5); break;
                case 6: MK_RUN(6); break;
                default: MK_RUN(7); break;
            }
        }
#undef MK_RUN
    }
}

// R consecutive outputs of plane b
template <typename T, int PB, int R>
__device__ __forceinline__ void store_outputs(T* dst, const typename FV<PB>::type (&acc)[R], int b) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int r0 = 0; r0 < R; r0 += 4) {
            const f32x4 v = {acc[r0][b], acc[r0 + 1][b], acc[r0 + 2][b], acc[r0 + 3][b]};
            *reinterpret_cast<f32x4*>(dst + r0) = v;
        }
    } else {
#pragma unroll
        for (int r0 = 0; r0 < R; r0 += 4) {
            u32x2 v;
            v[0] = pack_bf16x2(acc[r0][b], acc[r0 + 1][b]);
            v[1] = pack_bf16x2(acc[r0 + 2][b], acc[r0 + 3][b]);
            *reinterpret_cast<u32x2*>(dst + r0) = v;
        }
    }
}

// WV waves; work items (k, longitude segment of 64 lanes), id = k * NW + segment, dealt round-robin: with K = 9, NW = 3 and
// WV = 12 the four SIMDs of the CU (wave w runs on SIMD w % 4) get 7, 7, 7 and 6 items
template <typename T, typename IMG, int PB, int R, int NW, int WV>
__global__ __launch_bounds__(64 * WV) void disco_runs_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                                 const int* __restrict__ seg_off, const i32x4* __restrict__ runs,
                                                                 const float* __restrict__ vals, const int* __restrict__ lat_lo,
                                                                 const int* __restrict__ lat_n, int planes, int K, int nlat_in,
                                                                 int nlon, int nlat_out) {
    typedef typename FV<PB>::type fv;
    constexpr int SLOT = PB * (int)sizeof(IMG), SEGB = (64 * NW + 1) * SLOT, ROWB = R * SEGB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_r[];
    const int t = blockIdx.x, p0 = blockIdx.y * PB, tid = threadIdx.x;
    const int n4 = nlon / R;
    const int lo = lat_lo[t], nr = lat_n[t];
    stage_image<T, IMG, PB, R, SEGB, ROWB>(smem_r, x + (long long)lo * nlon, (long long)nlat_in * nlon, p0, planes, nr, nlon, n4,
                                           tid, 64 * WV);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const long long plane_out = (long long)nlat_out * nlon;
    for (int id = wave; id < K * NW; id += WV) {
        const int k = id / NW, wl = id - k * NW;
        const int slot = wl * 64 + lane;
        const unsigned laneB = (unsigned)min(slot, n4 - 1) * SLOT;
        fv acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = fv(0.f);
        walk_runs<IMG, PB, R, SEGB, ROWB>(acc, runs, seg_off[t * K + k], seg_off[t * K + k + 1], vals, (lds_cb*)smem_r, laneB,
                                          (unsigned)n4 * SLOT);
        if (slot < n4) {
#pragma unroll
            for (int b = 0; b < PB; ++b)
                if (p0 + b < planes)
                    store_outputs<T, PB, R>(y + ((long long)(p0 + b) * K + k) * plane_out + (long long)t * nlon + slot * R, acc, b);
        }
    }
}

template <typename T, typename IMG, int PB, int R, int NW, int LG>
__global__ __launch_bounds__(64 * NW * LG) void disco_runs_bwd_kernel(const T* __restrict__ gy, T* __restrict__ gx,
                                                                      const int* __restrict__ seg_off, const i32x4* __restrict__ runs,
                                                                      const float* __restrict__ vals, const int* __restrict__ t_lo,
                                                                      const int* __restrict__ t_n, int planes, int K, int nlat_in,
                                                                      int nlon, int nlat_out) {
    typedef typename FV<PB>::type fv;
    constexpr int SLOT = PB * (int)sizeof(IMG), SEGB = (64 * NW + 1) * SLOT, ROWB = R * SEGB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_r[];
    const int ig = blockIdx.x, p0 = blockIdx.y * PB, tid = threadIdx.x;
    const int n4 = nlon / R;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lg = wave / NW, wl = wave % NW;
    const int i = ig * LG + lg;
    const int slot = wl * 64 + lane;
    const unsigned laneB = (unsigned)min(slot, n4 - 1) * SLOT;
    const long long plane_out = (long long)nlat_out * nlon;
    fv acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = fv(0.f);
    for (int k = 0; k < K; ++k) {
        const int lo = t_lo[ig * K + k], nr = t_n[ig * K + k];
        if (nr == 0) continue;                                    // uniform over the workgroup
        __syncthreads();                                          // the previous basis function's rows are consumed
        stage_image<T, IMG, PB, R, SEGB, ROWB>(smem_r, gy + (long long)k * plane_out + (long long)lo * nlon, (long long)K * plane_out,
                                               p0, planes, nr, nlon, n4, tid, 64 * NW * LG);
        __syncthreads();
        if (i < nlat_in)
            walk_runs<IMG, PB, R, SEGB, ROWB>(acc, runs, seg_off[i * K + k], seg_off[i * K + k + 1], vals, (lds_cb*)smem_r, laneB,
                                              (unsigned)n4 * SLOT);
    }
    if (i < nlat_in && slot < n4) {
#pragma unroll
        for (int b = 0; b < PB; ++b)
            if (p0 + b < planes) store_outputs<T, PB, R>(gx + ((long long)(p0 + b) * nlat_in + i) * nlon + slot * R, acc, b);
    }
}


// ---- forward, all K basis functions at once ---------------------------------------------------------------------------
// For a fixed (output latitude, input row) the K filters live on the same longitude interval (the support is the disc, the
// basis functions differ in their values), so the window of row elements is shared: one stream per (t, row) whose groups
// carry K x R filter values.  R = 4 row elements fetched per group now feed K * R * R * PB multiply-adds (K = 9, PB = 2: 144
// packed FMAs per four 8-byte LDS reads) and the per-run costs (header, window fill) are paid once instead of K times.
// Runs are aligned to multiples of R by the host (zero taps in front), so every block is "the R classes of one slot" and the
// kernel needs neither the class switch nor the duplicate slot.  Workgroup = (LG consecutive output latitudes, PB planes):
// wave group lg of NW waves walks latitude tg * LG + lg; the row image is the union of what the LG latitudes touch.
template <typename T, int PB, int K, int NW, int LG>
__global__ __launch_bounds__(64 * NW * LG) void disco_fused_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                                       const int* __restrict__ seg_off, const i32x4* __restrict__ runs,
                                                                       const float* __restrict__ vals, const int* __restrict__ lat_lo,
                                                                       const int* __restrict__ lat_n, int planes, int nlat_in, int nlon,
                                                                       int nlat_out) {
    typedef typename FV<PB>::type fv;
    constexpr int R = 4;
    constexpr int SLOT = PB * 4, SEGB = (64 * NW + 1) * SLOT, ROWB = R * SEGB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_r[];
    const int tg = blockIdx.x, p0 = blockIdx.y * PB, tid = threadIdx.x;
    const int n4 = nlon / R;
    const int lo = lat_lo[tg], nr = lat_n[tg];
    stage_image<T, float, PB, R, SEGB, ROWB>(smem_r, x + (long long)lo * nlon, (long long)nlat_in * nlon, p0, planes, nr, nlon, n4,
                                             tid, 64 * NW * LG);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lg = wave / NW, wl = wave - lg * NW;
    const int t = tg * LG + lg;
    if (t >= nlat_out) return;
    const int slot = wl * 64 + lane;
    const unsigned laneB = (unsigned)min(slot, n4 - 1) * SLOT, n4B = (unsigned)n4 * SLOT;
    lds_cb* img = (lds_cb*)smem_r;
    fv acc[K][R];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[k][r] = fv(0.f);
    const int r0 = seg_off[t], r1 = seg_off[t + 1];
    i32x4 hnext = runs[min(r0, r1 - 1 < r0 ? r0 : r1 - 1)];
    for (int rr = r0; rr < r1; ++rr) {
        const i32x4 h = hnext;                                     // {row, first slot (js / R), value offset, groups}
        hnext = runs[min(rr + 1, r1 - 1)];                         // the next header travels while this run is multiplied
        lds_cb* rowp = img + h[0] * ROWB;
        const float* vp = vals + h[2];
        const int ng = h[3];
        unsigned s = laneB + (unsigned)h[1] * SLOT;
        unsigned tB = min(s, s - n4B);
        fv cur[R], nxt[R];
#pragma unroll
        for (int j = 0; j < R; ++j) cur[j] = load_slot<float, PB>(rowp + tB + j * SEGB);
        s = tB + SLOT;
        tB = min(s, s - n4B);
#pragma unroll
        for (int j = 0; j < R; ++j) nxt[j] = load_slot<float, PB>(rowp + tB + j * SEGB);
        for (int g = 0; g < ng; ++g) {
            fv fut[R];
            s = tB + SLOT;
            tB = min(s, s - n4B);
#pragma unroll
            for (int j = 0; j < R; ++j) fut[j] = load_slot<float, PB>(rowp + tB + j * SEGB);     // block g + 2
            const float* v = vp + g * (K * R);
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int tau = 0; tau < R; ++tau) {
                    const float val = v[k * R + tau];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const fv xe = (tau + r < R) ? cur[tau + r] : nxt[tau + r - R];
                        acc[k][r] = __builtin_elementwise_fma(fv(val), xe, acc[k][r]);
                    }
                }
#pragma unroll
            for (int j = 0; j < R; ++j) {
                cur[j] = nxt[j];
                nxt[j] = fut[j];
            }
        }
    }
    if (slot < n4) {
        const long long plane_out = (long long)nlat_out * nlon;
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int b = 0; b < PB; ++b)
                if (p0 + b < planes)
                    store_outputs<T, PB, R>(y + ((long long)(p0 + b) * K + k) * plane_out + (long long)t * nlon + slot * R, acc[k], b);
    }
}

// (NW, LG) of the fused forward kernel for a longitude count; false when it is not instantiated
inline bool fused_shape(int nlon, int& NW, int& LG) {
    if (nlon % 4) return false;
    const int lanes = nlon / 4;
    NW = (lanes + 63) / 64;
    if (lanes < 2 || !(NW <= 3 || NW == 6)) return false;
    LG = NW == 6 ? 2 : 4;
    return true;
}

constexpr size_t RUNS_LDS_CAP = 160 * 1024;       // the whole LDS of a CU: such a workgroup runs alone on it

template <typename KERN, typename... Args>
int runs_launch(KERN kern, dim3 grid, int threads, size_t lds, hipStream_t s, const char* what, Args... args) {
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(threads), lds, s, args...);
    return mk_check_launch(what);
}

// (R, NW) for a longitude count: N / R lanes in NW waves; 0 when the run form does not apply
inline bool runs_shape(int nlon, int& R, int& NW) {
    if (nlon % 4 == 0) {                          // R = 4: up to three waves per latitude circle, or six (1284 .. 1536 longitudes)
        const int lanes = nlon / 4;
        const int nw = (lanes + 63) / 64;
        if (lanes >= 2 && (nw <= 3 || nw == 6)) {
            R = 4;
            NW = nw;
            return true;
        }
    }
    if (nlon % 8 == 0) {                          // R = 8 with three waves covers 1032 .. 1536 longitudes
        const int lanes = nlon / 8;
        if (lanes > 128 && lanes <= 192) {
            R = 8;
            NW = 3;
            return true;
        }
    }
    return false;
}

inline int mk_runs_radix(int nlon) {
    int R, NW;
    return runs_shape(nlon, R, NW) ? R : 0;
}

template <int PB, int R, int NW>
constexpr size_t row_bytes(size_t elem) { return (size_t)R * (64 * NW + 1) * PB * elem; }

}  // namespace

// 1 when the run-form kernels take this shape (the host then builds run lists), 0 otherwise
extern "C" int mk_disco_runs_shape(int nlon, int max_rows, int planes, int dtype, int img_bf16, int* R_out, int* PB_out) {
    int R, NW;
    if (!runs_shape(nlon, R, NW) || planes < 2) return 0;
    const size_t elem = (dtype == MK_BF16 && img_bf16) ? 2 : 4;
    const int want = (PB_out && (*PB_out == 2 || *PB_out == 4)) ? *PB_out : 0;      // a preset *PB_out asks for exactly that
    for (int pb : {4, 2}) {
        if (planes < pb || (want && pb != want)) continue;
        if ((size_t)max_rows * R * (64 * NW + 1) * pb * elem <= RUNS_LDS_CAP) {
            if (R_out) *R_out = R;
            if (PB_out) *PB_out = pb;
            return 1;
        }
    }
    return 0;
}

#define MK_RUNS_DISPATCH(KERNEL, T, IMG, GROUPS, GROUPS8, what, ...)                                                                  \
    do {                                                                                                                      \
        const size_t lds = (size_t)max_rows * R * (64 * NW + 1) * PB * sizeof(IMG);                                           \
        const int g_ = R == 8 ? GROUPS8 : GROUPS;                                                                             \
        const int threads = fixed_waves ? 64 * g_ : 64 * NW * g_;                                                             \
        if (PB == 4 && R == 4 && NW == 1) return runs_launch(KERNEL<T, IMG, 4, 4, 1, GROUPS>, grid, threads, lds, s, what, __VA_ARGS__); \
        if (PB == 4 && R == 4 && NW == 2) return runs_launch(KERNEL<T, IMG, 4, 4, 2, GROUPS>, grid, threads, lds, s, what, __VA_ARGS__); \
        if (PB == 4 && R == 4 && NW == 3) return runs_launch(KERNEL<T, IMG, 4, 4, 3, GROUPS>, grid, threads, lds, s, what, __VA_ARGS__); \
        if (PB == 4 && R == 4 && NW == 6) return runs_launch(KERNEL<T, IMG, 4, 4, 6, GROUPS>, grid, threads, lds, s, what, __VA_ARGS__); \
        if (PB == 2 && R == 4 && NW == 6) return runs_launch(KERNEL<T, IMG, 2, 4, 6, GROUPS>, grid, threads, lds, s, what, __VA_ARGS__); \
        if (PB == 4 && R == 8 && NW == 3) return runs_launch(KERNEL<T, IMG, 4, 8, 3, GROUPS8>, grid, threads, lds, s, what, __VA_ARGS__); \
        if (PB == 2 && R == 4 && NW == 1) return runs_launch(KERNEL<T, IMG, 2, 4, 1, GROUPS>, grid, threads, lds, s, what, __VA_ARGS__); \
        if (PB == 2 && R == 4 && NW == 2) return runs_launch(KERNEL<T, IMG, 2, 4, 2, GROUPS>, grid, threads, lds, s, what, __VA_ARGS__); \
        if (PB == 2 && R == 4 && NW == 3) return runs_launch(KERNEL<T, IMG, 2, 4, 3, GROUPS>, grid, threads, lds, s, what, __VA_ARGS__); \
        if (PB == 2 && R == 8 && NW == 3) return runs_launch(KERNEL<T, IMG, 2, 8, 3, GROUPS8>, grid, threads, lds, s, what, __VA_ARGS__); \
        MK_REQUIRE(false, "%s: no kernel for PB=%d R=%d NW=%d", what, PB, R, NW);                                            \
    } while (0)


// 1 when the fused forward kernel (all K basis functions per stream) takes this shape; *LG_out = output latitudes per workgroup
// (the lists are built for it), *PB_out = planes per workgroup
extern "C" int mk_disco_fused_shape(int nlon, int K, int max_rows, int planes, int* LG_out, int* PB_out) {
    int NW, LG;
    if (K != 9 || planes < 2 || !fused_shape(nlon, NW, LG)) return 0;
    if ((size_t)max_rows * 4 * (64 * NW + 1) * 2 * 4 > RUNS_LDS_CAP) return 0;
    if (LG_out) *LG_out = LG;
    if (PB_out) *PB_out = 2;
    return 1;
}

// forward, all K = 9 basis functions per stream: seg_off (nlat_out + 1) indexes runs (n, 4) = {image row relative to
// lat_lo[t / LG], first slot = first longitude / 4, value offset, groups}; vals: per group K x 4 values ([k][tau]), runs
// aligned to multiples of 4 longitudes; lat_lo / lat_n per latitude GROUP (max_rows = their maximum row count)
extern "C" int mk_disco_fwd_fused(const void* x, void* y, int dtype, const int* seg_off, const int* runs, const float* vals,
                                  const int* lat_lo, const int* lat_n, int max_rows, int planes, int K, int nlat_in, int nlon,
                                  int nlat_out, void* stream) {
    MK_REQUIRE(x && y && seg_off && runs && vals && lat_lo && lat_n, "disco_fwd_fused: null pointer");
    int NW, LG;
    MK_REQUIRE(K == 9 && fused_shape(nlon, NW, LG), "disco_fwd_fused: K = 9 and a covered longitude count (got K = %d, %d)", K, nlon);
    constexpr int PB = 2;
    MK_REQUIRE(planes >= PB && (planes + PB - 1) / PB <= 65535, "disco_fwd_fused: bad plane count");
    const size_t lds = (size_t)max_rows * 4 * (64 * NW + 1) * PB * 4;
    MK_REQUIRE(lds <= RUNS_LDS_CAP, "disco_fwd_fused: %d rows do not fit the LDS", max_rows);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((nlat_out + LG - 1) / LG, (planes + PB - 1) / PB);
    const i32x4* rn = (const i32x4*)runs;
    const int threads = 64 * NW * LG;
#define MK_FUSED_GO(T, NW_, LG_)                                                                                              \
    return runs_launch(disco_fused_fwd_kernel<T, PB, 9, NW_, LG_>, grid, threads, lds, s, "mk_disco_fwd_fused", (const T*)x, (T*)y, \
                       seg_off, rn, vals, lat_lo, lat_n, planes, nlat_in, nlon, nlat_out)
    if (dtype == MK_F32) {
        if (NW == 1) MK_FUSED_GO(float, 1, 4);
        if (NW == 2) MK_FUSED_GO(float, 2, 4);
        if (NW == 3) MK_FUSED_GO(float, 3, 4);
        MK_FUSED_GO(float, 6, 2);
    }
    if (NW == 1) MK_FUSED_GO(u16, 1, 4);
    if (NW == 2) MK_FUSED_GO(u16, 2, 4);
    if (NW == 3) MK_FUSED_GO(u16, 3, 4);
    MK_FUSED_GO(u16, 6, 2);
#undef MK_FUSED_GO
}

// forward, run form.  seg_off (nlat_out * K + 1), runs (n, 4) = {row, first lon, value offset, groups}, vals: see the header.
// PB / R as returned by mk_disco_runs_shape (the lists are built for that R); img_bf16: keep bf16 tensors as bf16 in LDS.
extern "C" int mk_disco_fwd_runs(const void* x, void* y, int dtype, const int* seg_off, const int* runs, const float* vals,
                                 const int* lat_lo, const int* lat_n, int max_rows, int planes, int K, int nlat_in, int nlon,
                                 int nlat_out, int R, int PB, int img_bf16, void* stream) {
    MK_REQUIRE(x && y && seg_off && runs && vals && lat_lo && lat_n, "disco_fwd_runs: null pointer");
    int R2, NW;
    MK_REQUIRE(runs_shape(nlon, R2, NW) && R2 == R, "disco_fwd_runs: %d longitudes are not covered by R = %d", nlon, R);
    MK_REQUIRE(mk_runs_radix(nlon) == R, "disco_fwd_runs: lists built for R = %d", R);
    MK_REQUIRE(planes >= PB && (PB == 2 || PB == 4) && K > 0 && (planes + PB - 1) / PB <= 65535, "disco_fwd_runs: bad plane count");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(nlat_out, (planes + PB - 1) / PB);
    const i32x4* rn = (const i32x4*)runs;
    constexpr int KG = 12, KG8 = 8;              // waves per workgroup (R = 8 holds three blocks of 8 x PB row elements: 256 registers)
    const bool fixed_waves = true;
    if (dtype == MK_F32)
        MK_RUNS_DISPATCH(disco_runs_fwd_kernel, float, float, KG, KG8, "mk_disco_fwd_runs", (const float*)x, (float*)y, seg_off, rn, vals,
                         lat_lo, lat_n, planes, K, nlat_in, nlon, nlat_out);
    if (img_bf16)
        MK_RUNS_DISPATCH(disco_runs_fwd_kernel, u16, u16, KG, KG8, "mk_disco_fwd_runs", (const u16*)x, (u16*)y, seg_off, rn, vals, lat_lo,
                         lat_n, planes, K, nlat_in, nlon, nlat_out);
    MK_RUNS_DISPATCH(disco_runs_fwd_kernel, u16, float, KG, KG8, "mk_disco_fwd_runs", (const u16*)x, (u16*)y, seg_off, rn, vals, lat_lo,
                     lat_n, planes, K, nlat_in, nlon, nlat_out);
    return -1;
}

// adjoint, run form: segments per (input latitude, k), rows relative to t_lo[(i / lat_group) * K + k] (lat_group = 2 or 4
// consecutive latitudes share the staged image of a basis function; 4 x three waves fill the four SIMDs evenly)
extern "C" int mk_disco_bwd_runs(const void* gy, void* gx, int dtype, const int* seg_off, const int* runs, const float* vals,
                                 const int* t_lo, const int* t_n, int max_rows, int planes, int K, int nlat_in, int nlon,
                                 int nlat_out, int R, int PB, int img_bf16, int lat_group, void* stream) {
    MK_REQUIRE(gy && gx && seg_off && runs && vals && t_lo && t_n, "disco_bwd_runs: null pointer");
    int R2, NW;
    MK_REQUIRE(runs_shape(nlon, R2, NW) && R2 == R, "disco_bwd_runs: %d longitudes are not covered by R = %d", nlon, R);
    MK_REQUIRE(mk_runs_radix(nlon) == R, "disco_bwd_runs: lists built for R = %d", R);
    MK_REQUIRE(planes >= PB && (PB == 2 || PB == 4) && K > 0 && (planes + PB - 1) / PB <= 65535, "disco_bwd_runs: bad plane count");
    hipStream_t s = (hipStream_t)stream;
    MK_REQUIRE(lat_group == 2 || lat_group == 4, "disco_bwd_runs: latitude groups of 2 or 4");
    MK_REQUIRE(NW * lat_group <= 12, "disco_bwd_runs: %d waves per latitude x %d latitudes exceed a workgroup", NW, lat_group);
    const bool fixed_waves = false;
    const dim3 grid((nlat_in + lat_group - 1) / lat_group, (planes + PB - 1) / PB);
    const i32x4* rn = (const i32x4*)runs;
#define MK_BWD_GO(LG)                                                                                                          \
    do {                                                                                                                        \
        if (dtype == MK_F32)                                                                                                    \
            MK_RUNS_DISPATCH(disco_runs_bwd_kernel, float, float, LG, LG, "mk_disco_bwd_runs", (const float*)gy, (float*)gx, seg_off, \
                             rn, vals, t_lo, t_n, planes, K, nlat_in, nlon, nlat_out);                                          \
        if (img_bf16)                                                                                                           \
            MK_RUNS_DISPATCH(disco_runs_bwd_kernel, u16, u16, LG, LG, "mk_disco_bwd_runs", (const u16*)gy, (u16*)gx, seg_off, rn, \
                             vals, t_lo, t_n, planes, K, nlat_in, nlon, nlat_out);                                              \
        MK_RUNS_DISPATCH(disco_runs_bwd_kernel, u16, float, LG, LG, "mk_disco_bwd_runs", (const u16*)gy, (u16*)gx, seg_off, rn,   \
                         vals, t_lo, t_n, planes, K, nlat_in, nlon, nlat_out);                                                  \
    } while (0)
    if (lat_group == 4) MK_BWD_GO(4);
    MK_BWD_GO(2);
#undef MK_BWD_GO
    return -1;
}
