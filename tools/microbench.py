"""Per-kernel microbenchmarks at the BASELINE shapes (HIP events on the launch stream).
   python tools/microbench.py [fft] [legendre] [dhconv] [pointwise]"""
import os, sys, time, math, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from makani_amd import ops
import makani_amd as ma

dev = torch.device("cuda:0")


def timeit(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def fft():
    C = 384
    for nlat, nlon, mmax in ((721, 1440, 241), (240, 480, 241)):
        c = 2 * math.pi / nlon
        for dt in (torch.float32, torch.bfloat16):
            x = torch.rand(1, C, nlat, nlon, device=dev).to(dt)
            ms = timeit(lambda: ops.rfft_rows(x, mmax, C, (c, c, c)))
            nb = C * nlat * (nlon * x.element_size() + mmax * 8)
            print(f"rfft  {nlat}x{nlon} {str(dt)[6:]:9s} {ms:8.3f} ms  {nb/ms/1e6:8.1f} GB/s  ({nb/1e6:.0f} MB)")
            F = ops.rfft_rows(x, mmax, C, (c, c, c))
            ms = timeit(lambda: ops.irfft_rows(F, 1, C, nlon, dt, (1.0, 2.0, 1.0)))
            print(f"irfft {nlat}x{nlon} {str(dt)[6:]:9s} {ms:8.3f} ms  {nb/ms/1e6:8.1f} GB/s")
            del x, F


def legendre():
    C = 384
    for nlat, nlon, grid in ((721, 1440, "equiangular"), (240, 480, "legendre-gauss")):
        S = ma.RealSHT(nlat, nlon, lmax=240, mmax=241, grid=grid).to(dev)
        I = ma.InverseRealSHT(nlat, nlon, lmax=240, mmax=241, grid=grid).to(dev)
        F = torch.randn(241, nlat, 2, C, device=dev)
        Sc = torch.randn(240, 241, 2, C, device=dev)
        fl = 4.0 * C * nlat * 240 * 241

        def run(gen):
            a = ops.legendre_analysis(F, S.weights_t, 240)
            b = ops.legendre_synthesis(Sc, I.pct, nlat)
            ms = timeit(lambda: ops.legendre_analysis(F, S.weights_t, 240))
            print(f"gen{gen} analysis  K={nlat}: {ms:7.3f} ms  {fl/ms/1e9:7.1f} TF dense-equiv")
            ms = timeit(lambda: ops.legendre_synthesis(Sc, I.pct, nlat))
            print(f"gen{gen} synthesis K={nlat}: {ms:7.3f} ms  {fl/ms/1e9:7.1f} TF dense-equiv")
            return a, b
        run("2")


def dhconv():
    C, L, M = 384, 240, 241
    S = torch.randn(L, M, 2, C, device=dev)
    G = torch.randn(L, M, 2, C, device=dev)
    w = ops.native_w_empty(C, C, L, dev)
    w.copy_(torch.randn(1, C, C, L, dtype=torch.complex64, device=dev))
    fl = 8.0 * C * C * L * M
    tri = (torch.arange(L, device=dev)[:, None] >= torch.arange(M, device=dev)[None, :])[:, :, None, None]

    def run(gen):
        y = ops.dhconv_fwd(S, w, 1, C)
        gs = ops.dhconv_dgrad(G, w, 1, C, C)
        gw = ops.dhconv_wgrad(S, G, 1, native=True)
        ms = timeit(lambda: ops.dhconv_fwd(S, w, 1, C)); print(f"gen{gen} dhconv fwd  : {ms:7.3f} ms {fl/ms/1e9:7.1f} TF dense-equiv")
        ms = timeit(lambda: ops.dhconv_dgrad(G, w, 1, C, C)); print(f"gen{gen} dhconv dgrad: {ms:7.3f} ms {fl/ms/1e9:7.1f} TF dense-equiv")
        ms = timeit(lambda: ops.dhconv_wgrad(S, G, 1, native=True)); print(f"gen{gen} dhconv wgrad: {ms:7.3f} ms {fl/ms/1e9:7.1f} TF dense-equiv")
        return y, gs, torch.view_as_real(gw)
    run("2")


def pointwise():
    for H, W in ((240, 480), (721, 1440)):
        x = torch.randn(1, 384, H, W, device=dev).bfloat16().requires_grad_(True)
        g = torch.ones(384, device=dev); b = torch.zeros(384, device=dev)
        nb = x.numel() * 2
        for gelu in (False, True):
            tag = "instnorm+gelu" if gelu else "instnorm     "
            ms = timeit(lambda: ops.InstanceNormFn.apply(x, g, b, 1e-6, gelu))
            print(f"{tag} fwd {H}x{W}: {ms:7.3f} ms  {3*nb/ms/1e6:7.1f} GB/s (2 reads + 1 write)")
            y = ops.InstanceNormFn.apply(x, g, b, 1e-6, gelu)
            gy = torch.randn_like(y)
            ms = timeit(lambda: torch.autograd.grad(y, x, gy, retain_graph=True))
            print(f"{tag} bwd {H}x{W}: {ms:7.3f} ms  {5*nb/ms/1e6:7.1f} GB/s (4 reads + 1 write)")
        ms = timeit(lambda: ops.BiasGeluFn.apply(x, b))
        print(f"bias_gelu fwd     {H}x{W}: {ms:7.3f} ms  {2*nb/ms/1e6:7.1f} GB/s")


def cold():
    """streaming kernels on ROTATING buffers (12 x 88 MB inputs: nothing is left in the 256 MB memory-side cache or
    the L2s from the previous call) — what the kernels see inside the train step"""
    NB, C, H, W = 12, 384, 240, 480
    xs = [torch.randn(1, C, H, W, device=dev).bfloat16() for _ in range(NB)]
    ys = [torch.empty_like(x) for x in xs]
    g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    nb = xs[0].numel() * 2
    it = [0]
    def rot(fn):
        def f():
            i = it[0] = (it[0] + 1) % NB
            fn(i)
        return f
    ms = timeit(rot(lambda i: torch.add(xs[i], xs[(i + 5) % NB], out=ys[i])), reps=24, warm=12)
    print(f"cold torch add bf16       : {ms*1e3:7.1f} us  {3*nb/ms/1e6:7.1f} GB/s")
    ms = timeit(rot(lambda i: ys[i].copy_(xs[i])), reps=24, warm=12)
    print(f"cold torch copy bf16      : {ms*1e3:7.1f} us  {2*nb/ms/1e6:7.1f} GB/s")
    ms = timeit(rot(lambda i: ops._sum_planes(xs[i])), reps=24, warm=12)
    print(f"cold plane sums           : {ms*1e3:7.1f} us  {nb/ms/1e6:7.1f} GB/s")
    ms = timeit(rot(lambda i: xs[i].sum(dim=(0, 2, 3), dtype=torch.float32)), reps=24, warm=12)
    print(f"cold torch sum(0,2,3)     : {ms*1e3:7.1f} us  {nb/ms/1e6:7.1f} GB/s")
    for gelu in (False, True):
        ms = timeit(rot(lambda i: ops.InstanceNormFn.apply(xs[i], g, b, 1e-6, gelu)), reps=24, warm=12)
        print(f"cold instnorm fwd gelu={int(gelu)}  : {ms*1e3:7.1f} us  {3*nb/ms/1e6:7.1f} GB/s (2 reads + 1 write)")
    pbias = torch.randn(C, device=dev) * 0.1
    ms = timeit(rot(lambda i: ops.InstanceNormFn.apply(xs[i], g, b, 1e-6, False, pbias)), reps=24, warm=12)
    print(f"cold instnorm fwd pre_bias: {ms*1e3:7.1f} us  {3*nb/ms/1e6:7.1f} GB/s (2 reads + 1 write)")
    xr = [x.clone().requires_grad_(True) for x in xs]
    outs = [ops.InstanceNormFn.apply(x, g, b, 1e-6, False, pbias) for x in xr]
    ms = timeit(rot(lambda i: torch.autograd.grad(outs[i], xr[i], ys[i], retain_graph=True)), reps=24, warm=12)
    print(f"cold instnorm bwd pre_bias: {ms*1e3:7.1f} us  {5*nb/ms/1e6:7.1f} GB/s (4 reads + 1 write)")
    del outs
    for gelu in (False, True):
        outs = [ops.InstanceNormFn.apply(x, g, b, 1e-6, gelu) for x in xr]
        ms = timeit(rot(lambda i: torch.autograd.grad(outs[i], xr[i], ys[i], retain_graph=True)), reps=24, warm=12)
        print(f"cold instnorm bwd gelu={int(gelu)}  : {ms*1e3:7.1f} us  {5*nb/ms/1e6:7.1f} GB/s (4 reads + 1 write)")
        del outs
    c = 2 * math.pi / W
    Fs = [ops.rfft_rows(xs[i], 241, C, (c, c, c)) for i in range(NB)]
    nbf = C * H * (W * 2 + 241 * 8)
    ms = timeit(rot(lambda i: ops.rfft_rows(xs[i], 241, C, (c, c, c))), reps=24, warm=12)
    print(f"cold rfft 240x480 bf16    : {ms*1e3:7.1f} us  {nbf/ms/1e6:7.1f} GB/s")
    ms = timeit(rot(lambda i: ops.irfft_rows(Fs[i], 1, C, W, torch.bfloat16, (1.0, 2.0, 1.0))), reps=24, warm=12)
    print(f"cold irfft 240x480 bf16   : {ms*1e3:7.1f} us  {nbf/ms/1e6:7.1f} GB/s")
    ms = timeit(lambda: ops.rfft_rows(xs[0], 241, C, (c, c, c)), reps=24, warm=4)
    print(f"warm rfft 240x480 bf16    : {ms*1e3:7.1f} us  {nbf/ms/1e6:7.1f} GB/s")
    ms = timeit(lambda: ops.irfft_rows(Fs[0], 1, C, W, torch.bfloat16, (1.0, 2.0, 1.0)), reps=24, warm=4)
    print(f"warm irfft 240x480 bf16   : {ms*1e3:7.1f} us  {nbf/ms/1e6:7.1f} GB/s")


def conv():
    """forward / data-gradient channel GEMM: HIP kernel (MAKANI_AMD_CONV_NN=tile: the 128-row tile kernel instead of the
    ring kernel) against the library GEMM, plain and with the fused epilogues; each result checked against fp32"""
    per_step = {(768, 384, 721): 2, (384, 768, 721): 2, (384, 384, 721): 8, (768, 384, 240): 14, (384, 768, 240): 14, (384, 384, 240): 14}
    tot_hip = tot_lib = 0.0
    for (M, K, H, W) in ((768, 384, 721, 1440), (384, 768, 721, 1440), (384, 384, 721, 1440), (768, 384, 240, 480),
                         (384, 768, 240, 480), (384, 384, 240, 480), (73, 384, 721, 1440), (384, 73, 721, 1440)):
        torch.manual_seed(M + K)
        x = (torch.rand(1, K, H, W, device=dev) - 0.5).bfloat16()
        w = (torch.randn(M, K, device=dev) / K ** 0.5).bfloat16()
        bias = torch.randn(M, device=dev)
        A = ops.pad_weight_bf16(w)
        fl = 2.0 * M * K * H * W
        nb = 2.0 * H * W * (M + K)
        ref = torch.mm(w.float(), x.view(K, -1).float())
        y, _ = ops.conv1x1_nn(A, K, x)
        err = ((y.view(M, -1).float() - ref).norm() / ref.norm()).item()
        gsrc = torch.randn(1, M, H, W, device=dev).bfloat16()
        res = torch.randn(1, M, H, W, device=dev).bfloat16()
        y2, pre = ops.conv1x1_nn(A, K, x, bias=bias, act=True, want_pre=True, residual=res)
        pre_ref = (ref + bias[:, None]).bfloat16().float()
        ref2 = torch.nn.functional.gelu(pre_ref) + res.view(M, -1).float()
        err2 = ((y2.view(M, -1).float() - ref2).norm() / ref2.norm()).item()
        errp = ((pre.view(M, -1).float() - pre_ref).norm() / pre_ref.norm()).item()
        del ref2, pre_ref, y2, pre
        ms = timeit(lambda: ops.conv1x1_nn(A, K, x), reps=20, warm=3)
        ms_b = timeit(lambda: ops.conv1x1_nn(A, K, x, bias=bias, act=True, want_pre=True), reps=20, warm=3)
        ms_g = timeit(lambda: ops.conv1x1_nn(A, K, x, gelu_grad_of=gsrc), reps=20, warm=3)
        ms_r = timeit(lambda: ops.conv1x1_nn(A, K, x, residual=res), reps=20, warm=3)
        out = torch.empty(M, H * W, device=dev, dtype=torch.bfloat16)
        ms2 = timeit(lambda: torch.mm(w, x.view(K, H * W), out=out), reps=20, warm=3)
        n = per_step.get((M, K, H), 0)
        tot_hip += n * ms
        tot_lib += n * ms2
        print(f"conv M={M:3d} K={K:3d} {H}x{W}: hip {ms:7.3f} ms ({fl/ms/1e9:5.0f} TF, {nb/ms/1e6:5.0f} GB/s) | lib {ms2:7.3f} ms ({fl/ms2/1e9:5.0f} TF)"
              f" | +bias+gelu+pre {ms_b:6.3f}  *gelu'(G) {ms_g:6.3f}  +R {ms_r:6.3f} | rel err {err:.1e} fused {err2:.1e} pre {errp:.1e}")
        del x, gsrc, res, ref, out
    print(f"fwd+dgrad GEMMs of the step at these shapes: hip {tot_hip:.2f} ms, library {tot_lib:.2f} ms")


def wgrad():
    """channel-GEMM weight gradient: the nine shapes of the train step (+ ragged / batched ones), checked against an fp32
    GEMM of the same bf16 operands, then timed.  MAKANI_AMD_WGRAD=tile selects the 128 x 128 tile kernel."""
    shapes = [(768, 384, 1, 721, 1440), (384, 768, 1, 721, 1440), (384, 384, 1, 721, 1440), (384, 73, 1, 721, 1440),
              (73, 384, 1, 721, 1440), (73, 73, 1, 721, 1440), (768, 384, 1, 240, 480), (384, 768, 1, 240, 480),
              (384, 384, 1, 240, 480), (384, 200, 2, 91, 184), (300, 384, 1, 37, 72), (768, 384, 2, 60, 124)]
    per_step = {(768, 384, 721): 1, (384, 768, 721): 1, (384, 384, 721): 3, (384, 73, 721): 1, (73, 384, 721): 1, (73, 73, 721): 1,
                (768, 384, 240): 7, (384, 768, 240): 7, (384, 384, 240): 7}
    total = 0.0
    for (M, K, B, H, W) in shapes:
        torch.manual_seed(M + K + H)
        x = (torch.rand(B, K, H, W, device=dev) - 0.3).bfloat16()
        g = (torch.randn(B, M, H, W, device=dev) * 0.5).bfloat16()
        dW = ops.conv1x1_wgrad(g, x)
        ref = torch.einsum("bmn,bkn->mk", g.view(B, M, -1).float(), x.view(B, K, -1).float())
        err = ((dW - ref).norm() / ref.norm()).item()
        ms = timeit(lambda: ops.conv1x1_wgrad(g, x), reps=20, warm=3)
        nb = 2.0 * B * H * W * (M + K)
        n = per_step.get((M, K, H), 0)
        total += n * ms
        print(f"wgrad M={M:3d} K={K:3d} B={B} {H}x{W}: {ms:7.3f} ms  {nb/ms/1e6:7.0f} GB/s  {2.0*B*M*K*H*W/ms/1e9:6.0f} TF  rel err {err:.1e}"
              f"  {'x%d per step' % n if n else ''}")
        del x, g
    print(f"wgrad per step (29 launches): {total:.3f} ms  -> {16.72e3/total/1e3:.2f} TB/s algorithmic = {16.72e3/total/8e3:.3f} of 8 TB/s")


def spectral():
    """the HBM-streaming contractions (csrc/spectral_pointwise.hip): algorithmic bytes = activations in + out + weights once"""
    L, M = 240, 241
    for C in (64, 128):                       # diagonal: one C x C complex matrix per (l, m); 1.9 / 7.6 GB of weights
        S = torch.randn(L, M, 2, C, device=dev)
        w = torch.randn(1, C, C, L, M, dtype=torch.complex64, device=dev)
        wr = torch.view_as_real(w)
        T = torch.empty_like(S)
        gw = torch.empty_like(wr)
        nb = 8.0 * C * C * L * M + 2 * 8.0 * C * L * M
        live = 0.5 * 8.0 * C * C * L * M + 2 * 8.0 * C * L * M          # the l < m half of the weights is never read
        ms = timeit(lambda: ops.check(ops.lib().mk_spec_diag_apply(ops.ptr(S), ops.ptr(wr), ops.ptr(T), L, M, 1, C, C, C, C, 0, 0, 0, ops.stream())))
        print(f"diag fwd   C={C:3d}: {ms:7.3f} ms  {nb/ms/1e6:7.1f} GB/s dense  ({live/ms/1e6:7.1f} GB/s touched)")
        ms = timeit(lambda: ops.check(ops.lib().mk_spec_diag_apply(ops.ptr(S), ops.ptr(wr), ops.ptr(T), L, M, 1, C, C, C, C, 0, 0, 1, ops.stream())))
        print(f"diag dgrad C={C:3d}: {ms:7.3f} ms  {nb/ms/1e6:7.1f} GB/s dense  ({live/ms/1e6:7.1f} GB/s touched)")
        ms = timeit(lambda: ops.check(ops.lib().mk_spec_diag_wgrad(ops.ptr(S), ops.ptr(T), ops.ptr(gw), L, M, 1, C, C, C, C, 0, ops.stream())))
        print(f"diag wgrad C={C:3d}: {ms:7.3f} ms  {nb/ms/1e6:7.1f} GB/s (gradient written in full)")
        del S, w, wr, T, gw
    C = 384
    S = torch.randn(L, M, 2, C, device=dev)
    T = torch.empty_like(S)
    for Mw in (M, 1):
        Ws = torch.randn(L, Mw, 2, C, device=dev)
        gW = torch.empty_like(Ws)
        nb = 4.0 * (4 * C * L * M + 2 * C * L * Mw)
        ms = timeit(lambda: ops.check(ops.lib().mk_spec_sep_mul(ops.ptr(S), ops.ptr(Ws), ops.ptr(T), L, M, Mw, 1, C, 0, 0, ops.stream())))
        print(f"sep mul   Mw={Mw:3d}: {ms:7.3f} ms  {nb/ms/1e6:7.1f} GB/s dense")
        ms = timeit(lambda: ops.check(ops.lib().mk_spec_sep_wgrad(ops.ptr(S), ops.ptr(T), ops.ptr(gW), L, M, Mw, 1, C, 0, ops.stream())))
        print(f"sep wgrad Mw={Mw:3d}: {ms:7.3f} ms  {nb/ms/1e6:7.1f} GB/s dense")


def sht():
    """BASELINE's secondary metric "fwd SHT GB/s" (bench.py: sht_bandwidth): S1 ERA5-shaped (73 channels, full band),
    S2 model-shaped; plus the Legendre step of S1 alone (the narrow-operand form of the real split kernel, csrc/xgemm2.hip)"""
    for name, C, lmax, mmax in (("S1_c73_L721_M721", 73, 721, 721), ("S2_c384_L240_M241", 384, 240, 241)):
        S = ma.RealSHT(721, 1440, lmax=lmax, mmax=mmax, grid="equiangular").to(dev)
        x = torch.rand(1, C, 721, 1440, device=dev)
        ms = timeit(lambda: S(x), reps=10, warm=2)
        nb = C * 721 * 1440 * 4 + C * lmax * mmax * 8
        print(f"fwd SHT {name}: {ms:7.3f} ms  {nb/ms/1e6:7.1f} GB/s")
        if C == 73:
            R = ops.round4(C)
            F = torch.randn(mmax, 721, 2, R, device=dev)
            fl = 4.0 * C * 721 * lmax * mmax
            ms = timeit(lambda: ops.legendre_analysis(F, S.weights_t, lmax), reps=10, warm=2)
            print(f"   legendre analysis  C=73 full band: {ms:7.3f} ms  {fl/ms/1e9:7.1f} TF dense-equiv")
            c = 2 * math.pi / 1440
            ms = timeit(lambda: ops.rfft_rows(x, mmax, R, (c, c, c)), reps=10, warm=2)
            print(f"   rfft 1440 fp32 C=73 full spectrum : {ms:7.3f} ms")
        del S, x


if __name__ == "__main__":
    which = sys.argv[1:] or ["fft", "legendre", "dhconv", "pointwise"]
    for w in which:
        globals()[w]()
