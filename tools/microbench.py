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
        ms = timeit(lambda: ops.legendre_analysis(F, S.weights_t, 240))
        fl = 4.0 * C * nlat * 240 * 241
        print(f"analysis  K={nlat}: {ms:7.3f} ms  {fl/ms/1e9:7.1f} TF dense-equiv")
        Sc = torch.randn(240, 241, 2, C, device=dev)
        ms = timeit(lambda: ops.legendre_synthesis(Sc, I.pct, nlat))
        print(f"synthesis K={nlat}: {ms:7.3f} ms  {fl/ms/1e9:7.1f} TF dense-equiv")


def dhconv():
    C, L, M = 384, 240, 241
    S = torch.randn(L, M, 2, C, device=dev)
    w = torch.randn(1, C, C, L, dtype=torch.complex64, device=dev)
    W = ops.weight_to_wlayout(w)
    fl = 8.0 * C * C * L * M
    ms = timeit(lambda: ops.dhconv_fwd(S, W, 1, C)); print(f"dhconv fwd  : {ms:7.3f} ms {fl/ms/1e9:7.1f} TF dense-equiv")
    ms = timeit(lambda: ops.dhconv_dgrad(S, W, 1, C, C)); print(f"dhconv dgrad: {ms:7.3f} ms {fl/ms/1e9:7.1f} TF dense-equiv")
    ms = timeit(lambda: ops.dhconv_wgrad(S, S, 1)); print(f"dhconv wgrad: {ms:7.3f} ms {fl/ms/1e9:7.1f} TF dense-equiv")
    ms = timeit(lambda: ops.weight_to_wlayout(w)); print(f"weight->W   : {ms:7.3f} ms {2*w.numel()*8/ms/1e6:7.1f} GB/s")


def pointwise():
    for H, W in ((240, 480), (721, 1440)):
        x = torch.randn(1, 384, H, W, device=dev).bfloat16().requires_grad_(True)
        g = torch.ones(384, device=dev); b = torch.zeros(384, device=dev)
        nb = x.numel() * 2
        ms = timeit(lambda: ops.InstanceNormFn.apply(x, g, b, 1e-6, True))
        print(f"instnorm+gelu fwd {H}x{W}: {ms:7.3f} ms  {3*nb/ms/1e6:7.1f} GB/s (2 reads + 1 write)")
        y = ops.InstanceNormFn.apply(x, g, b, 1e-6, True)
        gy = torch.randn_like(y)
        ms = timeit(lambda: torch.autograd.grad(y, x, gy, retain_graph=True))
        print(f"instnorm+gelu bwd {H}x{W}: {ms:7.3f} ms  {5*nb/ms/1e6:7.1f} GB/s (4 reads + 1 write)")
        ms = timeit(lambda: ops.BiasGeluFn.apply(x, b))
        print(f"bias_gelu fwd     {H}x{W}: {ms:7.3f} ms  {2*nb/ms/1e6:7.1f} GB/s")


def conv():
    for (M, K, H, W) in ((384, 384, 721, 1440), (768, 384, 721, 1440), (384, 768, 721, 1440), (768, 384, 240, 480),
                         (384, 768, 240, 480), (384, 384, 240, 480), (384, 73, 721, 1440), (73, 384, 721, 1440)):
        x = torch.randn(1, K, H, W, device=dev).bfloat16()
        w = (torch.randn(M, K, device=dev) / K ** 0.5).bfloat16()
        A = ops.pad_weight_bf16(w)
        fl = 2.0 * M * K * H * W
        nb = 2.0 * H * W * (M + K)
        ms = timeit(lambda: ops.conv1x1_nn(A, K, x))
        ms2 = timeit(lambda: torch.mm(w, x.view(K, H * W)))
        g = torch.randn(1, M, H, W, device=dev).bfloat16()
        ms3 = timeit(lambda: ops.conv1x1_wgrad(g, x))
        ms4 = timeit(lambda: torch.mm(g.view(M, H * W), x.view(K, H * W).t()))
        print(f"conv M={M} K={K} {H}x{W}: nn hip {ms:7.3f} ms ({fl/ms/1e9:6.0f} TF, {nb/ms/1e6:6.0f} GB/s) | torch.mm {ms2:7.3f} ms ({fl/ms2/1e9:6.0f} TF)"
              f" || wgrad hip {ms3:7.3f} ms ({fl/ms3/1e9:6.0f} TF) | torch {ms4:7.3f} ms ({fl/ms4/1e9:6.0f} TF)")


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


if __name__ == "__main__":
    which = sys.argv[1:] or ["fft", "legendre", "dhconv", "pointwise"]
    for w in which:
        globals()[w]()
