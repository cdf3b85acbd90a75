"""Second step of the interference hunt (tools/two_stream_hunt.py reproduced it in ONE process): which kernel of the victims
goes wrong, and what do the wrong values look like?  Everything here goes through the C ABI on PRE-ALLOCATED buffers (no
allocator, no autograd): victim kernels on stream A, a culprit looping on stream B.

    python tools/two_stream_micro.py [--reps 40] [--culprit chan_gemm_f32]
"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")

import torch  # noqa: E402

from makani_amd import _lib, ops  # noqa: E402
from makani_amd._lib import check, lib, ptr, stream  # noqa: E402


def culprits(dev, which):
    out = []
    if which in ("chan_gemm_f32", "all"):
        w = torch.randn(384, 384, device=dev) / 384 ** 0.5
        x = torch.rand(1, 384, 181 * 1440, device=dev) - 0.5
        out.append(("chan_gemm_f32", lambda: ops.chan_gemm_f32(w, x)))
    if which in ("conv1x1_nn", "all"):
        xb = (torch.rand(1, 384, 181, 720, device=dev) - 0.5).bfloat16()
        A = ops.pad_weight_bf16((torch.randn(768, 384, device=dev) / 384 ** 0.5).bfloat16())
        out.append(("conv1x1_nn", lambda: ops.conv1x1_nn(A, 384, xb)[0]))
    if which in ("dhconv", "all"):
        S = torch.randn(240, 241, 2, 384, device=dev)
        wt = ops.native_w_empty(384, 384, 240, dev)
        wt.copy_(torch.randn(1, 384, 384, 240, dtype=torch.complex64, device=dev))
        out.append(("dhconv", lambda: ops.dhconv_fwd(S, wt, 1, 384, 0)))
    if which in ("torch_mm", "all"):
        a = torch.randn(8192, 8192, device=dev).bfloat16()
        b = torch.randn(8192, 8192, device=dev).bfloat16()
        out.append(("torch_mm", lambda: a @ b))
    if which in ("rfft", "all"):
        xr = torch.rand(1, 384, 721, 1440, device=dev).bfloat16()
        c = 2 * math.pi / 1440
        out.append(("rfft", lambda: ops.rfft_rows(xr, 241, 384, (c, c, c))))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--culprit", default="all")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    C, H, W = 384, 60, 480
    planes, hw = C, H * W
    L = lib()
    victims = {}
    for dt, code in ((torch.bfloat16, _lib.MK_BF16), (torch.float32, _lib.MK_F32)):
        x = torch.randn(1, C, H, W, device=dev).to(dt)
        gy = torch.randn(1, C, H, W, device=dev).to(dt)
        gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
        ch = L.mk_pointwise_chunks(hw, code, planes)
        bufs = dict(stats=torch.empty(planes, 2, device=dev), ws=torch.empty(planes * ch * 2, device=dev), y=torch.empty_like(x),
                    sums=torch.empty(2, planes, device=dev), ws2=torch.empty(planes * ch * 2, device=dev), gx=torch.empty_like(x),
                    stats_out=torch.empty(planes, 2, device=dev), ws3=torch.empty(planes * ch * 2, device=dev), sums2=torch.empty(2, planes, device=dev))
        # fixed statistics, computed once on the idle GPU
        check(L.mk_instnorm_fwd(ptr(x), ptr(bufs["y"]), code, ptr(bufs["stats"]), ptr(bufs["ws"]), ptr(gam), ptr(bet), None, None, 0.0,
                                planes, C, hw, 1e-6, 0, stream()), "fwd")
        torch.cuda.synchronize()
        tag = "bf16" if dt == torch.bfloat16 else "f32"

        def v_stats(x=x, code=code, b=bufs):
            check(L.mk_instnorm_stats(ptr(x), code, ptr(b["stats_out"]), ptr(b["ws3"]), planes, hw, 1e-6, None, 0.0, stream()), "stats")
            return [b["stats_out"], b["ws3"]]

        def v_bwd_reduce(x=x, gy=gy, code=code, b=bufs, gam=gam, bet=bet):
            check(L.mk_instnorm_bwd(ptr(x), ptr(gy), ptr(b["gx"]), code, ptr(b["stats"]), ptr(gam), ptr(bet), None, None, 0.0, ptr(b["sums"]),
                                    ptr(b["ws2"]), planes, C, hw, hw, 1, 0, stream()), "bwd1")
            return [b["sums"], b["ws2"]]

        def v_bwd_apply(x=x, gy=gy, code=code, b=bufs, gam=gam, bet=bet):
            # phase 2 on FIXED sums (sums2 filled below at idle)
            check(L.mk_instnorm_bwd(ptr(x), ptr(gy), ptr(b["gx"]), code, ptr(b["stats"]), ptr(gam), ptr(bet), None, None, 0.0, ptr(b["sums2"]),
                                    ptr(b["ws2"]), planes, C, hw, hw, 2, 0, stream()), "bwd2")
            return [b["gx"]]
        v_bwd_reduce()
        torch.cuda.synchronize()
        bufs["sums2"].copy_(bufs["sums"])
        victims[f"stats {tag}"] = v_stats
        victims[f"bwd_reduce(fixed stats) {tag}"] = v_bwd_reduce
        victims[f"bwd_apply(fixed stats+sums) {tag}"] = v_bwd_apply
    # irfft on fixed input / output buffers
    c = 2 * math.pi / 480
    xs = torch.rand(1, C, 240, 480, device=dev)
    F = ops.rfft_rows(xs, 241, C, (c, c, c))
    plan = ops.fft_plan(480, dev)
    xo = torch.empty(1, C, 240, 480, device=dev)

    def v_irfft():
        check(L.mk_irfft_rows(ptr(F), ptr(xo), _lib.MK_F32, ptr(plan.twiddle), plan.radix, plan.nradix, 1, C, C, 240, 480, 241, 1.0, 2.0, 1.0,
                              stream()), "irfft")
        return [xo]
    victims["irfft 240x480 f32"] = v_irfft
    torch.cuda.synchronize()

    refs = {}
    for n, f in victims.items():
        refs[n] = [t.clone() for t in f()]
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for cname, cf in culprits(dev, a.culprit):
        for n, f in victims.items():
            torch.cuda.synchronize()
            nbad, shown = 0, 0
            for r in range(a.reps):
                with torch.cuda.stream(sb):
                    for _ in range(3):
                        cf()
                with torch.cuda.stream(sa):
                    outs = [t.clone() for t in f()]
                sa.synchronize()
                bad = False
                for k, (o, rf) in enumerate(zip(outs, refs[n])):
                    ne = (o != rf) & ~(o.isnan() & rf.isnan())
                    if bool(ne.any()):
                        bad = True
                        if shown < 3:
                            shown += 1
                            idx = ne.reshape(-1).nonzero().reshape(-1)
                            of, rff = o.reshape(-1).float(), rf.reshape(-1).float()
                            show = ", ".join(f"[{int(i)}] {float(of[i]):.7g} vs {float(rff[i]):.7g}" for i in idx[:6])
                            print(f"   {cname} | {n} | rep {r} out{k}: {int(ne.sum())} of {ne.numel()} differ; first index {int(idx[0])} last {int(idx[-1])}: {show}", flush=True)
                nbad += int(bad)
            torch.cuda.synchronize()
            print(f"{cname:>14s} | {n:<40s} | {nbad} of {a.reps} runs differ", flush=True)


if __name__ == "__main__":
    main()
