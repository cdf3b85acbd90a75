"""Timing-dependent wrong results under CONTENTION: N processes share the GPU, each runs every hot-path operation REPS times
on fixed inputs and compares each result bit by bit with the first.  All kernels of the package are deterministic by
construction (no float atomics, fixed reduction orders), so ANY difference between two runs of the same operation on the same
inputs is a race in that operation's kernels (a missing barrier / wait that a lone process never loses).  Found this way in
round 5: see docs/LAB_NOTEBOOK.md.

    python tools/race_hunt.py [--procs 4] [--reps 6] [--only substring]
"""
import argparse
import math
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def _flat(out):
    if isinstance(out, torch.Tensor):
        return [out]
    return [t for o in out if o is not None for t in _flat(o)]


def _same(a, b):
    a = torch.view_as_real(a.resolve_conj()) if a.is_complex() else a
    b = torch.view_as_real(b.resolve_conj()) if b.is_complex() else b
    if a.dtype in (torch.float32, torch.float64):
        eq = (a == b) | (a.isnan() & b.isnan())
    else:
        eq = a.view(torch.int16) == b.view(torch.int16) if a.dtype == torch.bfloat16 else a == b
    nbad = int((~eq).sum())
    if nbad == 0:
        return 0, 0.0
    d = (a.double() - b.double())
    return nbad, float(d.norm() / b.double().norm().clamp_min(1e-30))


def _mask(t, tri):
    """entries with l < m are unspecified (the kernels skip the structurally empty tiles): compare the triangle only"""
    return torch.where(tri, t, torch.zeros((), dtype=t.dtype, device=t.device))


def cases(dev):
    from makani_amd import ops
    import makani_amd as ma
    C, L, M = 384, 240, 241
    out = []

    def add(name, fn):
        out.append((name, fn))

    # ---- fp32 channel GEMMs (the engine of the fp32 parity runs), shard-shaped pixel counts
    for (Mo, K, N) in ((384, 73, 181 * 1440), (384, 384, 181 * 1440), (73, 384, 181 * 1440), (768, 384, 60 * 480)):
        torch.manual_seed(Mo + K)
        w = torch.randn(Mo, K, device=dev) / K ** 0.5
        x = torch.rand(1, K, N, device=dev) - 0.5
        g = torch.randn(1, Mo, N, device=dev)
        add(f"chan_gemm_f32 m{Mo} k{K} n{N}", lambda w=w, x=x: ops.chan_gemm_f32(w, x))
        add(f"chan_gemm_f32^T m{Mo} k{K} n{N}", lambda w=w, g=g: ops.chan_gemm_f32(w, g, transposed=True))
        add(f"chan_wgrad_f32 m{Mo} k{K} n{N}", lambda g=g, x=x: ops.chan_wgrad_f32(g, x))
    # ---- Legendre transforms
    for nlat, nlon, grid in ((721, 1440, "equiangular"), (240, 480, "legendre-gauss")):
        S = ma.RealSHT(nlat, nlon, lmax=L, mmax=M, grid=grid).to(dev)
        I = ma.InverseRealSHT(nlat, nlon, lmax=L, mmax=M, grid=grid).to(dev)
        tri = (torch.arange(L, device=dev)[:, None] >= torch.arange(M, device=dev)[None, :])[:, :, None, None]
        for R in (384, 96):
            F = torch.randn(M, nlat, 2, R, device=dev)
            Fl = torch.randn(nlat, M, 2, R, device=dev)
            Sc = torch.randn(L, M, 2, R, device=dev) * tri
            add(f"legendre_analysis k{nlat} R{R}", lambda F=F, S=S, tri=tri: _mask(ops.legendre_analysis(F, S.weights_t, L), tri))
            add(f"legendre_analysis k{nlat} R{R} lat-major", lambda Fl=Fl, S=S, tri=tri: _mask(ops.legendre_analysis(Fl, S.weights_t, L, 0, True), tri))
            add(f"legendre_synthesis k{nlat} R{R}", lambda Sc=Sc, I=I: ops.legendre_synthesis(Sc, I.pct, nlat))
            add(f"legendre_synthesis k{nlat} R{R} lat-major", lambda Sc=Sc, I=I: ops.legendre_synthesis(Sc, I.pct, nlat, 0, True))
            add(f"legendre_analysis(adjoint matrix) k{nlat} R{R}", lambda F=F, I=I, tri=tri: _mask(ops.legendre_analysis(F, I.pct_t, L), tri))
            add(f"legendre_synthesis(adjoint matrix) k{nlat} R{R}", lambda Sc=Sc, S=S: ops.legendre_synthesis(Sc, S.weights, nlat))
        # ---- FFTs
        c = 2 * math.pi / nlon
        for dt in (torch.float32, torch.bfloat16):
            x = torch.rand(1, C, nlat, nlon, device=dev).to(dt)
            F = ops.rfft_rows(x, M, C, (c, c, c))
            add(f"rfft {nlat}x{nlon} {dt}", lambda x=x, c=c: ops.rfft_rows(x, M, C, (c, c, c)))
            add(f"irfft {nlat}x{nlon} {dt}", lambda F=F, dt=dt, nlon=nlon: ops.irfft_rows(F, 1, C, nlon, dt, (1.0, 2.0, 1.0)))
    # ---- dhconv
    for Ll, off in ((240, 0), (60, 180), (60, 0)):
        Ssp = torch.randn(Ll, M, 2, C, device=dev)
        G = torch.randn(Ll, M, 2, C, device=dev)
        w = ops.native_w_empty(C, C, Ll, dev)
        w.copy_(torch.randn(1, C, C, Ll, dtype=torch.complex64, device=dev))
        tri = (torch.arange(Ll, device=dev)[:, None] + off >= torch.arange(M, device=dev)[None, :])[:, :, None, None]
        add(f"dhconv_fwd L{Ll}+{off}", lambda Ssp=Ssp, w=w, off=off, tri=tri: _mask(ops.dhconv_fwd(Ssp, w, 1, C, off), tri))
        add(f"dhconv_dgrad L{Ll}+{off}", lambda G=G, w=w, off=off, tri=tri: _mask(ops.dhconv_dgrad(G, w, 1, C, C, off), tri))
        add(f"dhconv_wgrad L{Ll}+{off}", lambda Ssp=Ssp, G=G, off=off: ops.dhconv_wgrad(Ssp, G, 1, off, native=True))
    # ---- norms and pointwise, fp32 and bf16, serial and the distributed kernels (world-1 group)
    for H, W in ((181, 1440), (60, 480)):
        for dt in (torch.float32, torch.bfloat16):
            x = torch.randn(1, C, H, W, device=dev).to(dt)
            gy = torch.randn(1, C, H, W, device=dev).to(dt)
            gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1

            def norm(x=x, gy=gy, gam=gam, bet=bet, gelu=False, distd=False):
                xr = x.clone().requires_grad_(True)
                gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
                if distd:
                    y = ops.DistInstanceNormFn.apply(xr, gr, br, 1e-6, gelu, dist.group.WORLD)
                else:
                    y = ops.InstanceNormFn.apply(xr, gr, br, 1e-6, gelu)
                y.backward(gy)
                return y.detach(), xr.grad, gr.grad, br.grad
            for gelu in (False, True):
                add(f"instnorm gelu={int(gelu)} {H}x{W} {dt}", lambda f=norm, gelu=gelu: f(gelu=gelu))
                add(f"dist instnorm gelu={int(gelu)} {H}x{W} {dt}", lambda f=norm, gelu=gelu: f(gelu=gelu, distd=True))

            def bg(x=x, gy=gy, bet=bet):
                xr = x.clone().requires_grad_(True)
                y = ops.BiasGeluFn.apply(xr, bet)
                y.backward(gy)
                return y.detach(), xr.grad
            add(f"bias_gelu {H}x{W} {dt}", bg)
    # ---- bf16 channel GEMMs
    for (Mo, K, H, W) in ((768, 384, 181, 720), (384, 768, 181, 720), (384, 384, 181, 1440), (384, 73, 181, 1440), (73, 384, 181, 1440),
                          (768, 384, 60, 480), (384, 384, 240, 480)):
        torch.manual_seed(Mo + K + H)
        x = (torch.rand(1, K, H, W, device=dev) - 0.5).bfloat16()
        w = (torch.randn(Mo, K, device=dev) / K ** 0.5).bfloat16()
        bias = torch.randn(Mo, device=dev)
        A = ops.pad_weight_bf16(w)
        gsrc = torch.randn(1, Mo, H, W, device=dev).bfloat16()
        res = torch.randn(1, Mo, H, W, device=dev).bfloat16()
        add(f"conv1x1_nn m{Mo} k{K} {H}x{W}", lambda A=A, K=K, x=x: ops.conv1x1_nn(A, K, x)[0])
        add(f"conv1x1_nn+bias+gelu+pre m{Mo} k{K} {H}x{W}", lambda A=A, K=K, x=x, bias=bias: ops.conv1x1_nn(A, K, x, bias=bias, act=True, want_pre=True))
        add(f"conv1x1_nn*gelu' m{Mo} k{K} {H}x{W}", lambda A=A, K=K, x=x, gsrc=gsrc: ops.conv1x1_nn(A, K, x, gelu_grad_of=gsrc)[0])
        add(f"conv1x1_nn+R m{Mo} k{K} {H}x{W}", lambda A=A, K=K, x=x, res=res: ops.conv1x1_nn(A, K, x, residual=res)[0])
        add(f"conv1x1_wgrad m{Mo} k{K} {H}x{W}", lambda gsrc=gsrc, x=x: ops.conv1x1_wgrad(gsrc, x))
        add(f"conv1x1_wgrad+bias m{Mo} k{K} {H}x{W}", lambda gsrc=gsrc, x=x: ops.conv1x1_wgrad(gsrc, x, want_bias=True))
    # ---- probe kernels (tools/probes/lds_poison.hip; round 6): as a hog (RACE_HUNT_HOG_FILTER=probe_lds / probe_spin) they leave NaN
    # patterns in every compute unit's LDS and vector registers / stay co-resident WITHOUT touching memory or the matrix cores
    ppath = os.path.join(ROOT, "tools", "probes", "liblds_poison.so")
    if os.path.exists(ppath):
        import ctypes
        lib = ctypes.CDLL(ppath)
        lib.mk_probe_lds_poison.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
        lib.mk_probe_vgpr_poison.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        sink = torch.zeros(1024, dtype=torch.int32, device=dev)

        def poison(spin=0, lds=64 * 1024):
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            lib.mk_probe_lds_poison(0xFFFFFFFF, lds, 2048, 256, spin, ctypes.c_void_p(sink.data_ptr()), st)
            lib.mk_probe_vgpr_poison(0xFFFFFFFF, 4096, ctypes.c_void_p(sink.data_ptr()), st)
            return sink
        add("probe_lds poison 64K", lambda: poison())
        add("probe_spin 8K lds 20us", lambda: poison(spin=2000, lds=8 * 1024))
    return out


def torch_cases(dev):
    """the same producer -> consumer patterns in PURE torch kernels (no kernel of this package): elementwise producer, partial
    reduction, final reduction.  If these are not reproducible either, the box — not the package — loses writes between kernels."""
    out = []
    for n in (384 * 60 * 480, 384 * 181 * 1440):
        x = torch.randn(n, device=dev)
        y = torch.randn(n, device=dev)

        def f(x=x, y=y):
            p = (x * y).view(384, -1)
            part = p.view(384, 5, -1).sum(dim=2)          # partials
            tot = part.sum(dim=1)                         # final
            z = (p - tot[:, None] / p.shape[1])           # consumer of the totals
            return part, tot, z.bfloat16().float().sum(dim=1)
        out.append((f"torch partial->final n{n}", f))
    return out


def torch_norm_cases(dev):
    """the instance-norm forward + backward in PURE torch arithmetic on the tensors of the package's cases (control)"""
    out = []
    C = 384
    for H, W in ((60, 480), (181, 1440)):
        x = torch.randn(1, C, H, W, device=dev).bfloat16()
        gy = torch.randn(1, C, H, W, device=dev).bfloat16()
        gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1

        def f(x=x, gy=gy, gam=gam, bet=bet):
            xf, gf = x.float(), gy.float()
            mean = xf.mean(dim=(2, 3), keepdim=True)
            var = (xf - mean).square().mean(dim=(2, 3), keepdim=True)
            rstd = torch.rsqrt(var + 1e-6)
            n = (xf - mean) * rstd
            y = (n * gam[None, :, None, None] + bet[None, :, None, None]).bfloat16()
            s1 = gf.sum(dim=(2, 3), keepdim=True)
            s2 = (gf * n).sum(dim=(2, 3), keepdim=True)
            gx = (rstd * gam[None, :, None, None] * (gf - s1 / (H * W) - n * s2 / (H * W))).bfloat16()
            return y, gx, s2.reshape(-1), s1.reshape(-1)
        out.append((f"torch-arithmetic instnorm fwd+bwd {H}x{W} bf16", f))
    return out


def model_cases(dev):
    """the whole serial network, forward + backward, fp32 and bf16 autocast (catch-all)"""
    import makani_amd as ma
    from _fullsize import CONFIG2, perturb_affine
    torch.manual_seed(333)
    cfg = {**CONFIG2, "num_layers": 4}
    model = ma.SphericalFourierNeuralOperatorNet(**cfg)
    perturb_affine(model, 7)
    model = model.to(dev)
    x = torch.rand(1, 73, 721, 1440, device=dev)
    g = torch.randn(1, 73, 721, 1440, device=dev)

    def run(amp):
        model.zero_grad(set_to_none=True)
        xs = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            y = model(xs)
        (y.float() * g).sum().backward()
        names = ["y", "gx"] + [n for n, _ in model.named_parameters()]
        return names, [y.detach().float(), xs.grad] + [p.grad.detach().clone() for p in model.parameters()]
    return [("model 4 layers fp32", lambda: run(False)), ("model 4 layers bf16", lambda: run(True))]


def cu_mask_env(rank, world, ncu=256):
    """disjoint compute-unit ranges per process (HSA_CU_MASK, read by the ROCm runtime when it creates the process's queues):
    kernels of different processes then never share a compute unit"""
    per = ncu // world
    return f"0:{rank * per}-{(rank + 1) * per - 1}"


def worker(rank, world, ports, reps, only, with_model):
    if os.environ.get("RACE_HUNT_CU_MASK", "0") == "1":
        os.environ["HSA_CU_MASK"] = cu_mask_env(rank, world)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(ports[rank])
    dist.init_process_group("gloo", rank=0, world_size=1)                 # a world of one per process: the distributed kernels' group
    torch.set_num_threads(4)
    dev = torch.device("cuda:0")
    hog = os.environ.get("RACE_HUNT_HOGS", "0") == "1" and rank > 0      # every process but the first runs torch kernels only
    hogf = os.environ.get("RACE_HUNT_HOG_FILTER")                        # ... or only the operations whose name contains this
    todo = torch_cases(dev) + torch_norm_cases(dev) + ([] if hog else cases(dev))
    if hog:
        only, reps = "torch", reps * 4
    if with_model and not hog:
        todo += model_cases(dev)
    if hogf is not None:
        # culprit search: process 0 is the VICTIM (runs --only, typically the reduction-heavy norm kernels), every other process
        # loops over the operations matching the filter until the victim is done.  A class of operations whose presence makes the
        # victim irreproducible disturbs kernels of OTHER processes.
        tag = os.environ.get("RACE_HUNT_TAG", "x")
        flag = lambda r: f"/tmp/race_hunt_{tag}_{r}"
        open(flag(rank), "w").close()
        while not all(os.path.exists(flag(r)) for r in range(world)):
            time.sleep(0.05)
        if rank > 0:
            mine = [(n, f) for n, f in todo if hogf in n]
            it = 0
            while not os.path.exists(flag("done")):
                for n, f in mine:
                    f()
                torch.cuda.synchronize()
                it += 1
            print(f"[proc {rank}] hog '{hogf}': {len(mine)} operations x {it} rounds", flush=True)
            dist.destroy_process_group()
            return
    bad = 0
    t0 = time.time()

    def captured(fn):
        out = []
        for v in (fn.__defaults__ or ()):
            if isinstance(v, torch.Tensor):
                out.append(v)
            elif callable(v) and getattr(v, "__defaults__", None):
                out += [u for u in v.__defaults__ if isinstance(u, torch.Tensor)]
        return out

    def digest(ts):
        return [float(torch.view_as_real(t).double().sum()) if t.is_complex() else float(t.double().sum()) for t in ts]
    for name, fn in todo:
        if only and only not in name:
            continue
        names = None
        ins = captured(fn)
        d0 = digest(ins)
        ref = fn()
        if isinstance(ref, tuple) and len(ref) == 2 and isinstance(ref[0], list) and isinstance(ref[0][0], str):
            names, ref = ref
        ref = [t.clone() for t in _flat(ref)]
        torch.cuda.synchronize()
        worst = (0, 0.0, -1, "")
        nrep_bad = 0
        per_out = {}
        prev, changes = ref, []          # repetitions whose result differs from the PREVIOUS one (transient vs persistent)
        for r in range(reps):
            # shift the caching allocator's choices: otherwise a stale read of a scratch buffer finds the previous repetition's
            # (identical) values at the same address and goes unnoticed
            jitter = [torch.full((1 + (7919 * (r + 1) * (rank + 3)) % 100003,), float("nan"), device=dev) for _ in range(1 + r % 3)]
            o = fn()
            del jitter
            if names is not None:
                o = o[1]
            o = _flat(o)
            rep_bad = False
            for k, (a, b) in enumerate(zip(o, ref)):
                n, e = _same(a, b)
                rep_bad = rep_bad or n > 0
                if n:
                    d = per_out.setdefault(names[k] if names else f"out{k}", [0, 0, 0.0])
                    d[0] += 1
                    d[1] = max(d[1], n)
                    d[2] = max(d[2], e)
                if n and e >= worst[1]:
                    worst = (n, e, r, names[k] if names else f"out{k}")
            nrep_bad += int(rep_bad)
            if any(_same(a, b)[0] for a, b in zip(o, prev)):
                changes.append(r)
            prev = [t.clone() for t in o]
        d1 = digest(ins)
        if d0 != d1:
            print(f"[proc {rank}] INPUT CHANGED  {name}: checksums of the operation's fixed input tensors {d0} -> {d1}", flush=True)
        if worst[0]:
            bad += 1
            print(f"[proc {rank}] RACE  {name}: {nrep_bad} of {reps} repetitions differ from the first; worst: {worst[0]} elements "
                  f"(rel-L2 {worst[1]:.2e}) in rep {worst[2]}, {worst[3]}; per output (reps, max elements, max rel-L2): "
                  + ", ".join(f"{k}: {v[0]}/{v[1]}/{v[2]:.1e}" for k, v in list(per_out.items())[:8])
                  + f"; changed against the previous repetition in reps {changes[:40]}", flush=True)
        elif rank == 0:
            print(f"[proc {rank}] ok    {name}", flush=True)
    print(f"[proc {rank}] done: {bad} operation(s) not reproducible, {time.time() - t0:.0f} s", flush=True)
    if os.environ.get("RACE_HUNT_HOG_FILTER") is not None:
        open(f"/tmp/race_hunt_{os.environ.get('RACE_HUNT_TAG', 'x')}_done", "w").close()
    dist.destroy_process_group()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=4)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--only", default="")
    ap.add_argument("--model", action="store_true")
    a = ap.parse_args()
    socks = [socket.socket() for _ in range(a.procs)]
    for s in socks:
        s.bind(("127.0.0.1", 0))
    ports = [s.getsockname()[1] for s in socks]
    for s in socks:
        s.close()
    mp.spawn(worker, args=(a.procs, ports, a.reps, a.only, a.model), nprocs=a.procs, join=True)
