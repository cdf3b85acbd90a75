"""GPU tests of the optimizer path on the real kernels (mk_adamw_step / mk_adamw_multi / mk_adamw_advance / mk_grad_clip_coef):
ZeRO-1 (reduce-scattered gradients, sharded AdamW state, parameter all-gather: SURVEY.md §8f item 3) with N ranks sharing
cuda:0 over gloo against single-process torch.optim.AdamW, and the checkpoint round trip of the device-side step counter
(makani loads checkpoints with map_location="cpu": makani/utils/driver.py:436,507)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _r(t):
    return torch.view_as_real(t) if t.is_complex() else t


def _worker_zero_gpu(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _fullsize import share_gpu
    share_gpu(rank, world)              # ranks sharing ONE GPU get disjoint compute units, set before the first GPU call
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import makani_amd.comm as mcomm
        import makani_amd.distributed as thd
        import makani_amd.optim as mo
        from makani_amd import ops
        dev = torch.device("cuda:0")
        mcomm.init(1, 1)
        torch.manual_seed(0)
        model = torch.nn.Module()
        model.w = torch.nn.Parameter(torch.randn(1536, 1024, device=dev))                        # 6.3 MB fp32: sharded
        model.c = torch.nn.Parameter(torch.randn(640, 512, dtype=torch.complex64, device=dev))   # complex, sharded through its real view
        nat = ops.native_w_empty(64, 48, 120, dev)                                               # dhconv weight in the GEMM's memory order
        with torch.no_grad():
            nat.copy_(torch.randn(1, 64, 48, 120, dtype=torch.complex64, device=dev))
        model.n = torch.nn.Parameter(nat)
        assert not model.n.is_contiguous() and ops.is_native_w(model.n)
        model.odd = torch.nn.Parameter(torch.randn(1048583, device=dev))                          # a prime number of floats: stays replicated
        model.b = torch.nn.Parameter(torch.randn(37, device=dev))                                 # small: bucketed, multi-tensor kernel
        ref = {k: v.detach().clone().requires_grad_(True) for k, v in model.named_parameters()}
        # eps = 1e-3: with 1e-8 the first updates are lr * g / (|g| + eps) ~ lr * sign(g), and the handful of elements whose
        # MEAN gradient is ~1e-7 flip with the summation order of the collective (gloo's ring vs the reference's left-to-right
        # sum differ in the last bit once there are more than two ranks)
        kw = dict(lr=1e-2, betas=(0.9, 0.95), eps=1e-3, weight_decay=0.1)
        ropt = torch.optim.AdamW(list(ref.values()), **kw)
        net = thd.init_gradient_reduction_hooks(model, dev, zero=True)
        net.reducer.big_bytes = 1 << 20
        opt = mo.FusedAdamW(model.parameters(), **kw)
        for it in range(3):
            tg = {}
            for ik, (k, v) in enumerate(model.named_parameters()):    # every rank generates every rank's gradient
                gen = torch.Generator(device="cpu").manual_seed(1000 * it + ik)
                tg[k] = [(torch.randn(world, *_r(v).shape, generator=gen))[r].to(dev) for r in range(world)]
            for p in model.parameters():
                p.grad = None
            loss = sum((_r(p) * tg[k][rank]).sum() for k, p in model.named_parameters())
            loss.backward()
            for k in ("w", "c", "n"):
                z = mo.FusedAdamW.zero_shard(getattr(model, k))
                assert z is not None and z[0].numel() == _r(getattr(model, k)).numel() // world, k
            assert mo.FusedAdamW.zero_shard(model.odd) is None and mo.FusedAdamW.zero_shard(model.b) is None
            for k, v in ref.items():
                g = sum(tg[k]) / world
                v.grad = torch.view_as_complex(g.contiguous()) if v.is_complex() else g
            gn_ref = torch.sqrt(sum(_r(v.grad).double().pow(2).sum() for v in ref.values()))
            coef_ref = min(1.0, 30.0 / (float(gn_ref) + 1e-6))
            for v in ref.values():
                v.grad.mul_(coef_ref)
            ropt.step()
            cc = opt.clip_coef(30.0)
            assert abs(float(cc[1]) - float(gn_ref)) < 1e-5 * float(gn_ref), (float(cc[1]), float(gn_ref))
            assert abs(float(thd.total_grad_norm(model)) - float(gn_ref)) < 1e-5 * float(gn_ref)
            opt.step(grad_scale=cc[:1])
            with pytest.raises(RuntimeError):                         # the shard is consumed: p.grad is the unreduced local gradient
                opt.grad_norm()
            for k, p in model.named_parameters():
                a, b = _r(p.detach()), _r(ref[k].detach())
                assert torch.allclose(a, b, rtol=3e-5, atol=3e-6), (it, k, float((a - b).abs().max()))
        assert opt.state[model.w]["exp_avg"].numel() == 1536 * 1024 // world
        assert opt.state[model.n]["exp_avg"].numel() == 2 * 64 * 48 * 120 // world
        assert opt.state[model.odd]["exp_avg"].numel() == 1048583
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_zero1_on_the_hip_kernels_matches_single_process_adamw(world):
    mp.spawn(_worker_zero_gpu, args=(world, _free_port()), nprocs=world, join=True)


@pytest.mark.parametrize("source", ["fused", "torch"])
def test_fused_adamw_resumes_bias_correction_from_a_cpu_loaded_checkpoint(source, tmp_path):
    """save -> torch.load(map_location="cpu") -> load_state_dict -> step: the device-side step counter is rebuilt (from its
    saved tensor, or from state["step"] of a torch.optim.AdamW checkpoint), so the resumed run keeps torch.optim.AdamW's
    bias corrections instead of restarting them at step 1"""
    from makani_amd.optim import FusedAdamW
    dev = "cuda:0"
    torch.manual_seed(3)
    shapes = [(300, 200), (17,), (1200, 1024)]
    kw = dict(lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)

    def params():
        torch.manual_seed(3)
        return [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes] + \
               [torch.nn.Parameter(torch.randn(40, 30, dtype=torch.complex64, device=dev))]

    def grads(it, ps):
        gen = torch.Generator().manual_seed(50 + it)
        for p in ps:
            g = torch.randn(_r(p).shape, generator=gen).to(dev)
            p.grad = torch.view_as_complex(g) if p.is_complex() else g

    pa, pb = params(), params()
    ref = torch.optim.AdamW(pa, **kw)
    first = FusedAdamW(pb, **kw) if source == "fused" else torch.optim.AdamW(pb, **kw)
    for it in range(5):
        grads(it, pa), grads(it, pb)
        ref.step(), first.step()
    path = tmp_path / "opt.pt"
    torch.save(first.state_dict(), path)
    resumed = FusedAdamW(pb, **kw)
    resumed.load_state_dict(torch.load(path, map_location="cpu"))
    for it in range(5, 8):
        grads(it, pa), grads(it, pb)
        ref.step(), resumed.step()
    assert float(resumed.param_groups[0]["_mk_step_state"][0]) == 8.0
    for a, b in zip(pa, pb):
        assert torch.allclose(_r(a.detach()), _r(b.detach()), rtol=2e-5, atol=2e-6), float((_r(a) - _r(b)).abs().max())


def _dhconv_setup(C_in, C_out, L, M, seed=0):
    """one dhconv layer's worth of autograd graph on random spectral coefficients: returns (weight parameter, loss closure)"""
    from makani_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(seed)
    w = ops.native_w_empty(C_in, C_out, L, dev)
    with torch.no_grad():
        w.copy_(torch.randn(1, C_in, C_out, L, dtype=torch.complex64, generator=g).to(dev) / C_in ** 0.5)
    w = torch.nn.Parameter(w)
    x = torch.randn(1, C_in, L, M, dtype=torch.complex64, generator=g).to(dev)
    S = ops.complex_to_s(x)                                     # (L, M, 2, round4(C_in))
    G = torch.randn(L, M, 2, (C_out + 3) // 4 * 4, generator=g).to(dev)

    def loss():
        return (ops.DhconvFn.apply(S, w, 1, 0) * G).sum()
    return w, loss


@pytest.mark.parametrize("shape", [(64, 48, 40, 41), (384, 384, 24, 25), (132, 260, 17, 18)])   # (C_in, C_out, L, M)
def test_gradient_norm_from_the_sums_of_squares_of_the_weight_gradient_kernel(shape, monkeypatch):
    """FusedAdamW.clip_coef with the dhconv weight gradient's squares summed by the kernel that wrote it
    (mk_cgemm_split2_batched_ssq -> mk_grad_clip_coef_pre; makani/utils/training/training_helpers.py:123-165) against the same
    norm from reading the gradient, and against torch in fp64; the stand-in must be dropped as soon as the gradient changes."""
    import makani_amd.optim as mo
    from makani_amd import ops
    w, loss = _dhconv_setup(*shape)
    other = torch.nn.Parameter(torch.randn(1000, 77, device="cuda:0"))
    opt = mo.FusedAdamW([w, other], lr=1e-3)
    (loss() + (other ** 2).sum() * 1e-3).backward()
    flat = mo._flat(w.grad)
    part = ops.grad_ssq_lookup(flat, w.grad._version)
    assert part is not None and part.numel() > 0, "the weight-gradient launch did not register its sums of squares"
    want = torch.sqrt((torch.view_as_real(w.grad).double() ** 2).sum() + (other.grad.double() ** 2).sum()).item()
    # the partial sums themselves: their total is the gradient's squared norm
    assert abs(part.double().sum().item() - (torch.view_as_real(w.grad).double() ** 2).sum().item()) <= 1e-6 * want ** 2
    fused = opt.clip_coef(0.5 * want)
    monkeypatch.setattr(ops, "_GRAD_SSQ", {})
    plain = opt.clip_coef(0.5 * want)
    assert abs(fused[1].item() - want) <= 2e-6 * want and abs(plain[1].item() - want) <= 2e-6 * want
    assert abs(fused[0].item() - 0.5) < 1e-5 and abs(plain[0].item() - 0.5) < 1e-5
    monkeypatch.undo()
    # a torch-level write to the gradient invalidates the stand-in ...
    w.grad.mul_(3.0)
    assert ops.grad_ssq_lookup(mo._flat(w.grad), w.grad._version) is None
    want3 = torch.sqrt(9.0 * (torch.view_as_real(w.grad).double() ** 2).sum() / 9.0 + (other.grad.double() ** 2).sum()).item()
    assert abs(opt.clip_coef(None)[1].item() - want3) <= 2e-6 * want3
    # ... so does accumulation into an existing .grad (second backward without zero_grad) ...
    (loss() + (other ** 2).sum() * 1e-3).backward()
    acc = torch.sqrt((torch.view_as_real(w.grad).double() ** 2).sum() + (other.grad.double() ** 2).sum()).item()
    assert abs(opt.clip_coef(None)[1].item() - acc) <= 2e-6 * acc
    # ... and step() drops whatever is left
    opt.zero_grad(set_to_none=True)
    (loss() + (other ** 2).sum() * 1e-3).backward()
    assert ops.grad_ssq_lookup(mo._flat(w.grad), w.grad._version) is not None
    opt.step(max_grad_norm=1.0)
    assert not ops._GRAD_SSQ


def test_sums_of_squares_of_the_complex_split_gemm_cover_every_workgroup():
    """mk_cgemm_split2_batched_ssq directly: ragged tiles, batches that are not a multiple of the 8 XCDs (idle workgroups write 0),
    both limb counts; sum of the partials against the squares of C"""
    import ctypes as C
    from makani_amd import _lib, ops
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    for (Mm, Nn, K, batch), limbs in (((72, 52, 33, 5), 2), ((200, 132, 64, 11), 3), ((384, 384, 40, 16), 2)):
        A = torch.randn(batch, K, 2, Mm, device=dev)            # [b][k][re/im][row]
        Bm = torch.randn(batch, K, 2, Nn, device=dev)
        Cc = torch.empty(batch, Mm, Nn, 2, device=dev)          # interleaved complex result
        g = ops._gemm(A=A.data_ptr(), B=Bm.data_ptr(), C=Cc.data_ptr(), a_batch=K * 2 * Mm, a_row=1, a_k=2 * Mm, a_im=Mm,
                      b_batch=K * 2 * Nn, b_col=1, b_k=2 * Nn, b_im=Nn, c_batch=Mm * Nn * 2, c_row=Nn * 2, c_col=2, c_im=1,
                      M=Mm, N=Nn, K=K, batch=batch, inner=1, conj_a=1)
        n = _lib.lib().mk_cgemm_split2_ssq_count(C.byref(g))
        assert n > 0
        part = torch.full((n,), float("nan"), device=dev)
        _lib.check(_lib.lib().mk_cgemm_split2_batched_ssq(C.byref(g), limbs, _lib.ptr(part), _lib.stream()), "ssq")
        ref = torch.einsum("bkm,bkn->bmn", torch.complex(A[:, :, 0], -A[:, :, 1]).to(torch.complex128), torch.complex(Bm[:, :, 0], Bm[:, :, 1]).to(torch.complex128))
        got = torch.view_as_complex(Cc)
        assert ((got - ref).abs().max() / ref.abs().max()).item() < (1e-4 if limbs == 2 else 1e-5)
        assert torch.isfinite(part).all()
        tot, want = part.double().sum().item(), (Cc.double() ** 2).sum().item()
        assert abs(tot - want) <= 1e-6 * want, (Mm, Nn, K, batch, tot, want)
