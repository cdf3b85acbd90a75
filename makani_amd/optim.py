"""Fused AdamW + global-norm gradient clipping for the SFNO train step (one HBM pass per
parameter tensor; the 566 M real numbers of complex64 spectral weights dominate the step).

ZeRO-1 over the data-parallel group: ``makani_amd.distributed.GradReducer(..., zero=True)`` ends the
reduction of every large gradient with a reduce-scatter instead of an all-reduce (same bytes on the wire as half an
all-reduce); ``FusedAdamW`` then keeps ``exp_avg`` / ``exp_avg_sq`` only for this rank's 1/N slice of such a parameter,
updates that slice (1/N of the 28 B per element the update streams through HBM) and all-gathers the parameter in place.
With N = 8 the optimizer pass of the benchmark model goes from 16 GB of HBM traffic per rank to 2 GB + the gathered 2.3 GB."""
import ctypes as C

import torch
import torch.distributed as dist

from . import ops as _ops
from ._lib import MkAdamTensor, check, dense_view, lib, ptr, stream

SMALL = 1 << 20          # tensors below this many floats share multi-tensor launches


# The three device operations of the big-tensor path behind plain functions: the CPU tests of the ZeRO bookkeeping replace
# them with torch implementations (the product has no CPU path: these call the HIP library).
def _k_advance(sdev, b1, b2):
    check(lib().mk_adamw_advance(ptr(sdev), b1, b2, stream()), "mk_adamw_advance")


def _k_adamw(pr, gr, m, v, scale, lr, b1, b2, eps, wd, sdev):
    from .ops import _timed                      # (28 B per parameter: p, g, m, v read, p, m, v written)
    with _timed("adamw", nbytes=28.0 * pr.numel()):
        check(lib().mk_adamw_step(ptr(pr), ptr(gr), ptr(m), ptr(v), pr.numel(), ptr(scale), lr, b1, b2, eps, wd, 0, ptr(sdev),
                                  stream()), "mk_adamw_step")


def _descs(ts):
    arr = (MkAdamTensor * max(1, len(ts)))(*[MkAdamTensor(None, t.data_ptr(), None, None, t.numel(), None, None, 0, 0, 0) for t in ts])
    return C.cast(arr, C.c_void_p), arr


def _k_sumsq_clip(grads, max_grad_norm, pre=()):
    """(2,) device tensor [min(1, max_norm / (||g|| + 1e-6)), ||g||] over a list of flat fp32 tensors; ``pre``: buffers of partial
    sums of squares that stand in for gradients not in ``grads`` (``ops.grad_ssq_lookup``)"""
    dev = (grads[0] if grads else pre[0]).device
    ga, _keep_g = _descs(grads)
    pa, _keep_p = _descs(pre)
    nws = lib().mk_grad_norm_workspace_pre(ga, len(grads), pa, len(pre))
    ws = torch.empty((nws,), dtype=torch.float32, device=dev)
    out = torch.empty((2,), dtype=torch.float32, device=dev)
    from .ops import _timed
    with _timed("adamw_grad_norm", nbytes=4.0 * (sum(g.numel() for g in grads) + sum(t.numel() for t in pre))):
        check(lib().mk_grad_clip_coef_pre(ga, len(grads), pa, len(pre), float(max_grad_norm or 0.0), ptr(ws), ptr(out), stream()),
              "mk_grad_clip_coef_pre")
    return out


def _real(t):
    return torch.view_as_real(t) if t.is_complex() else t


def invalidate_weight_shadows(module_or_params):
    """Drop the bf16 weight / transposed-weight images ``FusedAdamW`` wrote for the channel GEMMs.  They are keyed on the
    parameter's autograd version and storage address, which every torch-level change bumps (``copy_``, ``load_state_dict``,
    optimizers, in-place ops on the parameter) — but a write THROUGH ``param.data`` (``p.data.mul_(0.5)``, some EMA / weight
    surgery utilities) bypasses the version counter and cannot be seen.  Call this after such a write; the next forward
    re-casts the weights, the next optimizer step re-creates the images."""
    params = module_or_params.parameters() if hasattr(module_or_params, "parameters") else module_or_params
    for p in params:
        for a in ("_mk_shadow", "_mk_shadow_t", "_mk_shadow_version", "_mk_shadow_ptr"):
            if hasattr(p, a):
                delattr(p, a)


def _flat(t, like=None):
    """fp32 view of ``t`` in memory order for the element-wise kernels; ``like``: a tensor it must share strides with"""
    r = dense_view(_real(t))
    if r is None or r.dtype != torch.float32 or (like is not None and t.stride() != like.stride()):
        raise RuntimeError("FusedAdamW needs dense fp32 / complex64 parameters with gradients and state in the same strides")
    return r


def _check_live_shard(p):
    """A ZeRO-sharded parameter whose reduced gradient shard was consumed by ``step()`` still carries the LOCAL, unreduced
    gradient in ``p.grad``: a gradient norm computed from it would be wrong and differ per rank.  Norms are defined between
    ``backward()`` and ``step()``."""
    z = getattr(p, "_mk_zero", None)
    if z is not None and z[0] is None and p.grad is not None:
        raise RuntimeError("gradient norm requested after FusedAdamW.step() consumed the ZeRO gradient shard: p.grad holds "
                           "this rank's unreduced gradient; take the norm between backward() and step()")


class FusedAdamW(torch.optim.Optimizer):
    """``torch.optim.AdamW`` semantics (state keys ``step`` / ``exp_avg`` / ``exp_avg_sq``), executed by
    ``mk_adamw_step``.  ``step(max_grad_norm=...)`` folds makani's global-norm clipping
    (``makani/utils/training/training_helpers.py:123-165``) into the same pass."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @staticmethod
    def zero_shard(p):
        """(gradient shard, group, ranks, rank) when ``GradReducer(zero=True)`` reduce-scattered this parameter's gradient"""
        z = getattr(p, "_mk_zero", None)
        return z if z is not None and z[0] is not None else None

    @torch.no_grad()
    def clip_coef(self, max_grad_norm):
        """(2,) device tensor: [min(1, max_norm / (||g|| + 1e-6)), ||g||] over the gradients of this data-parallel replica:
        whole gradients where they were all-reduced, this rank's shards (summed over the data group) where they were
        reduce-scattered."""
        full, pre, shards, group = [], [], [], None
        for g in self.param_groups:
            for p in g["params"]:
                z = self.zero_shard(p)
                _check_live_shard(p)
                if z is not None:
                    shards.append(z[0])
                    group = z[1]
                elif p.grad is not None:
                    f = _flat(p.grad)
                    part = _ops.grad_ssq_lookup(f, p.grad._version)      # squares already summed by the kernel that wrote this gradient
                    (pre if part is not None else full).append(part if part is not None else f)
        if not shards:
            return _k_sumsq_clip(full, max_grad_norm, pre)
        sq = _k_sumsq_clip(shards, None)[1:].square()
        dist.all_reduce(sq, group=group)
        if full or pre:
            sq = sq + _k_sumsq_clip(full, None, pre)[1:].square()
        norm = sq.sqrt()
        coef = torch.clamp(float(max_grad_norm) / (norm + 1e-6), max=1.0) if max_grad_norm else torch.ones_like(norm)
        return torch.cat([coef, norm])

    @torch.no_grad()
    def grad_norm(self):
        return self.clip_coef(None)[1]

    @torch.no_grad()
    def full_state_dict(self):
        """``state_dict()`` with every ZeRO-sharded moment all-gathered to the parameter's full size (COLLECTIVE over the data
        group: every data rank calls it; what makani's checkpointing then saves from ONE rank — ``driver.py`` writes the
        optimizer state of data rank 0 — restores on any rank and under any data-group size: ``step()`` slices full-size moments
        to the local shard).  Without ZeRO this is ``state_dict()``.  The moments are laid out in the parameter's own memory order
        (as non-ZeRO ``exp_avg`` tensors are)."""
        sd = self.state_dict()
        index = {}
        for gi, group in enumerate(self.param_groups):
            for pi, p in enumerate(group["params"]):
                index[id(p)] = sd["param_groups"][gi]["params"][pi]
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p, {})
                z = getattr(p, "_mk_zero", None)
                if "zero_shard" not in st or z is None:
                    continue
                _, zgroup, nranks, rank = z
                out = dict(sd["state"][index[id(p)]])
                out.pop("zero_shard", None)
                for key in ("exp_avg", "exp_avg_sq"):
                    mine = st[key]
                    if dist.get_backend(zgroup) == "gloo":
                        h = mine.cpu() if mine.is_cuda else mine.clone()
                        parts = [torch.empty_like(h) for _ in range(nranks)]
                        dist.all_gather(parts, h, group=zgroup)
                        flat = torch.cat(parts).to(mine.device)
                    else:
                        flat = torch.empty(mine.numel() * nranks, dtype=mine.dtype, device=mine.device)
                        dist.all_gather_into_tensor(flat, mine, group=zgroup)
                    full = torch.empty_like(p, memory_format=torch.preserve_format)
                    _flat(full, p).reshape(-1).copy_(flat)
                    out[key] = full
                sd["state"][index[id(p)]] = out
        return sd

    def _step_state(self, group, device):
        """device-side step counter + bias corrections of a parameter group (mk_adamw_advance)"""
        st = group.get("_mk_step_state")
        if torch.is_tensor(st) and st.device == device and st.numel() == 3:
            return st
        # No counter on this device: a resumed run.  ``Optimizer.load_state_dict`` does not move param_group tensors to the
        # device (makani loads checkpoints with map_location="cpu", driver.py:436/507), and checkpoints written by
        # torch.optim.AdamW / the round-1 FusedAdamW hold no counter at all.  Restarting at zero would restart the bias
        # correction (updates ~0.3x too small for thousands of steps at beta2 = 0.999), so the count is rebuilt from the
        # saved counter or, failing that, from the per-parameter ``state["step"]`` mirror.
        if torch.is_tensor(st) and st.numel() == 3:
            step = float(st.reshape(-1)[0])
        else:
            steps = [float(self.state[p]["step"]) for p in group["params"] if "step" in self.state.get(p, {})]
            step = max(steps) if steps else 0.0
        st = torch.tensor([step, 0.0, 0.0], dtype=torch.float32, device=device)      # [1], [2] are rewritten by mk_adamw_advance
        group["_mk_step_state"] = st
        return st

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm=None, grad_scale=None):
        """``max_grad_norm``: clip by the global norm of the local gradients; ``grad_scale``: a precomputed
        clipping coefficient (1-element device tensor) when the norm needs cross-rank reduction.

        The step number lives in device memory (one counter per parameter group, advanced by one tiny launch per
        ``step()``); the per-parameter ``state["step"]`` entries are kept as the host-side mirror torch expects.  No
        launch argument depends on the step number, so a captured graph of the train step can be replayed.  All
        parameters of a group that receive gradients share the group's counter (a parameter that skips steps would
        see a slightly different bias correction than torch's per-parameter count: not a case of this path)."""
        scale = grad_scale
        if scale is None and max_grad_norm is not None:
            scale = self.clip_coef(max_grad_norm)[:1]
        for group in self.param_groups:
            b1, b2 = group["betas"]
            live = [p for p in group["params"] if p.grad is not None]
            if not live:
                continue
            live = [p for p in group["params"] if p.grad is not None or self.zero_shard(p) is not None]
            sdev = self._step_state(group, live[0].device)
            _k_advance(sdev, b1, b2)
            descs, keep, shadowed = [], [], []
            for p in live:
                st = self.state[p]
                z = self.zero_shard(p)
                if z is not None:
                    # ZeRO-1: state and update for this rank's slice only, then the parameter is gathered in place
                    gshard, zgroup, nranks, rank = z
                    pr = _flat(p).reshape(-1)
                    chunk = pr.numel() // nranks
                    if gshard.numel() != chunk or p.grad is not None and p.grad.stride() != p.stride():
                        raise RuntimeError("ZeRO gradient shard does not match the parameter (size or memory order)")
                    if not st:
                        st["step"] = 0
                        st["exp_avg"] = torch.zeros(chunk, dtype=torch.float32, device=p.device)
                        st["exp_avg_sq"] = torch.zeros(chunk, dtype=torch.float32, device=p.device)
                    for key in ("exp_avg", "exp_avg_sq"):
                        full = _real(st[key])
                        if full.numel() == pr.numel():       # a non-ZeRO checkpoint (full-size moments): keep this rank's slice
                            st[key] = _flat(st[key], p).reshape(-1)[rank * chunk:(rank + 1) * chunk].clone()
                        elif full.numel() != chunk:
                            raise RuntimeError(f"ZeRO optimizer state of {full.numel()} elements does not fit a shard of {chunk}")
                    if st.setdefault("zero_shard", (nranks, rank)) != (nranks, rank):
                        raise RuntimeError(f"optimizer state belongs to ZeRO shard {st['zero_shard']}, this rank is {(nranks, rank)}: "
                                           "sharded state must be saved and restored per data rank")
                    st["step"] += 1
                    mine = pr[rank * chunk:(rank + 1) * chunk]
                    _k_adamw(mine, gshard, st["exp_avg"], st["exp_avg_sq"], scale, group["lr"], b1, b2, group["eps"],
                             group["weight_decay"], sdev)
                    if dist.get_backend(zgroup) == "gloo":           # (tests) gloo: no in-place gather, host memory only
                        h = mine.cpu() if mine.is_cuda else mine.clone()
                        parts = [torch.empty_like(h) for _ in range(nranks)]
                        dist.all_gather(parts, h, group=zgroup)
                        pr.copy_(torch.cat(parts))
                    else:
                        dist.all_gather_into_tensor(pr, mine, group=zgroup)
                    torch.autograd.graph.increment_version(p)
                    p._mk_zero = (None, zgroup, nranks, rank)         # the shard is consumed
                    continue
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                pr, gr = _flat(p), _flat(p.grad, p)
                m, v = _flat(st["exp_avg"], p), _flat(st["exp_avg_sq"], p)
                if pr.numel() < SMALL:
                    # parameters that asked for it (ops.want_bf16_shadow) get bf16(p) written by the same kernel — as the
                    # (M, round8(K)) operand of the forward GEMM and, transposed, the (K, round8(M)) operand of the
                    # data-gradient GEMM: what the bf16-autocast step would otherwise produce with cast / transpose
                    # kernels per weight
                    sh = sht = None
                    cols = ld = ldt = 0
                    if getattr(p, "_mk_want_bf16", False) and pr.dtype == torch.float32 and p.dim() >= 2:
                        Mr, Kc = p.shape[0], p.numel() // p.shape[0]
                        cols, ld, ldt = Kc, (Kc + 7) // 8 * 8, (Mr + 7) // 8 * 8
                        sh, sht = getattr(p, "_mk_shadow", None), getattr(p, "_mk_shadow_t", None)
                        if sh is None or sh.shape != (Mr, ld) or sh.device != p.device:
                            sh = torch.zeros((Mr, ld), dtype=torch.bfloat16, device=p.device)      # pad columns stay zero
                            sht = torch.zeros((Kc, ldt), dtype=torch.bfloat16, device=p.device)
                            p._mk_shadow, p._mk_shadow_t = sh, sht
                        shadowed.append(p)
                    descs.append(MkAdamTensor(pr.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), pr.numel(),
                                              sh.data_ptr() if sh is not None else None,
                                              sht.data_ptr() if sht is not None else None, cols, ld, ldt))
                    keep.append((pr, gr, m, v, sh, sht))
                else:
                    _k_adamw(pr, gr, m, v, scale, group["lr"], b1, b2, group["eps"], group["weight_decay"], sdev)
                # the update goes through raw pointers: tell autograd the parameter changed in place
                torch.autograd.graph.increment_version(p)
            if descs:
                arr = (MkAdamTensor * len(descs))(*descs)
                from .ops import _timed
                with _timed("adamw_multi", nbytes=28.0 * sum(d.n for d in descs)):
                    check(lib().mk_adamw_multi(C.cast(arr, C.c_void_p), len(descs), ptr(scale), group["lr"], b1, b2, group["eps"],
                                               group["weight_decay"], 0, ptr(sdev), stream()), "mk_adamw_multi")
            for p in shadowed:               # valid for exactly this version (and storage) of the parameter
                p._mk_shadow_version = p._version
                p._mk_shadow_ptr = p.data_ptr()
        _ops.grad_ssq_drop()                 # the producers' sums of squares belong to the gradients of THIS step
        return None
