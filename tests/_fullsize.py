"""Shared test infrastructure of the full-size (721 x 1440 x 73, 384 channels, 8 layers = BASELINE configs[1]) GPU tests.

The CPU oracle's forward + backward of the whole network costs about a minute of host time per pass; two test modules
need it (tests/test_gpu_headline.py: serial HIP against the oracle; tests/test_gpu_dist_fullsize.py: N ranks sharing one
GPU against the serial HIP model AND the oracle), and the worker processes of the second cannot inherit tensors from
the pytest process.  So the oracle's results are computed ONCE per box and kept as a ``torch.save`` file under the
system temp directory; every consumer ``torch.load(..., mmap=True)``s it (a worker touches only the pages of its own
shard).  Nothing here is product code.
"""
import os
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CACHE_DIR = os.environ.get("MAKANI_AMD_TEST_CACHE", os.path.join(tempfile.gettempdir(), "makani_amd_test_cache"))

# BASELINE configs[1] = SURVEY.md Appendix D (config/sfnonet.yaml:24-40)
CONFIG2 = dict(inp_shape=(721, 1440), out_shape=(721, 1440), inp_chans=73, out_chans=73, scale_factor=3, embed_dim=384,
               num_layers=8, mlp_ratio=2, operator_type="dhconv", normalization_layer="instance_norm",
               activation_function="gelu", big_skip=True, model_grid_type="equiangular", sht_grid_type="legendre-gauss")


def share_gpu(rank, world, ncu=256):
    """Round 5 gave the ranks that share ONE GPU in a test disjoint compute units (``HSA_CU_MASK``) because kernels of different
    processes on one compute unit disturbed each other's results.  Round 6 found the cause — packed-fp32 instructions that read a
    VGPR src1 through op_sel are unreliable on gfx950 while certain matrix-core kernels share the compute unit
    (docs/LAB_NOTEBOOK.md 6.1, tools/pk_hazard_probe.py) — and removed those instruction forms from every kernel
    (tools/pk_opsel_scan.py, tests/test_packed_forms.py).  The tests therefore run WITHOUT the mask: N ranks on one GPU share
    all compute units, which is the stronger check.  ``MAKANI_AMD_TEST_CU_MASK=1`` brings the mask back (must be in the
    environment before the process's first GPU call)."""
    if os.environ.get("MAKANI_AMD_TEST_CU_MASK", "0") != "1":
        return
    per = max(1, ncu // max(1, world))
    os.environ["HSA_CU_MASK"] = f"0:{rank * per}-{(rank + 1) * per - 1}"


def host_threads(cap=64):
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(n, cap)))


def perturb_affine(mod, seed):
    """non-trivial norm weights and biases (the initial ones are 1 / 0, which hides a wrong bias or affine path)"""
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in mod.named_parameters():
            if n.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=gen) * 0.1)
            elif ".norm" in n or n.startswith("norm"):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=gen))


def _save_atomic(obj, path):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    tmp = f"{path}.{os.getpid()}.tmp"
    torch.save(obj, tmp)
    os.replace(tmp, path)


def cached(name, compute):
    """``compute()`` -> a (nested) dict of CPU tensors, saved under CACHE_DIR/name and returned memory-mapped"""
    path = os.path.join(CACHE_DIR, name)
    if not os.path.exists(path):
        t0 = time.time()
        obj = compute()
        _save_atomic(obj, path)
        del obj
        print(f"[fullsize cache] {name}: computed and saved in {time.time() - t0:.0f} s")
    return torch.load(path, mmap=True, weights_only=True)


def _compute_config2_oracle():
    from oracle import sfno as osf
    host_threads()
    torch.manual_seed(333)
    omod = osf.SphericalFourierNeuralOperatorNet(**CONFIG2)
    perturb_affine(omod, 7)
    x = torch.rand(1, 73, 721, 1440)                    # DummyLoader-shaped U[0, 1) input (data_loader_dummy.py:264-277)
    g = torch.randn(1, 73, 721, 1440, generator=torch.Generator().manual_seed(99))
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):       # the reference's own op-by-op bf16 autocast, on the CPU
        yo_bf16 = omod(x).float()
    xo = x.clone().requires_grad_(True)
    # the bias gradients of the full-resolution convolutions are sums over 1 038 240 pixels: next to the fp32 sums autograd
    # returns, keep the SAME output gradients summed in fp64 (hooks on the same backward pass) — the yardstick that tells how much
    # of a bias-gradient difference is the oracle's own fp32 summation (VERDICT r5 weak #1a)
    bias64, hooks = {}, []
    for name, mod in omod.named_modules():
        if isinstance(mod, torch.nn.Conv2d) and mod.bias is not None:
            hooks.append(mod.register_full_backward_hook(
                lambda m, gi, go, name=name: bias64.__setitem__(name + ".bias", go[0].double().sum(dim=(0, 2, 3)))))
    yo = omod(xo)
    (yo * g).sum().backward()
    for h in hooks:
        h.remove()
    out = dict(bias_grads_fp64=bias64, state={k: v.detach().clone() for k, v in omod.state_dict().items()}, x=x, g=g, y=yo.detach(), y_bf16=yo_bf16,
               gx=xo.grad.detach(), grads={n: p.grad.detach().clone() for n, p in omod.named_parameters()})
    # ... and the reference's own bf16 arithmetic through the BACKWARD pass (op-by-op CPU bf16 autocast): the yardstick of the
    # bf16 gradient gates, as y_bf16 is of the forward gate
    omod.zero_grad(set_to_none=True)
    xb = x.clone().requires_grad_(True)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        yb = omod(xb)
    (yb.float() * g).sum().backward()
    out["bf16_gx"] = xb.grad.detach()
    out["bf16_grads"] = {n: p.grad.detach().clone() for n, p in omod.named_parameters()}
    return out


def _oracle_fingerprint():
    """what the cached results depend on: the oracle's sources, the configuration, this file's recipe (seeds) and the torch
    version — a change of any of them computes a new file instead of silently comparing against stale results"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "oracle", "*.py"))) + [os.path.abspath(__file__)]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(repr(sorted(CONFIG2.items())).encode())
    h.update(torch.__version__.encode())
    return h.hexdigest()[:12]


def config2_oracle():
    """the oracle side of the whole-network tests at 721 x 1440 x 73: state dict, input x, cotangent g, forward (fp32 and the
    reference's own CPU bf16 autocast), ONE backward pass of sum(y * g) in fp32 and one under bf16 autocast — input gradient
    and the gradient of every parameter (about three minutes on the GPU box's host the first time, then a file)"""
    return cached(f"config2_oracle_{_oracle_fingerprint()}.pt", _compute_config2_oracle)


def spawn(fn, args, nprocs, timeout_s):
    """``mp.spawn`` with a deadline: a schedule bug between ranks is a hang, and a hung GPU box is worse than a red test"""
    import torch.multiprocessing as mp
    ctx = mp.spawn(fn, args=args, nprocs=nprocs, join=False)
    deadline = time.time() + timeout_s
    while True:
        left = deadline - time.time()
        if left <= 0:
            for p in ctx.processes:
                if p.is_alive():
                    p.kill()
            raise TimeoutError(f"{fn.__name__}: {nprocs} ranks did not finish within {timeout_s} s")
        if ctx.join(timeout=min(left, 5.0)):
            return


def log_line(text):
    """measured numbers of the full-size distributed tests: appended to gpurun_out/dist_fullsize.txt (copied to profiles/)"""
    path = os.environ.get("MAKANI_AMD_DIST_LOG", os.path.join(ROOT, "gpurun_out", "dist_fullsize.txt"))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "a") as f:
            f.write(text.rstrip("\n") + "\n")
    except OSError:
        pass
    print(text, flush=True)
