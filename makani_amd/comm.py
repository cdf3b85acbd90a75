"""Process-group tree ``world -> data x (h x w)`` with the accessor names of ``makani/utils/comm.py:26-112``
(``get_size / get_rank / get_group / get_comm_names / is_distributed``).

Two ways to fill it:

* ``init(h, w)`` builds the tree on top of an initialised ``torch.distributed`` world exactly as
  ``makani/utils/comm.py:114-201`` lays ranks out (model instance ``d`` owns the ranks
  ``[d*h*w, (d+1)*h*w)``, h-major) — what ``bench.py`` and the tests use;
* ``adopt(comm_like)`` takes the groups from an object with makani's accessors; ``autodetect()`` does that with
  ``makani.utils.comm`` itself when makani is importable and initialised, so a plug-in network constructed by makani's
  ``Trainer`` finds makani's own ``h`` / ``w`` / ``spatial`` / ``data`` groups without any extra call.

One process per GPU; every collective of the package goes through the groups registered here (RCCL = backend
``"nccl"`` on ROCm, ``gloo`` in the CPU tests).
"""
from typing import Dict, Optional, Tuple

import torch.distributed as dist

# name -> (process group or None, size, rank in group)
_GROUPS: Dict[str, Tuple[Optional[object], int, int]] = {}
_SOURCE = None          # "init", "adopt" or None

MODEL_COMM_NAMES = ("h", "w", "spatial", "matmul", "fin", "fout", "model")


def share_gpu(rank_on_gpu: int, ranks_on_gpu: int, ncu: int = 256) -> str:
    """Give every rank that shares ONE GPU its own range of compute units (``HSA_CU_MASK``, read by the ROCm runtime when the
    process creates its queues — call this BEFORE the first GPU call of the process).  Returns the mask it set.

    History: round 5 needed this for functional runs with several ranks on one GPU — bf16 instance-norm backward and the
    inverse FFT returned transiently wrong values next to another process's matrix kernels.  Round 6 reproduced the effect in
    ONE process on two streams and traced it to an instruction form, not to process sharing: ``v_pk_mul/add/fma_f32`` with a
    VGPR src1 read through ``op_sel`` (low result lane <- src0.lo x src1.hi) returns wrong results on gfx950 while the split-bf16
    GEMM runs on the same compute unit (docs/LAB_NOTEBOOK.md 6.1; minimal repro: tools/pk_hazard_probe.py).  No kernel of the
    package contains that form any more (tools/pk_opsel_scan.py is part of the CPU test suite), the tests run without masks, and
    this helper remains for experiments that want the ranks of a shared GPU isolated (e.g. timing one rank's kernels)."""
    import os
    per = max(1, ncu // max(1, ranks_on_gpu))
    mask = f"0:{rank_on_gpu * per}-{(rank_on_gpu + 1) * per - 1}"
    os.environ["HSA_CU_MASK"] = mask
    return mask


def reset():
    global _SOURCE
    _GROUPS.clear()
    _SOURCE = None


def is_initialized() -> bool:
    return _SOURCE is not None


def get_size(name: str) -> int:
    return _GROUPS[name][1] if name in _GROUPS else 1


def get_rank(name: str) -> int:
    return _GROUPS[name][2] if name in _GROUPS else 0


def get_group(name: str):
    return _GROUPS[name][0] if name in _GROUPS else None


def get_comm_names():
    return list(_GROUPS)


def get_model_comm_names():
    return [n for n in _GROUPS if n not in ("world", "data", "ensemble", "batch")]


def is_distributed(name: str) -> bool:
    return name in _GROUPS


def get_world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_world_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _register(name, group, members, rank):
    _GROUPS[name] = (group if len(members) > 1 else None, len(members), members.index(rank) if rank in members else 0)


def init(h: int = 1, w: int = 1, ensemble: int = 1):
    """Build ``data x (h x w)`` over the current world.  Every rank creates every group in the same order
    (``dist.new_group`` is collective over the world) and keeps its own.  Groups of one rank are ``None``.
    ``ensemble``: the data group is ``batch x ensemble`` (makani's ``data_parallel_names = ["ensemble", "batch"]``,
    ``makani/utils/comm.py:114-201``): data index d = batch index * ensemble + ensemble index."""
    global _SOURCE
    reset()
    world, rank = get_world_size(), get_world_rank()
    msize = h * w
    if world % msize:
        raise ValueError(f"world size {world} is not a multiple of h*w = {msize}")
    dsize = world // msize
    if ensemble < 1 or dsize % ensemble:
        raise ValueError(f"the data group of {dsize} ranks cannot be split into ensembles of {ensemble}")
    d_idx, m_idx = rank // msize, rank % msize
    ih, iw = m_idx // w, m_idx % w

    def make(members):
        return dist.new_group(members) if (world > 1 and len(members) > 1) else None

    mine = {}
    for d in range(dsize):
        base = d * msize
        members = list(range(base, base + msize))
        g = make(members)
        if d == d_idx:
            mine["spatial"] = (g, members)
        for j in range(w):
            members = [base + i * w + j for i in range(h)]
            g = make(members)
            if d == d_idx and j == iw:
                mine["h"] = (g, members)
        for i in range(h):
            members = [base + i * w + j for j in range(w)]
            g = make(members)
            if d == d_idx and i == ih:
                mine["w"] = (g, members)
    for m in range(msize):
        members = [d * msize + m for d in range(dsize)]
        g = make(members)
        if m == m_idx:
            mine["data"] = (g, members)
    nb = dsize // ensemble
    for m in range(msize):
        for ib in range(nb):                                 # ensemble groups: same model rank, same batch index
            members = [(ib * ensemble + ie) * msize + m for ie in range(ensemble)]
            g = make(members)
            if rank in members:
                mine["ensemble"] = (g, members)
        for ie in range(ensemble):                           # batch groups: same model rank, same ensemble index
            members = [(ib * ensemble + ie) * msize + m for ib in range(nb)]
            g = make(members)
            if rank in members:
                mine["batch"] = (g, members)
    for name in ("h", "w", "spatial", "data", "ensemble", "batch"):
        g, members = mine[name]
        _register(name, g, members, rank)
    _GROUPS["model"] = _GROUPS["spatial"]            # no matmul / feature parallelism on this path: model = h x w
    for name in ("matmul", "fin", "fout"):
        _GROUPS[name] = (None, 1, 0)
    _SOURCE = "init"
    return d_idx, ih, iw


def adopt(comm_like):
    """Register the groups of an object exposing makani's accessors (``makani.utils.comm`` itself)."""
    global _SOURCE
    reset()
    names = list(comm_like.get_comm_names())
    for name in set(names) | {"h", "w", "spatial", "data", "model", "matmul"}:
        size = comm_like.get_size(name)
        _GROUPS[name] = (comm_like.get_group(name) if size > 1 else None, size, comm_like.get_rank(name))
    _SOURCE = "adopt"


def autodetect() -> bool:
    """Adopt ``makani.utils.comm`` when makani is importable and its distributed manager is up (the network is being
    constructed by makani's own driver).  Returns whether a tree is registered afterwards."""
    if is_initialized():
        return True
    import sys
    mk = sys.modules.get("makani.utils.comm")
    if mk is None:
        return False
    try:
        if mk.get_world_size() > 1 or getattr(mk, "_DM", None) is not None:
            adopt(mk)
            return True
    except Exception:
        return False
    return False
