"""The exchange schedule of the h x w distributed spherical harmonic transforms, built for xGMI (point-to-point links
between every pair of GPUs) and for kernels that address their operands per peer.

Reference schedule (torch-harmonics' DistributedRealSHT / DistributedInverseRealSHT [un-vendored]; in-tree twin
``makani/mpu/fft.py:148-182,214-249`` on ``makani/mpu/mappings.py:38-67``): four transposes per transform, each one
``split -> contiguous -> all_to_all -> cat``: two full copies of the tensor around every collective, and between the
longitude FFT and the Legendre transform the spectrum crosses the fabric TWICE (m <-> planes over the azimuth group, then
planes <-> latitude over the polar group).

This schedule (one rank = (ih, iw); P planes = batch x channels):

  analysis   x (P, lat_loc, lon_loc)
     (1) planes <-> longitude over the azimuth group: plane slabs leave as views (planes are outermost), the arriving pieces
         [peer][plane][lat][lon piece] are NOT concatenated: the FFT kernel reads a row from its w pieces
     (2) truncated rFFT -> writes, per destination rank (ih', iw') of the SPATIAL group, one contiguous slab
         [lat][m in M(iw')][re/im][planes of sub-block (iw, ih')]   (``MkFftSeg``, csrc/fft_fast.hip)
     (3) ONE all-to-all over the h x w spatial group replaces the two middle transposes: every coefficient crosses the
         fabric once ((hw - 1) / hw of the tensor instead of (w - 1) / w + (h - 1) / h: -30 % at h4 w2), and because latitude is the
         outermost index of a slab, the slabs of the ranks of one polar group land next to each other as the Legendre
         operand (latitude-major F) — no concatenation
     (4) Legendre analysis per plane block on the latitude-major operand (same GEMM kernel, other strides)
     (5) l <-> planes over the polar group, one exchange per plane block (l slabs are views); the arriving plane sub-blocks
         are joined into the channel-contiguous S layout the contraction kernels read: the one copy left on this side
  synthesis  the mirror image: plane sub-blocks are packed once out of the S layout (5'), Legendre synthesis writes
         latitude-major F (4') whose latitude ranges ARE the send slabs of the spatial all-to-all (3'), the inverse FFT
         reads its receive buffer per peer (2') and writes longitude pieces that leave as they are (1').

Overlap: steps (2)+(3) and (3')+(2') are cut into latitude chunks — transform chunk c+1 while chunk c is on the links —,
step (4)/(5) and (5')/(4') overlap across plane blocks.  Backward is the adjoint pipeline of the other direction with the
transposed matrices, scheduled the same way (both pipelines are explicit: one autograd node per transform).

Local compute goes through the backend object of ``makani_amd.distributed`` (``HipBackend``: the C ABI; tests install a
torch implementation of the same four operations to check the schedule with gloo on the CPU).
"""
import os

import torch
import torch.distributed as dist

from . import ops


def _offsets(sizes):
    out = [0]
    for s in sizes:
        out.append(out[-1] + s)
    return out


def _split(size, n):
    """ceil-div chunks, the last one smaller; floor-div fallback when the last would be empty (the reference's rule)"""
    if n == 1:
        return [size]
    chunk = (size + n - 1) // n
    last = max(0, size - chunk * (n - 1))
    if last == 0:
        chunk = size // n
        last = size - chunk * (n - 1)
    return [chunk] * (n - 1) + [last]


class Plan:
    """Sizes and slab offsets of one transform shape on this rank."""

    def __init__(self, T, P):
        self.P = P
        self.h, self.w, self.ih, self.iw = T.comm_size_polar, T.comm_size_azimuth, T.comm_rank_polar, T.comm_rank_azimuth
        h, w = self.h, self.w
        self.nlat, self.nlon, self.L, self.M = T.nlat, T.nlon, T.lmax, T.mmax
        self.lat, self.lon, self.l_shapes, self.m_shapes = T.lat_shapes, T.lon_shapes, T.l_shapes, T.m_shapes
        self.hl, self.wl = self.lat[self.ih], self.lon[self.iw]
        self.Ml, self.Ll = self.m_shapes[self.iw], self.l_shapes[self.ih]
        self.pw = _split(P, w)                                            # plane block of azimuth rank j (FFT phase)
        self.poff = _offsets(self.pw)
        # sub-blocks of block j over the polar group (Legendre phase), whole groups of 4 rows (16-byte row vectors)
        self.sub = [[4 * g for g in _split((self.pw[j] + 3) // 4, h)] for j in range(w)]
        self.suboff = [_offsets(s) for s in self.sub]
        self.valid = [[max(0, min(self.sub[j][i], self.pw[j] - self.suboff[j][i])) for i in range(h)] for j in range(w)]
        # send slabs of the forward FFT / receive slabs of the inverse FFT: spatial rank s = i * w + j, slab (j, i) =
        # [lat_loc][M(j)][2][sub(iw, i)]
        self.slab = [[self.hl * self.m_shapes[j] * 2 * self.sub[self.iw][i] for i in range(h)] for j in range(w)]
        base, off = [[0] * h for _ in range(w)], 0
        for i in range(h):
            for j in range(w):
                base[j][i] = off
                off += self.slab[j][i]
        self.base, self.f_total = base, off
        # every azimuth rank's plane block splits alike over the polar group (always, unless the plane count is ragged over w):
        # the w plane blocks of the Legendre phase then run as ONE batched GEMM and leave / arrive in ONE polar exchange
        self.uniform = all(sj == self.sub[0] for sj in self.sub) and os.environ.get("MAKANI_AMD_DIST_BLOCKS", "1") == "1"

        # latitude chunks of steps (2)+(3): the same count on every rank (each collective is entered by all of them) — the
        # count depends on an environment variable, so the ranks of the transform's groups compare it once per plan: ranks
        # that disagree would enter different numbers of collectives and hang
        nmin = min(self.lat)
        self.nc = max(1, min(int(os.environ.get("MAKANI_AMD_DIST_CHUNKS", "2")), nmin // 32 if nmin >= 64 else 1))
        lo, hi = agree_min_max(self.nc)
        if lo != hi:
            raise RuntimeError(f"MAKANI_AMD_DIST_CHUNKS differs between the ranks of the spatial group (chunk counts {lo}..{hi}): "
                               "every rank must run the same exchange schedule")

    def chunks(self, n):
        """latitude chunk boundaries of a rank with n local latitudes"""
        return [(c * n) // self.nc for c in range(self.nc + 1)]


def agree_min_max(value: int):
    """(min, max) of an integer over the ranks of the polar and the azimuth group (both all-reduces are entered by every rank
    of the h x w block, whatever its local value).  One tiny collective per group, at plan / eligibility time only."""
    from . import distributed as thd
    lo = hi = int(value)
    for group in (thd.polar_group(), thd.azimuth_group()):
        if group is None or dist.get_world_size(group) == 1:
            continue
        dev = "cpu" if dist.get_backend(group) == "gloo" else torch.device("cuda", torch.cuda.current_device())
        t = torch.tensor([-lo, hi], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        lo, hi = -int(t[0]), int(t[1])
    return lo, hi


FALLBACKS = []            # (transform class, grid, dtype, reason) of every transform that left the fused schedule


def eligible(T, x_dtype) -> bool:
    """the fused schedule needs the specialised FFT kernels, equal longitude pieces of whole 16-byte vectors and at most
    MK_FFT_SEG_MAX peers per direction; anything else runs the transpose-by-transpose schedule of distributed.py.
    The decision depends on environment variables and on the local group layout, so it is AGREED over the transform's groups
    the first time it is taken: the fused schedule runs only if every rank finds it eligible — ranks that decide
    differently would enter different collective sequences and hang instead of raising.
    A transform of a split group that does NOT run the fused schedule says so once (``logging`` warning + ``FALLBACKS``):
    the transpose-by-transpose schedule is correct but moves every coefficient twice, and a scaling run should not fall
    into it unnoticed."""
    from . import distributed as thd
    key = (x_dtype, os.environ.get("MAKANI_AMD_DIST_FUSED", "1"), os.environ.get("MAKANI_AMD_DIST_FORCE_FUSED", "0"), id(thd._BACKEND))
    cache = T.__dict__.setdefault("_fused_ok", {})
    if key not in cache:
        why = _ineligible(T, x_dtype)
        ok = why is None
        if T.comm_size_polar * T.comm_size_azimuth > 1:
            agreed = bool(agree_min_max(1 if ok else 0)[0])
            if ok and not agreed:
                why = "another rank of the transform's groups found the fused schedule ineligible"
            ok = agreed
            if not ok:
                rec = (type(T).__name__, f"{T.nlat}x{T.nlon}", str(x_dtype), why)
                FALLBACKS.append(rec)
                import logging
                logging.getLogger("makani_amd.dist").warning(
                    "%s %s (%s, h%d w%d) runs the transpose-by-transpose schedule, not the fused one: %s",
                    rec[0], rec[1], rec[2], T.comm_size_polar, T.comm_size_azimuth, why)
        cache[key] = ok
    return cache[key]


def _ineligible(T, x_dtype):
    """None when this rank can run the fused schedule for ``T``, else the reason it cannot"""
    from . import _lib
    from . import distributed as thd
    if os.environ.get("MAKANI_AMD_DIST_FUSED", "1") != "1":
        return "MAKANI_AMD_DIST_FUSED=0"
    if not getattr(thd._BACKEND, "segmented", False):
        return "the compute backend has no segmented FFT kernels"
    h, w = T.comm_size_polar, T.comm_size_azimuth
    # MAKANI_AMD_DIST_FORCE_FUSED=1: the fused pipeline also with ONE rank (h = w = 1) — every collective it issues (the list
    # all_to_all with async_op=True on contiguous slab views, the waits that order the compute stream behind it) then runs on a
    # process group of one rank: how the RCCL call signatures and the stream ordering are exercised on a one-GPU box
    force = os.environ.get("MAKANI_AMD_DIST_FORCE_FUSED", "0") == "1" and dist.is_available() and dist.is_initialized()
    if h > _lib.MK_FFT_SEG_MAX or w > _lib.MK_FFT_SEG_MAX:
        return f"more than MK_FFT_SEG_MAX = {_lib.MK_FFT_SEG_MAX} peers in one direction"
    if h * w == 1 and not force:
        return "one rank"
    if len(set(T.lon_shapes)) != 1:
        return f"unequal longitude pieces {T.lon_shapes}"
    ev = 8 if x_dtype == torch.bfloat16 else 4
    if T.lon_shapes[0] % ev:
        return f"longitude pieces of {T.lon_shapes[0]} points are not whole 16-byte vectors of {x_dtype}"
    if not thd._BACKEND.seg_supported(T.nlon):
        return f"no segmented FFT kernel for {T.nlon} longitudes"
    if h > 1 and w > 1:
        try:
            sp = thd.spatial_group()
        except ValueError:
            return "h and w are split but no spatial (h x w) group was given to init()"
        if sp is None or dist.get_world_size(sp) != h * w or dist.get_rank(sp) != T.comm_rank_polar * w + T.comm_rank_azimuth:
            return "the spatial group is not the h-major h x w block of the polar and azimuth groups"
    return None


def _eligible(T, x_dtype) -> bool:
    return _ineligible(T, x_dtype) is None


# --------------------------------------------------------------------------- #
# exchanges: lists of contiguous views in, a handle to wait on out
# --------------------------------------------------------------------------- #
class _Done:
    def wait(self):
        return None


def _exchange_async(recv, send, group):
    """all_to_all of two lists of contiguous tensors (flat views).  RCCL: asynchronous (the collective runs on the process
    group's stream; ``wait`` orders the current stream behind it).  gloo (CPU / one-GPU tests): served synchronously."""
    from .distributed import _exchange
    assert all(t.is_contiguous() for t in recv)                  # (a reshape of a strided tensor would receive into a copy)
    recv = [t.reshape(-1) for t in recv]
    send = [t.reshape(-1) for t in send]
    from .distributed import _count_exchange
    _count_exchange(send, group)
    if dist.get_backend(group) != "gloo":
        if recv[0].is_cuda and torch.cuda.is_current_stream_capturing():
            # inside a hipGraph capture (bench.py captures the whole h x w step): the SYNCHRONOUS form.  With RCCL 2.26 / torch 2.10
            # the send/recv-based collectives are captured and replayed correctly when issued synchronously, while their
            # async_op=True handles crash at the end of the capture (tools/probes/rccl_graph_probe.py, gpurun_out/r05l); the graph
            # then orders the collective between its neighbours on the capturing stream (no chunk overlap inside a replayed graph
            # until that is fixed upstream — the launch path it removes is worth far more: profiles/r05_shard_shapes.md)
            dist.all_to_all(recv, send, group=group)
            return _Done()
        return dist.all_to_all(recv, send, group=group, async_op=True)
    _exchange(recv, send, group)
    return _Done()


def _spatial_like_group(p: Plan):
    """the group of step (3): the spatial group, or the only split group when the other direction has one rank"""
    from . import distributed as thd
    if p.h > 1 and p.w > 1:
        return thd.spatial_group()
    return thd.polar_group() if p.h > 1 else thd.azimuth_group()


# --------------------------------------------------------------------------- #
# the two pipelines
# --------------------------------------------------------------------------- #
def analysis_shaped(x, p: Plan, matT, wts, m_off):
    """x (P, lat_loc, lon_loc) f32 | bf16 -> S (L_loc, M_loc, 2, round4(P)) through steps (1)-(5) with the Legendre matrix
    ``matT`` (M_loc, nlat, Lp) (the forward transform's quadrature-weighted matrix, or — backward of the inverse transform —
    the unweighted one) and the FFT weights ``wts``"""
    from . import distributed as thd
    be = thd._BACKEND
    h, w, ih, iw = p.h, p.w, p.ih, p.iw
    dev, Pw = x.device, p.pw[iw]
    fdt = torch.float64 if x.dtype == torch.float64 else torch.float32      # spectral side: fp32 (fp64 only in the CPU schedule tests)
    # (1) planes <-> longitude
    if w > 1:
        xbuf = torch.empty((w, Pw, p.hl, p.wl), dtype=x.dtype, device=dev)
        _exchange_async([xbuf[j] for j in range(w)], [x[p.poff[j]:p.poff[j + 1]] for j in range(w)], thd.azimuth_group()).wait()
    else:
        xbuf = x.reshape(1, Pw, p.hl, p.wl)
    # (2) + (3) in latitude chunks: transform chunk c + 1 while chunk c travels
    fs = torch.empty((p.f_total,), dtype=fdt, device=dev)
    blocks = p.uniform and hasattr(be, "analysis_lm_blocks")          # (the same decision on every rank: it shapes the exchanges)
    if blocks:        # one buffer for the w plane blocks: the Legendre phase is ONE launch over all of them
        Gall = torch.empty((w, p.nlat, p.Ml, 2, p.sub[0][ih]), dtype=fdt, device=dev)
        G = [Gall[j] for j in range(w)]
    else:
        G = [torch.empty((p.nlat, p.Ml, 2, p.sub[j][ih]), dtype=fdt, device=dev) for j in range(w)]
    lat_off = _offsets(p.lat)
    mine = p.chunks(p.hl)
    theirs = [p.chunks(n) for n in p.lat]
    group, works = _spatial_like_group(p), []
    for c in range(p.nc):
        a, b = mine[c], mine[c + 1]
        if Pw > 0 and b > a:                          # (a rank of the azimuth group may hold no plane: fewer planes than ranks)
            be.rfft_seg(xbuf, a, b, fs, p, wts)
        send, recv = [], []
        for i in range(h):
            for j in range(w):
                row = p.m_shapes[j] * 2 * p.sub[iw][i]
                send.append(fs[p.base[j][i] + a * row: p.base[j][i] + b * row])
                recv.append(G[j][lat_off[i] + theirs[i][c]: lat_off[i] + theirs[i][c + 1]])
        works.append(_exchange_async(recv, send, group))
    for wk in works:
        wk.wait()
    l_off = _offsets(p.l_shapes)
    pad = (-p.P) % 4
    if blocks:
        # (4) + (5), all plane blocks at once: one GEMM (L, w, M_loc, 2, sub) — degree outermost —, one polar exchange whose
        # l slabs are contiguous views over all blocks; the arriving [l][j][m][2][sub] pieces are joined into the
        # channel-contiguous S layout (the one copy on this side)
        Sall = be.analysis_lm_blocks(Gall, matT, p.L, m_off) if p.sub[0][ih] > 0 else torch.empty((p.L, w, p.Ml, 2, 0), dtype=fdt, device=dev)
        if h > 1:
            rec = [torch.empty((p.Ll, w, p.Ml, 2, p.sub[0][i]), dtype=fdt, device=dev) for i in range(h)]
            _exchange_async(rec, [Sall[l_off[i]:l_off[i + 1]] for i in range(h)], thd.polar_group()).wait()
        else:
            rec = [Sall]
        parts = [rec[i][:, j, :, :, :p.valid[j][i]] for j in range(w) for i in range(h) if p.valid[j][i] > 0]
        if pad:
            parts.append(torch.zeros((p.Ll, p.Ml, 2, pad), dtype=fdt, device=dev))
        return torch.cat(parts, dim=3) if len(parts) > 1 else parts[0].contiguous()
    # (4) + (5) per plane block: Legendre of block j + 1 while the l <-> planes exchange of block j travels
    pieces, works = [], []
    for j in range(w):
        if p.sub[j][ih] == 0:
            Sj = torch.empty((p.L, p.Ml, 2, 0), dtype=fdt, device=dev)
        else:
            Sj = be.analysis_lm(G[j], matT, p.L, m_off)                                   # (L, M_loc, 2, sub(j, ih))
        rec = [torch.empty((p.Ll, p.Ml, 2, p.sub[j][i]), dtype=fdt, device=dev) for i in range(h)]
        if h > 1:
            works.append(_exchange_async(rec, [Sj[l_off[i]:l_off[i + 1]] for i in range(h)], thd.polar_group()))
        else:
            rec = [Sj]
        pieces.append(rec)
    for wk in works:
        wk.wait()
    parts = [pieces[j][i][..., :p.valid[j][i]] for j in range(w) for i in range(h) if p.valid[j][i] > 0]
    if pad:
        parts.append(torch.zeros((p.Ll, p.Ml, 2, pad), dtype=fdt, device=dev))
    return torch.cat(parts, dim=3) if len(parts) > 1 else parts[0].contiguous()


def synthesis_shaped(S, p: Plan, mat, wts, m_off, out_dtype):
    """S (L_loc, M_loc, 2, >= P) -> x (P, lat_loc, lon_loc) through steps (5')-(1') with the Legendre matrix ``mat``
    (M_loc, L, Kp)"""
    from . import distributed as thd
    be = thd._BACKEND
    h, w, ih, iw = p.h, p.w, p.ih, p.iw
    dev, Pw = S.device, p.pw[iw]
    fdt = S.dtype
    l_off, lat_off = _offsets(p.l_shapes), _offsets(p.lat)
    group = _spatial_like_group(p)
    mine = p.chunks(p.hl)
    theirs = [p.chunks(n) for n in p.lat]
    fr = torch.empty((p.f_total,), dtype=fdt, device=dev)
    blocks = p.uniform and hasattr(be, "synthesis_lm_blocks")
    if blocks:
        # (5') + (4'), all plane blocks at once: the plane sub-blocks are packed out of the S layout into per-peer pieces
        # [l][j][m][2][sub] (the one pack on this side), ONE polar exchange delivers them as l slabs of T (L, w, M_loc, 2, sub),
        # ONE GEMM writes the latitude-major F of every block
        send = []
        for i in range(h):
            sb = p.sub[0][i]
            buf = torch.empty((p.Ll, w, p.Ml, 2, sb), dtype=fdt, device=dev)
            for j in range(w):
                r0, v = p.poff[j] + p.suboff[j][i], p.valid[j][i]
                if v > 0:
                    buf[:, j, :, :, :v].copy_(S[..., r0:r0 + v])
                if v < sb:
                    buf[:, j, :, :, v:].zero_()
            send.append(buf)
        if h > 1:
            Tall = torch.empty((p.L, w, p.Ml, 2, p.sub[0][ih]), dtype=fdt, device=dev)
            _exchange_async([Tall[l_off[i]:l_off[i + 1]] for i in range(h)], send, thd.polar_group()).wait()
        else:
            Tall = send[0]
        Gall = be.synthesis_lm_blocks(Tall, mat, p.nlat, m_off) if p.sub[0][ih] > 0 \
            else torch.empty((w, p.nlat, p.Ml, 2, 0), dtype=fdt, device=dev)        # (w, nlat, M_loc, 2, sub)
        G = [Gall[j] for j in range(w)]
    else:
        # (5') planes <-> l per plane block (the one pack on this side), (4') Legendre synthesis into latitude-major F
        Ts, works = [], []
        for j in range(w):
            Tj = torch.empty((p.L, p.Ml, 2, p.sub[j][ih]), dtype=fdt, device=dev)
            send = []
            for i in range(h):
                r0 = p.poff[j] + p.suboff[j][i]
                blk = S[..., r0:r0 + p.valid[j][i]]
                if p.valid[j][i] != p.sub[j][i]:
                    blk = torch.nn.functional.pad(blk, (0, p.sub[j][i] - p.valid[j][i]))
                send.append(blk.contiguous())
            if h > 1:
                works.append(_exchange_async([Tj[l_off[i]:l_off[i + 1]] for i in range(h)], send, thd.polar_group()))
            else:
                Tj = send[0]
            Ts.append(Tj)
        G = []
        for j in range(w):
            if h > 1:
                works[j].wait()
            G.append(be.synthesis_lm(Ts[j], mat, p.nlat, m_off) if p.sub[j][ih] > 0
                     else torch.empty((p.nlat, p.Ml, 2, 0), dtype=fdt, device=dev))
    # (3') + (2') in latitude chunks of the DESTINATION: inverse FFT of chunk c while chunk c + 1 travels
    xbuf = torch.empty((w, Pw, p.hl, p.wl), dtype=out_dtype, device=dev)
    works = []
    for c in range(p.nc):
        send, recv = [], []
        a, b = mine[c], mine[c + 1]
        for i in range(h):
            for j in range(w):
                send.append(G[j][lat_off[i] + theirs[i][c]: lat_off[i] + theirs[i][c + 1]])
                row = p.m_shapes[j] * 2 * p.sub[iw][i]
                recv.append(fr[p.base[j][i] + a * row: p.base[j][i] + b * row])
        works.append(_exchange_async(recv, send, group))
    for c in range(p.nc):
        works[c].wait()
        if Pw > 0 and mine[c + 1] > mine[c]:
            be.irfft_seg(fr, mine[c], mine[c + 1], xbuf, p, wts)
    # (1') longitude <-> planes: pieces leave as they are, plane slabs arrive in place
    if w > 1:
        x = torch.empty((p.P, p.hl, p.wl), dtype=out_dtype, device=dev)
        _exchange_async([x[p.poff[j]:p.poff[j + 1]] for j in range(w)], [xbuf[j] for j in range(w)], thd.azimuth_group()).wait()
        return x
    return xbuf[0]


# --------------------------------------------------------------------------- #
# autograd: one node per transform, backward = the other pipeline with the transposed matrix
# --------------------------------------------------------------------------- #
class DistAnalysisFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, T, plan):
        ctx.T, ctx.plan, ctx.dtype = T, plan, x.dtype
        return analysis_shaped(x.contiguous(), plan, T.weights_t, T._w, T.m_off)

    @staticmethod
    def backward(ctx, gS):
        T = ctx.T
        return synthesis_shaped(gS.contiguous(), ctx.plan, T.weights, T._w, T.m_off, ctx.dtype), None, None


class DistSynthesisFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, S, T, plan, out_dtype):
        ctx.T, ctx.plan = T, plan
        return synthesis_shaped(S.contiguous(), plan, T.pct, T._w, T.m_off, out_dtype)

    @staticmethod
    def backward(ctx, gx):
        T = ctx.T
        return analysis_shaped(gx.contiguous(), ctx.plan, T.pct_t, T._w, T.m_off), None, None, None
