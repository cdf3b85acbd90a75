"""Counted-wait checker of the weight-stationary channel GEMM (csrc/conv1x1.hip: conv_nn_astat_kernel).

The kernel keeps LDS-DMA pieces in flight across barriers and waits for them with counted ``s_waitcnt vmcnt(N)``: N must
equal the number of memory instructions a wave issued AFTER the pieces it is waiting for (gfx950 retires vmcnt in order,
loads and stores alike).  A count that is too large reads LDS before the data has landed (silently wrong numbers); one that is
too small only costs time.  There is no GPU in the development container, so the counts are checked here, two ways:

1. ``simulate(...)``: the issue / wait schedule of one wave, restated from the kernel source with the SAME template constants
   (NP, LOOK, NCH, NSTORE, operand pieces) and the SAME wait formulas, run over a stream of pixel tiles with an in-order
   retirement model: at every use of a chunk (and of the epilogue operand images) all of its pieces must be retired under the
   waits executed so far — for every instantiation the launcher can pick, every stream length from 0 tiles up, including the
   first tiles (fewer epilogues behind the wave) and the end of the stream (fewer chunks in flight).  The slack of every wait
   (how many instructions earlier than necessary it fires) is reported; the steady state of the variants without epilogue
   operand must have none.
2. ``check_isa(...)``: the compiled kernel (``hipcc -S``) must carry exactly the memory instructions the model counts — every
   run of consecutive LDS-DMA loads is one chunk (NP pieces) or one set of operand images, the epilogue stores come in
   multiples of NSTORE, no other vector memory instruction sits inside the tile loop (hipcc neither added nor removed one),
   and every ``vmcnt`` immediate in the kernel is one the model expects.

    python tools/vmcnt_check.py            # both, all instantiations (compiles csrc/conv1x1.hip to assembly, ~15 s)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Variant:
    """compile-time constants of one instantiation, as conv_nn_astat_kernel derives them"""

    def __init__(self, TM, KCH, PRE, EPI, KS=24):
        self.TM, self.KCH, self.PRE, self.EPI, self.KS = TM, KCH, PRE, EPI, KS
        self.SMALLK = KS < 24
        self.NCH = 1 if self.SMALLK else 384 // KCH
        self.CH = KCH * 128
        self.ROUNDS = TM
        self.EBYTES = self.ROUNDS * 128 * 128 if EPI else 0
        self.NSLOT = (8 if EPI else 5) if self.SMALLK else ((144 * 1024 - self.EBYTES) if EPI else 128 * 1024) // self.CH
        self.NP = KCH // 32
        self.NSTORE = self.ROUNDS * 4 * (2 if PRE else 1)
        self.EPIECES = self.ROUNDS * 4 if EPI else 0
        if self.SMALLK:
            self.LOOK = 2 if (PRE or EPI) else 3
        else:
            self.LOOK = 8 if KCH == 64 else (5 if self.NP * 4 + 2 * self.NSTORE <= 63 else 4)
        self.NEPI_MAX = (self.LOOK + self.NCH - 1) // self.NCH
        self.SMALLK_EPI_WAIT = self.NSTORE + (self.LOOK - 1) * (self.EPIECES + self.NP + self.NSTORE)

    @property
    def name(self):
        return f"<TM={self.TM}, KCH={self.KCH}, PRE={int(self.PRE)}, EPI_LOADS={int(self.EPI)}, KS={self.KS}>"

    def mangled(self):
        return f"20conv_nn_astat_kernelILi{self.TM}ELi{self.KCH}ELb{int(self.PRE)}ELb{int(self.EPI)}ELi{self.KS}EE"

    def static_asserts(self):
        assert self.LOOK <= self.NSLOT - 1, (self.name, "a slot is refilled only after every wave has left it")
        assert self.EPI or self.NP * (self.LOOK - 1) + self.NEPI_MAX * self.NSTORE <= 63, (self.name, "vmcnt is a 6-bit counter")
        assert not (self.SMALLK and self.EPI) or self.SMALLK_EPI_WAIT <= 63, self.name
        lds = self.NSLOT * self.CH + self.EBYTES + 128 * 128
        assert lds <= 160 * 1024, (self.name, lds)

    def chunk_wait(self, ts, kc, nchunks):
        """the vmcnt immediate in front of the multiplication of chunk c = ts * NCH + kc (None: no wait instruction)"""
        c = ts * self.NCH + kc
        if c + self.LOOK > nchunks:
            return 0
        if self.SMALLK and self.EPI:
            return 0 if ts < self.LOOK else self.SMALLK_EPI_WAIT
        nfull = (self.LOOK - kc + self.NCH - 1) // self.NCH
        if self.EPI:
            if ts < self.NEPI_MAX:
                return 0
            nf = min(nfull, 3)
            return min(63, self.NP * (self.LOOK - 1) + nf * self.NSTORE + (nf - (1 if kc == 0 else 0)) * self.EPIECES)
        return min(63, self.NP * (self.LOOK - 1) + min(nfull, ts, 3) * self.NSTORE)

    def epi_wait(self, ts, nchunks):
        if not self.EPI:
            return None
        return min(63, self.NP * self.NCH) if (ts + 1) * self.NCH - 1 + self.LOOK < nchunks else 0

    def expected_vmcnt(self):
        """every vmcnt immediate the kernel may contain"""
        vals = {0}
        for ts in range(0, 6):
            for kc in range(self.NCH):
                for nchunks in (10 ** 6,):
                    w = self.chunk_wait(ts, kc, nchunks)
                    if w is not None:
                        vals.add(w)
        if self.EPI:
            vals.add(min(63, self.NP * self.NCH))
        return vals


# the instantiations mk_conv1x1_nn can launch (csrc/conv1x1.hip: MK_ASTAT and the small-K branch)
VARIANTS = [Variant(3, kch, pre, epi) for kch in (64, 128) for pre in (False, True) for epi in (False, True) if not (epi and kch == 128)] + \
           [Variant(3, 96, False, True, 5), Variant(3, 96, True, False, 5), Variant(3, 96, False, False, 5)]


def simulate(v: Variant, tiles: int):
    """one wave's schedule over ``tiles`` pixel tiles.  Returns (max slack, steady-state slack) of the chunk waits; raises
    AssertionError when a chunk or an operand image is used before its pieces are guaranteed to have retired."""
    nchunks = tiles * v.NCH
    issued = []                 # tags in issue order
    retired = 0                 # ops [0, retired) are guaranteed complete

    def issue(tag, n):
        issued.extend([tag] * n)

    def wait(n):
        nonlocal retired
        retired = max(retired, len(issued) - n)

    def last_index(tag):
        idx = [i for i, t in enumerate(issued) if t == tag]
        assert idx, f"{v.name}: {tag} was never requested"
        return idx[-1]

    nxt = 0                     # next chunk to request

    def issue_next():
        nonlocal nxt
        issue(("X", nxt), v.NP)
        nxt += 1

    for _ in range(min(v.LOOK, nchunks)):
        issue_next()
    max_slack, steady = 0, 0
    for ts in range(tiles):
        for kc in range(v.NCH):
            c = ts * v.NCH + kc
            w = v.chunk_wait(ts, kc, nchunks)
            if w is not None:
                need = len(issued) - 1 - last_index(("X", c))       # ops issued after the chunk's last piece
                wait(w)
                slack = need - w
                assert slack >= 0, f"{v.name}: tile {ts} chunk {kc} of {tiles} tiles: vmcnt({w}) but {need} instructions follow the chunk"
                max_slack = max(max_slack, slack)
                if ts >= 4 and c + v.LOOK + v.NCH <= nchunks:
                    steady = max(steady, slack)
            assert last_index(("X", c)) < retired, f"{v.name}: chunk {c} multiplied before its pieces retired ({tiles} tiles)"
            if v.EPI and kc == 0:
                issue(("E", ts), v.EPIECES)
            if c + v.LOOK < nchunks:
                assert nxt == c + v.LOOK
                issue_next()
        if v.EPI:
            wait(v.epi_wait(ts, nchunks))
            assert last_index(("E", ts)) < retired, f"{v.name}: operand images of tile {ts} read before they landed ({tiles} tiles)"
        issue(("S", ts), v.NSTORE)
    return max_slack, steady


def check_model(verbose=True):
    for v in VARIANTS:
        v.static_asserts()
        worst, steady = 0, 0
        for tiles in list(range(0, 14)) + [63, 64, 200]:
            a, b = simulate(v, tiles)
            worst, steady = max(worst, a), max(steady, b)
        cap = max((v.NP * (v.LOOK - 1) + 3 * v.NSTORE + 3 * v.EPIECES) - 63, 0)       # counts beyond the 6-bit field are cut to 63
        assert steady <= cap, f"{v.name}: the steady-state wait fires {steady} instructions early"
        if verbose:
            print(f"model ok  {v.name:58s} NP={v.NP} LOOK={v.LOOK} NCH={v.NCH} NSTORE={v.NSTORE} slots={v.NSLOT}  "
                  f"slack: steady {steady}, worst {worst}")


# --------------------------------------------------------------------------- #
# the compiled kernel
# --------------------------------------------------------------------------- #
def assembly(path=None):
    if path and os.path.exists(path):
        return path
    out = path or os.path.join(tempfile.mkdtemp(prefix="mk_isa_"), "conv1x1.s")
    src = os.path.join(ROOT, "makani_amd", "csrc", "conv1x1.hip")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, src],
                   check=True, stderr=subprocess.DEVNULL)
    return out


def kernel_body(path, key):
    out, on = [], False
    for line in open(path):
        if re.match(r"^_Z\w+:", line):
            on = key in line
            continue
        if on:
            if line.startswith("\t.section") or line.startswith(".Lfunc_end"):
                break
            out.append(line)
    return out


def check_isa(path=None, verbose=True):
    path = assembly(path)
    for v in VARIANTS:
        body = kernel_body(path, v.mangled())
        assert body, f"{v.name}: kernel not found in {path}"
        ops = [m.group(1) + " " + l.strip() for l in body for m in [re.match(r"^\t([a-z_0-9]+)", l)] if m]
        # (1) runs of consecutive LDS-DMA loads (scalar bookkeeping between them ignored)
        runs, cur = [], 0
        for o in ops:
            op = o.split()[0]
            if op.startswith("buffer_load") and " lds" in o:
                cur += 1
            elif op.startswith(("s_", "v_readfirstlane", "v_mov", "v_add", "v_sub", "v_min", "v_max", "v_lshl", "v_mul", "v_mad", "v_and", "v_cndmask", "v_cmp")):
                continue
            elif cur:
                runs.append(cur)
                cur = 0
        if cur:
            runs.append(cur)
        ok_runs = {v.NP * k for k in range(1, v.LOOK + 1)} | ({v.EPIECES, 4} if v.EPI else set())
        bad = [r for r in runs if r not in ok_runs]
        assert not bad, f"{v.name}: LDS-DMA runs of {sorted(set(bad))} pieces (expected {sorted(ok_runs)})"
        # (2) vector memory instructions other than LDS-DMA loads and the epilogue's buffer stores: the weight / bias loads of
        #     the prologue only — none may follow the first DMA piece
        first_dma = next(i for i, o in enumerate(ops) if o.startswith("buffer_load") and " lds" in o)
        stray = [o for o in ops[first_dma:] if re.match(r"^(global_|flat_|scratch_)", o) or
                 (o.startswith("buffer_load") and " lds" not in o) or o.startswith("buffer_atomic")]
        assert not stray, f"{v.name}: vector memory instructions inside the stream: {stray[:3]}"
        nstores = sum(1 for o in ops if o.startswith("buffer_store_dwordx4"))
        assert nstores and nstores % v.NSTORE == 0, f"{v.name}: {nstores} epilogue stores, not a multiple of NSTORE = {v.NSTORE}"
        # (3) every vmcnt immediate behind the first DMA piece is one the schedule uses (before it: hipcc's own waits for the
        #     weight fragments of the prologue)
        imm = set()
        for o in ops[first_dma:]:
            if o.startswith("s_waitcnt"):
                m = re.search(r"vmcnt\((\d+)\)", o)
                if m:
                    imm.add(int(m.group(1)))
        exp = v.expected_vmcnt()
        assert imm <= exp, f"{v.name}: vmcnt immediates {sorted(imm - exp)} are not part of the schedule {sorted(exp)}"
        if verbose:
            print(f"isa ok    {v.name:58s} DMA runs {sorted(set(runs))}, {nstores} stores, vmcnt {sorted(imm)}")


# --------------------------------------------------------------------------- #
# the two-group form (conv_nn_astat2_kernel): both groups' instruction streams against the same retirement model
# --------------------------------------------------------------------------- #
class Variant2:
    """compile-time constants of conv_nn_astat2_kernel<PRE, EPI_LOADS>"""
    NP, NCH, PER, OFF = 2, 3, 7, 3
    # [group][kc]: chunk requests / read-back rounds / operand requests behind a chunk
    WA = ((3, 2, 1), (3, 2, 1))
    WB = ((3, 2, 1), (3, 1, 2))
    WE = ((1, 1, 1), (1, 0, 0))

    def __init__(self, PRE, EPI):
        self.PRE, self.EPI = PRE, EPI
        self.NSLOT = 4
        self.NS3 = 2 * (2 if PRE else 1)
        self.EPIECES = 6 if EPI else 0

    @property
    def name(self):
        return f"astat2<PRE={int(self.PRE)}, EPI_LOADS={int(self.EPI)}>"

    def mangled(self):
        return f"21conv_nn_astat2_kernelILb{int(self.PRE)}ELb{int(self.EPI)}EE"

    def wait_const(self, g, kc):
        return self.WA[g][kc] * self.NP + self.WB[g][kc] * self.NS3 + self.WE[g][kc] * self.EPIECES

    def chunk_wait(self, g, gts, gph, T):
        return 0 if (gts <= 1 or gts >= T - 1) else self.wait_const(g, gph)

    def epi_wait(self, g, lts, T):
        return 0 if (g == 0 or lts >= T - 2) else 3 * self.NP

    def expected_vmcnt(self):
        return {0} | {self.wait_const(g, k) for g in (0, 1) for k in range(3)} | ({3 * self.NP} if self.EPI else set())


VARIANTS2 = [Variant2(pre, epi) for pre in (False, True) for epi in (False, True)]


def simulate2(v: Variant2, T: int):
    """the streams of one wave of each group over T pixel tiles (ticks = workgroup barriers; group 1 runs OFF ticks behind).
    Checks at every chunk wait that the chunk's pieces of THIS wave have retired (every wave waits before the barrier behind
    which group 0 starts reading), at every operand wait that the images have, that a slot is only re-requested after group 1 read
    it, and returns the largest steady-state slack."""
    PER, OFF, NCH, NP = v.PER, v.OFF, v.NCH, v.NP
    nchunks = NCH * T
    steady = 0
    for grp in (0, 1):
        ops, retired = [], 0

        def issue(tag, n):
            ops.extend([tag] * n)

        def last(tag):
            idx = [i for i, t in enumerate(ops) if t == tag]
            assert idx, f"{v.name}: {tag} never requested"
            return idx[-1]

        requested = set()
        for c in range(min(v.NSLOT, nchunks)):
            issue(("X", c), NP)
            requested.add(c)
        for t in range(PER * T + OFF if T else 0):
            gph, gts = t % PER, t // PER
            lt = t - grp * OFF
            if v.EPI and 0 <= lt < PER * T and lt % PER == 4:
                # (0) IN FRONT of the tick's barrier: the epilogue-operand images of this group's tile.  vmcnt retires this wave's
                # own pieces only, and read-back round 0 (right behind the barrier) reads rows requested by sibling waves: every
                # wave waits here, the barrier then makes the landing group-wide (the wait used to sit behind the barrier)
                lts = lt // PER
                need = len(ops) - 1 - last(("E", lts))
                w = v.epi_wait(grp, lts, T)
                assert w <= need, f"{v.name}: group {grp} operand wait of tile {lts} of {T}: vmcnt({w}), {need} instructions follow"
                retired = max(retired, len(ops) - w)
                assert last(("E", lts)) < retired
            if gph < NCH and gts < T:                      # (1) every wave: the chunk group 0 multiplies in this tick
                c = NCH * gts + gph
                assert c in requested, f"{v.name}: chunk {c} waited for before it was requested (T={T})"
                need = len(ops) - 1 - last(("X", c))
                w = v.chunk_wait(grp, gts, gph, T)
                assert w <= need, f"{v.name}: group {grp} tile {gts} chunk {gph} of {T} tiles: vmcnt({w}) but only {need} instructions follow the chunk"
                retired = max(retired, len(ops) - w)
                assert last(("X", c)) < retired
                if 2 <= gts < T - 2:
                    steady = max(steady, need - w)
            # (2) this group's phase
            if 0 <= lt < PER * T:
                lts, lph = lt // PER, lt % PER
                if lph < NCH:
                    c = NCH * lts + lph                     # multiplied now: requested, and waited for (by every wave) at or before this tick
                    assert c in requested and PER * lts + lph <= t
                if lph == 0 and v.EPI:
                    issue(("E", lts), v.EPIECES)
                if lph >= 4:
                    issue(("S", lts, lph), v.NS3)
            if gph > OFF:                                  # (3) END of the tick: the slot group 1 read in the previous tick
                c = NCH * gts + (gph - OFF - 1) + v.NSLOT
                if c < nchunks:
                    # chunk c - NSLOT was multiplied by group 1 in tick t - 1 and by group 0 three ticks before that
                    assert (c - v.NSLOT) == NCH * ((t - 1 - OFF) // PER) + ((t - 1 - OFF) % PER) and (t - 1 - OFF) % PER < NCH
                    issue(("X", c), NP)
                    requested.add(c)
    return steady


def check_model2(verbose=True):
    for v in VARIANTS2:
        assert max(v.expected_vmcnt()) <= 63
        lds = v.NSLOT * 16384 + 2 * 192 * 128 + (2 * 192 * 128 if v.EPI else 0)
        assert lds <= 160 * 1024, (v.name, lds)
        steady = max(simulate2(v, T) for T in list(range(0, 12)) + [31, 64])
        assert steady == 0, f"{v.name}: a steady-state wait fires {steady} instructions early"
        if verbose:
            print(f"model ok  {v.name:40s} slots={v.NSLOT} waits " +
                  " ".join(f"g{g}k{k}={v.wait_const(g, k)}" for g in (0, 1) for k in range(3)) + f"  LDS {lds // 1024} KB")


def check_isa2(path=None, verbose=True):
    path = assembly(path)
    for v in VARIANTS2:
        body = kernel_body(path, v.mangled())
        assert body, f"{v.name}: kernel not found in {path}"
        ops = [m.group(1) + " " + l.strip() for l in body for m in [re.match(r"^\t([a-z_0-9]+)", l)] if m]
        first_dma = next(i for i, o in enumerate(ops) if o.startswith("buffer_load") and " lds" in o)
        stray = [o for o in ops[first_dma:] if re.match(r"^(global_|flat_|scratch_)", o) or
                 (o.startswith("buffer_load") and " lds" not in o) or o.startswith("buffer_atomic")]
        assert not stray, f"{v.name}: {len(stray)} vector memory instructions inside the stream (register spills?): {stray[:3]}"
        imm = set()
        for o in ops[first_dma:]:
            if o.startswith("s_waitcnt"):
                m = re.search(r"vmcnt\((\d+)\)", o)
                if m:
                    imm.add(int(m.group(1)))
        exp = v.expected_vmcnt()
        assert imm <= exp, f"{v.name}: vmcnt immediates {sorted(imm - exp)} are not part of the schedule {sorted(exp)}"
        nstores = sum(1 for o in ops if o.startswith("buffer_store_dwordx4"))
        assert nstores and nstores % v.NS3 == 0
        if verbose:
            print(f"isa ok    {v.name:40s} {nstores} stores, vmcnt {sorted(imm)}")


if __name__ == "__main__":
    check_model()
    check_model2()
    path = assembly(sys.argv[1] if len(sys.argv) > 1 else None)
    check_isa(path)
    check_isa2(path)
