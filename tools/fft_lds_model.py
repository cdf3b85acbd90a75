"""LDS bank-conflict model of the specialised FFT kernels (csrc/fft_fast.hip): LDS-array cycles of ONE work item, phase by phase.

Bank rules from /opt/skills/guides/MI355X_MICROARCH.md (section LDS): 64 banks of 4 bytes; a wave64 access is served in fixed lane
groups, one cycle per group when conflict-free, one more cycle per extra distinct address on a bank: ds_read_b64 = 2 groups of 32
lanes over 64 banks, ds_write_b64 = 4 x 16 lanes over 32 banks, ds_write_b128 = 8 x 8 lanes over 32 banks.

The functions restate the kernels' address arithmetic (pass_load / pass_compute_store / commit / untangle / spectrum stores) for a
"plan" = row stride LS[g] and padding D[g] per generation g of the work buffer (position p of row r at r * LS + p + (p // BS) * D,
BS = the block the writing pass produces) and lanes per row LPR[p] per pass (0 = butterflies dealt over consecutive threads).
`python tools/fft_lds_model.py` prints the current plans of the benchmark's four kernels next to the one-layout default;
`python tools/fft_lds_model.py search` re-derives the padding of the middle generations.  tests/test_lds_layouts.py pins the totals."""
import itertools
import sys
from collections import defaultdict


def _contig(n):
    return [list(range(g * n, (g + 1) * n)) for g in range(64 // n)]


KIND = {"r64": (_contig(32), 64, 8), "w64": (_contig(16), 32, 8), "w128": (_contig(8), 32, 16)}


def cycles(kind, addrs):
    """addrs: 64 byte addresses (None = inactive lane) -> (LDS-array cycles, cycles if conflict-free)"""
    groups, banks, nb = KIND[kind]
    cyc = ideal = 0
    for grp in groups:
        per = defaultdict(set)
        for lane in grp:
            a = addrs[lane]
            if a is None:
                continue
            for d in range(nb // 4):
                per[(a // 4 + d) % banks].add(a)
        if per:
            cyc += max(len(s) for s in per.values())
            ideal += 1
    return cyc, ideal


class Tot:
    def __init__(self):
        self.c, self.i = defaultdict(int), defaultdict(int)

    def add(self, phase, kind, addrs):
        c, i = cycles(kind, addrs)
        self.c[phase] += c
        self.i[phase] += i

    def total(self):
        return sum(self.c.values()), sum(self.i.values())

    def report(self, title):
        print(title)
        for k in self.c:
            print(f"   {k:18s} {self.c[k]:6d} / conflict-free {self.i[k]:6d} = {self.c[k] / max(1, self.i[k]):.2f}x")
        tc, ti = self.total()
        print(f"   {'TOTAL':18s} {tc:6d} / conflict-free {ti:6d} = {tc / ti:.2f}x")
        return tc, ti


def _lanes(NTH, RBH, NB, LPR, q, w):
    out = []
    for lane in range(64):
        idx = w * 64 + lane + q * NTH
        if LPR == 0:
            out.append(divmod(idx, NB) if idx < RBH * NB else None)
        else:
            row, j = divmod(idx, LPR)
            out.append((row, j) if (row < RBH and j < NB) else None)
    return out


def _rounds(NTH, RBH, NB, LPR):
    return (RBH * (LPR if LPR else NB) + NTH - 1) // NTH


def pass_loads(N2, R, RBH, NTH, LPR, LS, BS, D, skip=None):
    NB = N2 // R
    c = i = 0
    for q in range(_rounds(NTH, RBH, NB, LPR)):
        for w in range(NTH // 64):
            L = _lanes(NTH, RBH, NB, LPR, q, w)
            for r in range(R):
                ad = []
                for x in L:
                    pos = None if x is None else x[1] + r * NB
                    if pos is None or (skip and skip(pos)):
                        ad.append(None)
                    else:
                        ad.append(8 * (x[0] * LS + pos + ((pos // BS) * D if D else 0)))
                cc, ii = cycles("r64", ad)
                c += cc
                i += ii
    return c, i


def pass_stores(N2, R, NS, RBH, NTH, LPR, LS, D):
    NB, BS = N2 // R, NS * R
    c = i = 0
    for q in range(_rounds(NTH, RBH, NB, LPR)):
        for w in range(NTH // 64):
            L = _lanes(NTH, RBH, NB, LPR, q, w)
            for o in range(R):
                ad = [None if x is None else 8 * (x[0] * LS + (x[1] // NS) * (BS + D) + o * NS + x[1] % NS) for x in L]
                cc, ii = cycles("w64", ad)
                c += cc
                i += ii
    return c, i


def _ns(rad):
    NS = [1]
    for r in rad:
        NS.append(NS[-1] * r)
    return NS


def forward(N2, rad, RBH, NTH, VP, mmax, LS, D, LPR, swap=False):
    """rfft_fast_kernel: commit (16-byte stores of the converted row vectors), the passes, the untangle reads"""
    P, NS, t = len(rad), _ns(rad), Tot()
    VROW = N2 // VP
    for q in range((RBH * VROW + NTH - 1) // NTH):
        for w in range(NTH // 64):
            for part in range(VP // 2):
                ad = []
                for lane in range(64):
                    idx = w * 64 + lane + q * NTH
                    if idx < RBH * VROW:
                        row, c = divmod(idx, VROW)
                        ad.append(8 * (row * LS[0] + c * VP + 2 * (part ^ ((c >> 2) & 1) if swap else part)))
                    else:
                        ad.append(None)
                t.add("commit", "w128", ad)
    for p in range(P):
        c, i = pass_loads(N2, rad[p], RBH, NTH, LPR[p], LS[p], NS[p], D[p])
        t.c[f"pass {p + 1} loads"] += c
        t.i[f"pass {p + 1} loads"] += i
        c, i = pass_stores(N2, rad[p], NS[p], RBH, NTH, LPR[p], LS[p + 1], D[p + 1])
        t.c[f"pass {p + 1} stores"] += c
        t.i[f"pass {p + 1} stores"] += i
    tot = mmax * (RBH // 4)
    for base in range(0, tot, NTH):
        for w in range(NTH // 64):
            L = []
            for lane in range(64):
                idx = base + w * 64 + lane
                if idx < tot:
                    r0, m = (idx % (RBH // 4)) * 4, idx // (RBH // 4)
                    L.append((r0, 0 if m == N2 else m, 0 if m in (0, N2) else N2 - m))
                else:
                    L.append(None)
            for i_ in range(4):
                for s in (1, 2):
                    t.add("untangle reads", "r64", [None if x is None else 8 * ((x[0] + i_) * LS[P] + x[s]) for x in L])
    return t


def inverse(N2, rad, RBH, NTH, mmax, LS, D, LPR, pruned):
    """irfft_fast_kernel: spectrum stores (+ zero fill and in-place pre-twiddle when the spectrum is not pruned), the passes"""
    P, NS, t = len(rad), _ns(rad), Tot()
    tot = mmax * (RBH // 4)
    for q in range((tot + NTH - 1) // NTH):
        for w in range(NTH // 64):
            L = []
            for lane in range(64):
                idx = w * 64 + lane + q * NTH
                r0, m = (idx % (RBH // 4)) * 4, idx // (RBH // 4)
                L.append((r0, m) if m < mmax else None)
            for i_ in range(4):
                t.add("spectrum stores", "w64", [None if x is None else 8 * ((x[0] + i_) * LS[0] + x[1]) for x in L])
                if pruned:
                    t.add("spectrum stores", "w64", [None if (x is None or x[1] == 0) else 8 * ((x[0] + i_) * LS[0] + N2 - x[1]) for x in L])
    if not pruned:
        for base in range(mmax * RBH, (N2 + 1) * RBH, NTH):
            for w in range(NTH // 64):
                ad = []
                for lane in range(64):
                    idx = base + w * 64 + lane
                    ad.append(8 * ((idx % RBH) * LS[0] + idx // RBH) if idx < (N2 + 1) * RBH else None)
                t.add("zero fill", "w64", ad)
        H = N2 // 2 + 1
        for base in range(0, RBH * H, NTH):
            for w in range(NTH // 64):
                L = []
                for lane in range(64):
                    idx = base + w * 64 + lane
                    L.append(divmod(idx, H) if idx < RBH * H else None)
                A = lambda x, p: 8 * (x[0] * LS[0] + p)
                t.add("pre-twiddle reads", "r64", [None if x is None else A(x, x[1]) for x in L])
                t.add("pre-twiddle reads", "r64", [None if x is None else A(x, N2 - x[1]) for x in L])
                t.add("pre-twiddle stores", "w64", [None if x is None else A(x, x[1]) for x in L])
                t.add("pre-twiddle stores", "w64", [None if (x is None or x[1] == 0 or N2 - x[1] == x[1]) else A(x, N2 - x[1]) for x in L])
    skip = (lambda pos: mmax <= pos <= N2 - mmax) if pruned else None
    for p in range(P):
        c, i = pass_loads(N2, rad[p], RBH, NTH, LPR[p], LS[p], NS[p], D[p], skip if p == 0 else None)
        t.c[f"pass {p + 1} loads"] += c
        t.i[f"pass {p + 1} loads"] += i
        if p < P - 1:                      # the last pass stores to global memory
            c, i = pass_stores(N2, rad[p], NS[p], RBH, NTH, LPR[p], LS[p + 1], D[p + 1])
            t.c[f"pass {p + 1} stores"] += c
            t.i[f"pass {p + 1} stores"] += i
    return t


# (kernel, default plan, shipped plan = the LdsPlan specialisations of csrc/fft_fast.hip)
KERNELS = {
    "rfft 1440 bf16 (16 rows, 512 threads)": (lambda LS, D, LPR, sw: forward(720, (30, 24), 16, 512, 4, 241, LS, D, LPR, sw),
                                              ([722] * 3, [0, 0, 0], [0, 0], False), ([728, 744, 722], [0, 1, 0], [0, 32], False)),
    "irfft 1440 pruned (16 rows, 512 threads)": (lambda LS, D, LPR, sw: inverse(720, (30, 24), 16, 512, 241, LS, D, LPR, True),
                                                 ([721] * 2, [0, 0], [0, 0], False), ([721, 744], [0, 1], [0, 32], False)),
    "rfft 480 bf16 (one half: 16 rows, 256 threads)": (lambda LS, D, LPR, sw: forward(240, (10, 6, 4), 16, 256, 4, 241, LS, D, LPR, sw),
                                                       ([242] * 4, [0] * 4, [0, 0, 0], False), ([248, 264, 296, 242], [0, 1, 14, 0], [0, 0, 0], False)),
    "irfft 480 (32 rows, 512 threads)": (lambda LS, D, LPR, sw: inverse(240, (10, 6, 4), 32, 512, 241, LS, D, LPR, False),
                                         ([241] * 3, [0] * 3, [0, 0, 0], False), ([249, 264, 296], [0, 1, 14], [0, 0, 64], False)),
}


def search(N2, rad, RBH, NTH, inverse_kernel):
    """padding D and row stride of the middle generations that minimise (stores of the writing pass + loads of the reading pass)"""
    P, NS = len(rad), _ns(rad)
    best = None
    for LPRs in itertools.product(*[[0, 32, 64] for _ in rad]):
        if any(l and (l < N2 // r or _rounds(NTH, RBH, N2 // r, l) > _rounds(NTH, RBH, N2 // r, 0)) for l, r in zip(LPRs, rad)):
            continue
        total, plan = 0, []
        for g in range(1, P):
            bg = None
            for D in range(0, 17):
                need = (N2 // NS[g]) * (NS[g] + D)
                for LS in range(need, need + 32):
                    cs, _ = pass_stores(N2, rad[g - 1], NS[g - 1], RBH, NTH, LPRs[g - 1], LS, D)
                    cl, _ = pass_loads(N2, rad[g], RBH, NTH, LPRs[g], LS, NS[g], D)
                    if bg is None or cs + cl < bg[0]:
                        bg = (cs + cl, D, LS)
            total += bg[0]
            plan.append(bg)
        if best is None or total < best[0]:
            best = (total, LPRs, plan)
    return best


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "search":
        for name, args in (("1440 forward", (720, (30, 24), 16, 512, False)), ("480 forward half", (240, (10, 6, 4), 16, 256, False)),
                           ("480 inverse", (240, (10, 6, 4), 32, 512, True))):
            print(name, search(*args))
    else:
        for name, (fn, default, plan) in KERNELS.items():
            fn(*default).report(name + " — one layout for every generation")
            fn(*plan).report(name + " — shipped plan")
