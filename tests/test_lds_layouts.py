"""LDS layouts of the split-bf16 GEMM engines checked against the bank model of /opt/skills/guides/MI355X_MICROARCH.md
(section LDS: 64 banks of 4 bytes; a wave64 access is served in fixed lane groups, lanes of one group that touch different
addresses on one bank serialise).  The functions below restate the address arithmetic of csrc/xgemm2.hip
(`Stage2::lds_off`, `frag2`) and csrc/xsplit.h (`frag`, PK = 24); no GPU needed."""
import itertools

BK = 16          # k-depth of a limb image (xsplit.h)
PK = 24          # padded pitch of the [row][k] image in elements (xsplit.h)

# lane groups of the guide's table: ds_read_b128 = 4 x 16 lanes (not consecutive), ds_write_b64 = 4 x 16 consecutive lanes
READ_B128_GROUPS = [
    [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
    [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
    [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59],
    [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63],
]
WRITE_B64_GROUPS = [list(range(g * 16, g * 16 + 16)) for g in range(4)]


def worst_way(groups, byte_addr, nbytes, banks):
    """largest number of distinct addresses that meet on one bank inside one lane group"""
    worst = 1
    for grp in groups:
        per_bank = {}
        for lane in grp:
            a = byte_addr(lane)
            for d in range(nbytes // 4):
                per_bank.setdefault(((a // 4) + d) % banks, set()).add(a)
        worst = max(worst, max(len(s) for s in per_bank.values()))
    return worst


def swizzled_elem(row, k):            # Stage2<ROWS, KC, ., SW = true>::lds_off / frag2<ROWS, KC, SW = true>: unpadded rows of 16
    return row * BK + ((((k >> 3) ^ (row >> 3)) & 1) << 3) + (k & 7)


def padded_elem(row, k):              # the 48-byte pitch of the 128 x 128 tile
    return row * PK + k


def test_swizzle_is_a_bijection_of_every_row():
    for rows in (128, 256):
        seen = {swizzled_elem(r, k) for r in range(rows) for k in range(BK)}
        assert seen == set(range(rows * BK))


def test_fragment_reads_of_the_tall_tile_are_conflict_free():
    """frag2: lane -> row r0 + (lane & 31), k-block lane >> 5, 8 elements = 16 bytes (ds_read_b128)"""
    for r0 in range(0, 256, 32):
        way = worst_way(READ_B128_GROUPS, lambda lane: 2 * swizzled_elem(r0 + (lane & 31), (lane >> 5) * 8), 16, 64)
        assert way == 1, (r0, way)
    # the same rows WITHOUT the swizzle would meet two by two (32-byte rows: rows r and r + 8 share their banks)
    plain = worst_way(READ_B128_GROUPS, lambda lane: 2 * ((lane & 31) * BK + (lane >> 5) * 8), 16, 64)
    assert plain == 2
    # and the padded pitch of the square tile is conflict-free as documented in xsplit.h
    assert worst_way(READ_B128_GROUPS, lambda lane: 2 * padded_elem(lane & 31, (lane >> 5) * 8), 16, 64) == 1


def test_split_stores_of_the_tall_tile_are_conflict_free():
    """Stage2::store: thread f -> row f >> 2, k = (f & 3) * 4, 4 limbs = 8 bytes (ds_write_b64, banked mod 32)"""
    for wave, q in itertools.product(range(8), range(2)):
        f0 = wave * 64 + q * 512
        way = worst_way(WRITE_B64_GROUPS, lambda lane: 2 * swizzled_elem((f0 + lane) >> 2, ((f0 + lane) & 3) * 4), 8, 32)
        assert way == 1, (wave, q, way)


def test_two_stages_of_the_tall_tile_fit_the_lds():
    """xcgemm2_kernel<..., RT = 4>: (re, im) x 3 limbs x (256-row A + 128-column B), two stages, at most 160 KiB"""
    a_kc = 256 * BK                    # unpadded [row][k]
    b_kc = 128 * BK
    b_kr = BK * (128 + 32)             # [k][col] keeps its pitch (xsplit.h plane_elems<ROWS, false>)
    for plb in (b_kc, b_kr):
        assert 2 * (2 * 3 * (a_kc + plb)) * 2 <= 160 * 1024
    assert 2 * (2 * 3 * (256 * PK + 128 * PK)) * 2 > 160 * 1024        # the padded pitch would not


# ---- the FFT kernels' work-buffer layouts (csrc/fft_fast.hip LdsPlan, tools/fft_lds_model.py) -------------------------------
def _fft_model():
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fft_lds_model
    return fft_lds_model


def test_fft_lds_plans_against_the_bank_model():
    """LDS cycles of one work item under the bank model: the one-layout default against the plans the kernels ship with"""
    fm = _fft_model()
    want = {"rfft 1440": (5608, 3920), "irfft 1440": (2900, 2084), "rfft 480": (2568, 2032), "irfft 480": (4788, 3901)}
    for name, (fn, default, plan) in fm.KERNELS.items():
        key = " ".join(name.split()[:2])
        assert (fn(*default).total()[0], fn(*plan).total()[0]) == want[key], name
    # the forward 1440-point plan is conflict-free in the passes and the untangle step; the commit keeps its 2-way conflict (the
    # lane-dependent store order that removes it costs more vector instructions than the conflict costs time: LdsPlan::SWAP off)
    fn, _, (LS, D, LPR, _sw) = fm.KERNELS["rfft 1440 bf16 (16 rows, 512 threads)"]
    t = fn(LS, D, LPR, False)
    assert all(t.c[k] == t.i[k] for k in t.c if k != "commit")
    assert fn(LS, D, LPR, True).total()[0] == 3216


def test_fft_lds_plans_of_the_source_are_the_modelled_ones_and_consistent():
    """the LdsPlan specialisations in csrc/fft_fast.hip carry the constants the model was run with; every generation's map is a
    bijection of a row into its stride, and what a pass stores at the padded address is what the next pass loads there"""
    import os
    import re
    fm = _fft_model()
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "makani_amd", "csrc", "fft_fast.hip")).read()
    found = {}
    for m in re.finditer(r"struct LdsPlan<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (true|false), true> \{[^\n]*\n\s*static constexpr int "
                         r"LS0 = (\d+), LS1 = (\d+), LS2 = (\d+), LS3 = (\d+), D1 = (\d+), D2 = (\d+), LPR1 = (\d+), LPR2 = (\d+), LPR3 = (\d+);\n"
                         r"\s*static constexpr bool SWAP = (true|false|MK_FFT_SWAP != 0);", src):
        g = m.groups()
        found[(int(g[0]), int(g[4]), g[6] == "true")] = ([int(x) for x in g[7:11]], [0, int(g[11]), int(g[12])], [int(x) for x in g[13:16]], g[16] == "true")      # (MK_FFT_SWAP defaults to 0)
    shipped = {(720, 16, False): fm.KERNELS["rfft 1440 bf16 (16 rows, 512 threads)"][2], (720, 16, True): fm.KERNELS["irfft 1440 pruned (16 rows, 512 threads)"][2],
               (240, 16, False): fm.KERNELS["rfft 480 bf16 (one half: 16 rows, 256 threads)"][2], (240, 32, True): fm.KERNELS["irfft 480 (32 rows, 512 threads)"][2]}
    assert set(found) == set(shipped)
    for key, (LS, D, LPR, swap) in shipped.items():
        sLS, sD, sLPR, sswap = found[key]
        n = len(LS)
        assert sLS[:n] == LS and sD[:len(D)] == D[:3] and sLPR[:len(LPR)] == LPR and sswap == swap, key
    for N2, rad, LS, D in ((720, (30, 24), [728, 744, 722], [0, 1, 0]), (720, (30, 24), [721, 744, 721], [0, 1, 0]),
                           (240, (10, 6, 4), [248, 264, 296, 242], [0, 1, 14, 0]), (240, (10, 6, 4), [249, 264, 296, 249], [0, 1, 14, 0])):
        NS = [1]
        for r in rad:
            NS.append(NS[-1] * r)
        for p, R in enumerate(rad):
            BS, Dg, NB = NS[p] * R, D[p + 1], N2 // R
            pad = lambda pos: pos + (pos // BS) * Dg
            seen = set()
            for j in range(NB):
                k = j % NS[p]
                for o in range(R):
                    addr = (j // NS[p]) * (BS + Dg) + k + o * NS[p]                      # pass_compute_store
                    assert addr == pad((j - k) * R + k + o * NS[p]) and addr < LS[p + 1] and addr not in seen
                    seen.add(addr)
            assert len(seen) == N2
            if p + 1 < len(rad):
                NB2 = N2 // rad[p + 1]
                assert NB2 % BS == 0
                step = NB2 + (NB2 // BS) * Dg
                assert all(pad(j) + r * step == pad(j + r * NB2) for j in range(NB2) for r in range(rad[p + 1]))     # pass_load
