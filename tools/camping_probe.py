"""Do streaming kernels that walk several big tensors at the SAME offsets lose bandwidth when the tensors' base addresses are
congruent modulo the memory system's interleave period?  (round 6: the one-pass instance-norm backward measured 70 us in one
process and 99 us in another on the same box with identical code — the only difference was where the caching allocator had put
x, gy and gx.)  One buffer, three carved tensors at bases k * 2 MiB + delta_i, timed for a set of staggers delta.
    python tools/camping_probe.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from makani_amd import _lib  # noqa: E402
from makani_amd._lib import check, lib, ptr, stream  # noqa: E402

dev = torch.device("cuda:0")
C, H, W = 384, 240, 480
planes, hw = C, H * W
nbytes = planes * hw * 2
MB2 = 2 << 20
span = (nbytes + MB2 - 1) // MB2 * MB2 + MB2            # room for one tensor + stagger
NB = 12                                                  # rotating sets: nothing stays in the caches
pool = torch.empty(3 * NB * span + (64 << 20), dtype=torch.uint8, device=dev)
base = (pool.data_ptr() + MB2 - 1) // MB2 * MB2 - pool.data_ptr()
L = lib()
ch_f = L.mk_instnorm_fused_chunks(hw, _lib.MK_BF16, planes, 1)
slots = torch.full((max(planes * 64, 1 << 20),), -1, dtype=torch.int64, device=dev)
depart = torch.zeros((1 << 16,), dtype=torch.int32, device=dev)
gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
stats = torch.zeros(planes, 2, device=dev)
stats[:, 1] = 1.0
sums = torch.empty(2, planes, device=dev)


def carve(set_i, which, delta):
    off = base + (3 * set_i + which) * span + delta
    return pool[off:off + nbytes].view(torch.bfloat16).view(1, C, H, W)


def timeit(fn, reps=36, warm=12):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for name, deltas in (("all congruent mod 2 MiB", (0, 0, 0)), ("256 B / 512 B", (0, 256, 512)), ("4 KiB / 8 KiB", (0, 4096, 8192)),
                     ("64 KiB / 128 KiB", (0, 65536, 131072)), ("192 KiB / 448 KiB", (0, 196608, 458752)), ("1 MiB / 512 KiB", (0, 1 << 20, 1 << 19)),
                     ("683 KiB / 1365 KiB (thirds of 2 MiB)", (0, 699392, 1397760))):
    sets = [(carve(i, 0, deltas[0]), carve(i, 1, deltas[1]), carve(i, 2, deltas[2])) for i in range(NB)]
    for x, gy, gx in sets:
        x.normal_()
        gy.normal_()

    def bwd(i):
        x, gy, gx = sets[i % NB]
        check(L.mk_instnorm_bwd_fused(ptr(x), ptr(gy), ptr(gx), _lib.MK_BF16, ptr(stats), ptr(gam), ptr(bet), None, ptr(sums), ptr(slots), ptr(depart),
                                      planes, C, hw, 0, stream()), "bwd")

    def add(i):
        x, gy, gx = sets[i % NB]
        torch.add(x, gy, out=gx)
    t_b, t_a = timeit(bwd), timeit(add)
    print(f"{name:<40s}: one-pass norm backward {t_b:6.1f} us ({3 * nbytes / t_b / 1e6:5.2f} TB/s)   torch add (2 reads + 1 write) {t_a:6.1f} us ({3 * nbytes / t_a / 1e6:5.2f} TB/s)", flush=True)
