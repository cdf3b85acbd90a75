"""Plain vs segmented-addressing FFT kernels (csrc/fft_fast.hip) on one rank's share of the BASELINE config 5 transform
(h4 w2: 192 planes x 181 latitudes x 1440 longitudes, bf16 in, 241 modes), HIP-event timing."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from makani_amd import ops


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda:0")
    for (Cc, nlat, nlon, mmax, m_shapes, r_shapes, xseg) in [(192, 181, 1440, 241, [121, 120], [48] * 4, 2),
                                                            (192, 60, 480, 241, [121, 120], [48] * 4, 2),
                                                            (384, 721, 1440, 241, [241], [384], 1)]:
        w = (2.0 * math.pi / nlon,) * 3
        x = torch.randn(1, Cc, nlat, nlon, device=dev).bfloat16()
        Cp = ops.round4(Cc)
        wl = nlon // xseg
        xbuf = x[0].reshape(Cc, nlat, xseg, wl).permute(2, 0, 1, 3).contiguous()
        base, off = [[0] * len(r_shapes) for _ in m_shapes], 0
        for i in range(len(r_shapes)):
            for j in range(len(m_shapes)):
                base[j][i] = off
                off += nlat * m_shapes[j] * 2 * r_shapes[i]
        sg = ops.fft_seg_desc(m_shapes, r_shapes, base, xseg=xseg, x_stride=Cc * nlat * wl, x_nlat=nlat)
        fs = torch.empty((off,), device=dev)
        F = ops.rfft_rows(x, mmax, Cp, w)
        xo = torch.empty_like(xbuf)
        nbytes = Cc * nlat * (nlon * 2 + mmax * 8)
        t = {
            "rfft plain": timeit(lambda: ops.rfft_rows(x, mmax, Cp, w)),
            "rfft seg": timeit(lambda: ops.rfft_rows_seg(xbuf, 0, fs, Cc, nlat, nlon, mmax, w, sg)),
            "irfft plain": timeit(lambda: ops.irfft_rows(F, 1, Cc, nlon, torch.bfloat16, w)),
            "irfft seg": timeit(lambda: ops.irfft_rows_seg(fs, xo, 0, Cc, nlat, nlon, mmax, w, sg)),
        }
        print(f"{Cc} planes x {nlat} x {nlon}, {len(m_shapes)} x {len(r_shapes)} slabs, {xseg} pieces: " +
              ", ".join(f"{k} {v * 1e3:.1f} us ({nbytes / v / 1e6:.0f} GB/s)" for k, v in t.items()), flush=True)


if __name__ == "__main__":
    main()
