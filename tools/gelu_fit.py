"""Coefficients of the forward GELU of the bf16 kernels (csrc/common.h gelu_exp2_x2) and what they cost in accuracy.

    x Phi(x) = relu(x) - |x| Phi(-|x|),   Phi(-a) = erfc(a / sqrt 2) / 2 = 2^-(1 + a S(a)),   a = min(|x|, 9)

S: weighted minimax fit (Lawson iteration on Chebyshev least squares; weight a, because the relative error of Phi(-a) is
ln 2 * a * dS) of -log2(erfc(a / sqrt 2)) / a on [0, 9].  Prints, per degree, the fp32 coefficients, the fit error, and — evaluating
the formula in fp32 the way the kernel does — (1) the number of bf16 ARGUMENTS (all of them) whose bf16 result differs from the
correctly rounded GELU (fp64, erfc form), (2) the same rate for fp32 arguments drawn from N(0, sigma), next to the Abramowitz &
Stegun 7.1.26 erf form the kernels used before (erf_as_f, still used by GELU').  CPU only.
    python tools/gelu_fit.py [degree ...]"""
import sys

import numpy as np
import torch
from numpy.polynomial import chebyshev as C
from scipy.special import erfc

XMAX = 9.0


def fit(deg, xmax=XMAX, nodes=6000, iters=400):
    k = np.arange(nodes)
    a = np.maximum((np.cos(np.pi * (k + 0.5) / nodes) + 1) / 2 * xmax, 1e-9)
    S = -np.log2(erfc(a / np.sqrt(2))) / a
    w, best = a.copy(), None
    for _ in range(iters):
        ch = C.Chebyshev.fit(a, S, deg, w=w, domain=[0, xmax])
        err = np.abs(a * (ch(a) - S))
        if best is None or err.max() < best[1]:
            best = (ch, err.max())
        w = w * (1 + err / err.max())
        w /= w.max()
    return best[0].convert(kind=np.polynomial.Polynomial).coef, best[1] * np.log(2)


def f32(v):
    return np.asarray(v, dtype=np.float64).astype(np.float32)


def gelu_exp2(x, c):
    """the kernel's arithmetic: fp32 values, every fma rounded once"""
    x = x.astype(np.float32)
    a = np.minimum(np.abs(x), np.float32(XMAX))
    s = np.full_like(a, np.float32(c[-1]))
    for k in c[-2::-1]:
        s = f32(s.astype(np.float64) * a + np.float64(np.float32(k)))
    e = f32(a.astype(np.float64) * s + 1.0)
    h = f32(np.exp2(-e.astype(np.float64)))
    return f32(np.maximum(x, 0).astype(np.float64) - a.astype(np.float64) * h)


def gelu_as(x):
    x = x.astype(np.float32)
    z = np.abs(x) * np.float32(0.70710678118654752)
    t = (np.float32(1) / (np.float32(0.3275911) * z + np.float32(1))).astype(np.float32)
    e = np.exp(-(z * z)).astype(np.float32)
    p = np.float32(1.061405429) * t + np.float32(-1.453152027)
    for k in (1.421413741, -0.284496736, 0.254829592):
        p = (p * t + np.float32(k)).astype(np.float32)
    r = (np.float32(1) - p * t * e).astype(np.float32)
    return (np.float32(0.5) * x * (np.float32(1) + np.copysign(r, x))).astype(np.float32)


def exact(x):
    x = x.astype(np.float64)
    return 0.5 * x * erfc(-x / np.sqrt(2))


def bf16(v):
    return torch.from_numpy(np.asarray(v, dtype=np.float32)).bfloat16()


def main():
    degs = [int(a) for a in sys.argv[1:]] or [7, 9]
    allb = (np.arange(65536, dtype=np.uint32) << 16).view(np.float32)
    allb = allb[np.isfinite(allb)]
    rng = np.random.default_rng(1)
    for deg in degs:
        c, e = fit(deg)
        print(f"degree {deg}: relative error of Phi(-a) on [0, {XMAX:g}] {e:.2e}")
        print("   ", ", ".join(f"{np.float32(v):.9e}f" for v in c))
        ex = exact(allb)
        live = np.abs(ex) >= 1e-17
        for name, y in (("exp2 form", gelu_exp2(allb, c)), ("A&S erf form", gelu_as(allb))):
            diff = (bf16(y) != bf16(ex)).numpy() & live
            print(f"    all {int(live.sum())} bf16 arguments with |GELU| >= 1e-17, {name}: {int(diff.sum())} results differ from the correctly rounded one")
        for sig in (0.5, 1.0, 3.0):
            x = (rng.standard_normal(4_000_000) * sig).astype(np.float32)
            ex = exact(x)
            live = np.abs(ex) >= 1e-17
            rates = [float(np.mean((bf16(y) != bf16(ex)).numpy()[live])) for y in (gelu_exp2(x, c), gelu_as(x))]
            print(f"    fp32 arguments ~ N(0, {sig:g}): results one or more bf16 steps off: exp2 form {rates[0]:.2e}, A&S erf form {rates[1]:.2e}")


if __name__ == "__main__":
    main()
