"""Packed-fp32 forwarding probe (tools/probes/pk_hazard.hip) under the load that disturbs the package's packed-fp32 kernels
(docs/LAB_NOTEBOOK.md, round 6): the culprit loops on stream B, every (producer, gap, consumer) kernel runs on stream A.

    python tools/pk_hazard_probe.py [--culprit chan_gemm_f32] [--reps 10]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from two_stream_micro import culprits  # noqa: E402



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--culprit", default="all")
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--iters", type=int, default=4000)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "probes", "libpk_hazard.so"))
    lib.mk_probe_pk_hazard.argtypes = [ctypes.c_int] * 2 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.mk_probe_pk_form_name.restype = ctypes.c_char_p
    nforms = lib.mk_probe_pk_forms()
    blocks = 4096
    x = torch.randn(2 * blocks * 256, device=dev)
    out = torch.zeros(blocks * 256, dtype=torch.int32, device=dev)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    cul = [("idle", None)] + culprits(dev, a.culprit)
    torch.cuda.synchronize()
    for cname, cf in cul:
        for f in range(nforms):
            for g in range(2):
                tot, nthreads = 0, 0
                for r in range(a.reps):
                    if cf is not None:
                        with torch.cuda.stream(sb):
                            for _ in range(4):
                                cf()
                    with torch.cuda.stream(sa):
                        rc = lib.mk_probe_pk_hazard(f, g, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()), blocks, a.iters,
                                                    ctypes.c_void_p(sa.cuda_stream))
                        assert rc == 0, rc
                    sa.synchronize()
                    tot += int(out.sum())
                    nthreads += int((out != 0).sum())
                torch.cuda.synchronize()
                flag = "WRONG" if tot else "ok"
                print(f"{cname:>14s} | {lib.mk_probe_pk_form_name(f).decode():<30s} | gap {'s_nop 3' if g else 'none   '} | {flag:5s} {tot} wrong results in {nthreads} lanes "
                      f"of {a.reps * blocks * 256} ({a.reps * blocks * 256 * a.iters * 2:.2e} results)", flush=True)


if __name__ == "__main__":
    main()
