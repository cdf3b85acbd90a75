"""ONE process: do kernels of this package disturb each other when they share compute units?

Round 5 (docs/LAB_NOTEBOOK.md 5.1) found that with several PROCESSES on one GPU the bf16 instance-norm backward and the
inverse FFT return transiently wrong values whenever another process runs the bf16 channel GEMMs on the same compute units,
and answered with disjoint compute-unit masks.  Whether that is a race in the victims (exposed by any co-residency, and then
also by a second STREAM of the same process: RCCL kernels, the chunked FFTs of the distributed path) or a property of
multi-process sharing on these boxes was left open.  This tool settles it:

  poison    single stream.  A probe kernel fills every compute unit's LDS (160 KB per workgroup) / 200 vector registers per
            lane with NaN / Inf / 1.0 patterns right before each victim launch sequence.  A victim that reads LDS or
            registers it never wrote changes its results.
  streams   two streams.  Culprits (conv1x1_nn, conv1x1_wgrad, chan_gemm_f32 — the operation classes that disturbed other
            processes) loop on stream B while the victims run REPS times on stream A; every result is compared bit by bit
            with the victim's result on an idle GPU.  Control: the same with a non-culprit class (rfft) on stream B.

    python tools/two_stream_hunt.py [--reps 200] [--mode poison,streams]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from race_hunt import _flat, _same, cases  # noqa: E402

VICTIMS = ("instnorm", "irfft", "rfft 721", "bias_gelu")
CULPRITS = {"conv1x1_nn": ("conv1x1_nn m768 k384 181x720", "conv1x1_nn m384 k768 181x720", "conv1x1_nn+bias+gelu+pre m768 k384 181x720"),
            "conv1x1_wgrad": ("conv1x1_wgrad m768 k384 181x720", "conv1x1_wgrad+bias m384 k384 181x1440"),
            "chan_gemm_f32": ("chan_gemm_f32 m384 k384", "chan_wgrad_f32 m384 k384"),
            "control:rfft": ("rfft 721x1440 torch.bfloat16", "rfft 240x480 torch.float32")}


def probe_lib():
    path = os.path.join(ROOT, "tools", "probes", "liblds_poison.so")
    lib = ctypes.CDLL(path)
    lib.mk_probe_lds_poison.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
    lib.mk_probe_vgpr_poison.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return lib


def compare(name, outs, ref, tag, stats):
    bad = 0
    worst = 0.0
    for a, b in zip(_flat(outs), ref):
        n, e = _same(a, b)
        bad += n
        worst = max(worst, e)
    s = stats.setdefault((tag, name), [0, 0, 0.0])
    s[0] += 1
    s[1] += int(bad > 0)
    s[2] = max(s[2], worst)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--mode", default="poison,streams")
    a = ap.parse_args()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29577"
    dist.init_process_group("gloo", rank=0, world_size=1)
    dev = torch.device("cuda:0")
    allc = cases(dev)
    victims = [(n, f) for n, f in allc if any(v in n for v in VICTIMS)]
    torch.cuda.synchronize()
    refs = {}
    for n, f in victims:
        refs[n] = [t.clone() for t in _flat(f())]
    # idle repeatability first (must be 0 differences, else everything below is moot)
    stats = {}
    for n, f in victims:
        for _ in range(5):
            compare(n, f(), refs[n], "idle", stats)
    torch.cuda.synchronize()

    if "poison" in a.mode:
        lib = probe_lib()
        sink = torch.zeros(1024, dtype=torch.int32, device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for pat, pname in ((0xFFFFFFFF, "nan"), (0x7F800000, "inf"), (0x3F800000, "one"), (0x7F807F80, "bf16inf")):
            for n, f in victims:
                for _ in range(3):
                    rc = lib.mk_probe_lds_poison(pat, 160 * 1024, 1024, 256, 0, ctypes.c_void_p(sink.data_ptr()), st)
                    rc |= lib.mk_probe_lds_poison(pat, 64 * 1024, 2048, 256, 0, ctypes.c_void_p(sink.data_ptr()), st)
                    rc |= lib.mk_probe_vgpr_poison(pat, 8192, ctypes.c_void_p(sink.data_ptr()), st)
                    assert rc == 0, rc
                    compare(n, f(), refs[n], f"poison:{pname}", stats)
        torch.cuda.synchronize()

    if "streams" in a.mode:
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        for cname, pats in CULPRITS.items():
            cul = [(n, f) for n, f in allc if any(n.startswith(p) for p in pats)]
            assert cul, cname
            for n, f in victims:
                torch.cuda.synchronize()
                for r in range(a.reps):
                    with torch.cuda.stream(sb):
                        for _, cf in cul:
                            cf()
                    with torch.cuda.stream(sa):
                        out = f()
                        outs = [t.clone() for t in _flat(out)]
                    if r % 8 == 7:
                        torch.cuda.synchronize()
                    sa.synchronize()
                    compare(n, outs, refs[n], f"streams:{cname}", stats)
                torch.cuda.synchronize()

    tags = []
    for (tag, n) in stats:
        if tag not in tags:
            tags.append(tag)
    total_bad = 0
    for tag in tags:
        rows = [(n, s) for (t, n), s in stats.items() if t == tag]
        nbad = sum(s[1] for _, s in rows)
        total_bad += nbad if tag != "idle" else 0
        print(f"== {tag}: {sum(s[0] for _, s in rows)} victim runs, {nbad} differ from the idle-GPU result")
        for n, s in rows:
            if s[1]:
                print(f"   DIFF  {n}: {s[1]} of {s[0]} runs, worst rel-L2 {s[2]:.2e}")
    print(f"RESULT: {total_bad} differing victim runs under poison / second-stream load")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
