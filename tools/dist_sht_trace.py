"""N ranks on ONE GPU (gloo, host-staged exchanges) running the distributed SHT pair at BASELINE config 3 / 5 shapes
(721 x 1440, lmax 240, mmax 241), forward + backward, so that a rocprofv3 kernel trace of the run shows which kernels the
exchange schedule costs besides the transforms themselves:

    rocprofv3 --kernel-trace --stats -d out -- python tools/dist_sht_trace.py --h 4 --w 2 --fused 1
    python tools/dist_sht_trace.py --summarize out            # kernel classes per rank process

--fused 0 = the transpose-by-transpose schedule (split -> contiguous -> all_to_all -> cat around every exchange),
--fused 1 = makani_amd/dist_pipeline.py (the FFT kernels address per-peer slabs; one copy left per transform, on the S side)."""
import argparse
import collections
import csv
import glob
import os
import re
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, h, w, planes, reps, fused):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["MAKANI_AMD_DIST_FUSED"] = str(fused)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import makani_amd.comm as mcomm
    import makani_amd.distributed as thd
    from makani_amd import dist_pipeline as dp
    dev = torch.device("cuda:0")
    _, ih, iw = mcomm.init(h, w)
    assert thd.ensure_initialized()
    kw = dict(lmax=240, mmax=241, grid="equiangular")
    fwd = thd.DistributedRealSHT(721, 1440, **kw).to(dev)
    inv = thd.DistributedInverseRealSHT(721, 1440, **kw).to(dev)
    assert dp.eligible(fwd, torch.bfloat16) == bool(fused)
    hl, wl = fwd.lat_shapes[ih], fwd.lon_shapes[iw]
    torch.manual_seed(rank)
    x = torch.randn(1, planes, hl, wl, device=dev).bfloat16().requires_grad_(True)
    g = torch.randn(1, planes, hl, wl, device=dev).bfloat16()

    def step():
        S = fwd.analysis(x)
        y = inv.synthesis(S, 1, planes, out_dtype=torch.bfloat16)
        (y * g).sum().backward()
        x.grad = None
    step()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 0:
        print(f"h{h} w{w} planes {planes} fused={fused}: {(time.perf_counter() - t0) / reps * 1e3:.1f} ms per round trip "
              f"(forward + backward; exchanges host-staged through gloo)", flush=True)
    dist.destroy_process_group()


CLASSES = [("fft", r"rfft|irfft"), ("legendre_gemm", r"xgemm|sgemm|presplit"), ("concatenate", r"CatArray"),
           ("copy / cast / pad (torch elementwise)", r"elementwise|copy|fill|pad|Memcpy"), ("other", r".")]


def summarize(path):
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for f in sorted(glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True)):
        pid = re.findall(r"(\d+)_kernel_stats", os.path.basename(f))
        for row in csv.DictReader(open(f)):
            name = row.get("Name") or row.get("KernelName") or ""
            calls = int(float(row.get("Calls", 0) or 0))
            ns = float(row.get("TotalDurationNs", 0) or 0)
            for cls, pat in CLASSES:
                if re.search(pat, name):
                    per[pid[0] if pid else f][cls][0] += calls
                    per[pid[0] if pid else f][cls][1] += ns / 1e6
                    break
    shown = False
    for pid, d in per.items():
        if sum(v[0] for v in d.values()) < 50:
            continue                                  # the launcher process
        print(f"process {pid}: " + "; ".join(f"{k}: {v[0]} launches, {v[1]:.2f} ms" for k, v in d.items()))
        if not shown:                                 # the kernels of one rank, by time
            shown = True
            f = [g for g in glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True) if pid in os.path.basename(g)][0]
            rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))[:14]
            for r in rows:
                print(f"    {float(r['TotalDurationNs']) / 1e6:9.3f} ms {int(float(r['Calls'])):5d} x  {(r.get('Name') or '')[:110]}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=4)
    ap.add_argument("--w", type=int, default=2)
    ap.add_argument("--planes", type=int, default=96)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--fused", type=int, default=1)
    ap.add_argument("--summarize", default=None)
    a = ap.parse_args()
    if a.summarize:
        return summarize(a.summarize)
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(worker, args=(a.h * a.w, port, a.h, a.w, a.planes, a.reps, a.fused), nprocs=a.h * a.w, join=True)


if __name__ == "__main__":
    main()
