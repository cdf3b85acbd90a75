"""python bench.py ... | python tools/bench_brief.py [substr ...]: the bench line in short + the kernel-table rows whose name contains a substring"""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d.get("roofline", {})
print(f"{d['ms_per_step']:.2f} ms/step  {d['value']:.3f} {d['unit']}  dominant {r.get('kernel')} frac {r.get('frac')}  loss {d.get('final_loss')}")
for k, v in d.get("hip_kernels", {}).items():
    if any(s in k for s in sys.argv[1:]):
        print(f"   {k:44s} x{v['launches_per_step']:4.1f}  {v['ms_avg']:.4f} ms  {v['ms_per_step']:.3f} ms/step  {v.get('GBps_algorithmic')} GB/s")
