"""profiles/rNN_shard_shapes.md from the JSON files of tools/shadow_rank.py: per kernel family the time ONE rank of h2w1 / h4w1 /
h4w2 spends per step at its real shard shapes, its efficiency against the serial kernel (serial time / ranks = what perfect
shard-shape scaling would give), the host's launch path, what the rank puts on the links, and the projected step / speed-up.

    python tools/shard_table.py gpurun_out/r05x/shadow_h1w1.json gpurun_out/r05x/shadow_h2w1.json ... > profiles/r05_shard_shapes.md
"""
import json
import sys


def fam_group(name):
    for key, label in (("conv1x1_nn", "channel GEMM fwd / dgrad"), ("conv1x1_wgrad", "channel GEMM wgrad"), ("legendre", "Legendre"),
                       ("dhconv", "dhconv"), ("fft", "FFT"), ("instnorm", "instance norm"), ("adamw", "AdamW + clip"),
                       ("sumsq", "AdamW + clip"), ("clip", "AdamW + clip")):
        if key in name:
            return label
    return "other HIP kernels"


def main(paths):
    runs = [json.load(open(p)) for p in paths]
    serial = next((r for r in runs if r["parallelism"] == "h1w1"), None)
    if serial is None:
        raise SystemExit("needs the h1w1 run as the baseline")

    def groups(r):
        g = {}
        for k, v in r["families"].items():
            g[fam_group(k)] = g.get(fam_group(k), 0.0) + v["ms_per_step"]
        return g
    gs = groups(serial)
    order = sorted(gs, key=lambda k: -gs[k])
    print("# Shard-shaped kernels: one rank of the h x w split, alone on one MI355X (tools/shadow_rank.py)\n")
    print("Every collective is a phantom (sizes and ranks real, no data moved), so each kernel runs at the rank's real shape with nothing "
          "else on the GPU.  `eff` = (serial kernel time / ranks) / shard kernel time: 1.00 = the kernel keeps its full-size efficiency "
          "on the shard.  The heaviest polar rank (the last: all of its 60 degrees x all orders are live) is shown; dhconv work is 2x the "
          "average there (the triangle), so its `eff` is bounded by 0.5 x ranks / (h ...) — see DESIGN.md §8.\n")
    hdr = "| kernel family | serial ms |" + "".join(f" {r['parallelism']} ms | eff |" for r in runs if r is not serial)
    print(hdr)
    print("|---|---:|" + "---:|---:|" * (len(runs) - 1))
    for k in order:
        row = f"| {k} | {gs[k]:.2f} |"
        for r in runs:
            if r is serial:
                continue
            n = int(r["parallelism"][1]) * int(r["parallelism"][3])
            t = groups(r).get(k, 0.0)
            row += f" {t:.2f} | {(gs[k] / n / t if t else 0):.2f} |"
        print(row)
    row = f"| **sum of HIP kernels** | **{serial['hip_kernel_ms_per_step']:.2f}** |"
    for r in runs:
        if r is serial:
            continue
        n = int(r["parallelism"][1]) * int(r["parallelism"][3])
        row += f" **{r['hip_kernel_ms_per_step']:.2f}** | {serial['hip_kernel_ms_per_step'] / n / r['hip_kernel_ms_per_step']:.2f} |"
    print(row)
    print()
    print("| | " + " | ".join(r["parallelism"] for r in runs) + " |")
    print("|---|" + "---:|" * len(runs))
    print("| local grid | " + " | ".join(r["local_grid"] for r in runs) + " |")
    print("| C-ABI launches per step | " + " | ".join(f"{r['launches_per_step']:.0f}" for r in runs) + " |")
    print("| HIP kernel time per step (ms) | " + " | ".join(f"{r['hip_kernel_ms_per_step']:.2f}" for r in runs) + " |")
    print("| eager step, phantom collectives: wall (ms) | " + " | ".join(f"{r['wall_ms_per_step_eager_phantom']:.1f}" for r in runs) + " |")
    print("| the same step replayed as a hipGraph (ms) | " + " | ".join(f"{r.get('graph_ms_per_step_phantom') or float('nan'):.1f}" for r in runs) + " |")
    print("| ... of which the launching thread is busy (ms) | " + " | ".join(f"{r['host_launch_ms_per_step']:.1f}" for r in runs) + " |")
    print("| MB this rank sends per step | " + " | ".join(f"{sum(v['MB_sent'] for v in r['exchange_per_step'].values()):.0f}" for r in runs) + " |")
    print("| all-to-alls per step | " + " | ".join(f"{sum(v['all_to_alls'] for v in r['exchange_per_step'].values()):.0f}" for r in runs) + " |")
    print("| link time at 153 GB/s, peers in parallel (ms) | " + " | ".join(f"{r['link_ms_per_step_at_153GBs']:.2f}" for r in runs) + " |")
    print("| peak HBM (GB) | " + " | ".join(f"{r['peak_hbm_GB']:.1f}" for r in runs) + " |")
    print()
    print("Projected step per rank.  Eager launches: max(GPU work + link time, host launch path).  Captured (bench.py replays the step, "
          "RCCL collectives included, as one hipGraph): GPU work + the link time that is not hidden behind compute (shown for the two "
          "extremes: all of it exposed / all of it hidden).  GPU work = the replayed phantom step (every kernel incl. torch glue, no "
          "launch gaps); link time = bytes / (153 GB/s x peers in parallel).\n")
    print("| | " + " | ".join(r["parallelism"] for r in runs) + " |")
    print("|---|" + "---:|" * len(runs))

    def gpu(r):
        return r.get("graph_ms_per_step_phantom") or (r["hip_kernel_ms_per_step"] + 1.0)
    base = gpu(serial)
    print("| eager: max(GPU + links, host) (ms) | " + " | ".join(
        f"{max(gpu(r) + r['link_ms_per_step_at_153GBs'], r['host_launch_ms_per_step']):.1f}" for r in runs) + " |")
    print("| captured, links exposed (ms) | " + " | ".join(f"{gpu(r) + r['link_ms_per_step_at_153GBs']:.1f}" for r in runs) + " |")
    print("| captured, links hidden (ms) | " + " | ".join(f"{gpu(r):.1f}" for r in runs) + " |")
    print("| speed-up over the serial captured step: eager | " + " | ".join(
        f"{base / max(gpu(r) + r['link_ms_per_step_at_153GBs'], r['host_launch_ms_per_step']):.2f}" for r in runs) + " |")
    print("| ... captured, links exposed | " + " | ".join(f"{base / (gpu(r) + r['link_ms_per_step_at_153GBs']):.2f}" for r in runs) + " |")
    print("| ... captured, links hidden | " + " | ".join(f"{base / gpu(r):.2f}" for r in runs) + " |")


if __name__ == "__main__":
    main(sys.argv[1:])
