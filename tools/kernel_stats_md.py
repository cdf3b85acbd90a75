"""rocprofv3 `*kernel_stats.csv` -> the markdown table committed under profiles/.
   python tools/kernel_stats_md.py kernel_stats.csv <steps in the trace> "<title>" [sfno|fcn3] > profiles/rNN_..._kernel_stats.md"""
import csv
import re
import sys

FAMILIES = {
    "sfno": [("channel GEMM fwd/dgrad", r"conv_nn_"), ("channel GEMM wgrad", r"conv_wgrad_|reduce_splits"), ("dhconv", r"xcgemm2?_kernel"),
             ("Legendre", r"xgemm2?_kernel|sgemm"), ("FFT", r"fft_(fast_)?kernel"), ("instance norm", r"in_(stats|apply|bwd|fwd)"),
             ("AdamW + clip", r"adamw|sumsq|clip_coef"), ("plane sums", r"plane_sum|sum_chunks"), ("loss", r"quad_lp|spec_lp"),
             ("layout", r"weight_to_w|w_to_weight|slayout|complex_to"), ("torch glue", r"at::native|rocclr|Cijk")],
    "fcn3": [("DISCO contraction", r"disco_"), ("channel GEMM fwd/dgrad", r"conv_nn_"), ("channel GEMM wgrad", r"conv_wgrad_|reduce_splits"),
             ("SHT + dhconv (global blocks)", r"xc?gemm2?_kernel|fft_(fast_)?kernel|weight_to_w|w_to_weight|slayout|complex_to"),
             ("resampling", r"resample_"), ("grouped channel mix", r"group_mix"), ("library GEMMs (grouped channel mix, fp32)", r"Cijk"), ("plane sums", r"plane_sum|sum_chunks"),
             ("torch glue", r"at::native|rocclr")],
}


def main(path, steps, title, kind="sfno"):
    steps = float(steps)
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# {title}\n")
    print(f"{steps:g} executions of the step are in the trace: per-step = calls / {steps:g}.  Sum of kernel time {tot / 1e6 / steps:.2f} ms per step.\n")
    print("| kernel | calls | calls/step | total ms | avg us | ms/step | % |\n|---|---:|---:|---:|---:|---:|---:|")
    fam = {}
    for r in rows:
        name = re.sub(r"^void ", "", r["Name"]).replace("(anonymous namespace)::", "")
        short = name.split("(")[0][:100]
        ns = float(r["TotalDurationNs"])
        for f, rx in FAMILIES[kind]:
            if re.search(rx, name):
                fam[f] = fam.get(f, 0.0) + ns
                break
        else:
            fam["other"] = fam.get("other", 0.0) + ns
        if ns / tot > 2e-4:
            print(f"| `{short}` | {int(r['Calls'])} | {int(r['Calls']) / steps:.1f} | {ns / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | "
                  f"{ns / 1e6 / steps:.3f} | {100 * ns / tot:.2f} |")
    print("\nPer family (ms per step): " + ", ".join(f"{f} {v / 1e6 / steps:.2f}" for f, v in sorted(fam.items(), key=lambda kv: -kv[1])))


if __name__ == "__main__":
    main(*sys.argv[1:])
