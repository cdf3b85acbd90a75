"""Roofline budget of one sfno_sc3_layers8_edim384 train step (721x1440x73, B = 1) next to what the rocprofv3 trace of
bench.py measured: for every kernel family the algorithmic flops and bytes per step, the time the binding roof allows
(max of flops / matrix peak and bytes / 8 TB/s) and the measured time.

    python tools/step_model.py profiles/r01_final_bench_kernel_stats.md 13 > profiles/r01_step_roofline_budget.md

(second argument: number of steps in the trace).  Algorithmic work follows SURVEY.md §8d / Appendix B (dense formulation)."""
import re
import sys

C, Cin, HID = 384, 73, 768
BIG, SMALL = 721 * 1440, 240 * 480
L, M = 240, 241
PEAK_BF16, PEAK_X6, PEAK_HBM = 2500e12, 2500e12 / 6, 8e12


def conv_mix():
    """(out channels, in channels, pixels, launches per pass) of every 1x1 convolution of the network"""
    return [(HID, C, SMALL, 7), (C, HID, SMALL, 7), (C, C, SMALL, 7),          # fc1, fc2, outer skip of blocks 0-6
            (HID, C, BIG, 1), (C, HID, BIG, 1), (C, C, BIG, 1),                  # block 7
            (C, Cin, BIG, 1), (C, C, BIG, 1),                                    # encoder
            (C, C, BIG, 1), (Cin, C, BIG, 1),                                    # decoder
            (Cin, Cin, BIG, 1)]                                                  # residual_transform


def budget():
    rows = []
    fl = sum(2.0 * m * k * n * c for m, k, n, c in conv_mix())
    by = sum(2.0 * n * (m + k) * c for m, k, n, c in conv_mix())
    rows.append(("channel GEMMs forward + data gradient (library)", r"^Cijk_", 2 * fl, 2 * by, PEAK_BF16))
    rows.append(("channel GEMM weight gradient (conv_wgrad_kernel + reduce_splits)", r"conv_wgrad_kernel|reduce_splits", fl, by, PEAK_BF16))
    leg = 4.0 * C * L * M * (6 * 721 + 30 * 240)                    # 3 + 3 transforms at K = 721, 15 + 15 at K = 240 (fwd + bwd)
    leg_b = 4.0 * (6 * (2 * C * M * 721 + 2 * C * L * M + M * L * 721) + 30 * (2 * C * M * 240 + 2 * C * L * M + M * L * 240))
    rows.append(("Legendre analysis / synthesis (xgemm_kernel)", r"xgemm_kernel", leg, leg_b, PEAK_X6))
    dh = 24 * 8.0 * C * C * L * M
    dh_b = 24 * 4.0 * (4 * C * L * M + 2 * C * C * L)
    rows.append(("dhconv fwd / dgrad / wgrad (xcgemm_kernel)", r"xcgemm_kernel", dh, dh_b, PEAK_X6))
    fft_b = 6 * C * 721 * (1440 * 2 + M * 8) + 30 * C * 240 * (480 * 2 + M * 8)       # bf16 grid side, fp32 spectrum side
    rows.append(("longitude FFTs (rfft / irfft_fast_kernel)", r"fft_fast_kernel", 0.0, float(fft_b), None))
    plane = 2.0 * C * (14 * SMALL + 2 * BIG)                        # one pass over the 16 normalised bf16 tensors
    rows.append(("instance norm fwd (3 passes) + bwd (5 passes)", r"in_apply|in_stats|in_bwd|sum_chunks_final", 0.0, 8 * plane, None))
    act = 2.0 * (HID * (7 * SMALL + BIG) + 2 * C * BIG)             # GELU inputs: MLP hidden x 8, encoder, decoder
    rows.append(("bias + GELU fwd (2 passes) + bwd (3 passes)", r"bias_gelu", 0.0, 5 * act, None))
    npar = 572.5e6
    rows.append(("AdamW (28 B / parameter) + gradient norm (4 B)", r"adamw|sumsq|clip_coef", 0.0, 32 * npar, None))
    rows.append(("spectral weight re-layout (2 x read + write of 2.26 GB)", r"weight_to_w|w_to_weight_grad", 0.0, 4 * 8.0 * 8 * C * C * L, None))
    return rows


def measured(path, steps):
    out = []
    for line in open(path):
        m = re.match(r"\| `([^`]+)` \| (\d+) \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \|", line)      # the 5-column (kernel) table only
        if m:
            out.append((m.group(1), float(m.group(3)) / steps))
    return out


def main(path, steps):
    meas = measured(path, steps)
    used = set()
    print("| kernel family | GFLOP / step | GB / step | roof | roofline time (ms) | measured (ms / step) | fraction of roof |")
    print("|---|---:|---:|---|---:|---:|---:|")
    tot_i = tot_m = 0.0
    for name, pat, fl, by, peak in budget():
        t_f = fl / peak if peak else 0.0
        t_b = by / PEAK_HBM
        ideal = max(t_f, t_b) * 1e3
        roof = "HBM" if t_b >= t_f else ("bf16 MFMA" if peak == PEAK_BF16 else "split-bf16 MFMA")
        ms = 0.0
        for k, v in meas:
            if re.search(pat, k):
                ms += v
                used.add(k)
        tot_i += ideal
        tot_m += ms
        print(f"| {name} | {fl / 1e9:,.0f} | {by / 1e9:.1f} | {roof} | {ideal:.2f} | {ms:.2f} | {ideal / ms if ms else 0:.2f} |")
    rest = sum(v for k, v in meas if k not in used)
    print(f"| everything else (casts, adds, fills, loss, layout conversions) | | | | | {rest:.2f} | |")
    print(f"| **sum** | | | | **{tot_i:.1f}** | **{tot_m + rest:.1f}** (kernel time in the trace; the untraced step takes 51.8 ms) | {tot_i / (tot_m + rest):.2f} |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 13)
