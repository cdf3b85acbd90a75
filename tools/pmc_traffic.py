"""HBM bytes per launch of the bench's kernel families from two rocprofv3 PMC passes of the SAME bench command
(`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, summarised per kernel symbol by tools/pmc_summary.py).
   python tools/pmc_traffic.py fetch.md write.md out.json [sfno|fcn3]
FETCH_SIZE counts KiB and reports half of a coalesced 16 B/lane read stream on gfx950 (MI355X_MICROARCH.md, HBM): x2,
except for the inverse FFTs whose F-side reads are 32/64-byte runs (uncalibrated, taken as is: a lower bound).
WRITE_SIZE is KiB, calibrated on kernels of known traffic in the same runs (weight_to_w writes 283.1 MB = 2.765e5 KiB)."""
import json
import re
import sys

FAMILIES = [          # (family, regex on the kernel symbol, fetch correction)
    ("conv1x1_nn", r"^conv_nn_(astat2?|ring)?_?kernel", 2),
    ("conv1x1_wgrad", r"^conv_wgrad_(ring_)?kernel", 2),
    ("dhconv_fwd", r"^xcgemm2?_kernel<true, false", 2),
    ("dhconv_dgrad", r"^xcgemm2?_kernel<true, true", 2),
    ("dhconv_wgrad", r"^xcgemm2?_kernel<false, false", 2),
    ("legendre", r"^xgemm2?_kernel", 2),
    ("rfft_1440", r"^rfft_fast_kernel<720", 2),
    ("rfft_480", r"^rfft_fast_kernel<240", 2),
    ("irfft_1440", r"^irfft_fast_kernel<720", 1),
    ("irfft_480", r"^irfft_fast_kernel<240", 1),
    ("adamw", r"^adamw_kernel", 2),
]


FAMILIES_FCN3 = [     # bench.py --config fcn3_sc2_edim45_layers10
    ("disco_fwd", r"^disco_(fused_fwd|runs_fwd|fwd)_kernel", 2),
    ("disco_bwd", r"^disco_(runs_bwd|bwd_same|bwd)_kernel", 2),
    ("conv1x1_nn", r"^conv_nn_(astat2?|ring)?_?kernel", 2),
    ("conv1x1_wgrad", r"^conv_wgrad_(ring_)?kernel", 2),
    ("resample_bwd", r"^resample_bwd", 2),
    ("resample_fwd", r"^resample_fwd", 2),
    ("group_mix", r"^group_mix_kernel", 2),
]


def table(path):
    out = {}
    for line in open(path):
        if not line.startswith("| `"):
            continue
        c = [x.strip() for x in line.strip().strip("|").split("|")]
        out[c[0].strip("`")] = (int(c[1]), float(c[2]))
    return out


def main(fetch_md, write_md, out_json, kind="sfno"):
    global FAMILIES
    if kind == "fcn3":
        FAMILIES = FAMILIES_FCN3
    F, W = table(fetch_md), table(write_md)
    res = {"_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, two separate passes over `python bench.py --steps 2 --warmup 1 "
                      "--graph off` (the bench workload itself: sfno_sc3_layers8_edim384, 721x1440, B=1, bf16 autocast), MI355X",
           "_units": __doc__.split("\n", 3)[3].strip()}
    for fam, rx, corr in FAMILIES:
        ks = [k for k in F if re.match(rx, k)]
        n = sum(F[k][0] for k in ks)
        if not n:
            continue
        fetch = sum(F[k][0] * F[k][1] for k in ks) / n
        write = sum(W[k][0] * W[k][1] for k in ks if k in W) / n
        extra = 0.0
        if fam == "conv1x1_wgrad":        # the second-stage reduction belongs to the same C-ABI call
            for k in F:
                if k.startswith("reduce_splits"):
                    extra += F[k][0] * (F[k][1] * 2 + W.get(k, (0, 0.0))[1]) / n
        res[fam] = {"kernels": ks, "launches_in_profile": n, "fetch_size_kib": round(fetch, 1), "write_size_kib": round(write, 1),
                    "fetch_correction": corr, "hbm_bytes": int((fetch * corr + write + extra) * 1024)}
    json.dump(res, open(out_json, "w"), indent=1)
    for k, v in res.items():
        if not k.startswith("_"):
            print(f"{k:16s} {v['hbm_bytes'] / 1e6:9.1f} MB per launch  ({v['launches_in_profile']} launches)")


if __name__ == "__main__":
    main(*sys.argv[1:5])
