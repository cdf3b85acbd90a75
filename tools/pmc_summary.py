"""Per-kernel means of rocprofv3 `--pmc ... --output-format csv` counter files.
Usage: python tools/pmc_summary.py out.md file1_counter_collection.csv [file2 ...]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)[:90]


def main(out, files):
    agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                a = agg[short(row["Kernel_Name"])][row["Counter_Name"]]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    counters = sorted({c for k in agg.values() for c in k})
    lines = ["| kernel | dispatches | " + " | ".join(f"{c} (mean/dispatch)" for c in counters) + " |",
             "|---|---:|" + "---:|" * len(counters)]
    for k, d in sorted(agg.items(), key=lambda kv: -sum(v[1] for v in kv[1].values())):
        n = max(v[0] for v in d.values())
        lines.append(f"| `{k}` | {n} | " + " | ".join(f"{d[c][1] / d[c][0]:.4g}" if c in d else "" for c in counters) + " |")
    txt = "\n".join(lines) + "\n"
    open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
