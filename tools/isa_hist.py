"""Instruction-class histogram of one kernel in a hipcc -S dump (static counts; the FFT / contraction kernels are VALU-issue
bound, so the static mix of their fully unrolled item loop is the proxy that can be optimised without a GPU).
    python tools/isa_hist.py file.s <substring of the mangled kernel name> [top]"""
import collections
import re
import sys


def kernel_lines(path, key):
    out, on = [], False
    for line in open(path):
        if re.match(r"^_Z\w+:", line):
            on = key in line
            continue
        if on:
            if line.startswith("\t.section") or line.startswith(".Lfunc_end"):
                on = False
                continue
            out.append(line)
    return out


def classify(op):
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "v_mov"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    ops = collections.Counter()
    cls = collections.Counter()
    for line in kernel_lines(path, key):
        m = re.match(r"^\t([a-z_0-9]+)", line)
        if not m:
            continue
        op = m.group(1)
        ops[op] += 1
        cls[classify(op)] += 1
    print(dict(cls), "total", sum(cls.values()))
    for op, n in ops.most_common(top):
        print(f"  {n:6d} {op}")


if __name__ == "__main__":
    main()
