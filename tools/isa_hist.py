"""Instruction-class histogram of one kernel in a hipcc -S dump (static counts; the FFT / contraction kernels are VALU-issue
bound, so the static mix of their fully unrolled item loop is the proxy that can be optimised without a GPU).
    python tools/isa_hist.py file.s <substring of the mangled kernel name> [top] [--loops]
--loops: the same histogram for the three largest loops of the kernel (backward branches), i.e. without prologue / epilogue.
(Write the length-prefixed name, e.g. 16rfft_fast_kernelILi720, when one kernel name is a suffix of another.)"""
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


def histogram(lines, top):
    ops = collections.Counter()
    cls = collections.Counter()
    for line in lines:
        m = re.match(r"^\t([a-z_0-9]+)", line)
        if not m:
            continue
        op = m.group(1)
        ops[op] += 1
        cls[classify(op)] += 1
    print(dict(cls), "total", sum(cls.values()), "vector", cls["valu"] + cls["valu_pk"] + cls["v_mov"])
    for op, n in ops.most_common(top):
        print(f"  {n:6d} {op}")


def loops(lines):
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    out = []
    for i, l in enumerate(lines):
        m = re.match(r"^\ts_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            out.append((labels[m.group(1)], i))
    return sorted(out, key=lambda ab: ab[0] - ab[1])


def main():
    args = [a for a in sys.argv[1:] if a != "--loops"]
    path, key = args[0], args[1]
    top = int(args[2]) if len(args) > 2 else 25
    lines = kernel_lines(path, key)
    if "--loops" in sys.argv:
        for a, b in loops(lines)[:3]:
            print(f"loop: lines {a}..{b} of the kernel")
            histogram(lines[a:b + 1], top)
    else:
        histogram(lines, top)


if __name__ == "__main__":
    main()
