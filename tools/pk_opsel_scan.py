"""Scan compiled gfx950 code for the packed-fp32 instruction forms that are unreliable on MI355X (docs/LAB_NOTEBOOK.md round 6).

Finding (tools/pk_hazard_probe.py, one process, two streams): while the split-bf16 GEMM of csrc/xgemm.hip runs on the same
compute units, ``v_pk_mul_f32`` / ``v_pk_add_f32`` / ``v_pk_fma_f32`` return wrong results (≈2e-6 of the results) when their
**src1 is a VGPR pair whose HIGH half feeds the low result lane while src0's LOW half does** (``op_sel:[0,1...]``, any
``op_sel_hi``).  The same swizzle on src0, on src2, on both src0 and src1 (``op_sel:[1,1...]``: the complex-multiply form),
on SGPR / literal sources, and ``v_pk_mov_b32`` are reliable (0 wrong in 2.5e10 results each; 18 forms probed).
Kernels of this package must not contain the unreliable forms: hand-written packed code puts the swizzled operand first
(csrc/fft_packed.h), files of scalar code are compiled without the SLP vectoriser (``// MK_HIPCC_FLAGS``).

    python tools/pk_opsel_scan.py [makani_amd/libmakani_amd.so | file.o ...]      exit code 1 if any unreliable form is found
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
_PK = re.compile(r"\b(v_pk_(?:mul|add|fma)_f32)\s+(\S+),\s*(\S+),\s*([^,\s]+)(?:,\s*(\S+))?(?:\s+(.*?))?\s*(?://.*)?$")


def unreliable(line: str) -> bool:
    """the rule the probe's 18 forms pin down: a VGPR src1 whose HIGH half feeds the low result lane (op_sel bit of src1 set)
    while src0's LOW half does (op_sel bit of src0 clear)"""
    m = _PK.search(line)
    if not m:
        return False
    _, _, _, s1, _, mods = m.groups()
    mo = re.search(r"op_sel:\[(\d),(\d)(?:,\d)?\]", mods or "")
    return bool(mo) and mo.group(2) == "1" and mo.group(1) == "0" and s1.startswith("v")


def disassemble(path: str) -> str:
    """device ISA of a HIP object / shared library: every offload bundle of its .hip_fatbin section (a shared library holds one
    bundle per linked object), gfx950 entry"""
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    text = []
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", path, os.path.join(td, "unused")])
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        for i, a in enumerate(starts):
            one, co = os.path.join(td, f"b{i}"), os.path.join(td, f"co{i}")
            with open(one, "wb") as f:
                f.write(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--targets={TARGET}", f"--input={one}",
                                   f"--output={co}"], stderr=subprocess.DEVNULL)
            text.append(subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout)
    return "\n".join(text)


def scan(path: str) -> dict:
    """kernel symbol -> list of unreliable instructions"""
    out, fn = {}, None
    for line in disassemble(path).splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            fn = m.group(1)
            continue
        if unreliable(line):
            out.setdefault(fn, []).append(line.split("//")[0].strip())
    return out


if __name__ == "__main__":
    paths = sys.argv[1:] or [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "makani_amd", "libmakani_amd.so")]
    total = 0
    for p in paths:
        bad = scan(p)
        n = sum(len(v) for v in bad.values())
        total += n
        print(f"{p}: {n} unreliable packed-fp32 instruction(s) in {len(bad)} kernel(s)")
        for k, v in sorted(bad.items(), key=lambda kv: -len(kv[1]))[:12]:
            print(f"   {len(v):4d}  {k[:120]}\n         e.g. {v[0]}")
    sys.exit(1 if total else 0)
