"""Same-box A/B measurements of kernel variants (the GPU boxes of the pool differ by up to 8 % on identical code, so a
variant is only ever compared with the baseline inside ONE gpurun call).

1. here (no GPU needed, hipcc cross-compiles):
       python tools/ab.py build base  rb8:-DMK_FFT_RB1440=8  nt256:-DMK_FFT_NT1440=256
   builds makani_amd/libmakani_amd_<tag>.so for every `tag[:-Dmacro=value,...]` (the kernel source must read the macro);
2. on the GPU box, inside one gpurun call:
       python tools/ab.py run base rb8 nt256 -- python tools/microbench.py fft
   runs the command once per variant with MAKANI_AMD_LIB pointing at it and prints the outputs side by side.
`base` without defines is the default library rebuilt under its own tag, so that all variants come from the same sources."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def lib_of(tag):
    return os.path.join(ROOT, "makani_amd", f"libmakani_amd_{tag}.so")


def build(specs):
    from makani_amd import build as mb
    for spec in specs:
        tag, _, defs = spec.partition(":")
        path = mb.build(defines=[d for d in defs.split(",") if d], tag=tag, verbose=False)
        print(f"{tag}: {path}")


def run(tags, cmd):
    outs = {}
    for tag in tags:
        if not os.path.exists(lib_of(tag)):
            raise SystemExit(f"{lib_of(tag)} is missing: run `python tools/ab.py build {tag}[:-D...]` first")
        r = subprocess.run(cmd, env=dict(os.environ, MAKANI_AMD_LIB=lib_of(tag)), capture_output=True, text=True)
        outs[tag] = [l for l in (r.stdout + r.stderr).splitlines() if l.strip() and "amdgpu.ids" not in l]
        if r.returncode:
            print(f"[{tag}] exit code {r.returncode}")
    n = max(len(v) for v in outs.values())
    for i in range(n):
        for tag in tags:
            line = outs[tag][i] if i < len(outs[tag]) else ""
            print(f"{tag:>10s} | {line}")
        print()


if __name__ == "__main__":
    if len(sys.argv) < 3 or sys.argv[1] not in ("build", "run"):
        raise SystemExit(__doc__)
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        sep = sys.argv.index("--")
        run(sys.argv[2:sep], sys.argv[sep + 1:])
