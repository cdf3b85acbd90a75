"""Build libmakani_amd.so (gfx950) in-tree with hipcc.  No torch headers involved:
the library is a plain C-ABI shared object (include/makani_amd.h)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmakani_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def file_flags(src):
    """extra hipcc flags a source file asks for in a `// MK_HIPCC_FLAGS: ...` comment line (e.g. -fno-slp-vectorize for the
    files whose scalar arithmetic must not be auto-vectorised into packed-fp32 instructions: docs/LAB_NOTEBOOK.md, round 6)"""
    out = []
    with open(src) as f:
        for line in f:
            if line.startswith("// MK_HIPCC_FLAGS:"):
                out += line.split(":", 1)[1].split()
    return out


def _stale(obj, src):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [src] + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "makani_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, defines=(), tag=None):
    """Compile every csrc/*.hip to an object and link the shared library.
    ``tag`` / ``defines``: a variant build (``libmakani_amd_<tag>.so``, objects under ``build_<tag>/``) compiled with extra
    ``-D`` options — what tools/ab.py uses for A/B measurements inside one GPU call; the default build is untouched."""
    objdir = os.path.join(HERE, "build" + (f"_{tag}" if tag else ""))
    LIB = os.path.join(HERE, f"libmakani_amd_{tag}.so") if tag else globals()["LIB"]
    FLAGS = globals()["FLAGS"] + [d if d.startswith("-") else "-D" + d for d in defines]      # (-D..., or any other compiler flag)
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, src):
            cmd = [HIPCC] + FLAGS + file_flags(src) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    failed = [s for s, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError(f"hipcc failed for {failed}")
    if force or procs or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    argv = sys.argv[1:]
    tag = argv[argv.index("--tag") + 1] if "--tag" in argv else None
    print(build(force="--force" in argv, defines=[a for a in argv if a.startswith("-D")], tag=tag))
