"""The counted ``s_waitcnt vmcnt(N)`` of the weight-stationary channel GEMM (csrc/conv1x1.hip), checked without a GPU:
tools/vmcnt_check.py restates the kernel's issue / wait schedule with its template constants and verifies (i) against an
in-order retirement model that every LDS-DMA chunk and operand image has landed when it is used — all instantiations, stream
lengths from 0 tiles, first tiles and end of stream included — and that the steady-state waits carry no slack, (ii) against
the compiled kernel (hipcc -S) that the code holds exactly the memory instructions the model counts."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_wait_counts_of_every_instantiation_against_the_retirement_model():
    import vmcnt_check as vc
    vc.check_model(verbose=False)
    # the checker must be able to fail: a count one too large reads a chunk early
    v = vc.Variant(3, 128, False, False)
    orig = v.chunk_wait
    v.chunk_wait = lambda ts, kc, n: (orig(ts, kc, n) + 1) if orig(ts, kc, n) not in (None, 0) else orig(ts, kc, n)
    with pytest.raises(AssertionError):
        for tiles in range(0, 14):
            vc.simulate(v, tiles)


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_compiled_kernels_carry_exactly_the_memory_instructions_of_the_model():
    import vmcnt_check as vc
    vc.check_isa(verbose=False)
