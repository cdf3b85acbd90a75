"""The compiled kernels contain none of the packed-fp32 instruction forms that are unreliable on MI355X while a matrix-core
kernel shares the compute unit (tools/pk_opsel_scan.py, docs/LAB_NOTEBOOK.md round 6), and — on the GPU — the former victims
(bf16 instance-norm backward, inverse FFT) are bit-reproducible while the former culprits run on a second stream."""
import math
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pk_opsel_scan  # noqa: E402

LIB = os.path.join(ROOT, "makani_amd", "libmakani_amd.so")


def test_classifier_knows_the_unreliable_forms():
    bad = ["v_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel:[0,1] op_sel_hi:[1,0]",
           "v_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel:[0,1]",
           "v_pk_add_f32 v[2:3], v[4:5], v[6:7] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]   // 0000: D3B2",
           "v_pk_fma_f32 v[8:9], v[2:3], v[6:7], v[10:11] op_sel:[0,1,0]",
           "v_pk_fma_f32 v[8:9], v[2:3], v[6:7], v[10:11] op_sel:[0,1,0] op_sel_hi:[1,0,1]"]
    good = ["v_pk_mul_f32 v[2:3], v[4:5], v[6:7]",
            "v_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel_hi:[1,0]",                         # broadcast of src1.lo
            "v_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel:[1,0] op_sel_hi:[0,1]",            # the swizzle on src0
            "v_pk_mul_f32 v[2:3], v[4:5], s[6:7] op_sel:[0,1] op_sel_hi:[1,0]",            # SGPR src1
            "v_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[8:9] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]",
            "v_pk_add_f32 v[2:3], v[4:5], v[6:7] op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0] neg_hi:[1,0]",
            "v_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel:[1,1] op_sel_hi:[0,0]",            # both swizzled
            "v_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[8:9] op_sel:[0,0,1] op_sel_hi:[1,1,0]"]  # src2
    assert all(pk_opsel_scan.unreliable(l) for l in bad)
    assert not any(pk_opsel_scan.unreliable(l) for l in good)


@pytest.mark.skipif(not os.path.exists(LIB), reason="libmakani_amd.so not built")
def test_library_has_no_unreliable_packed_fp32_forms():
    bad = pk_opsel_scan.scan(LIB)
    assert not bad, {k: v[:2] for k, v in list(bad.items())[:5]}


@pytest.mark.gpu
def test_victims_are_bit_reproducible_next_to_matrix_kernels_on_a_second_stream():
    """round 5 saw the bf16 instance-norm backward and the inverse FFT return transiently wrong values whenever the split-bf16 /
    bf16 channel GEMMs ran on the same compute units (another process — or, round 6, another STREAM).  Guard for the
    chunk-overlapped distributed path: victims on stream A, culprits looping on stream B, every result bit-identical to the
    idle-GPU result."""
    from makani_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    C, H, W = 384, 60, 480
    x = torch.randn(1, C, H, W, device=dev).bfloat16()
    gy = torch.randn(1, C, H, W, device=dev).bfloat16()
    gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1

    def norm(gelu):
        xr = x.clone().requires_grad_(True)
        gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
        y = ops.InstanceNormFn.apply(xr, gr, br, 1e-6, gelu)
        y.backward(gy)
        return [y.detach(), xr.grad, gr.grad, br.grad]
    c = 2 * math.pi / 480
    F = ops.rfft_rows(torch.rand(1, C, 240, 480, device=dev), 241, C, (c, c, c))
    victims = {"instnorm bf16": lambda: norm(False), "instnorm+gelu bf16": lambda: norm(True),
               "irfft 240x480": lambda: [ops.irfft_rows(F, 1, C, 480, torch.float32, (1.0, 2.0, 1.0))]}
    w = torch.randn(384, 384, device=dev) / 384 ** 0.5
    xf = torch.rand(1, 384, 181 * 1440, device=dev) - 0.5
    xb = (torch.rand(1, 384, 181, 720, device=dev) - 0.5).bfloat16()
    A = ops.pad_weight_bf16((torch.randn(768, 384, device=dev) / 384 ** 0.5).bfloat16())
    bias = torch.randn(768, device=dev)

    def culprits():
        ops.chan_gemm_f32(w, xf)
        ops.conv1x1_nn(A, 384, xb, bias=bias, act=True, want_pre=True)
    torch.cuda.synchronize()
    refs = {n: [t.clone() for t in f()] for n, f in victims.items()}
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    bad = {}
    for n, f in victims.items():
        for r in range(25):
            with torch.cuda.stream(sb):
                culprits()
                culprits()
            with torch.cuda.stream(sa):
                outs = [t.clone() for t in f()]
            sa.synchronize()
            if any(not torch.equal(o.view(torch.int16) if o.dtype == torch.bfloat16 else o, rf.view(torch.int16) if rf.dtype == torch.bfloat16 else rf)
                   for o, rf in zip(outs, refs[n])):
                bad[n] = bad.get(n, 0) + 1
        torch.cuda.synchronize()
    assert not bad, f"results differ from the idle-GPU results while matrix kernels run on a second stream: {bad}"
