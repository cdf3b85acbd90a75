"""ctypes binding of libmakani_amd.so (the C ABI in include/makani_amd.h).

The product path has no CPU fallback: if the shared library is missing or a
tensor is not on a GPU, calls raise.  Build with ``python -m makani_amd.build``.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MAKANI_AMD_LIB: a variant of the library built by tools/ab.py (same-box A/B measurements); default: the in-tree build
LIB_PATH = os.environ.get("MAKANI_AMD_LIB") or os.path.join(_HERE, "libmakani_amd.so")

MK_F32, MK_BF16 = 0, 1
TRI_NONE, TRI_ROW_GE, TRI_K_GE, TRI_ROW_LE, TRI_K_LE = 0, 1, 2, 3, 4

c_ll, c_int, c_f, c_vp = C.c_longlong, C.c_int, C.c_float, C.c_void_p


class MkAdamTensor(C.Structure):
    """mirrors `struct MkAdamTensor` of include/makani_amd.h"""
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_longlong),
                ("p_bf16", C.c_void_p), ("p_bf16_t", C.c_void_p), ("cols", C.c_int), ("ld", C.c_int), ("ld_t", C.c_int)]


class MkGemm(C.Structure):
    _fields_ = [
        ("A", c_vp), ("B", c_vp), ("C", c_vp),
        ("a_batch", c_ll), ("a_row", c_ll), ("a_k", c_ll),
        ("b_batch", c_ll), ("b_col", c_ll), ("b_k", c_ll),
        ("c_batch", c_ll), ("c_row", c_ll), ("c_col", c_ll),
        ("a_inner", c_ll), ("b_inner", c_ll), ("c_inner", c_ll),
        ("a_im", c_ll), ("b_im", c_ll), ("c_im", c_ll),
        ("M", c_int), ("N", c_int), ("K", c_int), ("batch", c_int),
        ("inner", c_int), ("tri_mode", c_int), ("tri_off", c_int),
        ("conj_a", c_int), ("conj_b", c_int), ("beta", c_int),
    ]


MK_FFT_SEG_MAX = 8


class MkFftSeg(C.Structure):
    """mirrors `struct MkFftSeg` of include/makani_amd.h"""
    _fields_ = [("nw", c_int), ("nh", c_int), ("m_off", c_int * (MK_FFT_SEG_MAX + 1)), ("r_off", c_int * (MK_FFT_SEG_MAX + 1)),
                ("base", (c_ll * MK_FFT_SEG_MAX) * MK_FFT_SEG_MAX), ("xseg", c_int), ("x_stride", c_ll), ("x_nlat", c_int)]


_SIGS = {
    "mk_version": ([], c_int),
    "mk_sgemm_batched": ([C.POINTER(MkGemm), c_vp], c_int),
    "mk_cgemm_batched": ([C.POINTER(MkGemm), c_vp], c_int),
    "mk_sgemm_split_batched": ([C.POINTER(MkGemm), c_int, c_vp], c_int),
    "mk_cgemm_split_batched": ([C.POINTER(MkGemm), c_int, c_vp], c_int),
    "mk_cgemm_split2_batched": ([C.POINTER(MkGemm), c_int, c_vp], c_int),
    "mk_cgemm_split2_ssq_count": ([C.POINTER(MkGemm)], c_ll),
    "mk_cgemm_split2_batched_ssq": ([C.POINTER(MkGemm), c_int, c_vp, c_vp], c_int),
    "mk_sgemm_presplit_batched": ([C.POINTER(MkGemm), c_vp, c_ll, c_ll, c_ll, c_int, c_vp, c_vp, c_int, c_vp], c_int),
    "mk_rfft_rows": ([c_vp, c_int, c_vp, c_vp, C.POINTER(c_int), c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                      c_f, c_f, c_f, c_vp], c_int),
    "mk_irfft_rows": ([c_vp, c_vp, c_int, c_vp, C.POINTER(c_int), c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                       c_f, c_f, c_f, c_vp], c_int),
    "mk_fft_seg_supported": ([c_int], c_int),
    "mk_rfft_rows_seg": ([c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_f, c_f, c_f, C.POINTER(MkFftSeg), c_vp], c_int),
    "mk_irfft_rows_seg": ([c_vp, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_f, c_f, c_f, C.POINTER(MkFftSeg), c_vp], c_int),
    "mk_weight_to_wlayout": ([c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp], c_int),
    "mk_wlayout_to_weight_grad": ([c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp], c_int),
    "mk_slayout_to_complex": ([c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp], c_int),
    "mk_complex_to_slayout": ([c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp], c_int),
    "mk_spec_sep_mul": ([c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp], c_int),
    "mk_spec_sep_wgrad": ([c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp], c_int),
    "mk_spec_diag_apply": ([c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp], c_int),
    "mk_spec_diag_wgrad": ([c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp], c_int),
    "mk_pointwise_chunks": ([c_ll, c_int, c_ll], c_int),
    "mk_plane_sums": ([c_vp, c_int, c_vp, c_vp, c_ll, c_ll, c_vp], c_int),
    "mk_instnorm_stats": ([c_vp, c_int, c_vp, c_vp, c_ll, c_ll, c_f, c_vp, c_f, c_vp], c_int),
    "mk_instnorm_merge": ([c_vp, c_vp, c_vp, c_ll, c_int, c_f, c_vp], c_int),
    "mk_instnorm_apply": ([c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_ll, c_int, c_ll, c_int, c_vp], c_int),
    "mk_instnorm_fused_chunks": ([c_ll, c_int, c_ll, c_int], c_int),
    "mk_instnorm_fwd_fused": ([c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_ll, c_int, c_ll, c_f, c_int, c_vp], c_int),
    "mk_instnorm_bwd_fused": ([c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_ll, c_int, c_ll, c_int, c_vp], c_int),
    "mk_instnorm_bwd": ([c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_f, c_vp, c_vp, c_ll, c_int, c_ll, c_ll, c_int, c_int, c_vp], c_int),
    "mk_chan_layernorm_chunks": ([c_int, c_ll], c_int),
    "mk_chan_layernorm_fwd": ([c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_ll, c_f, c_vp], c_int),
    "mk_chan_layernorm_bwd": ([c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_ll, c_vp], c_int),
    "mk_chan_layernorm_wgrad": ([c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_ll, c_vp], c_int),
    "mk_bias_gelu_fwd": ([c_vp, c_vp, c_vp, c_int, c_ll, c_int, c_ll, c_vp], c_int),
    "mk_conv1x1_nn": ([c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_ll, c_int, c_vp], c_int),
    "mk_conv1x1_wgrad_workspace": ([c_int, c_int, c_int, c_ll], c_ll),
    "mk_conv1x1_wgrad": ([c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_ll, c_int, c_vp], c_int),
    "mk_conv1x1_wgrad_fuses_bias": ([c_int, c_int, c_int, c_ll], c_int),
    "mk_conv1x1_wgrad_bias": ([c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_ll, c_int, c_vp], c_int),
    "mk_adamw_step": ([c_vp, c_vp, c_vp, c_vp, c_ll, c_vp, c_f, c_f, c_f, c_f, c_f, c_int, c_vp, c_vp], c_int),
    "mk_adamw_advance": ([c_vp, c_f, c_f, c_vp], c_int),
    "mk_bias_gelu_bwd": ([c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_ll, c_int, c_ll, c_vp], c_int),
    "mk_adamw_multi": ([c_vp, c_int, c_vp, c_f, c_f, c_f, c_f, c_f, c_int, c_vp, c_vp], c_int),
    "mk_instnorm_fwd": ([c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f, c_ll, c_int, c_ll, c_f, c_int, c_vp], c_int),
    "mk_grad_norm_workspace": ([c_vp, c_int], c_ll),
    "mk_grad_clip_coef": ([c_vp, c_int, c_f, c_vp, c_vp, c_vp], c_int),
    "mk_grad_norm_workspace_pre": ([c_vp, c_int, c_vp, c_int], c_ll),
    "mk_grad_clip_coef_pre": ([c_vp, c_int, c_vp, c_int, c_f, c_vp, c_vp, c_vp], c_int),
    "mk_spec_lp_blocks": ([c_int, c_int], c_ll),
    "mk_spec_lp_fwd": ([c_vp, c_vp, c_vp, c_int, c_int, c_ll, c_int, c_int, c_f, c_f, c_f, c_vp], c_int),
    "mk_spec_lp_bwd": ([c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_ll, c_int, c_int, c_f, c_f, c_f, c_vp], c_int),
    "mk_crps_chunks": ([c_ll], c_int),
    "mk_crps": ([c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_ll, c_int, c_f, c_f, c_int, c_vp, c_vp], c_int),
    "mk_crps_complex": ([c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_ll, c_f, c_int, c_vp], c_int),
    "mk_disco_fwd": ([c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp], c_int),
    "mk_disco_bwd": ([c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp], c_int),
    "mk_disco_bwd_same": ([c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp], c_int),
    "mk_disco_runs_shape": ([c_int, c_int, c_int, c_int, c_int, c_vp, c_vp], c_int),
    "mk_disco_fused_shape": ([c_int, c_int, c_int, c_int, c_vp, c_vp], c_int),
    "mk_disco_fwd_fused": ([c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp], c_int),
    "mk_disco_fwd_runs": ([c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp], c_int),
    "mk_disco_bwd_runs": ([c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp], c_int),
    "mk_group_mix_supported": ([c_int, c_int], c_int),
    "mk_group_mix_blocks": ([c_ll, c_int, c_int], c_int),
    "mk_group_mix": ([c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_ll, c_vp], c_int),
    "mk_group_mix_wgrad": ([c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_ll, c_vp], c_int),
    "mk_resample_fwd": ([c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp], c_int),
    "mk_resample_bwd": ([c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp], c_int),
    "mk_quad_lp_chunks": ([c_ll], c_int),
    "mk_quad_lp_fwd": ([c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_ll, c_ll, c_int, c_f, c_vp], c_int),
    "mk_quad_lp_bwd": ([c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_ll, c_ll, c_int, c_f, c_vp], c_int),
}

EXPORTS = sorted(list(_SIGS) + ["mk_last_error"])

_lib = None


def lib():
    """Load (once) and return the shared library; raise loudly if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the MI355X HIP library has not been built "
                "(run `python -m makani_amd.build`).  There is no CPU fallback."
            )
        L = C.CDLL(LIB_PATH)
        for name, (args, res) in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = res
        L.mk_last_error.argtypes = []
        L.mk_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().mk_last_error().decode()
        raise RuntimeError(f"libmakani_amd {what} failed (code {rc}): {msg}")


def stream():
    return c_vp(torch.cuda.current_stream().cuda_stream)


def device_guard(fn):
    """Module ``forward``s of the package run with the CURRENT device = the device of their first tensor argument: every
    C-ABI launch goes to ``torch.cuda.current_stream()`` of the current device, so a model living on cuda:1 while the
    process's current device is cuda:0 would otherwise enqueue its kernels on the wrong device's stream.  (Backward nodes
    run on autograd's per-device worker threads, which set the device themselves.)"""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        # the first CUDA tensor among the positional, then the keyword arguments decides (model(x=inp), loss(prd=, tar=))
        for a in (*args, *kwargs.values()):
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if a.device.index != torch.cuda.current_device():
                    with torch.cuda.device(a.device):
                        return fn(self, *args, **kwargs)
                break
        return fn(self, *args, **kwargs)

    return wrapped


def ptr(t):
    if t is None:
        return c_vp(0)
    if not t.is_cuda:
        raise RuntimeError("makani_amd ops need GPU tensors (the HIP path has no CPU fallback)")
    return c_vp(t.data_ptr())


def dtype_code(t):
    if t.dtype == torch.float32:
        return MK_F32
    if t.dtype == torch.bfloat16:
        return MK_BF16
    raise TypeError(f"unsupported dtype {t.dtype} (float32 / bfloat16 only)")


def dense_view(t):
    """``t`` as a contiguous view in MEMORY order (its dims permuted by decreasing stride), or None when ``t`` has gaps or
    overlaps.  Element-wise kernels (optimizer, norms, all-reduce) run on this view of tensors that are dense but not
    C-contiguous — the native-order dhconv weight, its gradient and its optimizer state."""
    if t.is_contiguous():
        return t
    order = sorted(range(t.dim()), key=lambda d: (-t.stride(d), -t.size(d)))
    v = t.permute(order)
    return v if v.is_contiguous() else None
