"""Autoregressive rollout around the network: the train-time caller of the SFNO hot path when makani runs with
``--multistep_count > 1`` (``makani/models/stepper.py:176-345``).

makani's wrapper threads every step through ``Preprocessor2D`` (history normalisation, unpredicted / static feature
channels, input noise, bias correction — the data pipeline, SURVEY §2: OUT).  In the north-star configuration all
of those stages are identities (``history_normalization_mode: "none"``, no zenith / orography / land-mask channels,
no noise), and what remains is the part that decides the cost of a rollout on the GPU:

* the loop itself — ``n_future + 1`` network calls, the prediction of step k appended to the history window that
  feeds step k+1 (``stepper.py:236-281``, ``preprocessor.py:341-410``), all steps concatenated along the channel axis
  (``stepper.py:284``);
* **push-forward mode** — the input of every step is detached, gradients do not flow through the rollout
  (``stepper.py:246-247``);
* **rollout checkpointing** — only the network call of a step is recomputed in backward, so the activation
  footprint of backprop-through-time stays that of ONE step (``stepper.py:262-265``).  Measured at 721x1440 with
  ``bench.py --multistep-count 4``: 88.5 GB peak and 203 ms per 4-step sample without, 33.4 GB and 276 ms with
  checkpointing — on 288 GB of HBM the plain rollout is the faster default, checkpointing buys batch size.
  The HIP autograd functions hold no private RNG state and write their saved tensors only once, so recomputation is
  bit-identical to the first forward (``tests/test_gpu_model.py::test_rollout_checkpointing_is_exact_and_matches_manual_unroll``).

Evaluation mode performs ONE step (``stepper.py:286-313``): inference drives the rollout itself.
"""
import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint


def _private_generators(module):
    """class names of sub-modules that advance their own ``torch.Generator`` (their masks would not be restored by a
    checkpoint recompute — ``stepper.py:24-47``)"""
    return sorted({type(m).__name__ for m in module.modules()
                   if isinstance(getattr(m, "rng_cpu", None), torch.Generator)
                   or isinstance(getattr(m, "rng_gpu", None), torch.Generator)})


class MultiStepWrapper(nn.Module):
    """``MultiStepWrapper(model, n_future=, n_history=, push_forward=, multistep_checkpoint=)`` or, with makani's own
    calling convention, ``MultiStepWrapper.from_params(params, model_handle)`` (``model_registry.py:257-262``).

    ``forward(inp)``: ``inp`` (B, (n_history + 1)·C, H, W) → (B, (n_future + 1)·C_out, H, W) in training mode,
    (B, C_out, H, W) in evaluation mode.  ``update_state`` / ``replace_state`` are accepted for signature
    compatibility; there is no stochastic state to advance here."""

    def __init__(self, model, n_future=0, n_history=0, push_forward=False, multistep_checkpoint=False):
        super().__init__()
        if n_future < 0 or n_history < 0:
            raise ValueError(f"n_future ({n_future}) and n_history ({n_history}) must be >= 0")
        self.model = model
        self.n_future, self.n_history = int(n_future), int(n_history)
        self.push_forward_mode = bool(push_forward)
        self.multistep_checkpoint = bool(multistep_checkpoint)
        if self.multistep_checkpoint:
            offenders = _private_generators(model)
            if offenders:
                raise RuntimeError(f"multistep_checkpoint is incompatible with modules carrying private RNG generators "
                                   f"(found: {offenders}): their masks are not restored on the checkpoint recompute")

    @classmethod
    def from_params(cls, params, model_handle):
        """makani's ``(params, model_handle)`` constructor.  Reads ``n_future``, ``n_history``,
        ``multistep.push_forward``, ``multistep_checkpoint`` (``stepper.py:205-224``); preprocessor stages this package
        does not carry raise instead of being silently skipped."""
        get = params.get if hasattr(params, "get") else (lambda k, d=None: getattr(params, k, d))
        mode = get("history_normalization_mode", "none")
        if mode != "none":
            raise NotImplementedError(f"history_normalization_mode {mode!r}: only 'none' (the preprocessor is out of scope)")
        for key in ("input_noise", "bias_correction"):
            if get(key, None) is not None:
                raise NotImplementedError(f"params.{key} is set: that preprocessor stage is out of scope here")
        for key in ("add_zenith", "add_orography", "add_landmask", "add_soiltype", "add_copernicus_emb"):
            if get(key, False):
                raise NotImplementedError(f"params.{key}: unpredicted / static feature channels are out of scope here")
        multistep = get("multistep", None) or {"push_forward": False}
        return cls(model_handle(), n_future=get("n_future", 0), n_history=get("n_history", 0),
                   push_forward=multistep["push_forward"], multistep_checkpoint=get("multistep_checkpoint", False))

    # ---- history window (preprocessor.py:234-295,341-410) ----
    def append_history(self, window, pred):
        """drop the oldest time level of ``window`` (B, (n_history+1)·C, H, W), append ``pred`` (B, C, H, W)"""
        if self.n_history == 0:
            return pred
        b, ct, h, w = window.shape
        nh = self.n_history + 1
        if ct % nh:
            raise RuntimeError(f"append_history: channel dim {ct} is not divisible by n_history + 1 = {nh}")
        c = ct // nh
        if pred.shape[1] != c:
            raise RuntimeError(f"append_history: the prediction has {pred.shape[1]} channels, one time level has {c}")
        return torch.cat([window[:, c:], pred], dim=1)

    def _step(self, x):
        if self.multistep_checkpoint and self.training and torch.is_grad_enabled() and not self.push_forward_mode:
            return checkpoint(self.model, x, use_reentrant=False, preserve_rng_state=True)
        return self.model(x)

    def forward(self, inp, update_state=True, replace_state=True):
        if not self.training:
            return self.model(inp)
        result = []
        window = inp
        for step in range(self.n_future + 1):
            if self.push_forward_mode:
                window = window.detach()
            pred = self._step(window)
            result.append(pred)
            if step < self.n_future:
                window = self.append_history(window, pred)
        return torch.cat(result, dim=1) if len(result) > 1 else result[0]


class SingleStepWrapper(nn.Module):
    """one step; ``encode_process`` forwarded when the network has it (``stepper.py:50-173``)"""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, inp, update_state=True, replace_state=True):
        return self.model(inp)

    def encode_process(self, inp, update_state=True, replace_state=True):
        if not hasattr(self.model, "encode_process"):
            raise NotImplementedError(f"{type(self.model).__name__} does not expose encode_process().")
        return self.model.encode_process(inp)
