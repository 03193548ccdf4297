"""SpecUNet_2D for MI355X: the reference's module name, input/output contract and state_dict layout
(reference root/code/backend/pytorch_neural_nets.py:79-197), computed by libsoftspoken_hip.so.

The torch module here only *holds* the checkpoint tensors under the reference's 224 state_dict keys
(so `load_state_dict(checkpoint['model_state_dict'])` is strict-compatible, NNDetector.py:48); no
torch operator runs in forward().  forward() hands the windows to the HIP library, which runs the
mel front-end, the folded-BatchNorm residual U-Net and both heads on the GPU.
There is no CPU path: without the library or a gfx950 device forward() raises.
"""
from __future__ import annotations

import logging

import numpy as np
import torch
import torch.nn as nn

from root.code.backend import settings
from softspoken_amd import checkpoint as _ckpt
from softspoken_amd import native as _native
from softspoken_amd import layout as _layout

WINDOW_SAMPLES = settings.vad_resample * 3


class _Holder(nn.Module):
    """A module whose only job is to own parameters / buffers under given child names."""

    def forward(self, *a, **k):   # pragma: no cover - never called
        raise RuntimeError("parameter holder: computation happens in libsoftspoken_hip.so")


def _attach(root: nn.Module, dotted: str, tensor: torch.Tensor, is_param: bool):
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Holder())
        mod = mod._modules[p]
    if is_param:
        mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))
    else:
        mod.register_buffer(parts[-1], tensor)


class SpecUNet_2D(nn.Module):
    """forward(x: (B, 66150) float32 @ 22 050 Hz) -> (spec_output (B, 2, 128, 256), mask_output (B, 1, 256)).

    mask_output are raw logits (the reference's head ends in Conv1d(4,1,1), no sigmoid,
    pytorch_neural_nets.py:137-140).  `compute_spec_output=False` skips the spec head, which the
    reference computes and then drops (worker.py:78-79); forward then returns (None, mask).
    """

    def __init__(self, precision: str | None = None, device_index: int = 0, compute_spec_output: bool = True):
        super().__init__()
        self.n_mels = 128
        self.input_shape = (66150)
        self.output_shape = (2, 128, 256)
        self.precision = precision or settings.hip_precision
        self.device_index = device_index
        self.compute_spec_output = compute_spec_output
        for key, (shape, kind) in _layout.state_dict_layout().items():
            if kind == "window":
                t = torch.from_numpy(_layout.hann_window_512())
            elif kind == "fb":
                t = torch.from_numpy(_layout.mel_filterbank())
            elif kind == "bn_count":
                t = torch.zeros((), dtype=torch.long)
            elif kind in ("bn_var", "bn_gamma"):
                t = torch.ones(shape)
            elif kind in ("bn_mean", "bn_beta", "conv_b"):
                t = torch.zeros(shape)
            else:
                fan_in = int(np.prod(shape[1:]))
                t = torch.empty(shape).uniform_(-1.0, 1.0) * (1.0 / fan_in) ** 0.5
            is_param = kind in ("conv_w", "conv_b", "bn_gamma", "bn_beta")
            _attach(self, key, t, is_param)
        self._ctx = None
        self._ctx2 = None              # the second context of ProcessWorker.run's file pipeline (hip_context(1))
        self._ctx_version = None
        self._fp32_for = None          # weights version for which the f16x2 mode has reported SS_ERR_RANGE: those run in fp32

    # -- device context ---------------------------------------------------------------------------------
    def _weights_version(self):
        return tuple((k, v._version, v.data_ptr()) for k, v in self.state_dict().items())

    def effective_precision(self) -> str:
        """settings.hip_precision, or 'fp32' once the f16x2 mode has refused these weights (with_range_fallback)."""
        return "fp32" if (self.precision == "f16x2" and self._fp32_for == self._weights_version()) else self.precision

    def hip_context(self, which: int = 0) -> _native.Context:
        """The device context of these weights.  which = 1: a second context of the same weights and precision, created when first asked
        for -- ProcessWorker.run alternates its files between the two, so that the next file's launches are queued (on the other
        context's stream) before the current file's last launch ends."""
        ver = self._weights_version()
        prec = self.effective_precision()
        if self._ctx is None or ver != self._ctx_version or self._ctx.precision != prec:
            for c in (self._ctx, self._ctx2):
                if c is not None:
                    c.close()
            self._ctx = self._ctx2 = None
            blob = _ckpt.pack_state_dict(self.state_dict())
            chunk = settings.hip_chunk_windows or None
            try:
                self._ctx = _native.Context(blob, self.device_index, precision=prec, chunk=chunk)
            except _native.NativeError as e:            # a folded weight without an f16 representation: refused at creation
                if e.code != _native.SS_ERR_RANGE or prec != "f16x2":
                    raise
                self._note_fallback(ver, e)
                self._ctx = _native.Context(blob, self.device_index, precision="fp32", chunk=chunk)
            self._ctx_version = ver
        if which == 0:
            return self._ctx
        if self._ctx2 is None:
            self._ctx2 = _native.Context(_ckpt.pack_state_dict(self.state_dict()), self.device_index, precision=self._ctx.precision,
                                         chunk=settings.hip_chunk_windows or None)
        return self._ctx2

    def _note_fallback(self, ver, err):
        if self._fp32_for != ver:
            logging.warning("f16x2 mode cannot represent this checkpoint (%s): running it in the fp32 mode", err)
        self._fp32_for = ver

    def with_range_fallback(self, fn):
        """fn(context) -> result.  The reference computes in fp32 and cannot fail on magnitude (pytorch_neural_nets.py:142-197);
        the f16x2 mode reports SS_ERR_RANGE when a weight or an activation has no f16 representation.  Then -- once per set of
        weights, with one log line -- the detector switches to a fresh fp32 context in the same process and the call is run again."""
        try:
            return fn(self.hip_context())
        except _native.NativeError as e:
            if e.code != _native.SS_ERR_RANGE or self.effective_precision() != "f16x2":
                raise
            self._note_fallback(self._weights_version(), e)
            return fn(self.hip_context())

    def forward(self, x):
        if x.dim() != 2 or x.shape[1] != WINDOW_SAMPLES:
            raise ValueError(f"expected (B, {WINDOW_SAMPLES}) windows, got {tuple(x.shape)}")
        sig = np.ascontiguousarray(x.detach().to("cpu", torch.float32).numpy()).reshape(-1)
        starts = np.arange(x.shape[0], dtype=np.int64) * WINDOW_SAMPLES

        def run(ctx):
            ctx.reset()
            fid = ctx.add_f32_22k(sig, padded=True)      # windows back to back, stored as they are
            return ctx.infer_windows(fid, starts, want_spec=self.compute_spec_output)
        spec, mask = self.with_range_fallback(run)
        mask_t = torch.from_numpy(mask).to(x.device)
        spec_t = torch.from_numpy(spec).to(x.device) if spec is not None else None
        return spec_t, mask_t
