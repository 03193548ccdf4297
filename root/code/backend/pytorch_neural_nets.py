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
        self._fp32_for = None          # weights version for which the f16x2 mode is off (refused at creation, failed the load-time
                                       # check, or reported SS_ERR_RANGE for a second input): those run in fp32
        self._fp32_tmp = None          # fp32 context for the ONE input the f16x2 mode could not represent (with_range_fallback)
        self._range_inputs = set()     # inputs (keys) for which a run-time SS_ERR_RANGE was answered in fp32
        self._range_logged = False
        self.selfcheck_delta = None    # max |f16x2 - fp32| logit of the load-time check on these weights (None: not run)

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
            if self._fp32_tmp is not None:
                self._fp32_tmp.close()
                self._fp32_tmp = None
            if ver != self._ctx_version:
                self._range_inputs = set()
                self._range_logged = False
            try:
                self._ctx = _native.Context(blob, self.device_index, precision=prec, chunk=chunk)
            except _native.NativeError as e:            # a folded weight without an f16 representation: refused at creation
                if e.code != _native.SS_ERR_RANGE or prec != "f16x2":
                    raise
                self._note_fallback(ver, e)
                self._ctx = _native.Context(blob, self.device_index, precision="fp32", chunk=chunk)
            self._ctx_version = ver
            if self._ctx.precision == "f16x2" and getattr(settings, "hip_selfcheck", True) and not self._selfcheck(blob, ver):
                self._ctx.close()                       # f16x2 loses precision on THESE weights without overflowing: fp32 from here on
                self._ctx = _native.Context(blob, self.device_index, precision="fp32", chunk=chunk)
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

    # -- load-time check of the f16x2 mode on the weights actually loaded ------------------------------------
    SELFCHECK_TOLERANCE = 5e-5         # half the path's 1e-4 contract (BASELINE.json north_star)

    def _selfcheck(self, blob, ver) -> bool:
        """The f16x2 mode carries 22 significant bits per operand only while the values it stores sit where an f16 pair has them
        (DESIGN.md section 3); overflow and non-finite values are reported by the library (SS_ERR_RANGE), precision lost WITHOUT overflow
        is not.  So before the f16x2 context of a set of weights is used, a fixed handful of deterministic windows (loud, quiet,
        digital silence) goes through it and through a short-lived fp32 context of the same weights
        (reference arithmetic: NNDetector.py:21-53 loads the checkpoint and computes in fp32); beyond SELFCHECK_TOLERANCE the detector
        logs once and keeps fp32 for these weights.  < 1 s, once per set of weights."""
        from softspoken_amd import synth
        n = 8
        x = synth.synth_audio(4242, 3.0 * n, 22050, 1, bursts=n, with_silence=False).astype(np.float32).reshape(-1)[:n * WINDOW_SAMPLES]
        x[1 * WINDOW_SAMPLES:2 * WINDOW_SAMPLES] *= np.float32(1e-3)      # a quiet window (the log10(x + 1) cancellation)
        x[2 * WINDOW_SAMPLES:3 * WINDOW_SAMPLES] = 0.0                     # digital silence (exact-zero features)
        x[5 * WINDOW_SAMPLES:6 * WINDOW_SAMPLES] *= np.float32(3.0)       # beyond full scale (float WAVs may carry it)
        starts = np.arange(n, dtype=np.int64) * WINDOW_SAMPLES
        ref = None
        try:
            def logits(ctx):
                ctx.reset()
                return ctx.infer_windows(ctx.add_f32_22k(x, padded=True), starts, want_spec=False)[1]
            try:
                got = logits(self._ctx)
            except _native.NativeError as e:
                if e.code != _native.SS_ERR_RANGE:
                    raise
                self._note_fallback(ver, e)
                return False
            ref = _native.Context(blob, self.device_index, precision="fp32", chunk=n)
            want = logits(ref)
        finally:
            if ref is not None:
                ref.close()
            if self._ctx.alive:
                self._ctx.reset()
        ok = np.isfinite(got).all() and np.isfinite(want).all()
        # the contract is absolute (scores within 1e-4) for scores of ordinary size; a checkpoint whose scores are huge is held to
        # the same number of digits (fp32 itself carries no more)
        scale = max(1.0, float(np.abs(want).max())) if ok else 1.0
        self.selfcheck_delta = float(np.abs(got - want).max()) / scale if ok else float("inf")
        if self.selfcheck_delta > self.SELFCHECK_TOLERANCE:
            self._note_fallback(ver, "load-time check: f16x2 and fp32 scores of these weights differ by %.2e (> %.0e; score scale %.1e)"
                                % (self.selfcheck_delta, self.SELFCHECK_TOLERANCE, scale))
            return False
        return True

    # -- SS_ERR_RANGE at run time ----------------------------------------------------------------------------
    def fp32_context(self) -> _native.Context:
        """A context of these weights in the fp32 mode for the input the f16x2 mode could not represent (created when first needed;
        dropped with the weights)."""
        if self._fp32_tmp is None or not self._fp32_tmp.alive:
            self._fp32_tmp = _native.Context(_ckpt.pack_state_dict(self.state_dict()), self.device_index, precision="fp32",
                                             chunk=settings.hip_chunk_windows or None)
        return self._fp32_tmp

    def range_refused(self, key, err) -> bool:
        """Book a run-time SS_ERR_RANGE of the f16x2 mode for the input `key` -> True when the detector now runs everything in fp32.
        One input that the mode cannot represent (a NaN or Inf sample in a float WAV) says nothing about the checkpoint: that input
        alone is run again in fp32 (fp32_context) and f16x2 stays for the others.  A SECOND input with the status does: the
        detector switches for good, with the one log line."""
        self._range_inputs.add(key)
        if len(self._range_inputs) >= 2:
            self._note_fallback(self._weights_version(), err)
            return True
        if not self._range_logged:
            logging.warning("f16x2 mode cannot represent an input (%s): running that input in the fp32 mode", err)
            self._range_logged = True
        return False

    def with_range_fallback(self, fn, key=None):
        """fn(context) -> result.  The reference computes in fp32 and cannot fail on magnitude (pytorch_neural_nets.py:142-197);
        the f16x2 mode reports SS_ERR_RANGE when a weight or an activation has no f16 representation.  The call is then run again
        in fp32: for a first input (key: what names it; None = this call) on a side context, from the second input on with the
        detector switched to fp32 for these weights (range_refused)."""
        try:
            return fn(self.hip_context())
        except _native.NativeError as e:
            if e.code != _native.SS_ERR_RANGE or self.effective_precision() != "f16x2":
                raise
            if self.range_refused(key if key is not None else object(), e):
                return fn(self.hip_context())
            return fn(self.fp32_context())

    def drop_second_context(self):
        """ProcessWorker.run: the second file context's workspace does not fit (SS_ERR_NOMEM): give its memory back."""
        if self._ctx2 is not None:
            self._ctx2.close()
            self._ctx2 = None

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
